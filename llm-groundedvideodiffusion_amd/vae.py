"""VAE decode + tensor2vid on the HIP kernels (the step right after the denoising loop).

Reference call sites: `decode_latents` (/root/reference/models/controllable_pipeline_text_to_video_synth.py:374-400) and
`tensor2vid` (:66-88); the arithmetic is diffusers 0.27.2 `AutoencoderKL.decode` (post_quant_conv -> Decoder) and
`VaeImageProcessor.postprocess`, loaded by name from an `AutoencoderKL.state_dict()`.

Same design as the denoiser: every activation is a bf16 token matrix `[(frame, y, x), C]`, so
  * every 3x3 conv (and the nearest-x2 upsample in front of the three upsampler convs) is the implicit-im2col GEMM,
  * the 1x1 shortcut / post-quant convs and the attention projections are plain GEMMs with bias + residual epilogues,
  * GroupNorm(32, eps 1e-6)+SiLU are the two-pass norm kernels with one frame per sample,
  * the single-head, dim-512 attention of the mid block is three GEMMs per frame around a row softmax:
    S = (1/sqrt C)·Q·K^T (fp32 out), P = softmax(S), O = P·V with V^T produced directly by swapping the GEMM operands
    (V^T = W_v·X^T); V's bias is added after P·V (rows of P sum to 1), so no transpose kernel and no extra pass exist.
The decoder never materialises an NCHW tensor; `tokens_to_video` writes the (F, H, W, 3) fp32 frames in [0, 1].
"""
import math

import torch

from . import ops
from .weights import VAEConfig, pack_conv3x3


class HipVAEDecoder:
    def __init__(self, cfg: VAEConfig, state_dict, device="cuda"):
        self.cfg, self.dev = cfg, torch.device(device)
        self.w = {}
        bf = lambda t: t.to(self.dev, torch.bfloat16).contiguous()
        f32 = lambda t: t.to(self.dev, torch.float32).contiguous()
        for name, t in state_dict.items():
            if name.endswith(".bias") or t.dim() == 1:
                self.w[name] = f32(t)
            elif t.dim() == 4 and t.shape[-1] == 3:
                if t.shape[1] % 8:  # conv_in: pad the 4 latent channels to 8 (the token matrix is padded the same way)
                    t = torch.cat([t, t.new_zeros(t.shape[0], 8 - t.shape[1] % 8, 3, 3)], 1)
                self.w[name] = pack_conv3x3(t.to(self.dev))
            elif t.dim() == 4:  # 1x1 convs are Linear layers on token matrices
                self.w[name] = bf(t.reshape(t.shape[0], t.shape[1]))
            else:
                self.w[name] = bf(t)
        # conv_out: 3 output channels -> 4 (GEMM N % 4), post_quant_conv: 4 -> 8 in and out (K % 8, token width of conv_in)
        wo, bo = self.w["decoder.conv_out.weight"], self.w["decoder.conv_out.bias"]
        self.w["decoder.conv_out.weight"] = torch.cat([wo, wo.new_zeros(4 - wo.shape[0] % 4, wo.shape[1])], 0).contiguous() if wo.shape[0] % 4 else wo
        self.w["decoder.conv_out.bias"] = torch.cat([bo, bo.new_zeros(self.w["decoder.conv_out.weight"].shape[0] - bo.shape[0])]).contiguous()
        wq, bq = self.w["post_quant_conv.weight"], self.w["post_quant_conv.bias"]
        pq = wq.new_zeros(8, 8)
        pq[: wq.shape[0], : wq.shape[1]] = wq
        self.w["post_quant_conv.weight"] = pq.contiguous()
        self.w["post_quant_conv.bias"] = torch.cat([bq, bq.new_zeros(8 - bq.shape[0])]).contiguous()

    # ------------------------------------------------------------------ pieces
    def _gn_silu(self, x, name, rps, silu=True):
        return ops.groupnorm(x, self.w[name + ".weight"], self.w[name + ".bias"], rps, groups=self.cfg.norm_num_groups, eps=1e-6, silu=silu)

    def _conv(self, x, name, h, w, upsample=0, res=None):
        hin, win = (2 * h, 2 * w) if upsample else (h, w)
        return ops.gemm(x, self.w[name + ".weight"], bias=self.w[name + ".bias"], res=res, mode=ops.A_CONV3X3,
                        conv=ops.ConvGeom(hin, win, hin, win, 1, upsample))

    def _resnet(self, x, name, h, w):
        rps = h * w
        t = self._conv(self._gn_silu(x, name + ".norm1", rps), name + ".conv1", h, w)
        skip = x
        if name + ".conv_shortcut.weight" in self.w:
            skip = ops.gemm(x, self.w[name + ".conv_shortcut.weight"], bias=self.w[name + ".conv_shortcut.bias"])
        return self._conv(self._gn_silu(t, name + ".norm2", rps), name + ".conv2", h, w, res=skip)

    def _mid_attention(self, x, name, frames, hw):
        c = x.shape[1]
        t = self._gn_silu(x, name + ".group_norm", hw, silu=False)
        q = ops.gemm(t, self.w[name + ".to_q.weight"], bias=self.w[name + ".to_q.bias"])
        k = ops.gemm(t, self.w[name + ".to_k.weight"], bias=self.w[name + ".to_k.bias"])
        o = torch.empty_like(x)
        scores = torch.empty((hw, hw), dtype=torch.float32, device=self.dev)
        probs = torch.empty((hw, hw), dtype=torch.bfloat16, device=self.dev)
        vt = torch.empty((c, hw), dtype=torch.bfloat16, device=self.dev)
        wv, bv = self.w[name + ".to_v.weight"], self.w[name + ".to_v.bias"]
        for f in range(frames):
            rows = slice(f * hw, (f + 1) * hw)
            ops.gemm(wv, t[rows], out=vt)                                             # V^T = W_v · X^T   [C, HW]
            ops.gemm(q[rows], k[rows], out=scores, out_fp32=True, alpha=1.0 / math.sqrt(c))  # S = Q·K^T / sqrt(C)
            ops.softmax_rows(scores, out=probs)
            ops.gemm(probs, vt, bias=bv, out=o[rows])                                  # O = P·V + b_v (rows of P sum to 1)
        return ops.gemm(o, self.w[name + ".to_out.0.weight"], bias=self.w[name + ".to_out.0.bias"], res=x)

    # ------------------------------------------------------------------ decode
    def decode_tokens(self, latents):
        """(B, 4, F, h, w) fp32 latents -> (image tokens [(b,f,y,x), 4] bf16 in [-1, 1] nominal, frames, H, W)."""
        cfg = self.cfg
        B, _, F, h, w = latents.shape
        n = B * F
        z = ops.latents_to_tokens(latents.to(self.dev, torch.float32).contiguous(), cpad=8, scale=1.0 / cfg.scaling_factor)
        x = ops.gemm(z, self.w["post_quant_conv.weight"], bias=self.w["post_quant_conv.bias"])
        x = self._conv(x, "decoder.conv_in", h, w)
        x = self._resnet(x, "decoder.mid_block.resnets.0", h, w)
        x = self._mid_attention(x, "decoder.mid_block.attentions.0", n, h * w)
        x = self._resnet(x, "decoder.mid_block.resnets.1", h, w)
        nb = len(cfg.block_out_channels)
        for i in range(nb):
            for j in range(cfg.layers_per_block + 1):
                x = self._resnet(x, f"decoder.up_blocks.{i}.resnets.{j}", h, w)
            if i != nb - 1:
                x = self._conv(x, f"decoder.up_blocks.{i}.upsamplers.0.conv", h, w, upsample=1)
                h, w = 2 * h, 2 * w
        x = self._gn_silu(x, "decoder.conv_norm_out", h * w)
        return self._conv(x, "decoder.conv_out", h, w), n, h, w

    def decode(self, latents):
        """`decode_latents` + `tensor2vid(output_type="np")`: (B, 4, F, h, w) -> (B, F, 8h, 8w, 3) fp32 in [0, 1] (device tensor)."""
        B = latents.shape[0]
        tokens, n, H, W = self.decode_tokens(latents)
        return ops.tokens_to_video(tokens, n, H, W).reshape(B, n // B, H, W, 3)

    __call__ = decode
