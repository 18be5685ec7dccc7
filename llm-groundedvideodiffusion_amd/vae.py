"""VAE decode + tensor2vid on the HIP kernels (the step right after the denoising loop), and the encode
half for the video-to-video upsampler (row 4, `HipVAEEncoder` at the end of the file).

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


def _load_vae_weights(state_dict, dev, prefixes, flip_taps=()):
    """AutoencoderKL tensors under `prefixes` as kernel operands: 3x3 convs tap-major bf16 (input channels padded to a multiple
    of 8), 1x1 convs / Linear as bf16 matrices, vectors fp32.  `flip_taps`: name fragments whose 3x3 taps are stored rotated by
    180 degrees (the encoder's asymmetric stride-2 convs, see HipVAEEncoder._downsample)."""
    w = {}
    for name, t in state_dict.items():
        if not name.startswith(prefixes):
            continue
        if name.endswith(".bias") or t.dim() == 1:
            w[name] = t.to(dev, torch.float32).contiguous()
        elif t.dim() == 4 and t.shape[-1] == 3:
            if t.shape[1] % 8:  # conv_in: pad the 4 latent / 3 RGB channels to 8 (the token matrix is padded the same way)
                t = torch.cat([t, t.new_zeros(t.shape[0], 8 - t.shape[1] % 8, 3, 3)], 1)
            if any(f in name for f in flip_taps):
                t = t.flip(2, 3)
            w[name] = pack_conv3x3(t.to(dev))
        elif t.dim() == 4:  # 1x1 convs are Linear layers on token matrices
            w[name] = t.reshape(t.shape[0], t.shape[1]).to(dev, torch.bfloat16).contiguous()
        else:
            w[name] = t.to(dev, torch.bfloat16).contiguous()
    return w


class _VAEBlocks:
    """ResnetBlock2D / mid attention shared by the two halves of AutoencoderKL."""

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


class HipVAEDecoder(_VAEBlocks):
    def __init__(self, cfg: VAEConfig, state_dict, device="cuda"):
        self.cfg, self.dev = cfg, ops.use_device(device)
        self.w = _load_vae_weights(state_dict, self.dev, ("decoder.", "post_quant_conv."))
        # conv_out: 3 output channels -> 4 (GEMM N % 4), post_quant_conv: 4 -> 8 in and out (K % 8, token width of conv_in)
        wo, bo = self.w["decoder.conv_out.weight"], self.w["decoder.conv_out.bias"]
        self.w["decoder.conv_out.weight"] = torch.cat([wo, wo.new_zeros(4 - wo.shape[0] % 4, wo.shape[1])], 0).contiguous() if wo.shape[0] % 4 else wo
        self.w["decoder.conv_out.bias"] = torch.cat([bo, bo.new_zeros(self.w["decoder.conv_out.weight"].shape[0] - bo.shape[0])]).contiguous()
        wq, bq = self.w["post_quant_conv.weight"], self.w["post_quant_conv.bias"]
        pq = wq.new_zeros(8, 8)
        pq[: wq.shape[0], : wq.shape[1]] = wq
        self.w["post_quant_conv.weight"] = pq.contiguous()
        self.w["post_quant_conv.bias"] = torch.cat([bq, bq.new_zeros(8 - bq.shape[0])]).contiguous()

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


class HipVAEEncoder(_VAEBlocks):
    """`AutoencoderKL.encode(...).latent_dist.sample()` on the same kernels: the front of the video-to-video upsampler
    (/root/reference/scripts/upsample.py:49-77 -> diffusers VideoToVideoSDPipeline.prepare_latents).  Encoder of diffusers
    0.27.2: conv_in, four DownEncoderBlock2D (two ResnetBlock2D, then Downsample2D(padding=0) = zero-pad right/bottom by one +
    stride-2 3x3 conv on all but the last), UNetMidBlock2D, GroupNorm+SiLU+conv_out to the 8 moment channels, quant_conv.

    The asymmetric stride-2 conv is the symmetric pad-1 stride-2 loader applied to the image rotated by 180 degrees with
    180-degree-rotated taps (out[y] = sum_k w[k] in[2y+k] <=> out_r[y'] = sum_k w_r[k] in_r[2y'+k-1] with y' = H/2-1-y), so the
    conv kernel family is untouched; the two rotations are row gathers on the token matrix."""

    def __init__(self, cfg: VAEConfig, state_dict, device="cuda"):
        self.cfg, self.dev = cfg, ops.use_device(device)
        self.w = _load_vae_weights(state_dict, self.dev, ("encoder.", "quant_conv."), flip_taps=(".downsamplers.",))

    def _downsample(self, x, name, n, h, w):
        assert h % 2 == 0 and w % 2 == 0, "encoder input height/width must be divisible by 8"
        c = x.shape[1]
        xr = x.view(n, h, w, c).flip(1, 2).reshape(n * h * w, c)
        y = ops.gemm(xr, self.w[name + ".weight"], bias=self.w[name + ".bias"], mode=ops.A_CONV3X3, conv=ops.ConvGeom(h, w, h // 2, w // 2, 2, 0))
        return y.view(n, h // 2, w // 2, y.shape[1]).flip(1, 2).reshape(n * (h // 2) * (w // 2), y.shape[1])

    def moments(self, tokens, n, h, w):
        """image tokens [(f,y,x), 8] bf16 in [-1,1] (channels 3..7 zero) -> (mean, logvar) fp32 token matrices [(f,y',x'), 4]."""
        cfg = self.cfg
        x = self._conv(tokens, "encoder.conv_in", h, w)
        nb = len(cfg.block_out_channels)
        for i in range(nb):
            for j in range(cfg.layers_per_block):
                x = self._resnet(x, f"encoder.down_blocks.{i}.resnets.{j}", h, w)
            if i != nb - 1:
                x = self._downsample(x, f"encoder.down_blocks.{i}.downsamplers.0.conv", n, h, w)
                h, w = h // 2, w // 2
        x = self._resnet(x, "encoder.mid_block.resnets.0", h, w)
        x = self._mid_attention(x, "encoder.mid_block.attentions.0", n, h * w)
        x = self._resnet(x, "encoder.mid_block.resnets.1", h, w)
        x = self._conv(self._gn_silu(x, "encoder.conv_norm_out", h * w), "encoder.conv_out", h, w)
        m = ops.gemm(x, self.w["quant_conv.weight"], bias=self.w["quant_conv.bias"], out_fp32=True)
        L = cfg.latent_channels
        return m[:, :L], m[:, L:2 * L].clamp(-30.0, 20.0), h, w

    def encode(self, frames, eps=None, generator=None, size=None, resample="lanczos"):
        """frames uint8 (F, H, W, 3) -> scaling_factor * sample of the posterior, (1, 4, F, h, w) fp32.  `size` = (H', W'):
        PIL-exact resize first (`prepare_init_upsampled`, upsample.py:15-28); `eps` (F, 4, h, w) standard normal, drawn from
        `generator` when omitted (the reference draws it inside `latent_dist.sample`)."""
        cfg = self.cfg
        frames = torch.as_tensor(frames).to(self.dev).contiguous()
        F_, H, W, _ = frames.shape
        SH, SW = size if size is not None else (H, W)
        tokens = ops.frames_to_patches(frames, (SH, SW), 1, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5), kind=resample, width=8)
        mean, logvar, h, w = self.moments(tokens, F_, SH, SW)
        if eps is None:
            eps = torch.randn((F_, cfg.latent_channels, h, w), generator=generator, device=generator.device if generator is not None else self.dev)
        e = eps.to(self.dev, torch.float32).permute(0, 2, 3, 1).reshape(-1, cfg.latent_channels)
        z = ((mean + torch.exp(0.5 * logvar) * e) * cfg.scaling_factor).contiguous()
        return ops.tokens_to_latents(z, 1, cfg.latent_channels, F_, h, w)

    __call__ = encode
