"""CLIP text encoder on the HIP kernels (prompt and GLIGEN phrase encoding).

Reference call sites: `_encode_prompt` (/root/reference/models/controllable_pipeline_text_to_video_synth.py:197-372:
`self.text_encoder(input_ids, attention_mask=...)[0]`) and the GLIGEN phrase embeddings (:751-763, `.pooler_output`).  The
arithmetic is `transformers.CLIPTextModel` (third-party, pinned by the reference at 4.36.2): token + position embedding,
N pre-LN blocks of causal self-attention (head_dim 64, q/k/v/out with bias) and a GELU / quick-GELU MLP, final LayerNorm;
pooled output = the hidden state at the EOS token.  Weights load by name from `CLIPTextModel.state_dict()` (with or
without the `text_model.` prefix).

Runs once per prompt on 77 tokens: the point is that the conditioning is produced by the same kernels (fused QKV GEMM,
`lvdhip_attention_fwd` with the causal flag, LayerNorm, residual epilogues), not speed.  The embedding lookup is a
`torch` gather (one-off, [B,77] indices).
"""
from dataclasses import dataclass

import torch

from . import ops


@dataclass
class CLIPTextConfig:
    """Defaults = the OpenCLIP ViT-H text tower used by zeroscope / modelscope (SD 2.x text encoder, 23 layers kept)."""
    vocab_size: int = 49408
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 23
    num_attention_heads: int = 16
    max_position_embeddings: int = 77
    hidden_act: str = "gelu"
    layer_norm_eps: float = 1e-5
    eos_token_id: int = 2


class TextEncoderOutput(tuple):
    """`(last_hidden_state, pooler_output)` with the attribute names of transformers' BaseModelOutputWithPooling."""
    last_hidden_state = property(lambda self: self[0])
    pooler_output = property(lambda self: self[1])


def load_clip_layers(sd, prefix, n_layers, dev):
    """Pre-LN CLIP encoder layers (text and vision towers share the block): fused q/k/v weight, bf16 matrices, fp32 vectors."""
    bf = lambda t: t.to(dev, torch.bfloat16).contiguous()
    f32 = lambda t: t.to(dev, torch.float32).contiguous()
    layers = []
    for i in range(n_layers):
        p = f"{prefix}{i}."
        qkv = torch.cat([sd[p + f"self_attn.{n}_proj.weight"] for n in "qkv"], 0)
        qkv_b = torch.cat([sd[p + f"self_attn.{n}_proj.bias"] for n in "qkv"], 0)
        layers.append(dict(
            ln1=(f32(sd[p + "layer_norm1.weight"]), f32(sd[p + "layer_norm1.bias"])),
            ln2=(f32(sd[p + "layer_norm2.weight"]), f32(sd[p + "layer_norm2.bias"])),
            qkv=(bf(qkv), f32(qkv_b)),
            out=(bf(sd[p + "self_attn.out_proj.weight"]), f32(sd[p + "self_attn.out_proj.bias"])),
            fc1=(bf(sd[p + "mlp.fc1.weight"]), f32(sd[p + "mlp.fc1.bias"])),
            fc2=(bf(sd[p + "mlp.fc2.weight"]), f32(sd[p + "mlp.fc2.bias"]))))
    return layers


def run_clip_layers(x, layers, *, samples, seq, heads, causal, act, eps):
    """x: bf16 [samples*seq, C] residual stream -> same, through every layer (LN, fused QKV GEMM, attention, out-proj with the
    residual in the GEMM epilogue, LN, MLP)."""
    C = x.shape[1]
    rows = ops.RowMap(ninner=1, os=seq, is_=0, step=1)
    for ly in layers:
        h = ops.layernorm(x, *ly["ln1"], eps=eps)
        qkv = ops.gemm(h, ly["qkv"][0], bias=ly["qkv"][1])
        o = torch.empty((samples * seq, C), dtype=torch.bfloat16, device=x.device)
        ops.attention_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, samples=samples, heads=heads, sq=seq, skv=seq, qmap=rows, kvmap=rows,
                          scale=64 ** -0.5, causal=causal)
        x = ops.gemm(o, ly["out"][0], bias=ly["out"][1], res=x)
        h = ops.layernorm(x, *ly["ln2"], eps=eps)
        h = ops.gelu(ops.gemm(h, ly["fc1"][0], bias=ly["fc1"][1]), act)
        x = ops.gemm(h, ly["fc2"][0], bias=ly["fc2"][1], res=x)
    return x


class HipCLIPTextEncoder:
    def __init__(self, cfg: CLIPTextConfig, state_dict, device="cuda"):
        assert cfg.hidden_size % cfg.num_attention_heads == 0 and cfg.hidden_size // cfg.num_attention_heads == 64, "head_dim must be 64"
        assert cfg.hidden_act in ("gelu", "quick_gelu"), cfg.hidden_act
        self.cfg, self.dev = cfg, ops.use_device(device)
        sd = {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in state_dict.items()}
        bf = lambda t: t.to(self.dev, torch.bfloat16).contiguous()
        f32 = lambda t: t.to(self.dev, torch.float32).contiguous()
        self.tok = f32(sd["embeddings.token_embedding.weight"])
        self.pos = f32(sd["embeddings.position_embedding.weight"])
        self.layers = load_clip_layers(sd, "encoder.layers.", cfg.num_hidden_layers, self.dev)
        self.final_ln = (f32(sd["final_layer_norm.weight"]), f32(sd["final_layer_norm.bias"]))

    def __call__(self, input_ids, attention_mask=None, **_):
        """input_ids (B, L<=max_position_embeddings) -> (last_hidden_state (B, L, C) fp32, pooler_output (B, C) fp32).
        `attention_mask` is accepted for signature compatibility; like the reference's SD pipelines the padding tokens
        are attended (CLIP's text tower only applies the causal mask unless a mask is passed, and the reference passes None)."""
        cfg = self.cfg
        ids = torch.as_tensor(input_ids).to(self.dev)
        B, L = ids.shape
        C, H = cfg.hidden_size, cfg.num_attention_heads
        x = (self.tok[ids] + self.pos[:L][None]).reshape(B * L, C).to(torch.bfloat16).contiguous()
        x = run_clip_layers(x, self.layers, samples=B, seq=L, heads=H, causal=True, act=cfg.hidden_act, eps=cfg.layer_norm_eps)
        last = ops.layernorm(x, *self.final_ln, eps=cfg.layer_norm_eps).float().reshape(B, L, C)
        if cfg.eos_token_id == 2:  # transformers' legacy rule: EOS is the highest id of the CLIP vocabulary
            eos = ids.argmax(-1)
        else:
            eos = (ids == cfg.eos_token_id).int().argmax(-1)
        return TextEncoderOutput((last, last[torch.arange(B, device=self.dev), eos]))
