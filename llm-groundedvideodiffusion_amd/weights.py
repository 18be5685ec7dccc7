"""Weight handling for the HIP denoiser.

* ``unet_param_shapes`` enumerates the reference ``UNet3DConditionModel.state_dict()`` names and shapes
  for a config (models/unet_3d_condition.py:229-445 + diffusers 0.27.2 ResnetBlock2D / TemporalConvLayer /
  Downsample2D / Upsample2D / TimestepEmbedding parameter names), so reference checkpoints load by name.
* ``synthetic_state_dict`` draws seeded per-parameter random weights (there is no network: benchmarks and
  parity tests run on random-init weights of the real topology).  Values are bf16-representable so the
  fp32 oracle and the bf16 engine see identical parameters.  Tensors the reference zero-initialises
  (TemporalConvLayer.conv4, GLIGEN alpha_attn/alpha_dense, null features) get non-zero values, otherwise
  temporal convs and fusers would be identities and go untested.
* ``pack_*`` helpers re-lay weights for the kernels (tap-major implicit-GEMM conv weights, fused QKV,
  hidden/gate-interleaved GEGLU, transposed copies for the input-gradient GEMMs).
"""
import hashlib
import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Tuple

import torch


@dataclass
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attention_dim: int = 1024
    attention_head_dim: int = 64
    norm_num_groups: int = 32
    norm_eps: float = 1e-5
    attention_type: str = "default"  # "gated" adds the GLIGEN fusers + position_net
    down_block_types: Tuple[str, ...] = ("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D")
    up_block_types: Tuple[str, ...] = ("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D")
    sample_size: int = None
    transformer_in_heads: int = 8

    @property
    def gated(self):
        return self.attention_type == "gated"


TINY = dict(block_out_channels=(64, 128, 128, 128), layers_per_block=1, cross_attention_dim=64, attention_head_dim=64)


# ----------------------------------------------------------------------------- topology enumeration
def _lin(d, name, n_out, n_in, bias=True):
    d[name + ".weight"] = (n_out, n_in)
    if bias:
        d[name + ".bias"] = (n_out,)


def _norm(d, name, c):
    d[name + ".weight"] = (c,)
    d[name + ".bias"] = (c,)


def _attn(d, name, dim, kv_dim):
    _lin(d, name + ".to_q", dim, dim, bias=False)
    _lin(d, name + ".to_k", dim, kv_dim, bias=False)
    _lin(d, name + ".to_v", dim, kv_dim, bias=False)
    _lin(d, name + ".to_out.0", dim, dim)


def _ff(d, name, dim):
    _lin(d, name + ".net.0.proj", dim * 8, dim)
    _lin(d, name + ".net.2", dim, dim * 4)


def _tblock(d, name, dim, cross_dim, double_self, gated, fuser_ctx):
    # models/attention.py:64-177 registration order: norm1, attn1, norm2, attn2, norm3, ff, fuser
    _norm(d, name + ".norm1", dim)
    _attn(d, name + ".attn1", dim, dim)
    _norm(d, name + ".norm2", dim)
    _attn(d, name + ".attn2", dim, dim if double_self else cross_dim)
    _norm(d, name + ".norm3", dim)
    _ff(d, name + ".ff", dim)
    if gated:
        f = name + ".fuser"
        d[f + ".alpha_attn"] = ()  # own parameters precede sub-module parameters in state_dict()
        d[f + ".alpha_dense"] = ()
        _lin(d, f + ".linear", dim, fuser_ctx)
        _attn(d, f + ".attn", dim, dim)
        _ff(d, f + ".ff", dim)
        _norm(d, f + ".norm1", dim)
        _norm(d, f + ".norm2", dim)


def _transformer2d(d, name, c, cfg):
    _norm(d, name + ".norm", c)
    _lin(d, name + ".proj_in", c, c)
    _tblock(d, name + ".transformer_blocks.0", c, cfg.cross_attention_dim, False, cfg.gated, cfg.cross_attention_dim)
    _lin(d, name + ".proj_out", c, c)


def _transformer_temporal(d, name, c, inner):
    _norm(d, name + ".norm", c)
    _lin(d, name + ".proj_in", inner, c)
    _tblock(d, name + ".transformer_blocks.0", inner, None, True, False, None)
    _lin(d, name + ".proj_out", c, inner)


def _resnet(d, name, cin, cout, temb):
    _norm(d, name + ".norm1", cin)
    d[name + ".conv1.weight"] = (cout, cin, 3, 3)
    d[name + ".conv1.bias"] = (cout,)
    _lin(d, name + ".time_emb_proj", cout, temb)
    _norm(d, name + ".norm2", cout)
    d[name + ".conv2.weight"] = (cout, cout, 3, 3)
    d[name + ".conv2.bias"] = (cout,)
    if cin != cout:
        d[name + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
        d[name + ".conv_shortcut.bias"] = (cout,)


def _temp_conv(d, name, c):
    for k, conv_idx in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
        _norm(d, f"{name}.{k}.0", c)
        d[f"{name}.{k}.{conv_idx}.weight"] = (c, c, 3, 1, 1)
        d[f"{name}.{k}.{conv_idx}.bias"] = (c,)


def unet_param_shapes(cfg: UNetConfig) -> "OrderedDict[str, tuple]":
    d = OrderedDict()
    boc = cfg.block_out_channels
    temb = boc[0] * 4
    d["conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
    d["conv_in.bias"] = (boc[0],)
    _lin(d, "time_embedding.linear_1", temb, boc[0])
    _lin(d, "time_embedding.linear_2", temb, temb)
    _transformer_temporal(d, "transformer_in", boc[0], cfg.transformer_in_heads * cfg.attention_head_dim)
    out_c = boc[0]
    for i, btype in enumerate(cfg.down_block_types):
        in_c, out_c = out_c, boc[i]
        name = f"down_blocks.{i}"
        has_attn = btype == "CrossAttnDownBlock3D"
        for j in range(cfg.layers_per_block):
            _resnet(d, f"{name}.resnets.{j}", in_c if j == 0 else out_c, out_c, temb)
        for j in range(cfg.layers_per_block):
            _temp_conv(d, f"{name}.temp_convs.{j}", out_c)
        if has_attn:
            for j in range(cfg.layers_per_block):
                _transformer2d(d, f"{name}.attentions.{j}", out_c, cfg)
            for j in range(cfg.layers_per_block):
                _transformer_temporal(d, f"{name}.temp_attentions.{j}", out_c, out_c)
        if i != len(boc) - 1:
            d[f"{name}.downsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            d[f"{name}.downsamplers.0.conv.bias"] = (out_c,)
    rev = list(reversed(boc))
    out_c = rev[0]
    for i, btype in enumerate(cfg.up_block_types):
        prev_out = out_c
        out_c = rev[i]
        in_c = rev[min(i + 1, len(boc) - 1)]
        name = f"up_blocks.{i}"
        has_attn = btype == "CrossAttnUpBlock3D"
        nl = cfg.layers_per_block + 1
        for j in range(nl):
            skip = in_c if j == nl - 1 else out_c
            rin = prev_out if j == 0 else out_c
            _resnet(d, f"{name}.resnets.{j}", rin + skip, out_c, temb)
        for j in range(nl):
            _temp_conv(d, f"{name}.temp_convs.{j}", out_c)
        if has_attn:
            for j in range(nl):
                _transformer2d(d, f"{name}.attentions.{j}", out_c, cfg)
            for j in range(nl):
                _transformer_temporal(d, f"{name}.temp_attentions.{j}", out_c, out_c)
        if i != len(boc) - 1:
            d[f"{name}.upsamplers.0.conv.weight"] = (out_c, out_c, 3, 3)
            d[f"{name}.upsamplers.0.conv.bias"] = (out_c,)
    # mid_block is registered after up_blocks in the reference module (unet_3d_condition.py:323-356)
    c = boc[-1]
    _resnet(d, "mid_block.resnets.0", c, c, temb)
    _resnet(d, "mid_block.resnets.1", c, c, temb)
    _temp_conv(d, "mid_block.temp_convs.0", c)
    _temp_conv(d, "mid_block.temp_convs.1", c)
    _transformer2d(d, "mid_block.attentions.0", c, cfg)
    _transformer_temporal(d, "mid_block.temp_attentions.0", c, c)
    _norm(d, "conv_norm_out", boc[0])
    d["conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3)
    d["conv_out.bias"] = (cfg.out_channels,)
    if cfg.gated:
        pl = cfg.cross_attention_dim
        pos_dim = 8 * 2 * 4  # fourier_freqs * (sin, cos) * xyxy
        d["position_net.null_positive_feature"] = (pl,)
        d["position_net.null_position_feature"] = (pos_dim,)
        _lin(d, "position_net.linears.0", 512, pl + pos_dim)
        _lin(d, "position_net.linears.2", 512, 512)
        _lin(d, "position_net.linears.4", cfg.cross_attention_dim, 512)
    return d


def _seed_for(name, seed):
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:8], "little") % (2**63 - 1)


def synthetic_state_dict(cfg: UNetConfig, seed: int = 0, gain: float = 1.0, device="cpu"):
    """Per-parameter seeded fp32 tensors, rounded to bf16-representable values.

    device="cpu" (default) is bit-reproducible everywhere and is what the parity fixtures use;
    device="cuda" draws on the GPU (fast path for the 1.4 B-parameter benchmark model)."""
    sd = OrderedDict()
    for name, shape in unet_param_shapes(cfg).items():
        g = torch.Generator(device=device).manual_seed(_seed_for(name, seed))
        leaf = name.rsplit(".", 1)[-1]
        if shape == ():
            t = torch.tensor(0.6 if name.endswith("alpha_attn") else -0.4, device=device)
        elif leaf == "weight" and len(shape) == 1:  # norm gamma
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        elif leaf == "bias":
            t = 0.05 * torch.randn(shape, generator=g, device=device)
        elif len(shape) == 1:  # null features
            t = 0.5 * torch.randn(shape, generator=g, device=device)
        else:
            fan_in = 1
            for s in shape[1:]:
                fan_in *= s
            t = gain * torch.randn(shape, generator=g, device=device) / math.sqrt(fan_in)
        sd[name] = t.to(torch.bfloat16).to(torch.float32)
    return sd


# ----------------------------------------------------------------------------- VAE decoder
@dataclass
class VAEConfig:
    """diffusers AutoencoderKL config of the zeroscope / modelscope checkpoints (SD VAE)."""
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_num_groups: int = 32
    scaling_factor: float = 0.18215


VAE_TINY = dict(block_out_channels=(32, 64, 64, 64), layers_per_block=1)


def vae_decoder_param_shapes(cfg: VAEConfig) -> "OrderedDict[str, tuple]":
    """`AutoencoderKL.state_dict()` names of the decode half (diffusers 0.27.2 Decoder / UNetMidBlock2D / UpDecoderBlock2D)."""
    d = OrderedDict()

    def conv(name, cout, cin, k):
        d[name + ".weight"] = (cout, cin, k, k)
        d[name + ".bias"] = (cout,)

    def res(name, cin, cout):
        _norm(d, name + ".norm1", cin)
        conv(name + ".conv1", cout, cin, 3)
        _norm(d, name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cout, cin, 1)

    top = cfg.block_out_channels[-1]
    conv("decoder.conv_in", top, cfg.latent_channels, 3)
    a = "decoder.mid_block.attentions.0"
    _norm(d, a + ".group_norm", top)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(d, f"{a}.{nm}", top, top)
    res("decoder.mid_block.resnets.0", top, top)
    res("decoder.mid_block.resnets.1", top, top)
    rev = list(reversed(cfg.block_out_channels))
    cin = rev[0]
    for i, cout in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            res(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
        cin = cout
    _norm(d, "decoder.conv_norm_out", rev[-1])
    conv("decoder.conv_out", cfg.out_channels, rev[-1], 3)
    conv("post_quant_conv", cfg.latent_channels, cfg.latent_channels, 1)
    return d


def vae_encoder_param_shapes(cfg: VAEConfig) -> "OrderedDict[str, tuple]":
    """`AutoencoderKL.state_dict()` names of the encode half (diffusers 0.27.2 Encoder / DownEncoderBlock2D / UNetMidBlock2D
    + quant_conv): conv_in 3->c0, per level 2 resnets (+ stride-2 Downsample2D except at the last), mid block, 2*latent moments."""
    d = OrderedDict()

    def conv(name, cout, cin, k):
        d[name + ".weight"] = (cout, cin, k, k)
        d[name + ".bias"] = (cout,)

    def res(name, cin, cout):
        _norm(d, name + ".norm1", cin)
        conv(name + ".conv1", cout, cin, 3)
        _norm(d, name + ".norm2", cout)
        conv(name + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(name + ".conv_shortcut", cout, cin, 1)

    chans = cfg.block_out_channels
    conv("encoder.conv_in", chans[0], cfg.out_channels, 3)
    cin = chans[0]
    for i, cout in enumerate(chans):
        for j in range(cfg.layers_per_block):
            res(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != len(chans) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
        cin = cout
    top = chans[-1]
    a = "encoder.mid_block.attentions.0"
    _norm(d, a + ".group_norm", top)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        _lin(d, f"{a}.{nm}", top, top)
    res("encoder.mid_block.resnets.0", top, top)
    res("encoder.mid_block.resnets.1", top, top)
    _norm(d, "encoder.conv_norm_out", top)
    conv("encoder.conv_out", 2 * cfg.latent_channels, top, 3)
    conv("quant_conv", 2 * cfg.latent_channels, 2 * cfg.latent_channels, 1)
    return d


def synthetic_vae_state_dict(cfg: VAEConfig, seed: int = 0, gain: float = 1.0, device="cpu", encoder=False):
    """Seeded random decoder (and, with `encoder=True`, encoder) weights, bf16-representable, same recipe as `synthetic_state_dict`."""
    sd = OrderedDict()
    shapes = vae_decoder_param_shapes(cfg)
    if encoder:
        shapes.update(vae_encoder_param_shapes(cfg))
    for name, shape in shapes.items():
        g = torch.Generator(device=device).manual_seed(_seed_for("vae." + name, seed))
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "weight" and len(shape) == 1:
            t = 1.0 + 0.1 * torch.randn(shape, generator=g, device=device)
        elif leaf == "bias":
            t = 0.05 * torch.randn(shape, generator=g, device=device)
        else:
            fan_in = 1
            for s_ in shape[1:]:
                fan_in *= s_
            t = gain * torch.randn(shape, generator=g, device=device) / math.sqrt(fan_in)
        sd[name] = t.to(torch.bfloat16).to(torch.float32)
    return sd


# ----------------------------------------------------------------------------- packing helpers
def pack_conv3x3(w):
    """[cout,cin,3,3] -> [cout, 9*cin] tap-major (k = (ky*3+kx)*cin + c)."""
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous().to(torch.bfloat16)


def pack_conv3x3_dgrad(w):
    """Input-gradient of a stride-1 pad-1 conv as a conv over dY: Wd[ci, tap'*cout+co] = W[co,ci,2-ky',2-kx']."""
    return w.flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], -1).contiguous().to(torch.bfloat16)


def pack_conv3x3_dgrad_t2(w):
    """Input-gradient of the stride-2 pad-1 conv (LVD_A_CONV3X3_T2 loader): Wd[ci, tap*cout+co] = W[co,ci,ky,kx]."""
    return w.permute(1, 2, 3, 0).reshape(w.shape[1], -1).contiguous().to(torch.bfloat16)


def pack_tconv3(w):
    """Conv3d (3,1,1) weight [cout,cin,3,1,1] -> [cout, 3*cin] tap-major."""
    return w[..., 0, 0].permute(0, 2, 1).reshape(w.shape[0], -1).contiguous().to(torch.bfloat16)


def pack_tconv3_dgrad(w):
    """dX[f] = sum_t dY[f+1-t] W[t]  ->  taps flipped, in/out swapped: [cin, 3*cout]."""
    return w[..., 0, 0].flip(2).permute(1, 2, 0).reshape(w.shape[1], -1).contiguous().to(torch.bfloat16)


def interleave_geglu(w, b=None):
    """GEGLU proj [2*inner, K] (= [hidden ; gate]) -> rows interleaved in blocks of 32 (hidden32, gate32, ...)."""
    inner = w.shape[0] // 2
    assert inner % 32 == 0
    nb = inner // 32
    wi = torch.stack([w[:inner].reshape(nb, 32, -1), w[inner:].reshape(nb, 32, -1)], 1).reshape(2 * inner, -1).contiguous()
    if b is None:
        return wi
    bi = torch.stack([b[:inner].reshape(nb, 32), b[inner:].reshape(nb, 32)], 1).reshape(2 * inner).contiguous()
    return wi, bi


def deinterleave_geglu_cols(pre):
    """Inverse of the column order produced by an interleaved GEGLU projection: -> (hidden, gate)."""
    m, n2 = pre.shape
    v = pre.reshape(m, n2 // 64, 2, 32)
    return v[:, :, 0].reshape(m, n2 // 2), v[:, :, 1].reshape(m, n2 // 2)
