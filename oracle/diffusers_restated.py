"""ORACLE SUPPORT (test infrastructure only) — the six diffusers==0.27.2 arithmetic classes the reference imports but does not vendor
(models/unet_3d_blocks.py:181-199,330-349; models/unet_3d_condition.py:306-313), restated as nn.Modules from the public 0.27.2
definitions.  PARITY UNPINNED: diffusers is neither installed here nor present under /root/reference.

Used by (i) oracle/make_golden.py, whose import shim hands these classes to the REAL reference code so that its UNet can be built
and run to produce tests/golden/*.npz, and (ii) tests/test_oracle_restatements.py, which checks them op by op against the
functional restatement in oracle/unet_ref.py on independently seeded parameters — two transcriptions by construction, so a slip
in either one shows up.

Constructor defaults assumed (each is what the reference's call sites pass or leave at the 0.27.2 default):
  ResnetBlock2D(in_channels, out_channels, temb_channels, eps=norm_eps (1e-5 from UNet3DConditionModel), groups=32, dropout=0.0,
                time_embedding_norm="default", non_linearity="swish", output_scale_factor=1.0, pre_norm=True)   unet_3d_blocks.py:330-349
  TemporalConvLayer(in_dim, out_dim, dropout=0.1 (identity in eval), norm_num_groups=32); GroupNorm eps 1e-5; conv4 zero-initialised
                                                                                                                 unet_3d_blocks.py:181-199
  Downsample2D(channels, use_conv=True, out_channels, padding=1, name="op"): one 3x3 stride-2 pad-1 conv
  Upsample2D(channels, use_conv=True, out_channels): nearest x2 (or `output_size`) then 3x3 pad-1 conv
  Timesteps(block_out_channels[0], flip_sin_to_cos=True, downscale_freq_shift=0)                                 unet_3d_condition.py:306-313
  TimestepEmbedding(block_out_channels[0], 4 * block_out_channels[0], act_fn="silu")
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F


class Timesteps(nn.Module):
    def __init__(s, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        s.n, s.flip, s.shift = num_channels, flip_sin_to_cos, downscale_freq_shift

    def forward(s, t):
        half = s.n // 2
        ex = -math.log(10000) * torch.arange(half, dtype=torch.float32, device=t.device) / (half - s.shift)
        emb = t[:, None].float() * torch.exp(ex)[None, :]
        emb = torch.cat([emb.sin(), emb.cos()], -1)
        if s.flip:
            emb = torch.cat([emb[:, half:], emb[:, :half]], -1)
        return emb

class TimestepEmbedding(nn.Module):
    def __init__(s, i, o, act_fn="silu"):
        super().__init__()
        s.linear_1, s.act, s.linear_2 = nn.Linear(i, o), nn.SiLU(), nn.Linear(o, o)

    def forward(s, x, cond=None):
        return s.linear_2(s.act(s.linear_1(x)))


class ResnetBlock2D(nn.Module):
    def __init__(s, *, in_channels, out_channels=None, temb_channels=512, eps=1e-6, groups=32, dropout=0.0,
                 time_embedding_norm="default", non_linearity="swish", output_scale_factor=1.0, pre_norm=True):
        super().__init__()
        out_channels = out_channels or in_channels
        s.norm1 = nn.GroupNorm(groups, in_channels, eps=eps)
        s.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        s.time_emb_proj = nn.Linear(temb_channels, out_channels)
        s.norm2 = nn.GroupNorm(groups, out_channels, eps=eps)
        s.dropout = nn.Dropout(dropout)
        s.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        s.nonlinearity, s.osf = nn.SiLU(), output_scale_factor
        s.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None

    def forward(s, x, temb):
        h = s.conv1(s.nonlinearity(s.norm1(x)))
        h = h + s.time_emb_proj(s.nonlinearity(temb))[:, :, None, None]
        h = s.conv2(s.dropout(s.nonlinearity(s.norm2(h))))
        if s.conv_shortcut is not None:
            x = s.conv_shortcut(x)
        return (x + h) / s.osf

class TemporalConvLayer(nn.Module):
    def __init__(s, in_dim, out_dim=None, dropout=0.0, norm_num_groups=32):
        super().__init__()
        out_dim = out_dim or in_dim
        s.conv1 = nn.Sequential(nn.GroupNorm(norm_num_groups, in_dim), nn.SiLU(), nn.Conv3d(in_dim, out_dim, (3, 1, 1), padding=(1, 0, 0)))
        mk = lambda: nn.Sequential(nn.GroupNorm(norm_num_groups, out_dim), nn.SiLU(), nn.Dropout(dropout), nn.Conv3d(out_dim, in_dim, (3, 1, 1), padding=(1, 0, 0)))
        s.conv2, s.conv3, s.conv4 = mk(), mk(), mk()
        nn.init.zeros_(s.conv4[-1].weight)
        nn.init.zeros_(s.conv4[-1].bias)

    def forward(s, h, num_frames=1):
        h = h[None, :].reshape((-1, num_frames) + h.shape[1:]).permute(0, 2, 1, 3, 4)
        idt = h
        h = idt + s.conv4(s.conv3(s.conv2(s.conv1(h))))
        return h.permute(0, 2, 1, 3, 4).reshape((h.shape[0] * h.shape[2], -1) + h.shape[3:])

class Downsample2D(nn.Module):
    def __init__(s, channels, use_conv=False, out_channels=None, padding=1, name="conv"):
        super().__init__()
        s.conv = nn.Conv2d(channels, out_channels or channels, 3, stride=2, padding=padding)

    def forward(s, x):
        return s.conv(x)

class Upsample2D(nn.Module):
    def __init__(s, channels, use_conv=False, out_channels=None):
        super().__init__()
        s.conv = nn.Conv2d(channels, out_channels or channels, 3, padding=1)

    def forward(s, x, output_size=None):
        x = F.interpolate(x, scale_factor=2.0, mode="nearest") if output_size is None else F.interpolate(x, size=output_size, mode="nearest")
        return s.conv(x)
