"""ORACLE (test infrastructure only) — fp32 PyTorch restatement of the reference denoiser.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product
path (lvd_amd.*) never does.  It restates, functionally and on the reference's own state_dict names,
  UNet3DConditionModel.forward              /root/reference/models/unet_3d_condition.py:642-859
  CrossAttn{Down,Up}Block3D / mid / plain   /root/reference/models/unet_3d_blocks.py:255-291,396-437,505-520,619-662,724-745
  Transformer2DModel.forward                /root/reference/models/transformer_2d.py:310-370
  TransformerTemporalModel.forward          /root/reference/models/transformer_temporal.py:139-184
  BasicTransformerBlock / GEGLU / fuser     /root/reference/models/attention.py:44-60,179-276,355-376
  AttnProcessor (fast and probs-saving)     /root/reference/models/attention_processor.py:344-430,476-589
  PositionNet / FourierEmbedder             /root/reference/models/unet_3d_condition.py:47-179
and the un-vendored diffusers==0.27.2 pieces (ResnetBlock2D, TemporalConvLayer, Downsample2D, Upsample2D,
Timesteps, TimestepEmbedding) from their public definitions — PARITY UNPINNED for
those six classes: nothing under /root/reference tests them.  Pinned against the shimmed reference import
by tests/golden/*.npz (generator: oracle/make_golden.py).
"""
import math

import torch
import torch.nn.functional as F


PROBS_BYTES_PER_CHUNK = 1 << 30  # attention() materialises at most this many bytes of probabilities at a time when nobody asks for them


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def _gn(sd, name, x, groups, eps):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)


def _ln(sd, name, x):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], 1e-5)


def timestep_embedding(t, dim):
    """diffusers Timesteps(dim, flip_sin_to_cos=True, downscale_freq_shift=0): [cos | sin]."""
    half = dim // 2
    freq = torch.exp(-math.log(10000.0) * torch.arange(half, dtype=torch.float32, device=t.device) / half)
    ang = t.float()[:, None] * freq[None]
    return torch.cat([ang.cos(), ang.sin()], -1)


def attention(sd, name, x, ctx, heads, on_probs=None):
    """softmax(q k^T / sqrt(d)) v with to_q/k/v (no bias) and to_out.0 (bias); attention_processor.py:382-417,515-536."""
    q, k, v = _lin(sd, name + ".to_q", x), _lin(sd, name + ".to_k", ctx), _lin(sd, name + ".to_v", ctx)
    b, lq, c = q.shape
    d = c // heads
    sp = lambda t: t.reshape(b, -1, heads, d).permute(0, 2, 1, 3)
    q, k, v = sp(q), sp(k), sp(v)
    step = max(1, PROBS_BYTES_PER_CHUNK // (heads * lq * k.shape[2] * 4))
    if on_probs is None and not torch.is_grad_enabled() and step < b:
        # same arithmetic, batch rows taken a few at a time: the 576x320x24 forward would hold 8 GB of probabilities per level-0 layer
        o = torch.cat([(q[i:i + step] @ k[i:i + step].transpose(-1, -2) * d**-0.5).softmax(-1) @ v[i:i + step] for i in range(0, b, step)])
    else:
        probs = (q @ k.transpose(-1, -2) * d**-0.5).softmax(-1)
        if on_probs is not None:
            on_probs(probs)  # (batch, heads, queries, keys)
        o = probs @ v
    o = o.permute(0, 2, 1, 3).reshape(b, lq, c)
    return _lin(sd, name + ".to_out.0", o)


def feed_forward(sd, name, x):
    h, g = _lin(sd, name + ".net.0.proj", x).chunk(2, -1)
    return _lin(sd, name + ".net.2", h * F.gelu(g))


def fuser(sd, name, x, objs, heads):
    """GatedSelfAttentionDense.forward, models/attention.py:44-60."""
    n_vis = x.shape[1]
    o = _lin(sd, name + ".linear", objs)
    h = _ln(sd, name + ".norm1", torch.cat([x, o], 1))
    x = x + sd[name + ".alpha_attn"].tanh() * attention(sd, name + ".attn", h, h, heads)[:, :n_vis]
    x = x + sd[name + ".alpha_dense"].tanh() * feed_forward(sd, name + ".ff", _ln(sd, name + ".norm2", x))
    return x


def transformer_block(sd, name, x, ctx, heads, objs=None, on_probs=None):
    """BasicTransformerBlock.forward, models/attention.py:179-276 (layer_norm variant)."""
    h = _ln(sd, name + ".norm1", x)
    x = attention(sd, name + ".attn1", h, h, heads) + x
    if objs is not None and (name + ".fuser.linear.weight") in sd:
        x = fuser(sd, name + ".fuser", x, objs, heads)
    h = _ln(sd, name + ".norm2", x)
    x = attention(sd, name + ".attn2", h, h if ctx is None else ctx, heads, on_probs) + x
    x = feed_forward(sd, name + ".ff", _ln(sd, name + ".norm3", x)) + x
    return x


def transformer_2d(sd, name, x, ctx, heads, groups, objs=None, on_probs=None):
    n, c, hh, ww = x.shape
    h = _gn(sd, name + ".norm", x, groups, 1e-6).permute(0, 2, 3, 1).reshape(n, hh * ww, c)
    h = _lin(sd, name + ".proj_in", h)
    h = transformer_block(sd, name + ".transformer_blocks.0", h, ctx, heads, objs, on_probs)
    h = _lin(sd, name + ".proj_out", h)
    return h.reshape(n, hh, ww, c).permute(0, 3, 1, 2) + x


def transformer_temporal(sd, name, x, frames, heads, groups):
    bf, c, hh, ww = x.shape
    b = bf // frames
    h = x.reshape(b, frames, c, hh, ww).permute(0, 2, 1, 3, 4)
    h = _gn(sd, name + ".norm", h, groups, 1e-6)  # statistics over (C/g, F, H, W)
    h = h.permute(0, 3, 4, 2, 1).reshape(b * hh * ww, frames, c)
    h = _lin(sd, name + ".proj_in", h)
    h = transformer_block(sd, name + ".transformer_blocks.0", h, None, heads)
    h = _lin(sd, name + ".proj_out", h)
    h = h.reshape(b, hh, ww, frames, c).permute(0, 3, 4, 1, 2).reshape(bf, c, hh, ww)
    return h + x


def resnet(sd, name, x, temb, groups, eps):
    h = F.conv2d(F.silu(_gn(sd, name + ".norm1", x, groups, eps)), sd[name + ".conv1.weight"], sd[name + ".conv1.bias"], padding=1)
    h = h + _lin(sd, name + ".time_emb_proj", F.silu(temb))[:, :, None, None]
    h = F.conv2d(F.silu(_gn(sd, name + ".norm2", h, groups, eps)), sd[name + ".conv2.weight"], sd[name + ".conv2.bias"], padding=1)
    if (name + ".conv_shortcut.weight") in sd:
        x = F.conv2d(x, sd[name + ".conv_shortcut.weight"], sd[name + ".conv_shortcut.bias"])
    return x + h


def temporal_conv(sd, name, x, frames, groups):
    bf, c, hh, ww = x.shape
    h = x.reshape(bf // frames, frames, c, hh, ww).permute(0, 2, 1, 3, 4)
    idt = h
    for k, ci in (("conv1", 2), ("conv2", 3), ("conv3", 3), ("conv4", 3)):
        h = F.silu(_gn(sd, f"{name}.{k}.0", h, groups, 1e-5))
        h = F.conv3d(h, sd[f"{name}.{k}.{ci}.weight"], sd[f"{name}.{k}.{ci}.bias"], padding=(1, 0, 0))
    h = idt + h
    return h.permute(0, 2, 1, 3, 4).reshape(bf, c, hh, ww)


def position_net(sd, boxes, masks, positive_embeddings):
    m = masks.unsqueeze(-1)
    freq = 100.0 ** (torch.arange(8, dtype=torch.float32, device=boxes.device) / 8)
    e = freq[None, None, None] * boxes.unsqueeze(-1)  # (B,N,4,8)
    xyxy = torch.stack((e.sin(), e.cos()), -1).permute(0, 1, 3, 4, 2).reshape(*boxes.shape[:2], -1)
    xyxy = xyxy * m + (1 - m) * sd["position_net.null_position_feature"].view(1, 1, -1)
    pos = positive_embeddings * m + (1 - m) * sd["position_net.null_positive_feature"].view(1, 1, -1)
    h = torch.cat([pos, xyxy], -1)
    h = F.silu(_lin(sd, "position_net.linears.0", h))
    h = F.silu(_lin(sd, "position_net.linears.2", h))
    return _lin(sd, "position_net.linears.4", h)


def unet_forward(sd, cfg, sample, timestep, encoder_hidden_states, save_attn_to_dict=None, save_keys=None,
                 gligen=None, fuser_enabled=True, stop_after_key=None):
    """sample (B,4,F,h,w) fp32 -> (B,4,F,h,w).  `cfg` is an lvd_amd.weights.UNetConfig (plain dataclass).

    save_attn_to_dict / save_keys follow AttnProcessor.__call__ (attention_processor.py:459-474,553-586):
    probs (B·F, heads, HW, 77) are stored under the key tuple when the key is listed.
    gligen = dict(boxes, positive_embeddings, masks) as the pipeline builds it
    (controllable_pipeline_text_to_video_synth.py:736-814), already flattened to (B·F, 30, ·).
    """
    boc = cfg.block_out_channels
    groups, eps, dh = cfg.norm_num_groups, cfg.norm_eps, cfg.attention_head_dim
    b, _, frames, hh, ww = sample.shape
    t = timestep if torch.is_tensor(timestep) else torch.tensor([timestep], device=sample.device)
    t = t.reshape(-1).expand(b)
    emb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", timestep_embedding(t, boc[0]))))
    emb = emb.repeat_interleave(frames, 0)
    ctx = encoder_hidden_states.repeat_interleave(frames, 0)
    x = sample.permute(0, 2, 1, 3, 4).reshape(b * frames, -1, hh, ww)
    x = F.conv2d(x, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    x = transformer_temporal(sd, "transformer_in", x, frames, cfg.transformer_in_heads, groups)
    objs = None
    if gligen is not None and fuser_enabled:
        objs = position_net(sd, gligen["boxes"], gligen["masks"], gligen["positive_embeddings"])

    class _Stop(Exception):
        pass

    def saver(key):
        if save_attn_to_dict is None or (save_keys is not None and key not in save_keys):
            return None

        def f(probs):
            if save_keys is not None:
                save_attn_to_dict[key] = probs
            if stop_after_key is not None and key == stop_after_key:
                raise _Stop()
        return f

    def layer(prefix, j, key, x, has_attn, c):
        x = resnet(sd, f"{prefix}.resnets.{j}", x, emb, groups, eps)
        x = temporal_conv(sd, f"{prefix}.temp_convs.{j}", x, frames, groups)
        if has_attn:
            x = transformer_2d(sd, f"{prefix}.attentions.{j}", x, ctx, c // dh, groups, objs, saver(key))
            x = transformer_temporal(sd, f"{prefix}.temp_attentions.{j}", x, frames, c // dh, groups)
        return x

    try:
        skips = [x]
        for i, btype in enumerate(cfg.down_block_types):
            c = boc[i]
            for j in range(cfg.layers_per_block):
                x = layer(f"down_blocks.{i}", j, ("down", i, j, 0), x, btype == "CrossAttnDownBlock3D", c)
                skips.append(x)
            if i != len(boc) - 1:
                x = F.conv2d(x, sd[f"down_blocks.{i}.downsamplers.0.conv.weight"], sd[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
                skips.append(x)
        c = boc[-1]
        x = resnet(sd, "mid_block.resnets.0", x, emb, groups, eps)
        x = temporal_conv(sd, "mid_block.temp_convs.0", x, frames, groups)
        x = transformer_2d(sd, "mid_block.attentions.0", x, ctx, c // dh, groups, objs, saver(("mid", 0, 0, 0)))
        x = transformer_temporal(sd, "mid_block.temp_attentions.0", x, frames, c // dh, groups)
        x = resnet(sd, "mid_block.resnets.1", x, emb, groups, eps)
        x = temporal_conv(sd, "mid_block.temp_convs.1", x, frames, groups)
        rev = list(reversed(boc))
        for i, btype in enumerate(cfg.up_block_types):
            c = rev[i]
            for j in range(cfg.layers_per_block + 1):
                x = torch.cat([x, skips.pop()], 1)
                x = layer(f"up_blocks.{i}", j, ("up", i, j, 0), x, btype == "CrossAttnUpBlock3D", c)
            if i != len(boc) - 1:
                size = skips[-1].shape[2:]
                x = F.interpolate(x, size=size, mode="nearest") if tuple(size) != (x.shape[2] * 2, x.shape[3] * 2) else F.interpolate(x, scale_factor=2.0, mode="nearest")
                x = F.conv2d(x, sd[f"up_blocks.{i}.upsamplers.0.conv.weight"], sd[f"up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    except _Stop:
        return None
    x = F.silu(_gn(sd, "conv_norm_out", x, groups, eps))
    x = F.conv2d(x, sd["conv_out.weight"], sd["conv_out.bias"], padding=1)
    return x.reshape(b, frames, -1, hh, ww).permute(0, 2, 1, 3, 4)
