"""ORACLE (test infrastructure only) — fp32 PyTorch restatement of the VAE decode + tensor2vid step that follows the
denoising loop.

Only tests/ and __graft_entry__.smoke() may import this file; the product path (lvd_amd.*) never does.
The call site is in the reference (`decode_latents`, /root/reference/models/controllable_pipeline_text_to_video_synth.py:374-400;
`tensor2vid` :66-88) but the arithmetic lives in the un-vendored `diffusers==0.27.2` (`requirements.txt:5`):
`AutoencoderKL.decode` = `post_quant_conv` (1x1) -> `Decoder` (conv_in, UNetMidBlock2D with one single-head attention of
dim 512, four UpDecoderBlock2D of 3 ResnetBlock2D each with nearest-x2 Upsample2D between them, GroupNorm(32, eps 1e-6) +
SiLU + conv_out) and `VaeImageProcessor.postprocess` (x/2 + 0.5 clamped to [0,1], NCHW -> NHWC).  Restated here from the
public 0.27.2 definitions on the diffusers state_dict names.  PARITY UNPINNED: nothing under /root/reference holds a
fixture for it and the package is not installed in this image.
"""
import torch
import torch.nn.functional as F

EPS = 1e-6
GROUPS = 32


def _gn(sd, name, x):
    return F.group_norm(x, GROUPS, sd[name + ".weight"], sd[name + ".bias"], EPS)


def _conv(sd, name, x, pad):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=pad)


def resnet(sd, name, x):
    """ResnetBlock2D(temb_channels=None, eps=1e-6, output_scale_factor=1): norm-silu-conv twice + (1x1 shortcut) skip."""
    h = _conv(sd, name + ".conv1", F.silu(_gn(sd, name + ".norm1", x)), 1)
    h = _conv(sd, name + ".conv2", F.silu(_gn(sd, name + ".norm2", h)), 1)
    if name + ".conv_shortcut.weight" in sd:
        x = _conv(sd, name + ".conv_shortcut", x, 0)
    return x + h


def mid_attention(sd, name, x):
    """Attention(heads=1, dim_head=C, residual_connection=True, norm_num_groups=32, bias=True) on (B, C, H, W)."""
    b, c, h, w = x.shape
    t = _gn(sd, name + ".group_norm", x.reshape(b, c, h * w)).transpose(1, 2)  # (B, HW, C)
    q = F.linear(t, sd[name + ".to_q.weight"], sd[name + ".to_q.bias"])
    k = F.linear(t, sd[name + ".to_k.weight"], sd[name + ".to_k.bias"])
    v = F.linear(t, sd[name + ".to_v.weight"], sd[name + ".to_v.bias"])
    p = (q @ k.transpose(1, 2) * c**-0.5).softmax(-1)
    o = F.linear(p @ v, sd[name + ".to_out.0.weight"], sd[name + ".to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(b, c, h, w)


def decode(sd, cfg, z):
    """AutoencoderKL.decode(z).sample for z (N, 4, h, w) -> (N, 3, 8h, 8w)."""
    x = _conv(sd, "post_quant_conv", z, 0)
    x = _conv(sd, "decoder.conv_in", x, 1)
    x = resnet(sd, "decoder.mid_block.resnets.0", x)
    x = mid_attention(sd, "decoder.mid_block.attentions.0", x)
    x = resnet(sd, "decoder.mid_block.resnets.1", x)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            x = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x)
        if i != nb - 1:
            x = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", F.interpolate(x, scale_factor=2.0, mode="nearest"), 1)
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x))
    return _conv(sd, "decoder.conv_out", x, 1)


def decode_latents_to_video(sd, cfg, latents):
    """decode_latents + tensor2vid(output_type="np"): (B, 4, F, h, w) -> (B, F, 8h, 8w, 3) float32 in [0, 1]."""
    b, c, f, h, w = latents.shape
    z = (latents / cfg.scaling_factor).permute(0, 2, 1, 3, 4).reshape(b * f, c, h, w)
    img = decode(sd, cfg, z.float())
    img = (img / 2 + 0.5).clamp(0, 1)
    return img.reshape(b, f, 3, img.shape[-2], img.shape[-1]).permute(0, 1, 3, 4, 2).contiguous()


def encode_moments(sd, cfg, x):
    """AutoencoderKL.encode(x).latent_dist parameters for x (N, 3, H, W) in [-1, 1] -> (mean, logvar), each (N, 4, H/8, W/8).
    diffusers 0.27.2 Encoder: conv_in, DownEncoderBlock2D x4 (2 ResnetBlock2D each; Downsample2D(padding=0) = pad (0,1,0,1) then a
    stride-2 3x3 conv, on all but the last), UNetMidBlock2D, GroupNorm + SiLU + conv_out to 2*latent channels, quant_conv (1x1);
    DiagonalGaussianDistribution clamps logvar to [-30, 20].  PARITY UNPINNED like the decoder."""
    x = _conv(sd, "encoder.conv_in", x, 1)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            x = resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", x)
        if i != nb - 1:
            n = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), sd[n + ".weight"], sd[n + ".bias"], stride=2)
    x = resnet(sd, "encoder.mid_block.resnets.0", x)
    x = mid_attention(sd, "encoder.mid_block.attentions.0", x)
    x = resnet(sd, "encoder.mid_block.resnets.1", x)
    x = _conv(sd, "encoder.conv_out", F.silu(_gn(sd, "encoder.conv_norm_out", x)), 1)
    mean, logvar = _conv(sd, "quant_conv", x, 0).chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def encode_video(sd, cfg, frames_uint8, eps):
    """VideoToVideoSDPipeline.prepare_latents up to the noise: frames (F, H, W, 3) uint8 -> 2*(v/255)-1 -> latent_dist.sample
    with the given standard-normal `eps` (F, 4, h, w) -> scaling_factor * z as (1, 4, F, h, w)."""
    x = (2.0 * (torch.as_tensor(frames_uint8).float() / 255.0) - 1.0).permute(0, 3, 1, 2)
    mean, logvar = encode_moments(sd, cfg, x)
    z = mean + torch.exp(0.5 * logvar) * eps
    return (cfg.scaling_factor * z).permute(1, 0, 2, 3).unsqueeze(0)
