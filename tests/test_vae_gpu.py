"""VAE decode + tensor2vid on the HIP kernels vs the fp32 oracle restatement (oracle/vae_ref.py;
parity unpinned: diffusers is not vendored and the reference holds no fixture for it)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lvd_amd  # noqa: E402
from lvd_amd import ops  # noqa: E402
from lvd_amd.vae import HipVAEDecoder  # noqa: E402
from lvd_amd.weights import VAE_TINY, VAEConfig, synthetic_vae_state_dict, vae_decoder_param_shapes  # noqa: E402
from oracle import vae_ref  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    return ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-12)).item()


def test_softmax_rows_and_tokens_to_video():
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn(300, 2880, device=DEV, generator=g) * 4
    p = ops.softmax_rows(x)
    assert rel(p, x.softmax(-1)) < 4e-3
    assert (p.float().sum(-1) - 1).abs().max().item() < 2e-2
    t = (torch.randn(2 * 6 * 10, 4, device=DEV, generator=g) * 1.5).bfloat16()
    v = ops.tokens_to_video(t, 2, 6, 10)
    ref = (t[:, :3].float() / 2 + 0.5).clamp(0, 1).reshape(2, 6, 10, 3)
    assert torch.equal(v, ref)


@pytest.mark.parametrize("frames,h,w", [(2, 8, 8), (3, 4, 12)])
def test_vae_decode_matches_oracle_tiny(frames, h, w):
    cfg = VAEConfig(**VAE_TINY)
    sd = synthetic_vae_state_dict(cfg, seed=3)
    lat = torch.randn(1, 4, frames, h, w, generator=torch.Generator().manual_seed(5)) * cfg.scaling_factor * 4
    ref = vae_ref.decode_latents_to_video(sd, cfg, lat)
    dec = HipVAEDecoder(cfg, sd, device=DEV)
    vid = dec(lat.to(DEV))
    assert vid.shape == (1, frames, 8 * h, 8 * w, 3) and vid.dtype == torch.float32
    assert float(vid.min()) >= 0.0 and float(vid.max()) <= 1.0
    # bf16 storage through ~30 conv/norm layers against the fp32 oracle; the image is compared before the clamp as well
    tok, n, H, W = dec.decode_tokens(lat.to(DEV))
    img_ref = vae_ref.decode(sd, cfg, (lat / cfg.scaling_factor).permute(0, 2, 1, 3, 4).reshape(frames, 4, h, w))
    img = tok[:, :3].float().reshape(frames, H, W, 3).permute(0, 3, 1, 2).cpu()
    assert rel(img, img_ref) < 3e-2, rel(img, img_ref)
    assert (vid.cpu() - ref).abs().max().item() < 6e-2


def test_vae_state_dict_names_match_diffusers_layout():
    names = list(vae_decoder_param_shapes(VAEConfig()).keys())
    assert "decoder.mid_block.attentions.0.to_out.0.weight" in names and "decoder.up_blocks.2.resnets.0.conv_shortcut.weight" in names
    assert "decoder.up_blocks.3.upsamplers.0.conv.weight" not in names and "post_quant_conv.bias" in names
    assert sum(torch.Size(s).numel() for s in vae_decoder_param_shapes(VAEConfig()).values()) == 49490199


def test_vae_decode_full_size_finite():
    """The benchmark geometry: 24 frames of 40x72 latents -> 320x576 RGB."""
    cfg = VAEConfig()
    dec = HipVAEDecoder(cfg, synthetic_vae_state_dict(cfg, seed=0, device=DEV), device=DEV)
    lat = torch.randn(1, 4, 24, 40, 72, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)) * cfg.scaling_factor
    vid = dec(lat)
    assert vid.shape == (1, 24, 320, 576, 3)
    assert torch.isfinite(vid).all() and 0.05 < float(vid.mean()) < 0.95
