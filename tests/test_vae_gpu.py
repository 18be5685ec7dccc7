"""VAE decode + tensor2vid on the HIP kernels vs the fp32 oracle restatement (oracle/vae_ref.py;
parity unpinned: diffusers is not vendored and the reference holds no fixture for it)."""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lvd_amd  # noqa: E402
from lvd_amd import ops  # noqa: E402
from lvd_amd.vae import HipVAEDecoder, HipVAEEncoder  # noqa: E402
from lvd_amd.weights import VAE_TINY, VAEConfig, synthetic_vae_state_dict, vae_decoder_param_shapes, vae_encoder_param_shapes  # noqa: E402
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
    wide = torch.randn(40, 9216, device=DEV, generator=g) * 4  # 72x128 tokens: the block-per-row kernel
    pw = ops.softmax_rows(wide)
    assert rel(pw, wide.softmax(-1)) < 4e-3 and (pw.float().sum(-1) - 1).abs().max().item() < 2e-2


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


def test_lanczos_resize_to_token_matrix_matches_pillow():
    """`prepare_init_upsampled` + `preprocess_video` (upsample.py:15-28): Lanczos to a rectangular size, then 2*(v/255)-1, as the
    8-wide token matrix of the encoder's conv_in."""
    import numpy as np
    from PIL import Image
    frames = np.random.RandomState(0).randint(0, 256, (2, 40, 72, 3)).astype(np.uint8)
    tok, resized = ops.frames_to_patches(torch.from_numpy(frames).to(DEV), (72, 128), 1, (0.5,) * 3, (0.5,) * 3, return_resized=True,
                                         kind="lanczos", width=8)
    ref = np.stack([np.asarray(Image.fromarray(f).resize((128, 72), Image.LANCZOS)) for f in frames])
    assert np.array_equal(resized.cpu().numpy(), ref)
    want = torch.from_numpy(ref).float().reshape(-1, 3) * (2 / 255.0) - 1
    assert tok.shape == (2 * 72 * 128, 8) and tok[:, 3:].abs().max() == 0
    assert (tok[:, :3].float().cpu() - want).abs().max().item() < 8e-3  # bf16 rounding of values in [-1, 1]


@pytest.mark.parametrize("size", [None, (64, 96)])
def test_vae_encode_matches_oracle_tiny(size):
    import numpy as np
    from PIL import Image
    cfg = VAEConfig(**VAE_TINY)
    sd = synthetic_vae_state_dict(cfg, seed=4, encoder=True)
    rng = np.random.RandomState(1)
    frames = (np.kron(rng.randint(0, 256, (3, 4, 6, 3)), np.ones((1, 8, 8, 1))) * 0.7 + rng.randint(0, 77, (3, 32, 48, 3))).astype(np.uint8)
    src = frames if size is None else np.stack([np.asarray(Image.fromarray(f).resize(size[::-1], Image.LANCZOS)) for f in frames])
    h, w = src.shape[1] // 8, src.shape[2] // 8
    eps = torch.randn(3, 4, h, w, generator=torch.Generator().manual_seed(2))
    ref = vae_ref.encode_video(sd, cfg, src, eps)
    enc = HipVAEEncoder(cfg, sd, device=DEV)
    lat = enc.encode(torch.from_numpy(frames), eps=eps, size=size)
    assert lat.shape == ref.shape == (1, 4, 3, h, w)
    assert rel(lat.cpu(), ref) < 3e-2, rel(lat.cpu(), ref)
    # and back through the decoder: the round trip is a property of the pair, checked against the oracle pair
    vid = HipVAEDecoder(cfg, sd, device=DEV)(lat)
    assert (vid.cpu() - vae_ref.decode_latents_to_video(sd, cfg, ref)).abs().max().item() < 8e-2


def test_vae_encoder_state_dict_layout():
    names = vae_encoder_param_shapes(VAEConfig())
    assert "encoder.down_blocks.2.downsamplers.0.conv.weight" in names and "encoder.down_blocks.3.downsamplers.0.conv.weight" not in names
    assert names["encoder.conv_out.weight"] == (8, 512, 3, 3) and names["quant_conv.weight"] == (8, 8, 1, 1)
    assert sum(torch.Size(s).numel() for s in names.values()) == 34163664  # the SD VAE encoder
