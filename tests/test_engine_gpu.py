"""End-to-end parity of the HIP denoiser (through the C ABI) against the reference's golden outputs and the
fp32 oracle.  bf16 activations vs fp32 reference: tolerance rel-L2 <= 3e-2 on the noise prediction."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import lvd_amd  # noqa: E402
from lvd_amd.engine import HipUNet3D  # noqa: E402
from lvd_amd.weights import TINY, UNetConfig, synthetic_state_dict  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")
TOL = 3e-2


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_unet_tiny_vs_reference_golden():
    g = np.load(os.path.join(G, "unet_tiny.npz"))
    cfg = UNetConfig(**TINY)
    net = HipUNet3D(cfg, synthetic_state_dict(cfg, seed=0))
    out = net.forward(torch.from_numpy(g["sample"]).cuda(), int(g["timestep"]), torch.from_numpy(g["ehs"]).cuda())
    e = rel(out, g["out"])
    print("tiny unet rel-L2 vs reference:", e)
    assert e < TOL


def test_unet_tiny_gated_vs_reference_golden():
    g = np.load(os.path.join(G, "unet_tiny_gated.npz"))
    cfg = UNetConfig(attention_type="gated", **TINY)
    net = HipUNet3D(cfg, synthetic_state_dict(cfg, seed=1))
    gl = {k: torch.from_numpy(g[k]) for k in ("boxes", "masks", "positive_embeddings")}
    out = net.forward(torch.from_numpy(g["sample"]).cuda(), int(g["timestep"]), torch.from_numpy(g["ehs"]).cuda(), gligen=gl)
    e = rel(out, g["out"])
    print("tiny gated unet rel-L2 vs reference:", e)
    assert e < TOL


def test_unet_odd_batch_and_frames_vs_oracle():
    from oracle import unet_ref
    cfg = UNetConfig(block_out_channels=(64, 128, 192, 192), layers_per_block=2, cross_attention_dim=128, attention_head_dim=64)
    sd = synthetic_state_dict(cfg, seed=3)
    gen = torch.Generator().manual_seed(5)
    sample = torch.randn(1, 4, 6, 24, 8, generator=gen)
    ehs = torch.randn(1, 77, 128, generator=gen)
    with torch.no_grad():
        ref = unet_ref.unet_forward(sd, cfg, sample, 37, ehs)
    net = HipUNet3D(cfg, sd)
    out = net.forward(sample.cuda(), 37, ehs.cuda())
    e = rel(out, ref)
    print("2-layer unet rel-L2 vs oracle:", e)
    assert e < TOL


def test_unet_latent_size_not_divisible_by_8_vs_oracle():
    """BASELINE configs[0] territory (256x144 -> latent 18x32): the up path must resize to the skip connection's size
    (18 -> 9 -> 5 -> 3 and back 3 -> 5 -> 9 -> 18), like the reference's `upsample_size` (unet_3d_condition.py:711-730)."""
    from oracle import unet_ref
    cfg = UNetConfig(**TINY)
    sd = synthetic_state_dict(cfg, seed=4)
    gen = torch.Generator().manual_seed(6)
    sample = torch.randn(2, 4, 3, 18, 20, generator=gen)
    ehs = torch.randn(2, 77, cfg.cross_attention_dim, generator=gen)
    with torch.no_grad():
        ref = unet_ref.unet_forward(sd, cfg, sample, 123, ehs)
    out = HipUNet3D(cfg, sd).forward(sample.cuda(), 123, ehs.cuda())
    e = rel(out, ref)
    print("18x20 latent rel-L2 vs oracle:", e)
    assert out.shape == ref.shape and e < TOL


def test_resized_upsample_conv_backward_vs_autograd():
    """Input gradient of the size-explicit Upsample2D (nearest to 5x9 from 3x5, then conv3x3) against torch autograd."""
    import torch.nn.functional as F
    from lvd_amd.engine import Geom, Tape
    cfg = UNetConfig(**TINY)
    sd = synthetic_state_dict(cfg, seed=4)
    net = HipUNet3D(cfg, sd)
    name = "up_blocks.0.upsamplers.0.conv"
    C = sd[name + ".weight"].shape[1]
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(2 * 3 * 5, C, generator=gen).bfloat16()
    dy = torch.randn(2 * 5 * 9, sd[name + ".weight"].shape[0], generator=gen).bfloat16()
    xr = x.float().reshape(2, 3, 5, C).permute(0, 3, 1, 2).requires_grad_(True)
    yr = F.conv2d(F.interpolate(xr, size=(5, 9), mode="nearest"), sd[name + ".weight"], sd[name + ".bias"], padding=1)
    yr.backward(dy.float().reshape(2, 5, 9, -1).permute(0, 3, 1, 2))
    tape = Tape()
    xg = x.cuda()
    out, g = net._conv3x3(xg, name, Geom(1, 2, 3, 5), tape=tape, upsample=1, up_to=(5, 9))
    assert (g.H, g.W) == (5, 9) and rel(out, yr.detach().permute(0, 2, 3, 1).reshape(-1, out.shape[1])) < 1e-2
    tape.accumulate(out, dy.cuda())
    tape.backward()
    assert rel(tape.pop(xg), xr.grad.permute(0, 2, 3, 1).reshape(-1, C)) < 1e-2


def test_cfg_shared_prefix_equals_the_duplicated_batch():
    """forward(cfg_pairs=True) — the layers in front of the first text-dependent one once per sample, rows duplicated there — against the
    reference's form, a forward of torch.cat([latents] * 2): the same function (the two batch items are identical until attn2 of the first
    spatial transformer reads the text), so the only difference is the bf16 rounding of tile geometries picked for half the rows.  Plain and
    gated (GLIGEN fusers on: the split sits in front of the fuser) topologies, one and two samples."""
    from lvd_amd.weights import TINY
    for gated in (False, True):
        cfg = UNetConfig(attention_type="gated" if gated else "default", **TINY)
        net = HipUNet3D(cfg, synthetic_state_dict(cfg, seed=0))
        gen = torch.Generator().manual_seed(3)
        for V in (1, 2):
            Fr = 4
            lat = torch.randn(V, 4, Fr, 16, 16, generator=gen).cuda()
            ehs = torch.randn(2 * V, 77, cfg.cross_attention_dim, generator=gen).cuda()
            text = net.encode_text(ehs)
            gl = None
            if gated:
                gl = {"boxes": torch.rand(2 * V * Fr, 30, 4, generator=gen), "masks": (torch.rand(2 * V * Fr, 30, generator=gen) > 0.5).float(),
                      "positive_embeddings": torch.randn(2 * V * Fr, 30, cfg.cross_attention_dim, generator=gen)}
            full = net.forward(lat.repeat_interleave(2, 0).contiguous(), 500, text=text, gligen=gl)
            shared = net.forward(lat, 500, text=text, gligen=gl, cfg_pairs=True)
            assert shared.shape == full.shape == (2 * V, 4, Fr, 16, 16)
            err = ((shared - full).norm() / full.norm()).item()
            print(f"gated={gated} V={V}: shared-prefix vs duplicated batch rel-L2 {err:.2e}")
            # NOT bit-equal, at any V (tests/probes/shared_prefix_probe.py: every element differs): the prefix runs at half the rows, i.e. with
            # other tile geometries, K-split plans and GroupNorm chunkings, and those bf16 roundings pass through the rest of the random-weight
            # network.  Measured 1.9e-2 (V = 1) ... 2.2e-2 (V = 2, gated) on this topology — two bf16 runs of one function, about sqrt(2) x the
            # distance of either from the fp32 oracle (1.7e-2) — hence 4e-2 = measured x 2; the oracle comparison of the shared-prefix path itself
            # is tests/test_full_topology_gpu.py::test_forward_cfg_shared_prefix_vs_oracle_of_the_duplicated_batch (3e-2)
            assert err < 4e-2, err
            assert ((shared[0] - shared[1]).norm() / shared.norm()).item() > 1e-3  # the halves really differ (other text)
            net.cfg_shared_prefix = False
            assert torch.equal(net.forward_cfg(lat, 500, text=text, gligen=gl), full)  # the knob: exactly the duplicated batch
            net.cfg_shared_prefix = True
        # pairing order: two copies of ONE sample with the same (uncond, cond) texts must give the same pair twice
        lat1 = torch.randn(1, 4, 4, 16, 16, generator=gen).cuda()
        ehs1 = torch.randn(2, 77, cfg.cross_attention_dim, generator=gen).cuda()
        twice = net.forward(torch.cat([lat1, lat1]), 500, text=net.encode_text(torch.cat([ehs1, ehs1])), cfg_pairs=True)
        assert ((twice[0:2] - twice[2:4]).norm() / twice[0:2].norm()).item() < 1e-3
        once = net.forward(lat1, 500, text=net.encode_text(ehs1), cfg_pairs=True)
        assert ((twice[0:2] - once).norm() / once.norm()).item() < 4e-2
