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
