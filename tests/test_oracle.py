"""CPU tests: the oracle (oracle/*.py, a restatement) against golden vectors produced by importing the real
reference (tests/golden/*.npz, generator oracle/make_golden.py).  fp32 vs fp32: tolerance 2e-4 relative."""
import os

import numpy as np
import pytest
import torch

import lvd_amd  # noqa: F401
from lvd_amd.weights import TINY, UNetConfig, synthetic_state_dict, unet_param_shapes
from oracle import guidance_ref, scheduler_ref, unet_ref

G = os.path.join(os.path.dirname(__file__), "golden")


def rel(a, b):
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def test_param_names_match_reference():
    g = np.load(os.path.join(G, "unet_tiny.npz"))
    shapes = unet_param_shapes(UNetConfig(**TINY))
    assert list(shapes.keys()) == list(g["param_names"])
    numel = [int(np.prod(s)) if len(s) else 1 for s in shapes.values()]
    assert numel == list(g["param_numel"])


def test_zeroscope_param_count():
    # default (zeroscope) topology = 1411.2 M parameters, 1623.8 M with GLIGEN fusers
    n = sum(int(np.prod(s)) if len(s) else 1 for s in unet_param_shapes(UNetConfig()).values())
    ng = sum(int(np.prod(s)) if len(s) else 1 for s in unet_param_shapes(UNetConfig(attention_type="gated")).values())
    assert abs(n / 1e6 - 1411.2) < 0.5 and abs(ng / 1e6 - 1623.8) < 0.5, (n, ng)


def test_unet_forward_matches_reference():
    g = np.load(os.path.join(G, "unet_tiny.npz"))
    cfg = UNetConfig(**TINY)
    sd = synthetic_state_dict(cfg, seed=0)
    saved = {}
    keys = [("down", 1, 0, 0), ("down", 2, 0, 0), ("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 2, 1, 0)]
    with torch.no_grad():
        out = unet_ref.unet_forward(sd, cfg, torch.from_numpy(g["sample"]), int(g["timestep"]), torch.from_numpy(g["ehs"]),
                                    save_attn_to_dict=saved, save_keys=keys)
    assert rel(out, g["out"]) < 2e-4
    for k in keys:
        assert rel(saved[k], g["attn_" + "_".join(map(str, k))]) < 2e-4, k


def test_unet_gated_matches_reference():
    g = np.load(os.path.join(G, "unet_tiny_gated.npz"))
    cfg = UNetConfig(attention_type="gated", **TINY)
    sd = synthetic_state_dict(cfg, seed=1)
    gl = {k: torch.from_numpy(g[k]) for k in ("boxes", "masks", "positive_embeddings")}
    with torch.no_grad():
        out = unet_ref.unet_forward(sd, cfg, torch.from_numpy(g["sample"]), int(g["timestep"]), torch.from_numpy(g["ehs"]), gligen=gl)
        out_off = unet_ref.unet_forward(sd, cfg, torch.from_numpy(g["sample"]), int(g["timestep"]), torch.from_numpy(g["ehs"]), gligen=gl, fuser_enabled=False)
    assert rel(out, g["out"]) < 2e-4
    assert rel(out_off, g["out"]) > 1e-2  # the fusers really contribute (non-zero alpha in the synthetic weights)


@pytest.mark.parametrize("case,kw", [
    ("topk", dict(fg_top_p=0.75, bg_top_p=0.75, fg_weight=1.0, bg_weight=4.0, com_loss_scale=0.0)),
    ("com", dict(fg_top_p=0.25, bg_top_p=0.25, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03)),
])
def test_guidance_loss_matches_reference(case, kw):
    g = np.load(os.path.join(G, "guidance_loss.npz"))
    maps = {("down", 1, 0, 0): torch.from_numpy(g["maps_0"])[0].requires_grad_(True), ("up", 1, 1, 0): torch.from_numpy(g["maps_1"])[0].requires_grad_(True)}
    bboxes = g["bboxes"].tolist()
    loss = guidance_ref.compute_ca_loss(maps, bboxes, [[2, 3], [6]], list(maps.keys()), (8, 12), **kw)
    assert abs(loss.item() - float(g[f"loss_{case}"])) < 1e-4 * abs(float(g[f"loss_{case}"]))
    g0, g1 = torch.autograd.grad(loss, list(maps.values()))
    assert rel(g0, g[f"grad0_{case}"][0]) < 1e-4 and rel(g1, g[f"grad1_{case}"][0]) < 1e-4


LOSS_VARIANTS = {  # the cases of oracle/make_golden.py section (c), second box set
    "ratio": dict(use_ratio_based_loss=True, com_loss_scale=0.02),
    "sync": dict(fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, attn_sync_weight=3.0),
    "boxdiff": dict(fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, boxdiff_loss_scale=0.7, boxdiff_normed=True),
    "boxdiff_sum": dict(fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, boxdiff_loss_scale=0.05, boxdiff_normed=False, boxdiff_L=2),
    "all": dict(fg_top_p=0.3, bg_top_p=0.6, fg_weight=1.5, bg_weight=2.5, attn_sync_weight=1.0, boxdiff_loss_scale=0.4, com_loss_scale=0.03),
    "ce": dict(use_max_based_loss=False, use_ce_based_loss=True, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0),
    "ce_com": dict(use_max_based_loss=False, use_ce_based_loss=True, fg_top_p=0.25, bg_top_p=0.4, fg_weight=1.5, bg_weight=0.5, com_loss_scale=0.03),
    "smooth": dict(smooth_attn=True, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0),
    "renorm": dict(attn_renorm=True, num_tokens=9, renorm_scale=2.0, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03),
    "smooth_renorm": dict(smooth_attn=True, attn_renorm=True, num_tokens=8, renorm_scale=3.0, fg_top_p=0.3, bg_top_p=0.4, fg_weight=1.5, bg_weight=1.0,
                          boxdiff_loss_scale=0.4),
}


@pytest.mark.parametrize("case", sorted(LOSS_VARIANTS))
def test_optional_loss_terms_match_reference(case):
    """Ratio-based energy, attention sync, BoxDiff corner constraint (utils/guidance.py:312-323, 401-465): the oracle's restatement
    against the reference's own loss value and autograd gradients."""
    g = np.load(os.path.join(G, "guidance_loss.npz"))
    maps = {("down", 1, 0, 0): torch.from_numpy(g["maps_0"])[0].requires_grad_(True), ("up", 1, 1, 0): torch.from_numpy(g["maps_1"])[0].requires_grad_(True)}
    loss = guidance_ref.compute_ca_loss(maps, g["bboxes2"].tolist(), [[2, 3], [6]], list(maps.keys()), (8, 12), **LOSS_VARIANTS[case])
    assert abs(loss.item() - float(g[f"loss_{case}"])) < 1e-4 * abs(float(g[f"loss_{case}"]))
    g0, g1 = torch.autograd.grad(loss, list(maps.values()))
    assert rel(g0, g[f"grad0_{case}"][0]) < 1e-4 and rel(g1, g[f"grad1_{case}"][0]) < 1e-4


def test_guidance_step_matches_reference():
    g = np.load(os.path.join(G, "guidance_step.npz"))
    cfg = UNetConfig(**TINY)
    sd = synthetic_state_dict(cfg, seed=0)
    keys = [tuple(int(x) if x.isdigit() else x for x in k.split("_")) for k in g["keys"]]
    sched = scheduler_ref.DPMSolverPP2M()

    def unet_fn(lat, t, cond, save, save_keys):
        unet_ref.unet_forward(sd, cfg, lat, int(t), cond, save_attn_to_dict=save, save_keys=save_keys, stop_after_key=keys[-1])

    hp = dict(loss_scale=5.0, loss_threshold=0.01, max_index_step=10, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0,
              com_loss_scale=0.03, base_attn_dim=(16, 16), guidance_attn_keys=keys)
    for iters, suffix in ((1, "_1"), (2, "")):
        lat, loss = guidance_ref.latent_backward_guidance(unet_fn, sched.alphas_cumprod, torch.from_numpy(g["cond"]), 0, g["bboxes"].tolist(),
                                                          [[2]], int(g["t"]), torch.from_numpy(g["latents_in"]), 10000.0, max_iter=iters, **hp)
        assert abs(loss - float(g["loss" + suffix])) < 2e-4 * abs(float(g["loss" + suffix]))
        d_ref = torch.from_numpy(g["latents_out" + suffix]) - torch.from_numpy(g["latents_in"])
        assert rel(lat - torch.from_numpy(g["latents_in"]), d_ref) < 2e-3


def test_scheduler_properties():
    s = scheduler_ref.DPMSolverPP2M()
    s.set_timesteps(40)
    assert len(s.timesteps) == 40 and s.timesteps[0] == 999 and s.sigmas[-1] == 0
    # a perfect noise prediction makes every step land on the clean sample scaled by alpha of the next step
    x0 = torch.randn(3, 4, generator=torch.Generator().manual_seed(0))
    noise = torch.randn(3, 4, generator=torch.Generator().manual_seed(1))
    a, sg = s._alpha_sigma(float(s.sigmas[0]))
    x = a * x0 + sg * noise
    for _ in range(40):
        i = s.step_index
        a, sg = s._alpha_sigma(float(s.sigmas[i]))
        eps = (x - a * x0) / sg
        x = s.step(eps, x)
    assert torch.allclose(x, x0, atol=1e-4)


def test_vae_oracle_and_state_dict_layout():
    """VAE decode restatement (oracle/vae_ref.py, parity unpinned) runs on the diffusers-named decoder weights; the
    name/shape enumeration matches the 49.49 M-parameter decode half of the SD AutoencoderKL."""
    import lvd_amd  # noqa: F401
    from lvd_amd.weights import VAE_TINY, VAEConfig, synthetic_vae_state_dict, vae_decoder_param_shapes
    from oracle import vae_ref
    full = vae_decoder_param_shapes(VAEConfig())
    assert sum(torch.Size(s).numel() for s in full.values()) == 49490199
    assert "decoder.up_blocks.2.resnets.0.conv_shortcut.weight" in full and "decoder.up_blocks.3.upsamplers.0.conv.weight" not in full
    cfg = VAEConfig(**VAE_TINY)
    sd = synthetic_vae_state_dict(cfg, seed=1)
    lat = torch.randn(1, 4, 2, 4, 8, generator=torch.Generator().manual_seed(0)) * cfg.scaling_factor
    vid = vae_ref.decode_latents_to_video(sd, cfg, lat)
    assert vid.shape == (1, 2, 32, 64, 3) and float(vid.min()) >= 0 and float(vid.max()) <= 1
    # frames are decoded independently: permuting frames permutes the output
    vid2 = vae_ref.decode_latents_to_video(sd, cfg, lat.flip(2))
    assert torch.allclose(vid2.flip(1), vid, atol=1e-5)
