"""GPU parity of the guidance path: attention backward, fused loss fwd+bwd, and the whole
latent_backward_guidance step against the reference's golden vectors."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import lvd_amd  # noqa: E402
from lvd_amd import guidance, ops  # noqa: E402
from lvd_amd.engine import HipUNet3D  # noqa: E402
from lvd_amd.weights import TINY, UNetConfig, synthetic_state_dict  # noqa: E402
from oracle import guidance_ref, scheduler_ref  # noqa: E402

DEV = "cuda"
G = os.path.join(os.path.dirname(__file__), "golden")


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(DEV)


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("mode", ["spatial", "spatial_long", "temporal", "temporal_5", "temporal_32", "cross"])
def test_attention_backward(mode):
    H = 2
    C = H * 64
    if mode in ("spatial", "spatial_long"):
        S, sq, skv = (3, 70, 70) if mode == "spatial" else (2, 200, 200)  # >= 128 keys/queries: LDS-shared v2 kernels
        rows = S * sq
        qm = km = ops.RowMap(1, sq, 0, 1)
        perm = lambda t, L: t.reshape(S, L, H, 64).permute(0, 2, 1, 3)
        unperm = lambda t: t.permute(0, 2, 1, 3).reshape(-1, C)
    elif mode.startswith("temporal"):  # <= 32 frames: the single-pass dQ/dK/dV kernel
        Bt, Fr, hw = 2, {"temporal": 24, "temporal_5": 5, "temporal_32": 32}[mode], 10
        S, sq, skv = Bt * hw, Fr, Fr
        rows = Bt * Fr * hw
        qm = km = ops.RowMap(hw, Fr * hw, 1, hw)
        perm = lambda t, L: t.reshape(Bt, Fr, hw, H, 64).permute(0, 2, 3, 1, 4).reshape(S, H, Fr, 64)
        unperm = lambda t: t.reshape(Bt, hw, H, Fr, 64).permute(0, 3, 1, 2, 4).reshape(-1, C)
    else:
        Bt, Fr, sq, skv = 1, 3, 50, 77
        S = Bt * Fr
        rows = S * sq
        qm, km = ops.RowMap(1, sq, 0, 1), ops.RowMap(Fr, skv, 0, 1)
    q = rnd(rows, C, seed=1).bfloat16()
    if mode == "cross":
        k, v = rnd(skv, C, seed=2).bfloat16(), rnd(skv, C, seed=3).bfloat16()
    else:
        k, v = rnd(rows, C, seed=2).bfloat16(), rnd(rows, C, seed=3).bfloat16()
    do = rnd(rows, C, seed=4).bfloat16()
    o = torch.empty_like(q)
    lse = torch.empty(S, H, sq, device=DEV)
    kw = dict(samples=S, heads=H, sq=sq, skv=skv, qmap=qm, kvmap=km, scale=0.125)
    ops.attention_fwd(q, k, v, o, lse=lse, **kw)
    dq = torch.zeros_like(q)
    dk = torch.zeros_like(k) if mode != "cross" else None
    dv = torch.zeros_like(v) if mode != "cross" else None
    ops.attention_bwd(q, k, v, o, lse, do, dq, dk, dv, **kw)
    qa, ka, va = (t.float().requires_grad_(True) for t in (q, k, v))
    if mode == "cross":
        qs = qa.reshape(S, sq, H, 64).permute(0, 2, 1, 3)
        ks = ka.reshape(1, skv, H, 64).permute(0, 2, 1, 3).expand(S, -1, -1, -1)
        vs = va.reshape(1, skv, H, 64).permute(0, 2, 1, 3).expand(S, -1, -1, -1)
        out = ((qs @ ks.transpose(-1, -2) * 0.125).softmax(-1) @ vs).permute(0, 2, 1, 3).reshape(rows, C)
    else:
        qs, ks, vs = perm(qa, sq), perm(ka, skv), perm(va, skv)
        out = unperm((qs @ ks.transpose(-1, -2) * 0.125).softmax(-1) @ vs)
    gq, gk, gv = torch.autograd.grad(out, [qa, ka, va], do.float())
    assert rel(dq, gq) < 1.5e-2, ("dq", rel(dq, gq))
    if mode != "cross":
        assert rel(dk, gk) < 1.5e-2, ("dk", rel(dk, gk))
        assert rel(dv, gv) < 1.5e-2, ("dv", rel(dv, gv))


@pytest.mark.parametrize("com", [0.0, 0.03])
def test_fused_loss_kernels_vs_oracle(com):
    frames, heads, Hh, Ww, nt = 4, 3, 8, 12, 77
    P, C = Hh * Ww, heads * 64
    q = rnd(frames * P, C, seed=1, scale=1.5).bfloat16()
    kv = rnd(nt, 2 * C, seed=2).bfloat16()
    k = kv[:, :C]
    bboxes = [[[0.1, 0.2, 0.55, 0.8], [0.15, 0.2, 0.6, 0.8], [0.2, 0.2, 0.65, 0.8], [0.25, 0.2, 0.7, 0.8]],
              [[0.5, 0.5, 0.9, 0.95], [0.0, 0.0, 0.0, 0.0], [0.4, 0.45, 0.8, 0.9], [0.35, 0.4, 0.75, 0.85]]]
    pos = [[2, 3], [6]]
    hp = dict(fg_top_p=0.3, bg_top_p=0.4, fg_weight=1.0, bg_weight=2.0, com_loss_scale=com)
    loss_scale, nkeys = 5.0, 1
    qa = q.float().requires_grad_(True)
    probs = (qa.reshape(frames, P, heads, 64).permute(0, 2, 1, 3) @ k.float().reshape(nt, heads, 64).permute(1, 2, 0)[None] * 0.125).softmax(-1)
    ref = guidance_ref.compute_ca_loss({"k": probs}, bboxes, pos, ["k"], (Hh, Ww), **hp) * loss_scale
    (gq,) = torch.autograd.grad(ref, qa)
    lay = guidance.GuidanceLayout(bboxes, pos, frames, Hh, Ww, hp["fg_top_p"], hp["bg_top_p"], DEV)
    partial = torch.zeros(frames * heads * 3, device=DEV)
    gs = loss_scale / (len(bboxes) * nkeys)
    dq = guidance.ca_energy_loss_and_dq(q, k, heads, frames, lay, ntext=nt, grad_scale=gs, fg_weight=1.0, bg_weight=2.0,
                                        com_loss_scale=com, loss_partial=partial)
    loss = ops.reduce_sum(partial, gs).item()
    assert abs(loss - ref.item()) < 2e-4 * abs(ref.item()), (loss, ref.item())
    assert rel(dq, gq) < 1.5e-2, rel(dq, gq)   # dq is stored in bf16


@pytest.mark.parametrize("Hh,Ww,nt", [(8, 12, 130), (20, 36, 77), (24, 40, 40), (40, 72, 77)])
def test_fused_loss_map_sizes_and_prompt_lengths(Hh, Ww, nt):
    """The kernels' other instantiations: prompts past 96 text positions (general one-wave probs / dQ bodies instead of the three-tile
    ones) and maps of 720 / 960 / 2880 positions (12 / 24 / 64 map entries per lane in the wave-per-map select), BoxDiff on (it reads
    the map back by position) — same oracle, same bounds as the 8x12 case."""
    frames, heads = 2, 2
    P, C = Hh * Ww, heads * 64
    q = rnd(frames * P, C, seed=11, scale=1.5).bfloat16()
    k = rnd(nt, C, seed=12).bfloat16()
    bboxes = [[[0.1, 0.2, 0.55, 0.8], [0.15, 0.2, 0.6, 0.8]], [[0.5, 0.5, 0.9, 0.95], [0.4, 0.45, 0.8, 0.9]]]
    pos = [[2, 3], [6]]
    from test_oracle import LOSS_VARIANTS
    hp = dict(fg_top_p=0.3, bg_top_p=0.4, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03)
    hp.update(LOSS_VARIANTS["boxdiff"])
    qa = q.float().requires_grad_(True)
    probs = (qa.reshape(frames, P, heads, 64).permute(0, 2, 1, 3) @ k.float().reshape(nt, heads, 64).permute(1, 2, 0)[None] * 0.125).softmax(-1)
    ref = guidance_ref.compute_ca_loss({"k": probs}, bboxes, pos, ["k"], (Hh, Ww), **hp) * 5.0
    (gq,) = torch.autograd.grad(ref, qa)
    lay = guidance.GuidanceLayout(bboxes, pos, frames, Hh, Ww, hp["fg_top_p"], hp["bg_top_p"], DEV)
    partial = torch.zeros(frames * heads * 3, device=DEV)
    gs = 5.0 / len(bboxes)
    hip_kw = {kk: v for kk, v in hp.items() if kk not in ("fg_top_p", "bg_top_p")}
    dq = guidance.ca_energy_loss_and_dq(q, k, heads, frames, lay, ntext=nt, grad_scale=gs, loss_partial=partial, **hip_kw)
    loss = ops.reduce_sum(partial, gs).item()
    assert abs(loss - ref.item()) < 3e-4 * abs(ref.item()), (loss, ref.item())
    assert rel(dq, gq) < 1.5e-2, rel(dq, gq)


@pytest.mark.parametrize("case", ["ratio", "sync", "boxdiff", "boxdiff_sum", "all", "ce", "ce_com", "smooth", "renorm", "smooth_renorm"])
def test_fused_loss_optional_terms_vs_oracle_and_reference(case):
    """Ratio-based energy, attention sync, BoxDiff corner constraint in the fused kernel: (i) on projected Q/K vs autograd through
    the oracle, (ii) on the reference's own maps: the kernel's loss / dA against the golden loss / gradient of utils/guidance.py."""
    from test_oracle import LOSS_VARIANTS
    kw = dict(fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=1.0, com_loss_scale=0.0)
    kw.update(LOSS_VARIANTS[case])
    frames, heads, Hh, Ww, nt = 4, 3, 8, 12, 77
    P, C = Hh * Ww, heads * 64
    q = rnd(frames * P, C, seed=1, scale=1.5).bfloat16()
    k = rnd(nt, C, seed=2).bfloat16()
    g = np.load(os.path.join(G, "guidance_loss.npz"))
    bboxes, pos = g["bboxes2"].tolist(), [[2, 3], [6]]
    qa = q.float().requires_grad_(True)
    probs = (qa.reshape(frames, P, heads, 64).permute(0, 2, 1, 3) @ k.float().reshape(nt, heads, 64).permute(1, 2, 0)[None] * 0.125).softmax(-1)
    ref = guidance_ref.compute_ca_loss({"k": probs}, bboxes, pos, ["k"], (Hh, Ww), **kw) * 5.0
    (gq,) = torch.autograd.grad(ref, qa)
    lay = guidance.GuidanceLayout(bboxes, pos, frames, Hh, Ww, kw["fg_top_p"], kw["bg_top_p"], DEV)
    partial = torch.zeros(frames * heads * 3, device=DEV)
    gs = 5.0 / len(bboxes)
    hip_kw = {kk: v for kk, v in kw.items() if kk not in ("fg_top_p", "bg_top_p")}
    dq = guidance.ca_energy_loss_and_dq(q, k, heads, frames, lay, ntext=nt, grad_scale=gs, loss_partial=partial, **hip_kw)
    loss = ops.reduce_sum(partial, gs).item()
    assert abs(loss - ref.item()) < 3e-4 * abs(ref.item()), (case, loss, ref.item())
    assert rel(dq, gq) < 1.5e-2, (case, rel(dq, gq))


@pytest.mark.parametrize("case", ["smooth", "renorm", "smooth_renorm"])
def test_map_level_options_on_the_reference_maps_vs_golden(case):
    """`smooth_attn` / `attn_renorm` (utils/guidance.py:209-226; csrc/guidance_maps.hip) on the REFERENCE's own probability maps: loss and the
    gradient w.r.t. both maps against the golden values utils/guidance.py + autograd produced (oracle/make_golden.py section (c)).  fp32 in,
    fp32 out: the whole chain — reflect-padded 3x3 Gaussian and its adjoint, the second softmax and its backward, column gather / scatter, the
    unchanged selection kernel — to 2e-5."""
    from test_oracle import LOSS_VARIANTS
    g = np.load(os.path.join(G, "guidance_loss.npz"))
    kw = dict(LOSS_VARIANTS[case])
    bboxes, pos = g["bboxes2"].tolist(), [[2, 3], [6]]
    frames, heads, Hh, Ww = 4, 3, 8, 12
    fg, bgp, com = kw.pop("fg_top_p"), kw.pop("bg_top_p"), kw.pop("com_loss_scale", 0.0)
    lay = guidance.GuidanceLayout(bboxes, pos, frames, Hh, Ww, fg, bgp, DEV)
    gs = 1.0 / (len(bboxes) * 2)  # loss_scale 1, two objects, two keys (compute_ca_lossv3's normalisation, utils/guidance.py:569-572)
    total = 0.0
    for i in range(2):
        a0 = torch.from_numpy(g[f"maps_{i}"])[0].to(DEV).contiguous()  # (frames, heads, P, 10 tokens)
        partial = torch.zeros(frames * heads * 3, device=DEV)
        dm = guidance.ca_maps_loss_and_grad(a0, heads, frames, lay, grad_scale=gs, com_loss_scale=com, loss_partial=partial, **kw)
        total += ops.reduce_sum(partial, gs).item()
        want = torch.from_numpy(g[f"grad{i}_{case}"])[0]
        assert rel(dm, want) < 2e-5, (case, i, rel(dm, want))
    assert abs(total - float(g[f"loss_{case}"])) < 2e-5 * abs(float(g[f"loss_{case}"])), (case, total, float(g[f"loss_{case}"]))


def test_guidance_step_with_map_level_options_vs_oracle_loop():
    """`hip_latent_backward_guidance(..., smooth_attn=True, attn_renorm=True, num_tokens=..., renorm_scale=...)` through the whole guidance step on
    the tiny topology against the oracle's forward + compute_ca_loss + autograd with the same options; out-of-range and missing `num_tokens`
    raise as the reference's indexing would."""
    from oracle import unet_ref
    cfg = UNetConfig(**TINY)
    sd = synthetic_state_dict(cfg, seed=0)
    net = HipUNet3D(cfg, sd)
    gen = torch.Generator().manual_seed(22)
    lat0 = torch.randn(1, 4, 4, 16, 16, generator=gen)
    cond = torch.randn(1, 77, cfg.cross_attention_dim, generator=gen)
    keys = [("down", 1, 0, 0), ("up", 1, 1, 0), ("up", 2, 1, 0)]
    boxes, pos = [[[0.1 + 0.05 * f, 0.2, 0.6 + 0.05 * f, 0.8] for f in range(4)], [[0.5, 0.5, 1.0, 1.0]] * 3 + [[0.0] * 4]], [[2], [5, 6]]
    hp = dict(loss_scale=5.0, loss_threshold=0.01, max_index_step=10, fg_top_p=0.4, bg_top_p=0.3, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03,
              smooth_attn=True, attn_renorm=True, num_tokens=12, renorm_scale=2.0, guidance_attn_keys=keys)
    sched = scheduler_ref.DPMSolverPP2M()

    def unet_fn(x, tt, c, save, save_keys):
        unet_ref.unet_forward(sd, cfg, x, int(tt), c, save_attn_to_dict=save, save_keys=save_keys, stop_after_key=keys[-1])

    ref_lat, ref_loss = guidance_ref.latent_backward_guidance(unet_fn, sched.alphas_cumprod, cond, 0, boxes, pos, 801, lat0.clone(), 10000.0,
                                                              max_iter=1, base_attn_dim=(16, 16), **hp)
    run = lambda **over: guidance.hip_latent_backward_guidance(sched, net, cond.to(DEV), 0, boxes, pos, 801, lat0.clone().to(DEV), torch.tensor(10000.0),
                                                               max_iter=1, **dict(hp, **over))
    lat, loss = run()
    d, d_ref = lat.cpu() - lat0, ref_lat - lat0
    print(f"smooth + renorm: loss {float(loss):.4f} vs oracle {ref_loss:.4f}; update rel-L2 {rel(d, d_ref):.4f}")
    assert abs(float(loss) - ref_loss) < 2e-2 * abs(ref_loss)
    # the bf16-storage floor of exactly this problem is 0.108 (oracle/bf16_storage.py: the second softmax sharpens the maps, and with them every
    # rounding of the trunk); single realisations get 2 x the floor, as in test_guidance_update_within_bf16_noise_floor
    assert rel(d, d_ref) < 0.216
    with pytest.raises(TypeError, match="num_tokens"):
        run(num_tokens=None)
    with pytest.raises(IndexError):
        run(num_tokens=7)  # tokens 1 .. 5: object position 6 falls outside


@pytest.mark.parametrize("com", [0.0, 0.03])
def test_fused_loss_all_keys_in_one_launch_set_equals_key_by_key(com):
    """ca_*_multi (blockIdx.z = key): six keys of different resolutions / head counts in three launches give the bits of 18 single-key
    launches — loss partials and dQ."""
    frames, nt = 6, 77
    keysz = [(8, 12, 3), (4, 6, 5), (4, 6, 5), (8, 12, 3), (16, 24, 2), (4, 6, 5)]  # (H, W, heads)
    bboxes = [[[0.1, 0.2, 0.55 + 0.02 * f, 0.8] for f in range(frames)], [[0.5, 0.5, 0.9, 0.95] if f != 2 else [0.0] * 4 for f in range(frames)]]
    pos = [[2, 3], [6]]
    lays, items_a, items_b = {}, [], []
    ntok = 3
    tot = sum(frames * h * ntok for _, _, h in keysz)
    pa, pb = torch.zeros(tot, device=DEV), torch.zeros(tot, device=DEV)
    off = 0
    for i, (Hh, Ww, heads) in enumerate(keysz):
        q = rnd(frames * Hh * Ww, heads * 64, seed=10 + i, scale=1.5).bfloat16()
        k = rnd(nt, heads * 64, seed=30 + i).bfloat16()
        lay = lays.get((Hh, Ww)) or lays.setdefault((Hh, Ww), guidance.GuidanceLayout(bboxes, pos, frames, Hh, Ww, 0.3, 0.4, DEV))
        n = frames * heads * ntok
        items_a.append((q, k, heads, lay, pa[off:off + n]))
        items_b.append((q, k, heads, lay, pb[off:off + n]))
        off += n
    kw = dict(ntext=nt, grad_scale=0.4, fg_weight=1.0, bg_weight=2.0, com_loss_scale=com)
    dq_multi = guidance.ca_energy_loss_and_dq_all_keys(items_a, frames, **kw)
    dq_single = [guidance.ca_energy_loss_and_dq(q, k, heads, frames, lay, loss_partial=part, **kw) for q, k, heads, lay, part in items_b]
    assert torch.equal(pa, pb) and float(pa.abs().sum()) > 0
    for a, b in zip(dq_multi, dq_single):
        assert torch.equal(a, b) and float(a.float().abs().sum()) > 0


def test_fused_loss_many_object_tokens_and_bounds():
    """More object tokens than one launch keeps per query (csrc/guidance_loss.hip MAXTOK = 16) run in token chunks — loss and dQ
    are per-token / linear in the tokens; a token position beyond the text length raises IndexError like the reference's indexing."""
    frames, heads, Hh, Ww, nt = 2, 2, 8, 8, 77
    P, C = Hh * Ww, heads * 64
    q = rnd(frames * P, C, seed=1, scale=1.5).bfloat16()
    k = rnd(nt, C, seed=2).bfloat16()
    bboxes = [[[0.1, 0.2, 0.55, 0.8], [0.15, 0.2, 0.6, 0.8]], [[0.5, 0.5, 0.9, 0.95], [0.4, 0.45, 0.8, 0.9]]]
    pos = [list(range(1, 13)), list(range(20, 29))]  # 12 + 9 = 21 tokens
    hp = dict(fg_top_p=0.3, bg_top_p=0.4, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03)
    qa = q.float().requires_grad_(True)
    probs = (qa.reshape(frames, P, heads, 64).permute(0, 2, 1, 3) @ k.float().reshape(nt, heads, 64).permute(1, 2, 0)[None] * 0.125).softmax(-1)
    ref = guidance_ref.compute_ca_loss({"k": probs}, bboxes, pos, ["k"], (Hh, Ww), **hp) * 5.0
    (gq,) = torch.autograd.grad(ref, qa)
    lay = guidance.GuidanceLayout(bboxes, pos, frames, Hh, Ww, hp["fg_top_p"], hp["bg_top_p"], DEV)
    assert lay.ntok == 21 > guidance.MAX_TOKENS_PER_LAUNCH
    partial = torch.zeros(frames * heads * lay.ntok, device=DEV)
    gs = 5.0 / len(bboxes)
    dq = guidance.ca_energy_loss_and_dq(q, k, heads, frames, lay, ntext=nt, grad_scale=gs, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03,
                                        loss_partial=partial)
    loss = ops.reduce_sum(partial, gs).item()
    assert abs(loss - ref.item()) < 2e-4 * abs(ref.item()), (loss, ref.item())
    assert rel(dq, gq) < 2e-2, rel(dq, gq)   # bf16 dq, two chunks added in bf16
    bad = guidance.GuidanceLayout(bboxes, [[2], [77]], frames, Hh, Ww, 0.3, 0.4, DEV)
    with pytest.raises(IndexError):
        guidance.ca_energy_loss_and_dq(q, k, heads, frames, bad, ntext=nt, grad_scale=gs, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.0,
                                       loss_partial=torch.zeros(frames * heads * 2, device=DEV))


def test_return_saved_attn_first_and_last():
    """return_saved_attn of models/pipelines.py:85-97: the maps of the first / last iteration come back as the third element, equal to the
    softmax of the projected queries and text keys (the reference's AttnProcessor save path)."""
    cfg = UNetConfig(**TINY)
    sd = synthetic_state_dict(cfg, seed=0)
    net = HipUNet3D(cfg, sd)
    keys = [("down", 1, 0, 0), ("up", 1, 1, 0)]
    sched = scheduler_ref.DPMSolverPP2M()
    lat0 = rnd(1, 4, 4, 16, 16, seed=3)
    cond = rnd(1, 77, cfg.cross_attention_dim, seed=4)
    boxes = [[[0.1, 0.2, 0.6, 0.8]] * 4]
    hp = dict(loss_scale=5.0, loss_threshold=0.01, max_index_step=10, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, guidance_attn_keys=keys)
    for mode in ("first", "last"):
        lat, loss, maps = guidance.hip_latent_backward_guidance(sched, net, cond, 0, boxes, [[2]], 801, lat0.clone(), torch.tensor(10000.0), max_iter=2,
                                                                return_saved_attn=mode, **hp)
        assert set(maps) == set(keys)
        saved = {}
        from oracle import unet_ref
        x_in = lat0.cpu() if mode == "first" else None
        for key in keys:
            m = maps[key]
            assert m.shape[0] == 4 and m.shape[-1] == 77 and torch.allclose(m.sum(-1), torch.ones_like(m.sum(-1)), atol=1e-4)
        if mode == "first":
            unet_ref.unet_forward({k: v.cpu() for k, v in sd.items()}, cfg, x_in, 801, cond.cpu(), save_attn_to_dict=saved, save_keys=keys, stop_after_key=keys[-1])
            for key in keys:
                assert rel(maps[key], saved[key]) < 5e-2, (key, rel(maps[key], saved[key]))
    with pytest.raises(ValueError):
        guidance.hip_latent_backward_guidance(sched, net, cond, 0, boxes, [[2]], 801, lat0.clone(), torch.tensor(10000.0), max_iter=1, return_saved_attn="all", **hp)


def test_guidance_step_vs_reference_golden():
    """Latents after latent_backward_guidance (1 and 2 iterations) vs the reference run (fp32 CPU).
    Tolerance: loss 2e-2 relative, latent update rel-L2 0.08 (bf16 trunk forward+backward vs fp32)."""
    g = np.load(os.path.join(G, "guidance_step.npz"))
    cfg = UNetConfig(**TINY)
    net = HipUNet3D(cfg, synthetic_state_dict(cfg, seed=0))
    keys = [tuple(int(x) if x.isdigit() else x for x in k.split("_")) for k in g["keys"]]
    sched = scheduler_ref.DPMSolverPP2M()
    lat0 = torch.from_numpy(g["latents_in"]).to(DEV)
    cond = torch.from_numpy(g["cond"]).to(DEV)
    hp = dict(loss_scale=5.0, loss_threshold=0.01, max_index_step=10, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0,
              com_loss_scale=0.03, guidance_attn_keys=keys)
    for iters, suffix in ((1, "_1"), (2, "")):
        lat, loss = guidance.hip_latent_backward_guidance(sched, net, cond, 0, g["bboxes"].tolist(), [[2]], int(g["t"]), lat0.clone(),
                                                          torch.tensor(10000.0), max_iter=iters, **hp)
        ref_loss = float(g["loss" + suffix])
        d = lat - lat0
        d_ref = torch.from_numpy(g["latents_out" + suffix]).to(DEV) - lat0
        print(f"iters={iters}: loss {float(loss):.4f} vs ref {ref_loss:.4f}; update rel-L2 {rel(d, d_ref):.4f}; |d| {d.abs().mean().item():.3e} vs {d_ref.abs().mean().item():.3e}")
        assert abs(float(loss) - ref_loss) < 2e-2 * abs(ref_loss)
        assert rel(d, d_ref) < 0.07  # 1.5 x the bf16-storage noise floor of this topology (4.5 %, tests/test_noise_floor.py)


def test_guidance_step_with_the_ce_energy_vs_oracle_loop():
    """`hip_latent_backward_guidance(..., use_max_based_loss=False, use_ce_based_loss=True)` — the CE / NLL form of the top-k energy
    (utils/guidance.py:363-399, selected by the elif chain of :312,346,363) — through the whole guidance step on the tiny topology: recorded
    forward, fused loss in its CE form, hand-written backward, latent update, against the oracle's forward + compute_ca_loss + autograd with
    the same options (the oracle's CE restatement is pinned by the reference goldens loss_ce / loss_ce_com).  No form selected raises the
    reference's ValueError."""
    from oracle import unet_ref
    cfg = UNetConfig(**TINY)
    sd = synthetic_state_dict(cfg, seed=0)
    net = HipUNet3D(cfg, sd)
    gen = torch.Generator().manual_seed(21)
    lat0 = torch.randn(1, 4, 4, 16, 16, generator=gen)
    cond = torch.randn(1, 77, cfg.cross_attention_dim, generator=gen)
    keys = [("down", 1, 0, 0), ("up", 1, 1, 0), ("up", 2, 1, 0)]
    boxes, pos = [[[0.1 + 0.05 * f, 0.2, 0.6 + 0.05 * f, 0.8] for f in range(4)], [[0.5, 0.5, 1.0, 1.0]] * 3 + [[0.0] * 4]], [[2], [5, 6]]
    hp = dict(loss_scale=5.0, loss_threshold=0.01, max_index_step=10, fg_top_p=0.4, bg_top_p=0.3, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03,
              use_max_based_loss=False, use_ce_based_loss=True, guidance_attn_keys=keys)
    sched = scheduler_ref.DPMSolverPP2M()

    def unet_fn(x, tt, c, save, save_keys):
        unet_ref.unet_forward(sd, cfg, x, int(tt), c, save_attn_to_dict=save, save_keys=save_keys, stop_after_key=keys[-1])

    ref_lat, ref_loss = guidance_ref.latent_backward_guidance(unet_fn, sched.alphas_cumprod, cond, 0, boxes, pos, 801, lat0.clone(), 10000.0,
                                                              max_iter=1, base_attn_dim=(16, 16), **hp)
    lat, loss = guidance.hip_latent_backward_guidance(sched, net, cond.to(DEV), 0, boxes, pos, 801, lat0.clone().to(DEV), torch.tensor(10000.0),
                                                      max_iter=1, **hp)
    d, d_ref = lat.cpu() - lat0, ref_lat - lat0
    print(f"CE energy: loss {float(loss):.4f} vs oracle {ref_loss:.4f}; update rel-L2 {rel(d, d_ref):.4f}")
    assert abs(float(loss) - ref_loss) < 2e-2 * abs(ref_loss)
    # bf16-storage floor of exactly this problem (oracle/bf16_storage.py, fp32 oracle vs the same oracle with bf16-rounded stores): 5.35e-2 for the
    # CE form (3.85e-2 for the max-based energy on the same layout: -log a sends 1 / a into the gradient); measured here 8.2e-2 = 1.5 x the floor.
    # Bound = 2 x floor, the single-realisation bound of test_guidance_update_within_bf16_noise_floor.
    assert rel(d, d_ref) < 0.107
    plain, _ = guidance.hip_latent_backward_guidance(sched, net, cond.to(DEV), 0, boxes, pos, 801, lat0.clone().to(DEV), torch.tensor(10000.0), max_iter=1,
                                                     **{k: v for k, v in hp.items() if not k.startswith("use_")})
    assert rel(plain.cpu() - lat0, d_ref) > 3 * rel(d, d_ref)  # the max-based energy is another function: the option is not ignored
    with pytest.raises(ValueError, match="no loss selected"):
        guidance.hip_latent_backward_guidance(sched, net, cond.to(DEV), 0, boxes, pos, 801, lat0.clone().to(DEV), torch.tensor(10000.0), max_iter=1,
                                              **dict(hp, use_ce_based_loss=False))


@pytest.mark.parametrize("t,layout", [(801, "2obj"), (801, "1box"), (999, "1box")])
def test_guidance_update_within_bf16_noise_floor(t, layout):
    """The tolerance of the guidance update is not free-standing.  The fp32 oracle run with bf16 STORAGE of activations and activation
    gradients (oracle/bf16_storage.py: the rounding points of the product) differs from the plain fp32 oracle by 2.7-4.5 % on these
    problems: that is the floor any bf16 trunk pays.  The HIP path (bf16 storage, fp32 accumulation, hand-written backward) sits ON that
    floor: 1.02x on average over 18 (timestep, layout, seed) cases (0.76x .. 1.44x per case — the top-k selections of the energy are
    discontinuous, so single realisations scatter; tests/probes/noise_floor_probe.py, profiles/r03_noise_floor.txt).  Asserted here over
    SIX seeds per case: mean HIP error <= 1.25 x mean floor, every single error <= 2 x the floor of its own problem (or the mean floor,
    whichever is larger), and HIP is as close to the bf16-storage oracle as two independent roundings of one computation are (same bound).  (Three seeds until round 4: any change
    of summation order — the LayerNorm fold, the N-expanded temporal conv — re-draws the realisations, and with the documented 0.76x-1.44x
    scatter a three-seed mean lands on either side of 1.25: (999, 1box) read 1.18 with the tap-GEMM temporal conv and 1.28 with the expanded
    one, the outlier merely moving from seed 2 to seed 1.)"""
    from oracle import bf16_storage
    from test_noise_floor import HP, KEYS, guidance_problem
    hp = {k: v for k, v in HP.items() if k != "guidance_attn_keys"}
    floors, errs, errs16 = [], [], []
    net = None
    NS = 6
    for seed in range(NS):
        cfg, sd, lat, cond, boxes, pos = guidance_problem(seed)
        if layout == "1box":
            boxes, pos = [[[0.1, 0.2, 0.6, 0.8]] * lat.shape[2]], [[2]]
        u32, l32 = bf16_storage.oracle_guidance_update(cfg, sd, lat, cond, boxes, pos, t, "fp32", KEYS, **hp)
        u16, _ = bf16_storage.oracle_guidance_update(cfg, sd, lat, cond, boxes, pos, t, "bf16", KEYS, **hp)
        net = net or HipUNet3D(cfg, sd)
        new, loss = guidance.hip_latent_backward_guidance(scheduler_ref.DPMSolverPP2M(), net, cond.to(DEV), 0, boxes, pos, t, lat.clone().to(DEV),
                                                          torch.tensor(10000.0), **HP)
        d = new.cpu() - lat
        floors.append(rel(u16, u32)); errs.append(rel(d, u32)); errs16.append(rel(d, u16))
        assert abs(float(loss) - l32) < 1e-2 * abs(l32)
    mf = sum(floors) / NS
    print(f"t={t} {layout}: bf16-storage floors {[round(f, 4) for f in floors]}, HIP vs fp32 oracle {[round(e, 4) for e in errs]} "
          f"(mean ratio {sum(errs) / NS / mf:.2f}), HIP vs bf16-storage oracle {[round(e, 4) for e in errs16]}")
    assert sum(errs) / NS <= 1.25 * mf, (errs, floors)
    # single realisations: within twice the floor of their own problem (or the mean floor, whichever is larger)
    for e, e16, f in zip(errs, errs16, floors):
        assert e <= 2.0 * max(f, mf) and e16 <= 2.0 * max(f, mf), (errs, errs16, floors)


def test_batched_guidance_pass_keeps_the_samples_independent():
    """guidance.hip_latent_backward_guidance_many (throughput mode: V samples in ONE recorded forward / backward): per-sample layouts, losses and
    thresholds.  Two copies of one sample -> bit-identical updates and losses; two different samples -> each within the batch-consistency
    distance of its own batch-1 pass (other tile geometries at twice the rows: two bf16 realisations of one update, each about 3.6 % from the
    fp32 oracle (profiles/r05_noise_floor.txt), i.e. about 5 % from each other — measured 4.5 / 5.8 % — asserted < 8e-2; loss < 1e-3 relative) and far from the other sample's; a sample whose carried loss is already under
    the threshold leaves the batch untouched while the other one is updated exactly as in a two-sample batch of its own."""
    from lvd_amd import guidance
    from lvd_amd.weights import TINY, UNetConfig, synthetic_state_dict
    from lvd_amd.engine import HipUNet3D
    from oracle import scheduler_ref
    cfg = UNetConfig(**TINY)
    net = HipUNet3D(cfg, synthetic_state_dict(cfg, seed=0))
    sched = scheduler_ref.DPMSolverPP2M()
    keys = [("down", 1, 0, 0), ("up", 1, 1, 0), ("up", 2, 1, 0)]
    hp = dict(loss_scale=5.0, loss_threshold=0.01, max_iter=1, max_index_step=10, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0,
              com_loss_scale=0.03, guidance_attn_keys=keys)
    rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()

    def smp(seed, x0, pos):
        g = torch.Generator().manual_seed(seed)
        return (torch.randn(1, 4, 4, 16, 16, generator=g).cuda(), torch.randn(1, 77, cfg.cross_attention_dim, generator=g).cuda(),
                [[[x0 + 0.05 * f, 0.2, x0 + 0.5 + 0.05 * f, 0.8] for f in range(4)]], [pos])

    (la, ca, ba, pa), (lb, cb, bb, pb) = smp(1, 0.1, [2]), smp(2, 0.3, [4, 5])
    t = 801
    one = lambda l, c, b, p: guidance.hip_latent_backward_guidance(sched, net, c, 0, b, p, t, l.clone(), torch.tensor(10000.0), **hp)
    na, lossa = one(la, ca, ba, pa)
    nb, lossb = one(lb, cb, bb, pb)
    many = lambda ls, cs, bs, ps, losses: guidance.hip_latent_backward_guidance_many(sched, net, net.encode_text(torch.cat(cs)), 0, bs, ps, t,
                                                                                    [l.clone() for l in ls], losses, **hp)
    big = [torch.tensor(10000.0), torch.tensor(10000.0)]
    (d0, d1), (ld0, ld1) = many([la, la], [ca, ca], [ba, ba], [pa, pa], big)
    assert torch.equal(d0, d1) and float(ld0) == float(ld1), "two copies of one sample came out different"
    (ma, mb), (lma, lmb) = many([la, lb], [ca, cb], [ba, bb], [pa, pb], big)
    ea, eb = rel(ma - la, na - la), rel(mb - lb, nb - lb)
    print(f"batched guidance vs batch-1: update rel-L2 {ea:.3e} / {eb:.3e}; losses {float(lma):.5f} vs {float(lossa):.5f}, {float(lmb):.5f} vs {float(lossb):.5f}")
    assert ea < 8e-2 and eb < 8e-2
    assert abs(float(lma) - float(lossa)) < 1e-3 * abs(float(lossa)) and abs(float(lmb) - float(lossb)) < 1e-3 * abs(float(lossb))
    assert rel(ma - la, nb - lb) > 0.5
    (xb, xa), _ = many([lb, la], [cb, ca], [bb, ba], [pb, pa], big)
    assert torch.equal(xa, ma) and torch.equal(xb, mb), "swapping the samples did not swap the updates"
    # sample 0 already converged (carried loss under the threshold): untouched; sample 1 updated as a batch of one
    (ka, kb), (lka, lkb) = many([la, lb], [ca, cb], [ba, bb], [pa, pb], [torch.tensor(0.0), torch.tensor(10000.0)])
    assert torch.equal(ka, la) and float(lka) == 0.0
    assert rel(kb - lb, nb - lb) < 8e-2


@pytest.mark.parametrize("samples,heads,P,ntext,spk", [(6, 2, 50, 77, 1), (8, 5, 180, 77, 4), (3, 10, 720, 77, 3), (2, 1, 33, 96, 1), (4, 3, 129, 20, 2)])
def test_full_probability_maps_and_their_product_with_v(samples, heads, P, ntext, spk):
    """The slow path of the plug-in on the C ABI (models/attention_processor.py:515-552): `lvdhip_ca_probs_full` materialises
    softmax(scale Q K^T) over all text positions as (samples, heads, P, tokens) fp32; `lvdhip_ca_apply_probs` multiplies a map held in
    memory with V.  Against fp32 torch on the same bf16 inputs: the map to 2e-3 absolute (bf16 MFMA scores, exp2), rows sum to 1; the
    product to bf16 rounding.  Tile tails (P not a multiple of 32 or 64), shared text keys (samples_per_key = frames) and short prompts."""
    gen = torch.Generator().manual_seed(samples * 1000 + P)
    q = torch.randn(samples * P, heads * 64, generator=gen).cuda().bfloat16()
    k = torch.randn((samples // spk) * ntext, heads * 64, generator=gen).cuda().bfloat16()
    v = torch.randn((samples // spk) * ntext, heads * 64, generator=gen).cuda().bfloat16()
    probs = guidance.ca_probability_maps(q, k, samples=samples, heads=heads, positions=P, ntext=ntext, samples_per_key=spk)
    qf = q.float().reshape(samples, P, heads, 64).permute(0, 2, 1, 3)
    sel = lambda t: t.float().reshape(samples // spk, ntext, heads, 64).permute(0, 2, 1, 3).repeat_interleave(spk, 0)
    ref = (qf @ sel(k).transpose(-1, -2) * 0.125).softmax(-1)
    assert probs.shape == ref.shape and float((probs - ref).abs().max()) < 2e-3
    assert float((probs.sum(-1) - 1).abs().max()) < 1e-5
    noisy = (ref + 0.01 * torch.rand(ref.shape, generator=torch.Generator().manual_seed(1)).cuda()).contiguous()  # any fp32 map, not only a softmax
    out = guidance.ca_apply_probabilities(noisy, v, samples=samples, heads=heads, positions=P, ntext=ntext, samples_per_key=spk)
    want = (noisy @ sel(v)).permute(0, 2, 1, 3).reshape(samples * P, heads * 64)
    assert rel(out, want) < 4e-3
