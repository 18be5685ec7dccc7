"""Size-independent properties at BASELINE's full sizes (Zeroscope 576x320x24: latent 40x72, 24 frames, C=320..1280),
where the fp32 oracle would take minutes: identities, linearity, row-stochasticity, batch consistency, descent."""
import pytest
import torch

pytestmark = pytest.mark.gpu

import lvd_amd  # noqa: E402
from lvd_amd import guidance, ops  # noqa: E402
from lvd_amd.engine import HipUNet3D  # noqa: E402
from lvd_amd.weights import UNetConfig, synthetic_state_dict  # noqa: E402

DEV = "cuda"
B, FR, H, W = 2, 24, 40, 72
ROWS = B * FR * H * W  # 138240 tokens


def rnd(*shape, seed=0):
    return torch.randn(*shape, generator=torch.Generator(device=DEV).manual_seed(seed), device=DEV)


def test_gemm_identity_and_linearity_full_rows():
    C = 320
    x, y = rnd(ROWS, C, seed=1).bfloat16(), rnd(ROWS, C, seed=2).bfloat16()
    eye = torch.eye(C, device=DEV).bfloat16()
    assert torch.equal(ops.gemm(x, eye), x), "identity weight must reproduce the input bit-exactly"
    w = (rnd(640, C, seed=3) * 0.05).bfloat16()
    s = ops.gemm((x.float() + y.float()).bfloat16(), w, out_fp32=True)
    t = ops.gemm(x, w, out_fp32=True) + ops.gemm(y, w, out_fp32=True)
    assert ((s - t).norm() / t.norm()).item() < 5e-3  # only the bf16 rounding of (x+y) separates the two sides


def test_conv_and_tconv_delta_kernels_are_identities():
    C = 320
    x = rnd(ROWS, C, seed=4).bfloat16()
    w = torch.zeros(C, 9, C, device=DEV)
    w[:, 4] = torch.eye(C, device=DEV)  # centre tap
    out = ops.gemm(x, w.reshape(C, 9 * C).bfloat16(), mode=ops.A_CONV3X3, conv=ops.ConvGeom(H, W, H, W))
    assert torch.equal(out, x)
    wt = torch.zeros(C, 3, C, device=DEV)
    wt[:, 1] = torch.eye(C, device=DEV)
    out = ops.gemm(x, wt.reshape(C, 3 * C).bfloat16(), mode=ops.A_TCONV3, frames=FR, hw=H * W)
    assert torch.equal(out, x)
    # shifted temporal tap: frame f takes frame f+1, last frame of every batch item sees the zero padding
    ws = torch.zeros(C, 3, C, device=DEV)
    ws[:, 2] = torch.eye(C, device=DEV)
    out = ops.gemm(x, ws.reshape(C, 3 * C).bfloat16(), mode=ops.A_TCONV3, frames=FR, hw=H * W).reshape(B, FR, H * W, C)
    xr = x.reshape(B, FR, H * W, C)
    assert torch.equal(out[:, :-1], xr[:, 1:]) and out[:, -1].abs().max() == 0


@pytest.mark.parametrize("variant", [41, 45, 47])
def test_halo_conv_every_tap_is_a_shift_full_rows(variant):
    """LDS-resident im2col kernels at the headline size: a one-hot tap turns the conv into a shift of the token matrix with zero
    padding at the image borders — bit-exact for all nine taps (halo rows, validity masks, tiles that straddle images), and for all
    three taps of the temporal conv (frame -1 / F padding, pixel blocks that do not divide HW)."""
    C = 320
    x = rnd(ROWS, C, seed=5).bfloat16()
    xi = x.reshape(B * FR, H, W, C)
    eye = torch.eye(C, device=DEV)
    for t in range(9):
        w = torch.zeros(C, 9, C, device=DEV)
        w[:, t] = eye
        out = ops.gemm(x, w.reshape(C, 9 * C).bfloat16(), mode=ops.A_CONV3X3, conv=ops.ConvGeom(H, W, H, W), variant=variant).reshape(B * FR, H, W, C)
        dy, dx = t // 3 - 1, t % 3 - 1
        want = torch.zeros_like(xi)
        want[:, max(0, -dy):H - max(0, dy), max(0, -dx):W - max(0, dx)] = xi[:, max(0, dy):H + min(0, dy), max(0, dx):W + min(0, dx)]
        assert torch.equal(out, want), f"variant {variant}, tap {t}"
    xr = x.reshape(B, FR, H * W, C)
    for t in range(3):
        w = torch.zeros(C, 3, C, device=DEV)
        w[:, t] = eye
        out = ops.gemm(x, w.reshape(C, 3 * C).bfloat16(), mode=ops.A_TCONV3, frames=FR, hw=H * W, variant=variant).reshape(B, FR, H * W, C)
        d = t - 1
        want = torch.zeros_like(xr)
        want[:, max(0, -d):FR - max(0, d)] = xr[:, max(0, d):FR + min(0, d)]
        assert torch.equal(out, want), f"variant {variant}, temporal tap {t}"


def test_attention_rows_are_stochastic_full_sequence():
    heads, hw = 5, H * W
    C = heads * 64
    S = B * FR
    qk = rnd(S * hw, 2 * C, seed=5).bfloat16()
    ones = torch.ones(S * hw, C, device=DEV, dtype=torch.bfloat16)
    o = torch.empty_like(ones)
    lse = torch.empty(S, heads, hw, device=DEV)
    ops.attention_fwd(qk[:, :C], qk[:, C:], ones, o, samples=S, heads=heads, sq=hw, skv=hw, qmap=ops.RowMap(1, hw, 0, 1),
                      kvmap=ops.RowMap(1, hw, 0, 1), scale=0.125, lse=lse)
    assert (o.float() - 1).abs().max().item() < 8e-3  # sum_j P_ij = 1 (bf16 P and output rounding)
    assert torch.isfinite(lse).all()
    # temporal addressing: V = frame index -> output is a convex combination of 0..F-1
    fidx = torch.arange(FR, device=DEV).repeat_interleave(hw).repeat(B)[:, None].expand(-1, C).to(torch.bfloat16).contiguous()
    tm = ops.RowMap(hw, FR * hw, 1, hw)
    ops.attention_fwd(qk[:, :C], qk[:, C:], fidx, o, samples=B * hw, heads=heads, sq=FR, skv=FR, qmap=tm, kvmap=tm, scale=0.125)
    assert o.float().min().item() >= -1e-3 and o.float().max().item() <= FR - 1 + 0.2


def test_groupnorm_output_statistics_full_rows():
    C = 640
    x = (rnd(ROWS // 4, C, seed=6) * 3 + 1.5).bfloat16()
    ones, zeros = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    rps = FR * (H // 2) * (W // 2)  # 5-D norm: statistics across the 24 frames of a batch item
    y = ops.groupnorm(x, ones, zeros, rps).float().reshape(B, rps, 32, C // 32)
    assert y.mean(dim=(1, 3)).abs().max().item() < 2e-2
    assert (y.var(dim=(1, 3), unbiased=False) - 1).abs().max().item() < 2e-2


@pytest.fixture(scope="module")
def full_unet():
    cfg = UNetConfig()
    return HipUNet3D(cfg, synthetic_state_dict(cfg, seed=0, device=DEV), device=DEV)


def test_unet_full_size_batch_consistency(full_unet):
    lat = rnd(1, 4, FR, H, W, seed=7)
    ehs = rnd(1, 77, 1024, seed=8)
    one = full_unet.forward(lat, 500, ehs)
    two = full_unet.forward(lat.expand(2, -1, -1, -1, -1).contiguous(), 500, ehs.expand(2, -1, -1).contiguous())
    assert torch.isfinite(one).all()
    e01 = ((two[0] - two[1]).norm() / two[0].norm()).item()
    e = ((two[0] - one[0]).norm() / one[0].norm()).item()
    # Rows of one product may take different tile geometries (autotuned per M, whole-round head vs K-split tail), which
    # changes fp32 summation order and where the bf16 rounding of a residual add falls; through ~60 random-weight layers
    # that is the same ~2e-2 noise floor as bf16 storage vs the fp32 oracle (tests/test_engine_gpu.py), not a coupling
    # between batch items.
    assert e01 < 3e-2 and e < 3e-2, (e01, e)


def test_guidance_descent_and_empty_layout_full_size(full_unet):
    lat = rnd(1, 4, FR, H, W, seed=9)
    text = full_unet.encode_text(rnd(1, 77, 1024, seed=10))
    keys = [("down", 1, 0, 0), ("down", 2, 0, 0), ("down", 2, 1, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 2, 2, 0)]
    boxes = [[[0.1 + 0.02 * f, 0.3, 0.4 + 0.02 * f, 0.8] for f in range(FR)]]
    hp = dict(loss_scale=2.5, fg_top_p=0.25, bg_top_p=0.25, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03)
    l0, g = guidance.guidance_loss_and_grad(full_unet, lat, 801, text, boxes, [[2]], keys, **hp)
    assert torch.isfinite(g).all() and g.abs().max() > 0
    step = 0.05 / g.abs().max()
    l1, _ = guidance.guidance_loss_and_grad(full_unet, lat - step * g, 801, text, boxes, [[2]], keys, **hp)
    assert l1.item() < l0.item(), (l0.item(), l1.item())  # a small step along -grad lowers the energy

    class S:
        alphas_cumprod = torch.linspace(0.999, 0.01, 1000)
    out, loss = guidance.hip_latent_backward_guidance(S(), full_unet, text, 0, [], [], 801, lat, 10000.0, loss_scale=2.5, loss_threshold=1.0,
                                                      max_iter=1, guidance_attn_keys=keys)
    assert out is lat and float(loss) == 0.0  # zero boxes: no guidance, no crash (reference would raise in autograd)
