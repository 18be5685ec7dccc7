"""Developer probe: what the plain-loader asm-DMA GEMMs do on the K = taps x Cin products of the deep UNet levels (M <= 8640) if the
im2col matrix existed — the bound for an asm-DMA convolution loader — next to the tap-GEMM / split-K plans the step runs today.
    python tools/deep_probe.py"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

dev = "cuda"
F = 24


def rnd(*s):
    return torch.randn(*s, device=dev).bfloat16()


def timeit(fn, rounds=5):
    fn(); fn()
    ts = []
    for _ in range(rounds):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(3):
            fn()
        e.record()
        e.synchronize()
        ts.append(s.elapsed_time(e) / 3 * 1e3)
    return statistics.median(ts)


for M, hw, h, w in ((1080, 45, 5, 9), (2160, 45, 5, 9), (4320, 180, 10, 18), (8640, 180, 10, 18)):
    for taps, kind in ((3, "tconv"), (9, "conv")):
        cin = cout = 1280
        K = taps * cin
        x = rnd(M, cin)
        wt = rnd(cout, K) * 0.02
        bias = torch.randn(cout, device=dev)
        res = rnd(M, cout)
        if kind == "tconv":
            cur = lambda: ops.gemm(x, wt, bias=bias, res=res, mode=ops.A_TCONV3, frames=F, hw=hw)
        else:
            cur = lambda: ops.gemm(x, wt, bias=bias, res=res, mode=ops.A_CONV3X3, conv=ops.ConvGeom(h, w, h, w))
        t_cur = timeit(cur)
        acol = rnd(M, K)
        plain = lambda: ops.gemm(acol, wt, bias=bias, res=res)
        t_plain = timeit(plain)
        fl = 2.0 * M * cout * K
        if kind == "tconv":
            al, wl = rnd(M, 1280), rnd(1280, 1280) * 0.02
            t_l = timeit(lambda: ops.gemm(al, wl, bias=bias, res=res))
            print(f"linear M={M:5d} K= 1280: today {t_l:6.1f} us {2.0 * M * 1280 * 1280 / t_l / 1e6:5.0f} TF/s")
        key = [k for k in ops.gemm_autotune_table() if k[1] == M and k[3] == K]
        var = {k[0]: ops.gemm_autotune_table()[k] for k in key}
        print(f"{kind:5s} M={M:5d} K={K:5d}: today {t_cur:6.1f} us {fl / t_cur / 1e6:5.0f} TF/s | plain GEMM on a materialised im2col {t_plain:6.1f} us "
              f"{fl / t_plain / 1e6:5.0f} TF/s (+ {2 * M * K * 2 / 1e6:.0f} MB of im2col traffic if materialised) variants {var}", flush=True)

print("# temporal conv: tap GEMM / K-split plan of the step vs N-expanded plain product (3N columns, fp32) + combine pass")
for M, hw in ((1080, 45), (2160, 45), (4320, 180), (8640, 180), (17280, 720), (34560, 720)):
    cin = cout = 1280 if hw <= 180 else 640
    x = rnd(M, cin)
    wt = rnd(cout, 3 * cin) * 0.02
    bias = torch.randn(cout, device=dev)
    res = rnd(M, cout)
    we = ops.tconv_expand_weight(wt)
    cur = lambda: ops.gemm(x, wt, bias=bias, res=res, mode=ops.A_TCONV3, frames=F, hw=hw)
    new = lambda: ops.tconv_expanded(x, we, frames=F, hw=hw, bias=bias, res=res)
    err = ((new().float() - cur().float()).norm() / cur().float().norm()).item()
    t0, t1 = timeit(cur), timeit(new)
    print(f"tconv M={M:6d} C={cin:4d}: tap GEMM {t0:6.1f} us | expanded + combine {t1:6.1f} us  (rel diff {err:.1e})", flush=True)
