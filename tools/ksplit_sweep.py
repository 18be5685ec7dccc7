"""Developer probe: K-slice count of the K-split variants on the deep-level products of a guided 576x320x24 step — the library's plan (ksplit = 0)
against pinned slice counts.    python tools/ksplit_sweep.py"""
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

ops.set_gemm_autotune(False)


def rnd(*s, seed=0, scale=1.0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*s, generator=g) * scale).cuda()


shapes = [(4320, 1280, 1280, 0), (4320, 1280, 1280, 1), (1080, 3840, 1280, 0), (2160, 3840, 1280, 0), (4320, 3840, 1280, 0), (8640, 1280, 1280, 1), (4320, 1280, 3840, 0),
          (1080, 1280, 1280, 0), (2160, 1280, 1280, 1), (4320, 1280, 5120, 1), (8640, 1280, 5120, 1), (17280, 640, 640, 0), (2160, 1280, 2560, 0), (4320, 5120, 1280, 0)]
for M, N, K, hasres in shapes:
    a, w, bias = rnd(M, K, seed=1).bfloat16(), rnd(N, K, seed=2, scale=0.03).bfloat16(), rnd(N, seed=3)
    res = rnd(M, N, seed=4).bfloat16() if hasres else None
    cands = [(v, ks) for v in (120, 220, 125, 225) for ks in (0, 2, 3, 4, 5, 6, 8)] + [(105, 0), (205, 0), (111, 0), (211, 0), (231, 0)]
    t = {c: [] for c in cands}
    live = []
    for c in cands:
        try:
            ops.gemm(a, w, bias=bias, res=res, variant=c[0], ksplit=c[1])
            ops.gemm(a, w, bias=bias, res=res, variant=c[0], ksplit=c[1])
            live.append(c)
        except RuntimeError:
            pass
    torch.cuda.synchronize()
    for _ in range(5):
        for c in live:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                ops.gemm(a, w, bias=bias, res=res, variant=c[0], ksplit=c[1])
            e.record()
            e.synchronize()
            t[c].append(s.elapsed_time(e) / 3 * 1e3)
    med = {c: statistics.median(t[c]) for c in live}
    best = min(med, key=med.get)
    line = f"M={M:6d} N={N:5d} K={K:5d} r{hasres} best v{best[0]}/ks{best[1]} {med[best]:6.1f}us |"
    for v in (120, 220, 125, 225):
        line += f" v{v}: " + " ".join(f"{med[(v, ks)]:5.1f}" if (v, ks) in med else "  -  " for ks in (0, 2, 3, 4, 5, 6, 8)) + " |"
    line += " 105/205/111/211/231: " + " ".join(f"{med[(v, 0)]:5.1f}" if (v, 0) in med else "  -  " for v in (105, 205, 111, 211, 231))
    print(line, flush=True)
