"""Developer probe: how much of the short-K, large-M linears' time is tile-round quantisation, how much the kernel's structure, and
where the memory floor of the same bytes sits.  For each (M, N, K, residual) the step's variants run on M as in the step and on an M
that fills whole rounds of every geometry (multiple of 256 x 256 rows); a LayerNorm over the same matrix and a device copy are the
streaming references (same bytes in, same bytes out).
    python tools/quant_probe.py [--variants 109,211,111,117,105]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="109,211,111,117,105,231,225")
ap.add_argument("--rounds", type=int, default=5)
args = ap.parse_args()
variants = [int(v) for v in args.variants.split(",")]
dev = "cuda"


def rnd(*s):
    return torch.randn(*s, device=dev).bfloat16()


def timeit(fn, rounds):
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


for M0, N, K, hasres in [(138240, 320, 320, 1), (138240, 320, 320, 0), (138240, 960, 320, 0), (69120, 320, 320, 1), (34560, 640, 640, 1), (17280, 640, 640, 1)]:
    for M in (M0, (M0 // 65536) * 65536 if M0 >= 65536 else (M0 // 8192) * 8192):
        a, w = rnd(M, K), rnd(N, K) * 0.03
        bias = torch.randn(N, device=dev)
        res = rnd(M, N) if hasres else None
        fl = 2.0 * M * N * K
        byts = (M * K + M * N * (2 if hasres else 1)) * 2
        line = f"M={M:6d} N={N:4d} K={K:4d} r{hasres} {byts / 1e6:6.1f} MB |"
        for v in variants:
            try:
                us = timeit(lambda: ops.gemm(a, w, bias=bias, res=res, variant=v), args.rounds)
                line += f" v{v}: {us:6.1f}us {fl / us / 1e6:4.0f}TF {byts / us / 1e6:4.2f}TB/s |"
            except Exception as ex:  # noqa: BLE001
                line += f" v{v}: n/a |"
        if K in (320, 640, 1280):
            g, b = torch.ones(K, device=dev), torch.zeros(K, device=dev)
            y = torch.empty_like(a)
            us = timeit(lambda: ops.layernorm(a, g, b, out=y), args.rounds)
            line += f" LN(MxK): {us:6.1f}us {2 * M * K * 2 / us / 1e6:4.2f}TB/s |"
        y2 = torch.empty_like(a)
        us = timeit(lambda: y2.copy_(a), args.rounds)
        line += f" copy(MxK): {us:6.1f}us {2 * M * K * 2 / us / 1e6:4.2f}TB/s"
        print(line, flush=True)
