"""Developer probe: the 64-deep two-slot ping-pong GEMM (variant + 200) against the 32-deep asm-DMA ring (variant + 100) on the guide's
square shapes and on the linear shapes of the zeroscope step; interleaved rounds in one process, results compared with the 32-deep form.
    python tools/p64_bench.py [--variants 111,211] [--rounds 7] [--set square|step|all]
    LVD_LIB=build/abl/liblvdhip_abl1.so python tools/p64_bench.py ...     (ablation builds: timing only, results are wrong by design)"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="111,211,131,231")
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--set", default="all")
ap.add_argument("--nocheck", action="store_true")
args = ap.parse_args()
variants = [int(v) for v in args.variants.split(",")]
dev = "cuda"
square = [(4096, 4096, 4096, 0, 0), (8192, 8192, 8192, 0, 0)]
# (M, N, K, geglu, residual)
step = [(34560, 640, 2560, 0, 1), (138240, 320, 1280, 0, 1), (8640, 1280, 5120, 0, 1), (4320, 1280, 10240, 0, 1), (8640, 10240, 1280, 1, 0),
        (34560, 5120, 640, 1, 0), (138240, 2560, 320, 1, 0), (138240, 320, 320, 0, 1), (34560, 640, 640, 0, 1), (138240, 960, 320, 0, 0),
        (34560, 1920, 640, 0, 0), (8640, 1280, 1280, 0, 1), (4320, 1280, 1280, 0, 1), (8640, 3840, 1280, 0, 0), (17280, 640, 640, 0, 1),
        (69120, 320, 320, 0, 1)]
shapes = {"square": square, "step": step, "all": square + step}[args.set]


def rnd(*s):
    return torch.randn(*s, device=dev).bfloat16()


for M, N, K, geglu, hasres in shapes:
    a, w = rnd(M, K), rnd(N, K) * 0.03
    bias = torch.randn(N, device=dev)
    res = rnd(M, N) if hasres else None
    run = lambda v: ops.gemm(a, w, bias=bias, res=res, act=ops.ACT_GEGLU if geglu else ops.ACT_NONE, variant=v)
    ref = run(variants[0]).float()
    times = {v: [] for v in variants}
    errs = {}
    for v in variants:
        out = run(v).float()
        errs[v] = 0.0 if args.nocheck else ((out - ref).norm() / ref.norm()).item()
        run(v)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for v in variants:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                run(v)
            e.record()
            e.synchronize()
            times[v].append(s.elapsed_time(e) / 3 * 1e3)
    fl = 2.0 * M * N * K
    line = f"M={M:6d} N={N:5d} K={K:5d} g{geglu} r{hasres} |"
    best = min(variants, key=lambda v: statistics.median(times[v]))
    for v in variants:
        us = statistics.median(times[v])
        line += f" v{v}: {us:7.1f}us {fl / us / 1e6:5.0f}TF e={errs[v]:.0e}{'*' if v == best else ' '}|"
    print(line, flush=True)
