"""Developer probe: 3x3 conv GEMM variants on the conv shapes of the zeroscope step, interleaved A/B rounds in one process.
    python tools/conv_bench.py [--variants 11,31,25,41,45,47] [--rounds 5] [--quick]
Prints median microseconds and TF/s per (shape, variant) and checks every variant against variant 11 (rel-L2)."""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="11,31,25,41,45,47")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--quick", action="store_true")
ap.add_argument("--tconv", action="store_true", help="temporal (3,1,1) conv shapes instead of the 3x3 conv")
args = ap.parse_args()
variants = [int(v) for v in args.variants.split(",")]
dev = "cuda"
F = 24
LEVELS = {0: (40, 72, 320), 1: (20, 36, 640), 2: (10, 18, 1280), 3: (5, 9, 1280)}
# (level, batch, cin) — forward CFG pass (B=2) and the guidance pass (B=1, forward + dgrad have the same shapes when cin == cout)
shapes = [(0, 2, 320), (0, 2, 640), (0, 2, 960), (0, 1, 320), (1, 2, 640), (1, 2, 1280), (1, 2, 1920), (1, 1, 640), (1, 2, 320),
          (2, 2, 1280), (2, 2, 2560), (2, 1, 1280), (2, 2, 640), (3, 2, 1280), (3, 2, 2560), (3, 1, 1280)]
if args.tconv:
    shapes = [(0, 2, 320), (0, 1, 320), (1, 2, 640), (1, 1, 640), (2, 2, 1280), (2, 1, 1280), (3, 2, 1280), (3, 1, 1280)]
if args.quick:
    shapes = [(0, 2, 320), (0, 2, 960), (1, 2, 640), (2, 2, 1280), (3, 1, 1280)] if not args.tconv else [(0, 2, 320), (1, 2, 640), (3, 1, 1280)]
T = 3 if args.tconv else 9


def rnd(*s):
    return torch.randn(*s, device=dev).bfloat16()


for lvl, B, cin in shapes:
    h, w, cout = LEVELS[lvl]
    M = B * F * h * w
    x = rnd(M, cin)
    wt = rnd(cout, T * cin) * 0.02
    bias = torch.randn(cout, device=dev)
    geo = ops.ConvGeom(h, w, h, w)
    if args.tconv:
        run = lambda v: ops.gemm(x, wt, bias=bias, mode=ops.A_TCONV3, frames=F, hw=h * w, variant=v)
    else:
        run = lambda v: ops.gemm(x, wt, bias=bias, mode=ops.A_CONV3X3, conv=geo, variant=v)
    ref = run(11).float()
    times = {v: [] for v in variants}
    errs = {}
    for v in variants:
        out = run(v).float()
        errs[v] = ((out - ref).norm() / ref.norm()).item()
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
    fl = 2.0 * M * cout * T * cin
    line = f"L{lvl} B={B} M={M:6d} N={cout:4d} cin={cin:4d} |"
    best = min(variants, key=lambda v: statistics.median(times[v]))
    for v in variants:
        us = statistics.median(times[v])
        line += f" v{v}: {us:7.1f}us {fl / us / 1e6:6.0f}TF e={errs[v]:.0e}{'*' if v == best else ' '}|"
    print(line, flush=True)
