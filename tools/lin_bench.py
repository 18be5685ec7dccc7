"""Developer probe: plain-loader GEMM variants on the linear shapes of the zeroscope step, interleaved A/B rounds in one process.
    python tools/lin_bench.py [--variants 11,31,25,51,57,55] [--rounds 5]"""
import argparse
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

ap = argparse.ArgumentParser()
ap.add_argument("--variants", default="5,105,9,109,17,117,111,131,125")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--deep", action="store_true", help="the deep-level products (M <= 8640) of a guided step")
ap.add_argument("--ln", action="store_true", help="the LayerNorm-folded products of the step (to_qkv / to_q / ff.net.0) instead of the plain ones")
args = ap.parse_args()
variants = [int(v) for v in args.variants.split(",")]
dev = "cuda"
# (M, N, K, geglu, residual) — the heaviest plain-loader products of a guided step (tools/step_gemm_profile.py)
shapes = [(138240, 2560, 320, 1, 0), (138240, 320, 320, 0, 1), (4320, 1280, 1280, 0, 1), (34560, 5120, 640, 1, 0), (8640, 10240, 1280, 1, 0),
          (34560, 640, 640, 0, 1), (17280, 640, 640, 0, 1), (8640, 1280, 1280, 0, 1), (138240, 960, 320, 0, 0), (34560, 1920, 640, 0, 0),
          (138240, 320, 1280, 0, 1), (8640, 1280, 5120, 0, 1), (69120, 320, 320, 0, 1), (34560, 640, 2560, 0, 1), (8640, 3840, 1280, 0, 0),
          (4320, 1280, 10240, 0, 1), (1080, 1280, 1280, 0, 1)]


if args.deep:
    shapes = [(1080, 3840, 1280, 0, 0), (1080, 1280, 1280, 0, 0), (1080, 1280, 1280, 0, 1), (2160, 3840, 1280, 0, 0), (2160, 1280, 1280, 0, 1),
              (2160, 1280, 2560, 0, 0), (4320, 1280, 1280, 0, 0), (4320, 1280, 1280, 0, 1), (4320, 3840, 1280, 0, 0), (4320, 1280, 3840, 0, 0),
              (4320, 1280, 5120, 0, 1), (8640, 1280, 1280, 0, 1), (8640, 3840, 1280, 0, 0), (1080, 1280, 5120, 0, 1), (2160, 1280, 5120, 0, 1),
              (4320, 3840, 1280, 0, 2), (4320, 1280, 1280, 0, 2), (1080, 3840, 1280, 0, 2), (17280, 640, 640, 0, 0), (17280, 640, 640, 0, 1)]
if args.ln:  # (M, N, K, geglu, "ln")
    shapes = [(138240, 960, 320, 0, 2), (138240, 2560, 320, 1, 2), (34560, 1920, 640, 0, 2), (34560, 5120, 640, 1, 2), (8640, 3840, 1280, 0, 2),
              (69120, 960, 320, 0, 2), (138240, 320, 320, 0, 2)]


def rnd(*s):
    return torch.randn(*s, device=dev).bfloat16()


for M, N, K, geglu, hasres in shapes:
    a, w = rnd(M, K), rnd(N, K) * 0.03
    bias = torch.randn(N, device=dev)
    res = rnd(M, N) if hasres == 1 else None
    ln_kw = dict(ln_stats=ops.layernorm_stats(a), ln_colsum=w.float().sum(1).contiguous()) if hasres == 2 else {}
    run = lambda v: ops.gemm(a, w, bias=bias, res=res, act=ops.ACT_GEGLU if geglu else ops.ACT_NONE, variant=v, **ln_kw)
    ref = run(111 if args.ln else 11).float()
    times = {v: [] for v in variants}
    errs = {}
    ok = []
    for v in variants:
        try:
            out = run(v).float()
        except (RuntimeError, AssertionError) as e:  # a geometry that cannot take this product (e.g. a LayerNorm fold on a tail variant)
            print(f"  v{v} refused M={M} N={N} K={K}: {str(e)[:100]}", flush=True)
            continue
        ok.append(v)
        errs[v] = ((out - ref).norm() / ref.norm()).item()
        run(v)
    torch.cuda.synchronize()
    for _ in range(args.rounds):
        for v in ok:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(3):
                run(v)
            e.record()
            e.synchronize()
            times[v].append(s.elapsed_time(e) / 3 * 1e3)
    fl = 2.0 * M * N * K
    line = f"M={M:6d} N={N:5d} K={K:5d} g{geglu} r{hasres} |"
    best = min(ok, key=lambda v: statistics.median(times[v]))
    for v in ok:
        us = statistics.median(times[v])
        line += f" v{v}: {us:7.1f}us {fl / us / 1e6:5.0f}TF e={errs[v]:.0e}{'*' if v == best else ' '}|"
    print(line, flush=True)
