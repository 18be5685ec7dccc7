"""Developer probe: the memory-bound kernels (GroupNorm fwd/bwd, LayerNorm fwd/bwd, GEGLU fwd/bwd, add) at the shapes of the zeroscope step:
microseconds and effective TB/s (algorithmic bytes / time) per level and batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

dev = "cuda"
F = 24
LEVELS = [(40 * 72, 320), (20 * 36, 640), (10 * 18, 1280), (5 * 9, 1280)]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / n * 1e3


def rnd(*s):
    return torch.randn(*s, device=dev).bfloat16()


print("# memory-bound kernels of the step, per UNet level (L0 = 40x72 ... L3 = 5x9) and batch: microseconds and algorithmic bytes / time; "
      "reference figures: 8.0 TB/s HBM3E spec, 6.3 TB/s measured float4 copy (MI355X_MICROARCH.md)")
for B in (2, 1):
    for lvl, (hw, C) in enumerate(LEVELS):
        M = B * F * hw
        x, dy = rnd(M, C), rnd(M, C)
        g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        line = f"B={B} L{lvl} M={M:6d} C={C:4d} |"
        for name, rps in (("gn2d", hw), ("gn3d", F * hw)):
            y, mr = ops.groupnorm_auto(x, g, b, rps, silu=True)
            t = timeit(lambda: ops.groupnorm_auto(x, g, b, rps, silu=True))
            line += f" {name} fwd {t:6.1f}us {3 * M * C * 2 / t / 1e6:4.1f}TB/s"
            t = timeit(lambda: ops.groupnorm_bwd(x, dy, g, b, mr, rps, silu=True))
            line += f" bwd {t:6.1f}us {5 * M * C * 2 / t / 1e6:4.1f}TB/s |"
        y, mr = ops.layernorm(x, g, b, return_stats=True)
        t = timeit(lambda: ops.layernorm(x, g, b))
        line += f" ln fwd {t:6.1f}us {2 * M * C * 2 / t / 1e6:4.1f}TB/s"
        t = timeit(lambda: ops.layernorm_bwd(x, dy, g, mr))
        line += f" bwd {t:6.1f}us {3 * M * C * 2 / t / 1e6:4.1f}TB/s"
        t = timeit(lambda: ops.layernorm_stats(x))
        line += f" stats-only {t:6.1f}us {M * C * 2 / t / 1e6:4.1f}TB/s |"
        pre, dh = rnd(M, 8 * C), rnd(M, 4 * C)
        t = timeit(lambda: ops.geglu_fwd(pre))
        line += f" geglu fwd {t:6.1f}us {12 * M * C * 2 / t / 1e6:4.1f}TB/s"
        t = timeit(lambda: ops.geglu_bwd(pre, dh))
        line += f" bwd {t:6.1f}us {20 * M * C * 2 / t / 1e6:4.1f}TB/s |"
        o = torch.empty_like(x)
        t = timeit(lambda: ops.add(x, dy, out=o))
        line += f" add {t:5.1f}us {3 * M * C * 2 / t / 1e6:4.1f}TB/s"
        print(line, flush=True)
