"""Developer probe: launch GEMM shapes of the zeroscope step a few times each (for rocprofv3 --pmc / --kernel-trace runs).
    python tools/gemm_pmc.py [--shape i] [--reps 3]      (LVD_GEMM_VARIANT=v pins a geometry; default: autotuned like the step)
Shapes: 0-2 linear, 3-4 3x3 conv, 5 temporal conv."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

SHAPES = [(138240, 960, 320, "linear"), (138240, 320, 1280, "linear"), (34560, 640, 2560, "linear"), (138240, 320, 2880, "conv3x3"),
          (34560, 640, 5760, "conv3x3"), (138240, 320, 960, "tconv3")]


def algorithmic_bytes(M, N, K, kind):
    cin = K // {"linear": 1, "conv3x3": 9, "tconv3": 3}[kind]
    return 2 * (M * cin + N * K + M * N)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", type=int, default=-1)
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    dev = "cuda"
    if os.environ.get("LVD_GEMM_VARIANT"):
        ops.set_gemm_autotune(False)  # pinned geometry: no tuning launches in the counter pass
    rnd = lambda *s: torch.randn(*s, device=dev).bfloat16()
    B, F = 2, 24
    for i, (M, N, K, kind) in enumerate(SHAPES):
        if args.shape >= 0 and i != args.shape:
            continue
        w = rnd(N, K) * 0.05
        if kind == "linear":
            a = rnd(M, K)
            f = lambda: ops.gemm(a, w)
        elif kind == "conv3x3":
            cin = K // 9; hw = M // (B * F); h = {2880: 40, 720: 20}[hw]; wd = hw // h
            a = rnd(M, cin)
            f = lambda: ops.gemm(a, w, mode=ops.A_CONV3X3, conv=ops.ConvGeom(h, wd, h, wd))
        else:
            a = rnd(M, K // 3)
            f = lambda: ops.gemm(a, w, mode=ops.A_TCONV3, frames=F, hw=M // (B * F))
        f()  # first call: autotune (when not pinned); the marker launch below separates it from the measured ones
        torch.cuda.synchronize()
        ops.silu(rnd(64, 64))  # marker kernel
        for _ in range(args.reps):
            f()
    torch.cuda.synchronize()
