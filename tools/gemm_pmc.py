"""Developer probe: launch GEMM products of the zeroscope step a few times each (for rocprofv3 --pmc / --kernel-trace runs), in the
call form the step uses (residual / LayerNorm fold as in engine.py) and with the SHIPPED autotune table loaded, so that the counter tables
describe the kernels the step runs (a bare call had tuned itself to a kernel the step never launches).
    python tools/gemm_pmc.py [--shape i] [--reps 3]      (LVD_GEMM_VARIANT=v pins a geometry instead)
Shapes: 0 QKV (LayerNorm-folded), 1-2 feed-forward output (+ residual), 3-4 3x3 conv (+ residual), 5 temporal conv."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

SHAPES = [(138240, 960, 320, "linear"), (138240, 320, 1280, "linear"), (34560, 640, 2560, "linear"), (138240, 320, 2880, "conv3x3"),
          (34560, 640, 5760, "conv3x3"), (138240, 320, 960, "tconv3")]
FORM = {0: "ln", 1: "res", 2: "res", 3: "res", 4: "res", 5: ""}  # how engine.py issues each product
TABLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "gemm_autotune_576x320x24.json")


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
    elif os.path.exists(TABLE):
        ops.load_gemm_autotune_table(TABLE)
    known = set(ops.gemm_autotune_table())
    rnd = lambda *s: torch.randn(*s, device=dev).bfloat16()
    B, F = 2, 24
    for i, (M, N, K, kind) in enumerate(SHAPES):
        if args.shape >= 0 and i != args.shape:
            continue
        w = rnd(N, K) * 0.05
        bias = torch.randn(N, device=dev)
        res = rnd(M, N) if FORM[i] == "res" else None
        if kind == "linear":
            a = rnd(M, K)
            if FORM[i] == "ln":
                mr, cs = ops.layernorm_stats(a), w.float().sum(1).contiguous()
                f = lambda: ops.gemm(a, w, bias=bias, ln_stats=mr, ln_colsum=cs)
            else:
                f = lambda: ops.gemm(a, w, bias=bias, res=res)
        elif kind == "conv3x3":
            cin = K // 9; hw = M // (B * F); h = {2880: 40, 720: 20}[hw]; wd = hw // h
            a = rnd(M, cin)
            f = lambda: ops.gemm(a, w, bias=bias, res=res, mode=ops.A_CONV3X3, conv=ops.ConvGeom(h, wd, h, wd))
        else:
            a = rnd(M, K // 3)
            f = lambda: ops.gemm(a, w, bias=bias, mode=ops.A_TCONV3, frames=F, hw=M // (B * F))
        f()  # first call: autotune (when the table does not hold the product); the marker launch below separates it from the measured ones
        new = set(ops.gemm_autotune_table()) - known
        if new and not os.environ.get("LVD_GEMM_VARIANT"):
            print(f"NOTE shape {i}: not in the shipped table, tuned here: {sorted(new)}", file=sys.stderr)
        known |= new
        torch.cuda.synchronize()
        ops.silu(rnd(64, 64))  # marker kernel
        for _ in range(args.reps):
            f()
    torch.cuda.synchronize()
