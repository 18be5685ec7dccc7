"""Developer probe: per-shape throughput of the GEMM family on the shapes of the zeroscope step."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd
from lvd_amd import ops

dev = "cuda"
if os.environ.get("LVD_GEMM_VARIANT"):
    ops.set_gemm_autotune(False)
def rnd(*s): return torch.randn(*s, device=dev).bfloat16()

HW0, F, B = 2880, 24, 2
shapes = []
for lvl, (C, hw) in enumerate([(320, 2880), (640, 720), (1280, 180), (1280, 45)]):
    M = B * F * hw
    shapes += [("lin qkv", M, 3 * C, C, "plain"), ("lin out", M, C, C, "plain"), ("geglu", M, 8 * C, C, "geglu"), ("ff2", M, C, 4 * C, "plain"),
               ("conv", M, C, 9 * C, "conv"), ("tconv", M, C, 3 * C, "tconv")]
shapes += [("conv up 2560->1280", B * F * 180, 1280, 9 * 2560, "conv"), ("conv up 960->320", B * F * 2880, 320, 9 * 960, "conv"),
           ("conv up 1920->640", B * F * 720, 640, 9 * 1920, "conv")]

def run(name, M, N, K, kind, iters=10):
    w = rnd(N, K) * 0.05
    bias = torch.randn(N, device=dev)
    if kind in ("plain", "geglu"):
        a = rnd(M, K)
        f = lambda: ops.gemm(a, w, bias=bias, act=ops.ACT_GEGLU if kind == "geglu" else ops.ACT_NONE)
    elif kind == "conv":
        cin = K // 9
        hw = M // (B * F)
        h = {2880: 40, 720: 20, 180: 10, 45: 5}[hw]; wd = hw // h
        a = rnd(M, cin)
        f = lambda: ops.gemm(a, w, bias=bias, mode=ops.A_CONV3X3, conv=ops.ConvGeom(h, wd, h, wd))
    else:
        cin = K // 3
        a = rnd(M, cin)
        f = lambda: ops.gemm(a, w, bias=bias, mode=ops.A_TCONV3, frames=F, hw=M // (B * F))
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): f()
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / iters * 1e3
    tf = 2.0 * M * N * K / us / 1e6
    print(f"{name:22s} M={M:7d} N={N:6d} K={K:6d} {kind:6s} {us:9.1f} us  {tf:7.1f} TF/s", flush=True)

for sh in shapes:
    run(*sh)
