"""Developer probe: launch a few GEMM shapes once each (for rocprofv3 --pmc runs)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd
from lvd_amd import ops
dev = "cuda"
if os.environ.get("LVD_GEMM_VARIANT"):
    ops.set_gemm_autotune(False)  # pinned geometry: no tuning launches in the counter pass
def rnd(*s): return torch.randn(*s, device=dev).bfloat16()
B, F = 2, 24
for (M, N, K, kind) in [(138240, 960, 320, "plain"), (138240, 320, 1280, "plain"), (34560, 640, 2560, "plain"), (138240, 320, 2880, "conv"), (34560, 640, 5760, "conv")]:
    w = rnd(N, K) * 0.05
    if kind == "plain":
        a = rnd(M, K)
        for _ in range(3): ops.gemm(a, w)
    else:
        cin = K // 9; hw = M // (B * F); h = {2880: 40, 720: 20}[hw]; wd = hw // h
        a = rnd(M, cin)
        for _ in range(3): ops.gemm(a, w, mode=ops.A_CONV3X3, conv=ops.ConvGeom(h, wd, h, wd))
torch.cuda.synchronize()
