"""Developer probe: one GEMM shape, every tile geometry, with and without the residual epilogue."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd
from lvd_amd import ops

ops.set_gemm_autotune(False)
dev = "cuda"
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]] or [(138240, 320, 320), (34560, 640, 640), (69120, 320, 320)]
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)  # evict L2 / Infinity Cache between launches
for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).bfloat16(); w = (torch.randn(N, K, device=dev) * 0.05).bfloat16()
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev).bfloat16()
    for use_res in (False, True):
        line = []
        for v in (1, 5, 11, 17, 31, 37):
            ts = []
            for it in range(6):
                flush.zero_()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record(); ops.gemm(a, w, bias=bias, res=res if use_res else None, variant=v); e.record()
                torch.cuda.synchronize()
                if it >= 2: ts.append(s.elapsed_time(e) * 1e3)
            line.append(f"v{v}:{min(ts):6.1f}")
        print(f"M={M} N={N} K={K} res={int(use_res)}  cold-cache us  " + "  ".join(line), flush=True)
