"""Developer probe: phase timeline of the ping-pong GEMM main loops from the LVD_TRACE build (tools/build_ablations.sh):
    LVD_LIB=build/abl/liblvdhip_trace.so python tools/phase_trace.py
Per phase pair of waves 0 (group 0) and 4 (group 1) of one workgroup: L = reads + DMA issue + waits, B1 = wait at the barrier behind L,
M = the MFMAs, B2 = wait at the barrier behind M (shader cycles; the stamps themselves cost ~40-100 cycles each)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

dev = "cuda"
for (M, N, K, v) in [(65536, 4096, 4096, 111), (65536, 4096, 4096, 211), (138240, 320, 2880, 111), (138240, 320, 2880, 211), (138240, 320, 320, 211)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, device=dev)
    ops.gemm(a, w, bias=bias, variant=v)
    ws = ops._splitk_ws[a.device]
    ws.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.gemm(a, w, bias=bias, variant=v)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3
    raw = ws[:256].view(torch.int32).cpu().numpy().astype("uint32")
    print(f"== M={M} N={N} K={K} variant {v}: {us:.1f} us, {2.0 * M * N * K / us / 1e6:.0f} TF/s")
    for g in range(2):
        t = raw[g * 128:(g + 1) * 128]
        n, life = int(t[127]), int(t[126])
        st = t[:n].astype("int64")
        d = (st[1:] - st[:-1]) & 0xffffffff
        print(f"  group {g}: {n} stamps, workgroup lifetime {life} cycles")
        # stamps per phase pair: S0 (start L) S1 (end L) S2 (after barrier) S3 (after MFMAs)
        rows = []
        for i in range(0, n - 4, 4):
            L, B1, Mm, B2 = d[i], d[i + 1], d[i + 2], d[i + 3]
            rows.append((L, B1, Mm, B2))
        for r in rows:
            print("    L %5d  B1 %5d  M %5d  B2 %5d   sum %5d" % (*r, sum(r)))
