"""Developer probe: timeline of one item of the persistent walker (gemm_stream.hip) from the LVD_TRACE build (tools/build_ablations.sh):
    LVD_LIB=build/abl/liblvdhip_strace.so python tools/stream_trace.py
Stamps of waves 0 (group 0) and 4 (group 1) of the middle workgroup, second item onwards; tags: 1 L start, 2 L end (fragments read, DMA issued,
waited), 3 behind the barrier, 4 MFMAs issued, 5 end of the iteration (behind the second barrier and, at an item's last K tile, the epilogue);
10 boundary start, 11 next K tile pre-issued, 12 waited, 13 a block converted into the strip, 14 a block stored, 19 epilogue done, 20 accumulators
re-initialised.  Shader cycles (the stamps cost 40-100 cycles each)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

dev = "cuda"
ops.set_gemm_autotune(False)
NAMES = {1: "L", 2: "Lend", 3: "B1", 4: "M", 5: "end", 10: "bnd", 11: "pre", 12: "wait", 13: "conv", 14: "store", 19: "epi", 20: "init"}
for (M, N, K, geglu, hasres, ln) in [(138240, 960, 320, 0, 0, 1), (138240, 960, 320, 0, 0, 0), (138240, 2560, 320, 1, 0, 1), (138240, 320, 320, 0, 1, 0), (34560, 1920, 640, 0, 0, 1)]:
    a = torch.randn(M, K, device=dev).bfloat16()
    w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
    bias = torch.randn(N, device=dev)
    res = torch.randn(M, N, device=dev).bfloat16() if hasres else None
    kw = dict(bias=bias, res=res, act=ops.ACT_GEGLU if geglu else ops.ACT_NONE)
    if ln:
        kw["ln_stats"] = ops.layernorm_stats(a)
        kw["ln_colsum"] = w.float().sum(1).contiguous()
    ws = torch.zeros(1 << 16, device=dev)
    ops._debug_ws, ops._DEBUG_HOOKS = ws, True
    ops.gemm(a, w, variant=161, **kw)
    ws.zero_()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    ops.gemm(a, w, variant=161, **kw)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3
    raw = ws[:512].view(torch.int32).cpu().numpy().astype("uint32")
    print(f"== M={M} N={N} K={K} geglu={geglu} res={hasres} ln={ln}: {us:.1f} us, {2.0 * M * N * K / us / 1e6:.0f} TF/s")
    for g in range(2):
        t = raw[g * 256:(g + 1) * 256]
        n = int(t[255])
        st = t[0:n:2].astype("int64")
        tg = t[1:n:2]
        if n == 0:
            print(f"  group {g}: no stamps")
            continue
        d = (st[1:] - st[:-1]) & 0xffffffff
        line = f"  group {g} ({n // 2} stamps):"
        for i in range(len(d)):
            if tg[i] == 1:
                print(line)
                line = "     "
            line += f" {NAMES.get(int(tg[i]), tg[i])}->{NAMES.get(int(tg[i + 1]), tg[i + 1])} {d[i]:5d} |"
        print(line)
ops._debug_ws = None
