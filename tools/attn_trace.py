"""Developer probe: timeline of a few 64-key tiles of one wave of attn_fwd_v3_kernel (build with -DLVD_ATTN_TRACE, see tools/build_ablations.sh):
    LVD_LIB=build/abl/liblvdhip_atrace.so python tools/attn_trace.py
tags: 1 tile start, 2 score MFMAs issued, 3 row maximum known, 4 exponentials + sums done, 5 P.V MFMAs issued, 6 next tile stored to LDS, 7 behind the barrier."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import ops

dev = "cuda"
NAMES = {1: "top", 2: "qk", 3: "max", 4: "exp", 5: "pv", 6: "store", 7: "bar"}
for (S, H, sq) in [(48, 5, 2880), (48, 10, 720)]:
    C = H * 64
    qkv = torch.randn(S * sq, 3 * C, device=dev).bfloat16()
    o = torch.empty(S * sq, C, device=dev, dtype=torch.bfloat16)
    lse_big = torch.zeros(S * H * sq + 512, device=dev)
    lse = lse_big[:S * H * sq].view(S, H, sq)
    kw = dict(samples=S, heads=H, sq=sq, skv=sq, qmap=ops.RowMap(1, sq, 0, 1), kvmap=ops.RowMap(1, sq, 0, 1), scale=0.125, lse=lse)
    for _ in range(2):
        ops.attention_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], o, **kw)
    torch.cuda.synchronize()
    t = lse_big[S * H * sq:].cpu().numpy()
    st, tg = t[0:200:2].astype("int64"), t[1:200:2].astype("int64")
    n = int((tg > 0).sum())
    d = (st[1:n] - st[:n - 1]) & 0xffffff
    print(f"== spatial self-attention S={S} H={H} sq={sq}: {n} stamps")
    line = "   "
    for i in range(n - 1):
        if tg[i] == 1 and i:
            print(line)
            line = "   "
        line += f" {NAMES.get(int(tg[i]), tg[i])}->{NAMES.get(int(tg[i + 1]), tg[i + 1])} {d[i]:5d} |"
    print(line)
