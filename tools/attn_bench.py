"""Developer probe: attention forward/backward throughput on the shapes of the zeroscope step."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd
from lvd_amd import ops
dev = "cuda"
def rnd(*s): return torch.randn(*s, device=dev).bfloat16()
B, F = 2, 24
def run(name, samples, heads, sq, skv, qmap, kvmap, rows_q, rows_kv, bwd=True, cross=False):
    C = heads * 64
    qkv = rnd(rows_q, 3 * C)
    kv = rnd(rows_kv, 2 * C) if cross else None
    q = qkv[:, :C]; k = kv[:, :C] if cross else qkv[:, C:2*C]; v = kv[:, C:] if cross else qkv[:, 2*C:]
    o = torch.empty(rows_q, C, device=dev, dtype=torch.bfloat16)
    lse = torch.empty(samples, heads, sq, device=dev)
    kw = dict(samples=samples, heads=heads, sq=sq, skv=skv, qmap=qmap, kvmap=kvmap, scale=0.125)
    f = lambda: ops.attention_fwd(q, k, v, o, lse=lse, **kw)
    def timeit(fn, it=10):
        for _ in range(2): fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); s.record()
        for _ in range(it): fn()
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / it * 1e3
    us = timeit(f)
    fl = 4.0 * samples * heads * sq * skv * 64
    line = f"{name:14s} fwd {us:8.1f} us {fl/us/1e6:7.1f} TF/s"
    if bwd:
        do = rnd(rows_q, C); dqkv = torch.empty_like(qkv)
        if cross:
            g = lambda: ops.attention_bwd(q, k, v, o, lse, do, dqkv[:, :C], None, None, **kw)
            flb = 6.0 * samples * heads * sq * skv * 64
        else:
            g = lambda: ops.attention_bwd(q, k, v, o, lse, do, dqkv[:, :C], dqkv[:, C:2*C], dqkv[:, 2*C:], **kw)
            flb = 14.0 * samples * heads * sq * skv * 64
        usb = timeit(g, 5)
        line += f" | bwd {usb:8.1f} us {flb/usb/1e6:7.1f} TF/s(incl. recompute)"
    print(line, flush=True)

for lvl, (C, hw) in enumerate([(320, 2880), (640, 720), (1280, 180)]):
    S = B * F
    run(f"spatial L{lvl}", S, C // 64, hw, hw, ops.RowMap(1, hw, 0, 1), ops.RowMap(1, hw, 0, 1), S * hw, S * hw, bwd=True)
    run(f"cross L{lvl}", S, C // 64, hw, 77, ops.RowMap(1, hw, 0, 1), ops.RowMap(F, 77, 0, 1), S * hw, B * 77, bwd=True, cross=True)
    tm = ops.RowMap(hw, F * hw, 1, hw)
    run(f"temporal L{lvl}", B * hw, C // 64, F, F, tm, tm, S * hw, S * hw, bwd=True)
