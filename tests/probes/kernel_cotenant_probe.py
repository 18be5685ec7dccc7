"""Developer probe: bit-reproducibility of every kernel family next to a co-tenant process (see gemm_cotenant_probe.py)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import lvd_amd
from lvd_amd import ops
from lvd_amd.weights import pack_conv3x3, pack_tconv3

dev = "cuda"
load = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "probes", "gemm_cotenant_probe.py"), "load", "200"])
time.sleep(10)
g = torch.Generator(device=dev).manual_seed(1)
rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20


def check(name, fn):
    ref = fn()
    ref = [t.clone() for t in (ref if isinstance(ref, (tuple, list)) else (ref,)) if torch.is_tensor(t)]
    bad = 0
    for _ in range(REPS):
        out = fn()
        out = [t for t in (out if isinstance(out, (tuple, list)) else (out,)) if torch.is_tensor(t)]
        bad += any(not torch.equal(a, b) for a, b in zip(out, ref))
    print(f"{name}: {bad}/{REPS} differ", flush=True)


B, Fr, H, W = 2, 24, 40, 72
for C, Co in ((64, 64), (128, 128)):
    hh, ww = (H, W) if C == 64 else (H // 2, W // 2)
    rows = B * Fr * hh * ww
    x = rn(rows, C).bfloat16()
    wc = pack_conv3x3(rn(Co, C, 3, 3, sc=0.05)).to(dev)
    bias = rn(Co)
    res = rn(rows, Co).bfloat16()
    for v in (5, 11, 41, 45, 47, 20, 25):
        check(f"conv3x3 C{C} {hh}x{ww} v{v}", lambda: ops.gemm(x, wc, bias=bias, res=res, mode=ops.A_CONV3X3, conv=ops.ConvGeom(hh, ww, hh, ww), variant=v))
    wt = pack_tconv3(rn(Co, C, 3, 1, 1, sc=0.05)).to(dev)
    for v in (5, 11, 41, 45, 47):
        check(f"tconv C{C} v{v}", lambda: ops.gemm(x, wt, bias=bias, res=res, mode=ops.A_TCONV3, frames=Fr, hw=hh * ww, variant=v))
    gam, bet = 1 + 0.1 * rn(C), 0.1 * rn(C)
    check(f"groupnorm2d C{C}", lambda: ops.groupnorm(x, gam, bet, hh * ww, groups=32, silu=True))
    check(f"groupnorm5d C{C}", lambda: ops.groupnorm(x, gam, bet, Fr * hh * ww, groups=32, silu=True))
    y, mr = ops.groupnorm_auto(x, gam, bet, hh * ww, groups=32, silu=True)
    dy = rn(rows, C).bfloat16()
    check(f"groupnorm_bwd C{C}", lambda: ops.groupnorm_bwd(x, dy, gam, bet, mr, hh * ww, groups=32, silu=True)[0])
    check(f"layernorm C{C}", lambda: ops.layernorm(x, gam, bet))
    yl, mrl = ops.layernorm(x, gam, bet, return_stats=True)
    check(f"layernorm_bwd C{C}", lambda: ops.layernorm_bwd(x, dy, gam, mrl))
    heads = C // 64
    q, k, v_ = rn(rows, C).bfloat16(), rn(rows, C).bfloat16(), rn(rows, C).bfloat16()
    o = torch.empty_like(q)
    for nm, samples, seq, rmap in (("spatial", B * Fr, hh * ww, ops.RowMap(1, hh * ww, 0, 1)), ("temporal", B * hh * ww, Fr, ops.RowMap(hh * ww, Fr * hh * ww, 1, hh * ww))):
        lse = torch.empty(samples, heads, seq, device=dev)
        kw = dict(samples=samples, heads=heads, sq=seq, skv=seq, qmap=rmap, kvmap=rmap, scale=0.125)
        check(f"attention_fwd {nm} C{C}", lambda: ops.attention_fwd(q, k, v_, torch.empty_like(q), lse=lse, **kw))
        ops.attention_fwd(q, k, v_, o, lse=lse, **kw)
        do = rn(rows, C).bfloat16()
        def bwd():
            dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
            ops.attention_bwd(q, k, v_, o, lse, do, dq, dk, dv, **kw)
            return dq, dk, dv
        check(f"attention_bwd {nm} C{C}", bwd)
    kt, vt = rn(B * 77, C).bfloat16(), rn(B * 77, C).bfloat16()
    kwc = dict(samples=B * Fr, heads=heads, sq=hh * ww, skv=77, qmap=ops.RowMap(1, hh * ww, 0, 1), kvmap=ops.RowMap(Fr, 77, 0, 1), scale=0.125)
    check(f"attention_fwd cross C{C}", lambda: ops.attention_fwd(q, kt, vt, torch.empty_like(q), **kwc))
# transformer_in sized linears (8 heads)
a = rn(B * Fr * H * W, 512).bfloat16()
for (N, K, geglu) in ((1536, 512, 0), (4096, 512, 1), (512, 2048, 0), (64, 512, 0)):
    aa = a if K == 512 else rn(B * Fr * H * W, K).bfloat16()
    w = rn(N, K, sc=0.05).bfloat16()
    for v in (1, 5, 9, 11, 17, 31, 37, 111, 131, 211, 231, 120, 125):
        check(f"linear N{N} K{K} g{geglu} v{v}", lambda: ops.gemm(aa, w, act=ops.ACT_GEGLU if geglu else ops.ACT_NONE, variant=v))
load.terminate()
