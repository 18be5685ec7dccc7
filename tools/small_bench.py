import sys, os, statistics
sys.path.insert(0, '/root/repo')
import torch, lvd_amd
from lvd_amd import ops
dev='cuda'
variants=[10,1,105,106,120,125,109,117,111,131]
shapes=[(4320,1280,1280,1),(1080,1280,1280,1),(17280,640,640,1),(34560,640,640,1),(8640,1280,1280,1),(69120,320,320,1),(2160,1280,1280,1),(4320,1280,5120,1),(4320,3840,1280,0)]
rnd=lambda *s: torch.randn(*s,device=dev).bfloat16()
for M,N,K,hasres in shapes:
    a,w=rnd(M,K),rnd(N,K)*0.03
    bias=torch.randn(N,device=dev); res=rnd(M,N) if hasres else None
    run=lambda v: ops.gemm(a,w,bias=bias,res=res,variant=v)
    times={v:[] for v in variants}
    for v in variants: run(v); run(v)
    torch.cuda.synchronize()
    for _ in range(5):
        for v in variants:
            s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            s.record()
            for _ in range(5): run(v)
            e.record(); e.synchronize()
            times[v].append(s.elapsed_time(e)/5*1e3)
    fl=2.0*M*N*K
    line=f"M={M:6d} N={N:5d} K={K:5d} |"
    best=min(variants,key=lambda v: statistics.median(times[v]))
    for v in variants:
        us=statistics.median(times[v]); line+=f" v{v}:{us:6.1f}{'*' if v==best else ' '}"
    print(line, f"| best {fl/statistics.median(times[best])/1e6:.0f} TF", flush=True)
