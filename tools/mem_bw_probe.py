"""Developer probe: achieved bandwidth of fill / copy / elementwise kernels on this GPU (sanity check of the HBM rooflines)."""
import torch, time
def t(fn, it=10):
    for _ in range(3): fn()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/it*1e3
for mb in (88, 265, 1024):
    n=mb*1024*1024//2
    a=torch.empty(n,dtype=torch.bfloat16,device="cuda"); b=torch.empty_like(a)
    us=t(lambda: a.zero_()); print(f"fill  {mb:5d} MB: {us:7.1f} us  {mb*1.048576/us*1e3/1e3:6.2f} TB/s written")
    us=t(lambda: b.copy_(a)); print(f"copy  {mb:5d} MB: {us:7.1f} us  {2*mb*1.048576/us*1e3/1e3:6.2f} TB/s moved")
    us=t(lambda: a.sum()); print(f"read  {mb:5d} MB: {us:7.1f} us  {mb*1.048576/us*1e3/1e3:6.2f} TB/s read")
