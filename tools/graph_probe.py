"""Developer probe: does replaying the CFG UNet forward as a captured HIP graph shorten the step (inter-kernel gaps), versus eager launches?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd.engine import HipUNet3D
from lvd_amd.weights import UNetConfig, synthetic_state_dict

cfg = UNetConfig()
engine = HipUNet3D(cfg, synthetic_state_dict(cfg, seed=0, device="cuda"))
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(2, 4, 24, 40, 72, device="cuda", generator=g)
ehs = torch.randn(2, 77, 1024, device="cuda", generator=g)
text = engine.encode_text(ehs)
t = torch.full((1,), 500.0, device="cuda")
for _ in range(3):
    ref = engine.forward(x, t, text=text)
torch.cuda.synchronize()


def timeit(fn, n=10):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / n


eager = timeit(lambda: engine.forward(x, t, text=text))
graph = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    engine.forward(x, t, text=text)
torch.cuda.current_stream().wait_stream(side)
with torch.cuda.graph(graph):
    out = engine.forward(x, t, text=text)
torch.cuda.synchronize()
graph.replay()
torch.cuda.synchronize()
print("graph output equals eager:", torch.equal(out, ref))
rep = timeit(graph.replay)
eager2 = timeit(lambda: engine.forward(x, t, text=text))
print(f"CFG forward eager {eager:.2f} / {eager2:.2f} ms, graph replay {rep:.2f} ms")
