"""Developer probe: the CFG UNet forward (batch 2 = [uncond, cond]) as ONE batch-2 pass versus TWO independent batch-1 passes on two
HIP streams (eager from one host thread, eager from two host threads, and as two captured graphs).  The two halves share nothing but the
weights, so the hardware can fill one pass's tile tails and under-filled deep-level grids with the other pass's workgroups."""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd.engine import HipUNet3D, TextCache
from lvd_amd.weights import UNetConfig, synthetic_state_dict

cfg = UNetConfig()
engine = HipUNet3D(cfg, synthetic_state_dict(cfg, seed=0, device="cuda"))
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(2, 4, 24, 40, 72, device="cuda", generator=g)
ehs = torch.randn(2, 77, 1024, device="cuda", generator=g)
text = engine.encode_text(ehs)
halves = [TextCache(text.tokens[i * 77:(i + 1) * 77], {p: kv[i * 77:(i + 1) * 77] for p, kv in text.kv.items()}, 1, 77) for i in range(2)]
xs = [x[0:1].contiguous(), x[1:2].contiguous()]
t = torch.full((1,), 500.0, device="cuda")
for _ in range(2):
    ref = engine.forward(x, t, text=text)
    r1 = [engine.forward(xs[i], t, text=halves[i]) for i in range(2)]
torch.cuda.synchronize()
print("batch-1 halves vs batch-2 rel-L2:", [float((r1[i] - ref[i:i + 1]).norm() / ref[i:i + 1].norm()) for i in range(2)])


def timeit(fn, n=8):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / n


streams = [torch.cuda.Stream(), torch.cuda.Stream()]
main = torch.cuda.current_stream()


def fork_join(body):
    for s in streams:
        s.wait_stream(main)
    body()
    for s in streams:
        main.wait_stream(s)


def eager_one_thread():
    def body():
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                engine.forward(xs[i], t, text=halves[i])
    fork_join(body)


def eager_two_threads():
    def body():
        def run(i):
            with torch.cuda.stream(streams[i]):
                engine.forward(xs[i], t, text=halves[i])
        th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
        for h in th:
            h.start()
        for h in th:
            h.join()
    fork_join(body)


b2 = timeit(lambda: engine.forward(x, t, text=text))
b1 = timeit(lambda: engine.forward(xs[0], t, text=halves[0]))
e1 = timeit(eager_one_thread)
e2 = timeit(eager_two_threads)
print(f"batch-2 one pass {b2:.2f} ms | one batch-1 pass alone {b1:.2f} ms | two batch-1 passes on two streams: one host thread {e1:.2f} ms, two host threads {e2:.2f} ms", flush=True)

graphs, outs = [], []
for i in range(2):
    gr = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        engine.forward(xs[i], t, text=halves[i])
    main.wait_stream(side)
    with torch.cuda.graph(gr):
        outs.append(engine.forward(xs[i], t, text=halves[i]))
    graphs.append(gr)
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    out2 = engine.forward(x, t, text=text)
torch.cuda.synchronize()


def graphs_two_streams():
    def body():
        for i in range(2):
            with torch.cuda.stream(streams[i]):
                graphs[i].replay()
    fork_join(body)


gb2 = timeit(g2.replay)
gs = timeit(lambda: (graphs[0].replay(), graphs[1].replay()))
gp = timeit(graphs_two_streams)
print("graph halves equal eager halves:", [bool(torch.equal(outs[i], r1[i])) for i in range(2)])
print(f"graphs: batch-2 replay {gb2:.2f} ms | two batch-1 graphs back to back on one stream {gs:.2f} ms | on two streams {gp:.2f} ms")
