"""Developer probe: time the full-size CFG forward (not the official bench)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd
from lvd_amd.engine import HipUNet3D
from lvd_amd.weights import UNetConfig, synthetic_state_dict

gated = "--gated" in sys.argv
cfg = UNetConfig(attention_type="gated" if gated else "default")
t0 = time.time()
sd = synthetic_state_dict(cfg, seed=0, device="cuda")
net = HipUNet3D(cfg, sd)
del sd
torch.cuda.synchronize()
print("weights ready %.1fs, mem %.1f GB" % (time.time() - t0, torch.cuda.memory_allocated() / 1e9))
B, F, H, W = 2, 24, 40, 72
g = torch.Generator(device="cuda").manual_seed(0)
x = torch.randn(B, 4, F, H, W, device="cuda", generator=g)
ehs = torch.randn(B, 77, 1024, device="cuda", generator=g)
text = net.encode_text(ehs)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    out = net.forward(x, 500, text=text)
    t1 = time.time()
    torch.cuda.synchronize(); t2 = time.time()
    print("fwd %d: host %.1f ms, total %.1f ms, peak mem %.1f GB  finite=%s absmean=%.3f" % (it, (t1 - t0) * 1e3, (t2 - t0) * 1e3,
          torch.cuda.max_memory_allocated() / 1e9, bool(torch.isfinite(out).all()), out.abs().mean().item()))
print("TFLOP/s at 42.79 TF: %.1f" % (42.79 / (t2 - t0)))
