"""Developer probe: VAE decode of one 24-frame 576x320 video (random-init SD-VAE decoder) — time and kernel mix."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd
from lvd_amd.vae import HipVAEDecoder
from lvd_amd.weights import VAEConfig, synthetic_vae_state_dict

cfg = VAEConfig()
dec = HipVAEDecoder(cfg, synthetic_vae_state_dict(cfg, seed=0, device="cuda"))
lat = torch.randn(1, 4, 24, 40, 72, device="cuda") * cfg.scaling_factor
for _ in range(2):
    dec(lat)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    v = dec(lat)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
# decoder FLOPs: 2*MAC over convs / linears / the mid attention (counted from the shapes below)
def flops():
    tot = 0
    h, w, n = 40, 72, 24
    c = 512
    conv = lambda cin, cout, hh, ww, k=9: 2 * n * hh * ww * cin * cout * k
    tot += conv(4, 512, h, w) + 4 * conv(512, 512, h, w)                      # conv_in + 2 mid resnets
    tot += 4 * conv(512, 512, h, w, 1) + 2 * 2 * n * (h * w) ** 2 * 512       # attention projections + QK^T / PV
    chans = [(512, 512), (512, 512), (512, 256), (256, 128)]
    for i, (cin, cout) in enumerate(chans):
        tot += conv(cin, cout, h, w) + conv(cout, cout, h, w) + 4 * conv(cout, cout, h, w)
        if cin != cout:
            tot += conv(cin, cout, h, w, 1)
        if i != 3:
            h, w = 2 * h, 2 * w
            tot += conv(cout, cout, h, w)
    tot += conv(128, 3, h, w)
    return tot
tf = flops() / 1e12
print(f"VAE decode 24x320x576: {dt * 1e3:.1f} ms  ({tf:.1f} TFLOP -> {tf / dt:.0f} TFLOP/s, {24 / dt:.0f} frames/s)")
