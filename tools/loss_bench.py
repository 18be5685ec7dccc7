"""Developer probe: the fused guidance loss (csrc/guidance_loss.hip) on the six keyed layers of the zeroscope 576x320x24 step, the bench's
3-object / 4-token layout: time per key and per iteration, HBM GB/s against the floor (read Q + write dQ of the keyed layers: 177 MB)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import guidance
import bench

dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
F = 24
# (H, W, heads) of generation/lvd.py:66-73's keys: down1 (20x36, 10 heads), down2 x2 / up1 x2 (10x18, 20 heads), up2 (20x36, 10 heads)
layers = [(20, 36, 10), (10, 18, 20), (10, 18, 20), (10, 18, 20), (10, 18, 20), (20, 36, 10)]
bboxes, positions = bench.demo_layout()
ntok = sum(len(p) for p in positions)
qs = [(torch.randn(F * h * w, heads * 64, device=dev, generator=g)).bfloat16() for h, w, heads in layers]
ks = [(torch.randn(77, heads * 64, device=dev, generator=g)).bfloat16() for h, w, heads in layers]
lay = {}
partial = torch.empty(sum(F * heads * ntok for _, _, heads in layers), device=dev)


def items():
    off, out = 0, []
    for (h, w, heads), q, k in zip(layers, qs, ks):
        L = lay.get((h, w)) or lay.setdefault((h, w), guidance.GuidanceLayout(bboxes, positions, F, h, w, 0.25, 0.25, dev))
        n = F * heads * ntok
        out.append((q, k, heads, L, partial[off:off + n]))
        off += n
    return out


KW = dict(ntext=77, grad_scale=1.0, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.0)


def key_by_key():
    for q, k, heads, L, part in items():
        guidance.ca_energy_loss_and_dq(q, k, heads, F, L, loss_partial=part, **KW)


def all_keys():
    guidance.ca_energy_loss_and_dq_all_keys(items(), F, **KW)


mb = sum(2 * q.numel() * 2 for q in qs) / 1e6
for name, fn, launches in (("key by key", key_by_key, 18), ("all keys per launch", all_keys, 3)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    t0 = time.perf_counter()
    for _ in range(20):
        fn()
    host = (time.perf_counter() - t0) / 20 * 1e6
    e.record()
    e.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    print(f"guidance loss, 6 keys, {ntok} object tokens, {name} ({launches} launches): {us:.0f} us per iteration on the device timeline "
          f"(host issue {host:.0f} us per iteration: back-to-back iterations are host-bound when that is the larger figure; the step issues "
          f"the loss behind a long forward); algorithmic traffic {mb:.0f} MB (read Q + write dQ) -> {mb / us:.3f} TB/s = {mb / us / 8.0:.3f} "
          f"of the 8 TB/s HBM figure")
