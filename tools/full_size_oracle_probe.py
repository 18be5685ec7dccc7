"""Developer probe (GPU box, host side only): how many threads the fp32 oracle wants at the headline size, and the bf16-storage noise floor
of ONE full-size guidance update (oracle/bf16_storage.py) — the yardstick for tests/test_full_size_gpu.py's update bound."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd.weights import UNetConfig, synthetic_state_dict
from oracle import bf16_storage, guidance_ref, scheduler_ref, unet_ref

cfg = UNetConfig()
sd = synthetic_state_dict(cfg, seed=0)
gen = torch.Generator().manual_seed(0)
x = torch.randn(2, 4, 24, 40, 72, generator=gen)
ehs = torch.randn(2, 77, cfg.cross_attention_dim, generator=gen)
for n in (32, 64, 16):
    torch.set_num_threads(n)
    t0 = time.time()
    with torch.no_grad():
        unet_ref.unet_forward(sd, cfg, x, 500, ehs)
    print(f"threads {n}: full-size CFG forward {time.time() - t0:.1f} s", flush=True)
if "--floor" in sys.argv:
    torch.set_num_threads(32)
    KEYS = [("down", 1, 0, 0), ("down", 2, 0, 0), ("down", 2, 1, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 2, 2, 0)]
    gen = torch.Generator().manual_seed(1)
    lat0 = torch.randn(1, 4, 24, 40, 72, generator=gen)
    cond = torch.randn(1, 77, cfg.cross_attention_dim, generator=gen)
    bear = [[0.0 + 0.8301 * f / 23, 0.5, 0.1953 + 0.8301 * f / 23, 0.6953] for f in range(24)]
    ball = [([0.45, 0.7, 0.6, 0.9] if not 9 <= f < 15 else [0.0] * 4) for f in range(24)]
    boxes, pos = [bear, ball], [[2], [7, 8]]
    hp = dict(loss_scale=2.5, loss_threshold=0.0, max_iter=1, max_index_step=10, fg_top_p=0.25, bg_top_p=0.25, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03)
    t0 = time.time()
    d32, _ = bf16_storage.oracle_guidance_update(cfg, sd, lat0, cond, boxes, pos, 801, "fp32", KEYS, base_attn_dim=(40, 72), **hp)
    d16, _ = bf16_storage.oracle_guidance_update(cfg, sd, lat0, cond, boxes, pos, 801, "bf16", KEYS, base_attn_dim=(40, 72), **hp)
    print(f"bf16-storage floor of the full-size guidance update: rel-L2 {((d16 - d32).norm() / d32.norm()).item():.4f} ({time.time() - t0:.0f} s)")
