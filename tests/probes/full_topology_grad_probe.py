"""Where does the bf16 backward's distance to fp32 autograd come from?  One guidance iteration of the full zeroscope topology
(128x128x4 clip) per guidance key: the gradient path gets longer key by key (down1 -> ... -> up2), so noise grows smoothly
with depth while a wrong layer would show as a jump at the first key whose path crosses it."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lvd_amd  # noqa: E402,F401
from lvd_amd import guidance  # noqa: E402
from lvd_amd.engine import HipUNet3D  # noqa: E402
from lvd_amd.weights import UNetConfig, synthetic_state_dict  # noqa: E402
from oracle import guidance_ref, scheduler_ref, unet_ref  # noqa: E402  (test-side diagnostic: lives under tests/ because only tests may use the oracle)

rel = lambda a, b: ((a.double().cpu() - b.double().cpu()).norm() / b.double().cpu().norm()).item()
cfg = UNetConfig()
sd = synthetic_state_dict(cfg, seed=0, device="cuda")
net = HipUNet3D(cfg, sd, device="cuda")
sd = {k: v.cpu() for k, v in sd.items()}
all_keys = [("down", 1, 0, 0), ("down", 2, 0, 0), ("down", 2, 1, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 2, 2, 0)]
sched = scheduler_ref.DPMSolverPP2M()
for size in (16, 32):
    gen = torch.Generator().manual_seed(1)
    lat0 = torch.randn(1, 4, 4, size, size, generator=gen)
    cond = torch.randn(1, 77, cfg.cross_attention_dim, generator=gen)
    boxes, pos = [[[0.1 + 0.1 * f, 0.2, 0.6 + 0.1 * f, 0.8] for f in range(4)], [[0.5, 0.5, 1.0, 1.0]] * 2 + [[0.0] * 4] * 2], [[2], [5, 6]]
    for keys in [[k] for k in all_keys] + [all_keys]:
        hp = dict(loss_scale=5.0, loss_threshold=0.01, max_index_step=10, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0,
                  com_loss_scale=0.03, guidance_attn_keys=keys)

        def unet_fn(x, tt, c, save, save_keys):
            unet_ref.unet_forward(sd, cfg, x, int(tt), c, save_attn_to_dict=save, save_keys=save_keys, stop_after_key=keys[-1])

        ref_lat, ref_loss = guidance_ref.latent_backward_guidance(unet_fn, sched.alphas_cumprod, cond, 0, boxes, pos, 801, lat0.clone(), 10000.0,
                                                                  max_iter=1, base_attn_dim=(size, size), **hp)
        lat, loss = guidance.hip_latent_backward_guidance(sched, net, cond.cuda(), 0, boxes, pos, 801, lat0.clone().cuda(), torch.tensor(10000.0),
                                                          max_iter=1, **hp)
        d, dr = lat.cpu() - lat0, ref_lat - lat0
        cos = torch.nn.functional.cosine_similarity(d.flatten().double(), dr.flatten().double(), dim=0).item()
        print(f"latent {size}x{size} keys {['_'.join(map(str, k)) for k in keys]}: loss {float(loss):.4f} vs {ref_loss:.4f}, update rel-L2 {rel(d, dr):.4f}, "
              f"cos {cos:.5f}, |d| ratio {(d.norm() / dr.norm()).item():.4f}", flush=True)
