"""Developer probe: HIP guidance-update error vs the bf16-storage noise floor of the oracle across timesteps, layouts and seeds."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import lvd_amd
from lvd_amd import guidance
from lvd_amd.engine import HipUNet3D
from lvd_amd.weights import TINY, UNetConfig, synthetic_state_dict
from oracle import bf16_storage, scheduler_ref

cfg = UNetConfig(**TINY)
sd = synthetic_state_dict(cfg, seed=0)
net = HipUNet3D(cfg, sd)
keys = [("down", 1, 0, 0), ("up", 1, 1, 0), ("up", 2, 1, 0)]
hp = dict(loss_scale=5.0, loss_threshold=0.01, max_iter=1, max_index_step=10, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03)
rel = lambda a, b: ((a.double() - b.double()).norm() / b.double().norm()).item()
layouts = {"1box": ([[[0.1, 0.2, 0.6, 0.8]] * 4], [[2]]),
           "2obj": ([[[0.1 + 0.05 * f, 0.2, 0.6 + 0.05 * f, 0.8] for f in range(4)], [[0.5, 0.5, 1.0, 1.0]] * 4], [[2], [5, 6]])}
for t in (999, 801, 401):
    for lname, (boxes, pos) in layouts.items():
        for seed in range(3):
            g = torch.Generator().manual_seed(seed)
            lat = torch.randn(1, 4, 4, 16, 16, generator=g)
            cond = torch.randn(1, 77, cfg.cross_attention_dim, generator=g)
            u32, l32 = bf16_storage.oracle_guidance_update(cfg, sd, lat, cond, boxes, pos, t, "fp32", keys, **hp)
            u16, _ = bf16_storage.oracle_guidance_update(cfg, sd, lat, cond, boxes, pos, t, "bf16", keys, **hp)
            new, loss = guidance.hip_latent_backward_guidance(scheduler_ref.DPMSolverPP2M(), net, cond.cuda(), 0, boxes, pos, t, lat.clone().cuda(),
                                                              torch.tensor(10000.0), guidance_attn_keys=keys, **hp)
            d = new.cpu() - lat
            print(f"t={t} {lname} seed={seed}: floor {rel(u16, u32):.4f}  HIP-vs-fp32 {rel(d, u32):.4f}  HIP-vs-bf16emu {rel(d, u16):.4f}  |u| {u32.norm():.3e}", flush=True)
