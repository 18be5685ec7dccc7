"""Developer probe: host-side (Python) cost of one guided step — cProfile around the step with the GPU running async."""
import sys, os, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd
from lvd_amd import guidance, ops
from lvd_amd.engine import HipUNet3D
from lvd_amd.sampler import DPMSolverPP2MSchedule, HipSampler
from lvd_amd.weights import UNetConfig, synthetic_state_dict
import bench

cfg = UNetConfig()
engine = HipUNet3D(cfg, synthetic_state_dict(cfg, seed=0, device="cuda"))
g = torch.Generator(device="cuda").manual_seed(0)
latents = torch.randn(1, 4, 24, 40, 72, device="cuda", generator=g)
ehs = torch.randn(2, 77, 1024, device="cuda", generator=g)
text_cfg, text_cond = engine.encode_text(ehs), engine.encode_text(ehs[1:2])
bboxes, positions = bench.demo_layout()
sched = DPMSolverPP2MSchedule(); sched.set_timesteps(40)
sampler = HipSampler(engine, sched); sampler.reset(latents)
hp = dict(loss_scale=2.5, fg_top_p=0.25, bg_top_p=0.25, fg_weight=1.0, bg_weight=2.0)
def step():
    sched.step_index, sched.lower_order_nums = 1, 1
    t = int(sched.timesteps[1])
    loss, grad = guidance.guidance_loss_and_grad(engine, latents, t, text_cond, bboxes, positions, bench.GUIDANCE_KEYS, **hp)
    sampler.cfg_step(latents.clone(), 1, text_cfg)
for _ in range(3): step()
torch.cuda.synchronize()
# host-only time: issue one step and measure until the last launch returns (GPU still busy)
t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host issue time {1e3*(t1-t0):.1f} ms, until GPU done {1e3*(t2-t0):.1f} ms")
pr = cProfile.Profile(); pr.enable(); step(); pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
