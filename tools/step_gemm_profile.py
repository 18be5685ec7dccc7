"""Developer probe: per-shape GEMM time of one guided step (HIP events around every launch)."""
import sys, os, collections
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
for _ in range(2): step()
rec = []
orig = ops.gemm
def timed(a1, w, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); out = orig(a1, w, **kw); e.record()
    conv = kw.get("conv")
    rec.append(((kw.get("mode", 0), out.shape[0], w.shape[0], w.shape[1], kw.get("act", 0), (conv.stride, conv.upsample) if conv else None,
                 a1.shape[1], bool(kw.get("accumulate"))), s, e))
    return out
ops.gemm = timed
step()
torch.cuda.synchronize()
agg = collections.OrderedDict()
for key, s, e in rec:
    d = agg.setdefault(key, [0, 0.0]); d[0] += 1; d[1] += s.elapsed_time(e)
tot = sum(v[1] for v in agg.values())
print(f"total GEMM ms {tot:.1f} over {len(rec)} launches")
tab = ops.gemm_autotune_table()
for key, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(os.environ.get('TOP', '400'))]:
    mode, M, N, K, act, cv, c1, acc = key
    tf = 2.0 * M * N * K * n / ms / 1e9
    var = [v for k, v in tab.items() if k[0] == mode and k[1] == M and k[2] == N and k[3] == K and k[4] == act]
    print(f"mode{mode} M={M:7d} N={N:6d} K={K:6d} act={act} conv={cv} c1={c1:5d} acc={int(acc)} x{n:3d}  {ms:7.2f} ms  {tf:6.1f} TF/s  variant={var[:1]}")
