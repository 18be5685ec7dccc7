"""Per-shape roofline model of the plain (linear-loader) GEMMs of one guided 576x320x24 step, next to what the shipped kernels measure:
    python tools/gemm_roofline_model.py [--table profiles/gemm_autotune_576x320x24.json] > profiles/r05_gemm_roofline_model.txt
For every (M, N, K, epilogue) of the step: launches, measured us per launch (HIP events around each launch, the shipped autotune table pinned),
T_mfma = 2MNK / 1.25 PF/s (the random-operand matrix-pipe wall of this chip), T_mem = algorithmic bytes / 5.6 TB/s (the rate a
LayerNorm streams a matrix at: A + W + output, + the residual where there is one), their SUM (what a kernel pays when its memory phase and its
matrix phase do not overlap) and their MAX (the bound), and where the measurement sits between the two.  Roll-ups by the families VERDICT r4 names."""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import lvd_amd  # noqa: F401
from lvd_amd import guidance, ops
from lvd_amd.engine import HipUNet3D
from lvd_amd.sampler import DPMSolverPP2MSchedule, HipSampler
from lvd_amd.weights import UNetConfig, synthetic_state_dict
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--table", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "gemm_autotune_576x320x24.json"))
args = ap.parse_args()
if args.table and os.path.exists(args.table):
    ops.load_gemm_autotune_table(args.table)
PF, TBS = 1.25e15, 5.6e12

cfg = UNetConfig()
engine = HipUNet3D(cfg, synthetic_state_dict(cfg, seed=0, device="cuda"))
g = torch.Generator(device="cuda").manual_seed(0)
latents = torch.randn(1, 4, 24, 40, 72, device="cuda", generator=g)
ehs = torch.randn(2, 77, 1024, device="cuda", generator=g)
text_cfg, text_cond = engine.encode_text(ehs), engine.encode_text(ehs[1:2])
bboxes, positions = bench.demo_layout()
sched = DPMSolverPP2MSchedule.from_ddim_config()
sched.set_timesteps(40)
sampler = HipSampler(engine, sched)
sampler.reset(latents)
hp = dict(loss_scale=2.5, fg_top_p=0.25, bg_top_p=0.25, fg_weight=1.0, bg_weight=2.0)


def step():
    sched.step_index, sched.lower_order_nums = 1, 1
    t = int(sched.timesteps[1])
    guidance.guidance_loss_and_grad(engine, latents, t, text_cond, bboxes, positions, bench.GUIDANCE_KEYS, **hp)
    sampler.cfg_step(latents.clone(), 1, text_cfg)


for _ in range(2):
    step()
rec = []
orig = ops.gemm


def timed(a1, w, **kw):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    out = orig(a1, w, **kw)
    e.record()
    if kw.get("mode", 0) == ops.A_PLAIN:
        N, K = (w.shape if kw.get("n") is None else (kw["n"], kw["k"]))
        rec.append(((out.shape[0], N, K, int(kw.get("act", 0)), kw.get("res") is not None, kw.get("ln_stats") is not None, bool(kw.get("accumulate")),
                     bool(kw.get("out_fp32"))), s, e))
    return out


ops.gemm = timed
for _ in range(3):
    step()
torch.cuda.synchronize()
ops.gemm = orig
agg = collections.OrderedDict()
for key, s, e in rec:
    d = agg.setdefault(key, [0, 0.0])
    d[0] += 1
    d[1] += s.elapsed_time(e) * 1e3
tab = ops.gemm_autotune_table()
rows = []
for (M, N, K, act, res, ln, acc, f32), (n, us) in agg.items():
    n_step, us1 = n / 3, us / n
    nout = N // 2 if act else N
    byts = 2.0 * M * K + 2.0 * N * K + (4.0 if f32 else 2.0) * M * nout + (2.0 * M * nout if res or acc else 0.0)
    t_mfma, t_mem = 2.0 * M * N * K / PF * 1e6, byts / TBS * 1e6
    var = [v for k, v in tab.items() if k[0] == 0 and k[1] == M and k[2] == N and k[3] == K and k[4] == act and k[9] == res and (len(k) > 14) == ln]
    rows.append(dict(M=M, N=N, K=K, act=act, res=res, ln=ln, acc=acc, n=n_step, us=us1, mfma=t_mfma, mem=t_mem, var=var[:1]))
rows.sort(key=lambda r: -r["n"] * r["us"])
print(f"# plain GEMMs of one guided step (guidance iteration + CFG forward), MI355X, table {os.path.relpath(args.table) if args.table else None}; us per launch")
print(f"# T_mfma at {PF / 1e15:.2f} PF/s, T_mem at {TBS / 1e12:.1f} TB/s; pos = (measured - max) / (sum - max): 0 = at the bound, 1 = fully serialised, > 1 = beyond")
print(f"# {'M':>7s} {'N':>6s} {'K':>6s} epi      x/step  measured   T_mfma    T_mem      sum      max    pos   TF/s  variant")
tot = collections.defaultdict(lambda: [0.0, 0.0, 0.0])  # family -> measured ms, sum ms, max ms


def fam(r):
    if r["M"] <= 8640:
        return "M <= 8640 (deep levels)"
    if r["K"] <= 640:
        return "K <= 640, M >= 17280, N >= 960" if r["N"] >= 960 else "K <= 640, M >= 17280, N <= 640"
    return "K > 640, M >= 17280"


for r in rows:
    s_, m_ = r["mfma"] + r["mem"], max(r["mfma"], r["mem"])
    pos = (r["us"] - m_) / max(s_ - m_, 1e-9)
    epi = ("geglu" if r["act"] else "") + ("+res" if r["res"] else "") + ("+ln" if r["ln"] else "") + ("+acc" if r["acc"] else "") or "-"
    print(f"  {r['M']:7d} {r['N']:6d} {r['K']:6d} {epi:8s} {r['n']:6.1f}  {r['us']:8.1f} {r['mfma']:8.1f} {r['mem']:8.1f} {s_:8.1f} {m_:8.1f}  {pos:5.2f}  {2.0 * r['M'] * r['N'] * r['K'] / r['us'] / 1e6:5.0f}  {r['var']}")
    f = tot[fam(r)]
    f[0] += r["n"] * r["us"] / 1e3
    f[1] += r["n"] * s_ / 1e3
    f[2] += r["n"] * m_ / 1e3
print("# family roll-up (ms per guided step): measured | serialised model (sum) | bound (max)")
for k, (a, b, c) in sorted(tot.items()):
    print(f"#   {k:34s} {a:6.2f} | {b:6.2f} | {c:6.2f}")
a, b, c = (sum(v[i] for v in tot.values()) for i in range(3))
print(f"#   {'all plain GEMMs':34s} {a:6.2f} | {b:6.2f} | {c:6.2f}")
