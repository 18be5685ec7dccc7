#!/usr/bin/env python
"""Headline benchmark: denoise-step frames/s, LVD-Zeroscope 576x320x24 with guidance (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one GUIDED denoising step of one video, exactly what the reference's loop body does for step index < 10
(models/controllable_pipeline_text_to_video_synth.py:836-950 with the README weak-guidance setting max_iter=1):
    guidance iteration  = recorded cond-branch UNet forward (B=1) up to the last guidance key + fused
                          cross-attention-energy loss + hand-scheduled backward to the latents + latent update
    CFG UNet forward    = batch 2 (uncond, cond), full UNet
    CFG combine + DPM-Solver++(2M) update
on the real topology (UNet3DConditionModel defaults = zeroscope, 1411 M params, random-init weights — no network for
checkpoints), bf16 storage / fp32 accumulation, synthetic latents and text states resident in HBM before timing.
Each rank samples its own video (independent (prompt, seed) samples shard with no data-path collective): weak scaling,
value = N * frames / max-over-ranks(step time).
Also reported: the unguided step, the 40-step schedule mean (10 guided + 30 unguided), the MFMA roofline of the dominant
kernel (HIP-event timed inside this script) and a CPU baseline (the fp32 oracle on the host cores, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import lvd_amd  # noqa: E402
from lvd_amd import guidance, ops  # noqa: E402
from lvd_amd.engine import HipUNet3D  # noqa: E402
from lvd_amd.sampler import DPMSolverPP2MSchedule, HipSampler  # noqa: E402
from lvd_amd.weights import UNetConfig, synthetic_state_dict  # noqa: E402

# algorithmic work per unit (BASELINE.md §2 / SURVEY §8d; counted with torch.utils.flop_counter on the reference module)
TF_CFG_FWD = 42.79          # unguided step: CFG forward, B=2
TF_GUIDANCE_ITER = 31.8     # 15.92 fwd (to the last key) + ~15.92 dgrad backward
PEAK_BF16_TFLOPS = 2500.0   # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
# What the shared CFG prefix (engine.forward(cfg_pairs=True)) does NOT execute a second time, per video: conv_in 0.0016 + transformer_in 0.777
# (per token 2*320*512 + 2 x (2*512*1536 + 2*512*512 + 4*24*512) + 2*512*4096 + 2*2048*512 + 2*512*320) + the first ResnetBlock2D 0.2548
# (two 3x3 convs 320 -> 320) + the first TemporalConvLayer 0.1699 (four (3,1,1) convs) + proj_in / to_qkv / 2880-key self-attention / to_out of
# the first spatial transformer 0.3256 (per token 2*320*320 + 2*320*960 + 4*2880*320 + 2*320*320), all x 69120 token rows
TF_CFG_SHARED_PREFIX = 1.53
GUIDANCE_KEYS = [("down", 1, 0, 0), ("down", 2, 0, 0), ("down", 2, 1, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 2, 2, 0)]  # generation/lvd.py:66-73
FRAMES, LAT_H, LAT_W = 24, 40, 72


def demo_layout():
    """cache/cache_demo_v0.1_gpt-4-1106-preview.json: one 'bear' box moving left->right, token index 2,
    plus two synthetic objects (one disappears for 6 frames) so the multi-object path is timed too."""
    bear = [[0.0 + 0.8301 * f / 23, 0.5, 0.1953 + 0.8301 * f / 23, 0.6953] for f in range(FRAMES)]
    bird = [[0.6 - 0.4 * f / 23, 0.1, 0.85 - 0.4 * f / 23, 0.35] for f in range(FRAMES)]
    ball = [([0.45, 0.7, 0.6, 0.9] if not 9 <= f < 15 else [0.0, 0.0, 0.0, 0.0]) for f in range(FRAMES)]
    return [bear, bird, ball], [[2], [7, 8], [12]]


class GemmTimer:
    """HIP-event timing of every MFMA GEMM class (linear loader, 3x3 conv, temporal conv, transposed conv) over the timed
    region; the dominant class is the one with the most time.  Events are recorded on torch's current stream, which is the
    stream the kernels are launched on.  An event pair is a barrier packet on each side of the launch (~3 us of idle GPU
    each, measured in the rocprof trace: 6 % of the step when every launch is bracketed), so every `stride`-th GEMM launch
    is bracketed; a step has 1133 GEMM launches (1133 % 8 == 5, coprime with 8), so the sampled positions rotate every step
    and 8 timed steps cover every launch site exactly once."""

    def __init__(self, modes=(ops.A_PLAIN, ops.A_CONV3X3, ops.A_TCONV3, ops.A_CONV3X3_T2), stride=8):
        self.rec = []
        self.orig = ops.gemm
        self.modes = modes
        self.stride = stride
        self.count = 0
        self.per_mode = {}

    def __enter__(self):
        def timed(a1, w, **kw):
            mode = kw.get("mode", ops.A_PLAIN)
            if mode not in self.modes:
                return self.orig(a1, w, **kw)
            self.count += 1
            self.per_mode[mode] = self.per_mode.get(mode, 0) + 1
            if self.count % self.stride:
                return self.orig(a1, w, **kw)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = self.orig(a1, w, **kw)
            e.record()
            N, K = w.shape
            self.rec.append((mode, s, e, 2.0 * out.shape[0] * N * K))
            return out
        ops.gemm = timed
        return self

    def __exit__(self, *a):
        ops.gemm = self.orig

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for mode, s, e, fl in self.rec:
            d = agg.setdefault(mode, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += s.elapsed_time(e) * 1e-3
            d[2] += fl
        return agg


def _pick_threads(sd_cpu, cfg, ehs, g):
    """The oracle is memory- and launch-bound on many-core hosts: all hardware threads is rarely the fastest setting (round 2 ran 128
    threads at 0.11-0.21 TFLOP/s where 8 cores reach 0.6).  One small CFG forward per candidate thread count, keep the fastest."""
    from oracle import unet_ref
    hw = os.cpu_count() or 8
    cands = sorted({c for c in (8, 16, 32, 64) if 1 <= c <= hw} or {hw})  # beyond 64 the probe itself takes minutes on a 256-thread host (128: 18 s, 256: 343 s)
    x = torch.randn(2, 4, 4, 18, 32, generator=g)
    timings = {}
    with torch.no_grad():
        torch.set_num_threads(cands[-1])
        unet_ref.unet_forward(sd_cpu, cfg, x, 500, ehs)  # page the weights in once, outside the comparison
        for c in cands:
            torch.set_num_threads(c)
            t0 = time.time()
            unet_ref.unet_forward(sd_cpu, cfg, x, 500, ehs)
            timings[c] = time.time() - t0
    best = min(timings, key=timings.get)
    torch.set_num_threads(best)
    return best, {str(k): round(v, 2) for k, v in timings.items()}


def cpu_baseline(sd_cpu, cfg, budget_s=25.0, loop_budget_s=70.0):
    """fp32 oracle (restatement of the reference, oracle/) on the host cores, three bounded legs, at the thread count that
    maximises its throughput (measured first, printed):
      A. BASELINE config 0 end to end: 256x144x8 (latent 18x32), DPM-Solver++ CFG sampling loop, all 10 steps unless the leg's own budget
         (`loop_budget_s`, stated in the line) runs out — the steps are identical work, the per-step time is what is reported;
      B. ONE CFG UNet forward at the largest of (8x18x32, 12x24x40, 16x32x32, 24x40x72) predicted to fit the budget;
      C. ONE guidance iteration (cond-branch forward with saved attention maps + compute_ca_loss + autograd backward to the
         latents) at the largest size predicted to fit the budget — a different FLOP/s regime than a forward.
    `value` extrapolates the guided 576x320x24 step from B (CFG forward part) and C (guidance part) by algorithmic FLOPs,
    each leg with its own measured rate; every extrapolation factor is spelled out in `sample`."""
    from oracle import guidance_ref, scheduler_ref, unet_ref
    g = torch.Generator().manual_seed(0)
    fl_per_cell = TF_CFG_FWD / (2 * FRAMES * LAT_H * LAT_W)       # TFLOP per (batch item, frame, latent pixel) of a forward
    gfl_per_cell = TF_GUIDANCE_ITER / (FRAMES * LAT_H * LAT_W)    # guidance iteration (fwd to the last key + backward), B=1
    ehs = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    cores, thread_probe = _pick_threads(sd_cpu, cfg, ehs, g)

    # ---- A: config 0 loop
    sch = scheduler_ref.DPMSolverPP2M()
    sch.set_timesteps(10)
    lat = torch.randn(1, 4, 8, 18, 32, generator=g)
    t0 = time.time()
    done = 0
    with torch.no_grad():
        for i, t in enumerate(sch.timesteps):
            eps = unet_ref.unet_forward(sd_cpu, cfg, lat.expand(2, -1, -1, -1, -1), int(t), ehs)
            lat = sch.step(eps[0:1] + 9.0 * (eps[1:2] - eps[0:1]), lat)
            done += 1
            if time.time() - t0 > loop_budget_s:
                break
    t_a = (time.time() - t0) / done
    tf_a = 2 * 8 * 18 * 32 * fl_per_cell
    rate_fwd = tf_a / t_a

    # ---- B: one CFG forward at the largest size that fits the budget
    sizes = [(8, 18, 32), (12, 24, 40), (16, 32, 32), (24, 40, 72)]
    fit = [z for z in sizes if 2 * z[0] * z[1] * z[2] * fl_per_cell / rate_fwd <= budget_s] or sizes[:1]
    fb = fit[-1]
    x = torch.randn(2, 4, *fb, generator=g)
    with torch.no_grad():
        t0 = time.time()
        unet_ref.unet_forward(sd_cpu, cfg, x, 500, ehs)
        t_b = time.time() - t0
    tf_b = 2 * fb[0] * fb[1] * fb[2] * fl_per_cell
    rate_b = tf_b / t_b

    # ---- C: one guidance iteration (autograd through the oracle)
    sizes_c = [(8, 16, 32), (12, 24, 40), (16, 32, 32), (24, 40, 72)]  # latent H, W divisible by 8 (the reference's attention-map geometry, utils/guidance.py)
    fitc = [z for z in sizes_c if z[0] * z[1] * z[2] * gfl_per_cell / (0.6 * rate_fwd) <= budget_s] or sizes_c[:1]
    fc = fitc[-1]
    bboxes, positions = demo_layout()
    bboxes = [bx[:fc[0]] for bx in bboxes]
    latc = torch.randn(1, 4, *fc, generator=g)

    def unet_fn(xx, tt, cond, save, save_keys):
        unet_ref.unet_forward(sd_cpu, cfg, xx, int(tt), cond, save_attn_to_dict=save, save_keys=save_keys, stop_after_key=GUIDANCE_KEYS[-1])

    t0 = time.time()
    guidance_ref.latent_backward_guidance(unet_fn, sch.alphas_cumprod, ehs[1:2], 0, bboxes, positions, 500, latc, 10000.0, loss_scale=2.5,
                                          loss_threshold=0.0, max_iter=1, max_index_step=10, guidance_attn_keys=GUIDANCE_KEYS,
                                          base_attn_dim=(fc[1], fc[2]), fg_top_p=0.25, bg_top_p=0.25, fg_weight=1.0, bg_weight=2.0)
    t_c = time.time() - t0
    tf_c = fc[0] * fc[1] * fc[2] * gfl_per_cell
    rate_c = tf_c / t_c

    t_guided = TF_CFG_FWD / rate_b + TF_GUIDANCE_ITER / rate_c
    return {"value": round(FRAMES / t_guided, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "host_threads_available": os.cpu_count(), "thread_probe_s": thread_probe,
            "legs": {"config0_loop": {"steps_run": done, "of": 10, "budget_s": loop_budget_s, "s_per_step": round(t_a, 2), "tflops": round(rate_fwd, 3),
                                      "frames_per_s": round(8 / t_a, 3), "workload": "256x144x8, CFG DPM-Solver++ step, fp32"},
                     "cfg_forward": {"frames_h_w": list(fb), "s": round(t_b, 2), "tflop": round(tf_b, 2), "tflops": round(rate_b, 3)},
                     "guidance_iteration": {"frames_h_w": list(fc), "s": round(t_c, 2), "tflop": round(tf_c, 2), "tflops": round(rate_c, 3)}},
            "sample": f"fp32 oracle on {cores} threads (fastest of {thread_probe} s per probe forward): (A) config 0 loop 256x144x8: {done}/10 CFG steps, {t_a:.1f} s/step = {rate_fwd:.3f} TFLOP/s; "
                      f"(B) one CFG forward at {fb[0]}x{fb[1]}x{fb[2]} latents ({tf_b:.2f} TFLOP) {t_b:.1f} s = {rate_b:.3f} TFLOP/s; "
                      f"(C) one guidance iteration with autograd at {fc[0]}x{fc[1]}x{fc[2]} ({tf_c:.2f} TFLOP) {t_c:.1f} s = {rate_c:.3f} TFLOP/s; "
                      f"guided 576x320x24 step extrapolated by algorithmic FLOPs: {TF_CFG_FWD}/{rate_b:.3f} + {TF_GUIDANCE_ITER}/{rate_c:.3f} s = {t_guided:.0f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unguided-steps", type=int, default=4, help="extra (untimed-for-value) unguided steps for the breakdown")
    ap.add_argument("--gligen", action="store_true", help="BASELINE config 3 instead of the default config 2: gated topology (1624M params), "
                    "GLIGEN fusers on in the CFG forward (52.88 TFLOP); not the headline metric")
    ap.add_argument("--gemm_autotune_table", default=os.path.join(ROOT, "profiles", "gemm_autotune_576x320x24.json"),
                    help="per-shape GEMM tile-geometry choices (shipped: measured once on MI355X): the same kernels, and the same bf16 bits, in every "
                         "run and rank; shapes the table does not hold are tuned on first use")
    ap.add_argument("--retune", action="store_true", help="ignore the table and time the candidates again (--save_autotune_table writes the result)")
    ap.add_argument("--save_autotune_table", default=None)
    ap.add_argument("--guidance-one-by-one", action="store_true", help="throughput mode A/B: the V guidance passes as V batch-1 passes (round 4) instead of one batch-V pass")
    ap.add_argument("--videos-per-gpu", type=int, default=1, help="throughput mode, NOT the headline: V independent (prompt, seed) samples per GPU, "
                    "their CFG forwards batched (B = 2V); guidance stays one recorded pass per sample")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher — one rank per GPU under torch.distributed.run on this node (rendezvous on
        # 127.0.0.1; the container hostname may not resolve).  Rank 0 of the children prints the one JSON line; the exit code is theirs.
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        # stdout of this command is the ONE JSON line of rank 0: anything else the ranks' libraries write there (gloo's "[Gloo] Rank 0 is
        # connected to ..." banner of the rehearsal backend, for one) is passed on to stderr
        child = subprocess.Popen(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")),
                                 stdout=subprocess.PIPE, text=True, bufsize=1)
        for line in child.stdout:
            (sys.stdout if line.lstrip().startswith("{") else sys.stderr).write(line)
            sys.stdout.flush()
        sys.exit(child.wait())

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local %= torch.cuda.device_count()  # one GPU per rank on a node; the modulo only matters for the single-GPU rehearsal below
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    backend = os.environ.get("LVD_BENCH_BACKEND", "nccl")  # "gloo": rehearse the N>1 control flow with all ranks on one GPU
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # RCCL over xGMI
        else:
            dist.init_process_group(backend)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run for N>1)"

    table_loaded = None
    if not args.retune and args.gemm_autotune_table and os.path.exists(args.gemm_autotune_table):
        ops.load_gemm_autotune_table(args.gemm_autotune_table)
        table_loaded = os.path.relpath(args.gemm_autotune_table, ROOT)
    V = args.videos_per_gpu
    cfg = UNetConfig(attention_type="gated") if args.gligen else UNetConfig()
    sd = synthetic_state_dict(cfg, seed=0, device=dev)
    engine = HipUNet3D(cfg, sd, device=dev)
    sd_cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        sd_cpu = {k: v.float().cpu() for k, v in sd.items()}
    del sd

    g = torch.Generator(device=dev).manual_seed(1234 + rank)  # each rank = its own (prompt, seed) sample
    latents = torch.randn(1, 4, FRAMES, LAT_H, LAT_W, device=dev, generator=g)
    ehs = torch.randn(2, 77, cfg.cross_attention_dim, device=dev, generator=g)  # [negative; positive]
    text_cfg = engine.encode_text(ehs)
    text_cond = engine.encode_text(ehs[1:2])
    # throughput mode: V - 1 further samples with their own latents and prompts; CFG batch = [uncond_0, cond_0, uncond_1, cond_1, ...]
    more = [(torch.randn(1, 4, FRAMES, LAT_H, LAT_W, device=dev, generator=g), torch.randn(2, 77, cfg.cross_attention_dim, device=dev, generator=g))
            for _ in range(V - 1)]
    if V > 1:
        assert not args.gligen, "--videos-per-gpu with --gligen is not wired"
        text_cfg_all = engine.encode_text(torch.cat([ehs] + [e for _, e in more]))
        text_cond_more = [engine.encode_text(e[1:2]) for _, e in more]
        text_cond_all = engine.encode_text(torch.cat([ehs[1:2]] + [e[1:2] for _, e in more]))  # the V cond embeddings: one batch-V guidance pass
        x0_prev_more = [torch.zeros_like(l) for l, _ in more]
    bboxes, positions = demo_layout()
    gligen = None
    if args.gligen:  # controllable_pipeline_text_to_video_synth.py:736-814: 30 slots per frame, [unconditional; conditional]
        gb = torch.zeros(2 * FRAMES, 30, 4)
        gm = torch.zeros(2 * FRAMES, 30)
        gb[FRAMES:, :len(bboxes)] = torch.tensor(bboxes).permute(1, 0, 2)
        gm[FRAMES:, :len(bboxes)] = 1.0
        gligen = {"boxes": gb, "masks": gm, "positive_embeddings": torch.randn(2 * FRAMES, 30, cfg.cross_attention_dim, device=dev, generator=g).cpu()}
    sched = DPMSolverPP2MSchedule.from_ddim_config()  # the reference's effective schedule: t = 961, 937, ..., 25
    sched.set_timesteps(40)
    sampler = HipSampler(engine, sched, guidance_scale=9.0)
    sampler.reset(latents)
    hp = dict(loss_scale=2.5, fg_top_p=0.25, bg_top_p=0.25, fg_weight=1.0, bg_weight=2.0)  # README.md:68 weak guidance

    state = {"i": 0, "loss": 10000.0, "loss_more": [10000.0] * (V - 1)}

    gkw = dict(loss_scale=hp["loss_scale"], loss_threshold=0.0, max_iter=1, max_index_step=10, guidance_attn_keys=GUIDANCE_KEYS,
               **{k: v for k, v in hp.items() if k != "loss_scale"})

    def cfg_all(i):
        """CFG forward + fused update of all V samples in one batch-2V pass."""
        t = int(sched.timesteps[i])
        lats = [latents] + [l for l, _ in more]
        eps = engine.forward_cfg(torch.cat(lats).contiguous(), t, text=text_cfg_all)
        a_t, s_t, c_x, c_0, c_1 = sched.coefficients(i)
        for v, (l, xp) in enumerate(zip(lats, [sampler.x0_prev] + x0_prev_more)):
            ops.cfg_dpm_step(eps[2 * v:2 * v + 1], eps[2 * v + 1:2 * v + 2], sampler.guidance_scale, l, xp, a_t, s_t, c_x, c_0, c_1)
        sched.advance()

    def guided_step():
        """The product's own step: `hip_latent_backward_guidance` (recorded forward, fused loss, hand-written backward, latent update; the
        reference's loss.item() of models/pipelines.py:134 is read by the next step's entry check) followed by the CFG forward and the
        fused CFG / DPM-Solver++ update."""
        i = state["i"] % 10  # guidance is active for step indices < max_index_step=10
        sched.step_index, sched.lower_order_nums = i, min(i, 2)
        t = int(sched.timesteps[i])
        # the carried loss is the tensor the previous guided step returned (controllable_pipeline_text_to_video_synth.py keeps `loss` across
        # steps): its entry check waits for the pinned-memory copy queued behind the previous backward, so that wait is inside the timed region
        if V > 1 and not args.guidance_one_by_one:
            # what pipeline.sample_many runs for V samples with the stock guidance function: ONE recorded forward / backward of batch V
            lats, losses = guidance.hip_latent_backward_guidance_many(sched, engine, text_cond_all, i, [bboxes] * V, [positions] * V, t,
                                                                      [latents] + [l for l, _ in more], [state["loss"]] + state["loss_more"], **gkw)
            for dst, new in zip([latents] + [l for l, _ in more], lats):
                dst.copy_(new)
            state["loss"], state["loss_more"] = losses[0], list(losses[1:])
            loss = losses[0]
        else:
            new, loss = guidance.hip_latent_backward_guidance(sched, engine, text_cond, i, bboxes, positions, t, latents, state["loss"], **gkw)
            latents.copy_(new)
            state["loss"] = loss
            for v, ((l, _), tc) in enumerate(zip(more, text_cond_more if V > 1 else [])):
                nl, lv = guidance.hip_latent_backward_guidance(sched, engine, tc, i, bboxes, positions, t, l, state["loss_more"][v], **gkw)
                l.copy_(nl)
                state["loss_more"][v] = lv
        if V > 1:
            cfg_all(i)
        else:
            sampler.cfg_step(latents, i, text_cfg, gligen=gligen)
        state["i"] += 1
        return loss

    def unguided_step():
        i = 10 + state["i"] % 29
        sched.step_index, sched.lower_order_nums = i, 2
        if V > 1:
            cfg_all(i)
        else:
            sampler.cfg_step(latents, i, text_cfg, gligen=gligen)  # --gligen: fusers on, as in steps 10..15 of the 40 (beta = 0.4)
        state["i"] += 1

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def keep_finite():
        # random-init weights are not a denoiser: re-draw the latents so timing never runs on inf/nan (untimed)
        for v, l in enumerate([latents] + [m for m, _ in more]):
            xp = sampler.x0_prev if v == 0 else x0_prev_more[v - 1]
            if not bool(torch.isfinite(l).all()) or float(l.abs().max()) > 50 or not bool(torch.isfinite(xp).all()):
                l.copy_(torch.randn(l.shape, device=dev, generator=g))
                if v == 0:
                    sampler.reset(latents)      # a re-drawn sample starts a fresh multistep history
                    state["loss"] = 10000.0
                else:
                    xp.zero_()
                    state["loss_more"][v - 1] = 10000.0

    guided_step()  # untimed preparation, independent of --warmup: shapes the autotune table does not hold are tuned on first use
    unguided_step()
    keep_finite()
    if args.save_autotune_table and rank == 0:
        ops.save_gemm_autotune_table(args.save_autotune_table)
    for _ in range(args.warmup):
        guided_step()
        keep_finite()
    # timed region: exactly K guided steps; every 8th launch of the dominant GEMM class is bracketed by HIP events on the
    # launch stream (GemmTimer; no extra synchronisation)
    gt = GemmTimer()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]  # step boundaries on the launch stream
    guidance.loss_kernel_events = []  # the fused guidance-loss launches (3 per iteration) between events on the launch stream
    from lvd_amd import hip as _hip
    sync()
    calls0 = _hip.calls
    t0 = time.perf_counter()
    with gt:
        marks[0].record()
        for k in range(args.steps):
            last_loss = guided_step()
            marks[k + 1].record()
    sync()
    dt = time.perf_counter() - t0
    abi_calls_per_step = (_hip.calls - calls0) / args.steps
    loss_events, guidance.loss_kernel_events = guidance.loss_kernel_events, None
    per_step = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(args.steps))
    med = per_step[len(per_step) // 2] if len(per_step) % 2 else 0.5 * (per_step[len(per_step) // 2 - 1] + per_step[len(per_step) // 2])
    if dist is not None:
        tt = torch.tensor([dt, med], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt, med = float(tt[0].item()), float(tt[1].item())
    ms_mean = dt / args.steps * 1e3   # wall clock of the K steps (barrier + synchronize on both sides), max over ranks
    ms_guided = med                   # median of the K per-step times (HIP events at the step boundaries), max over ranks
    finite = bool(torch.isfinite(last_loss).all())
    keep_finite()

    # breakdown (not part of `value`): unguided step
    unguided_step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.unguided_steps):
        unguided_step()
    sync()
    ms_unguided = (time.perf_counter() - t0) / max(args.unguided_steps, 1) * 1e3
    keep_finite()

    # roofline of the dominant kernel class over the timed region
    roof = None
    if rank == 0:
        agg = gt.summary()
        names = {ops.A_PLAIN: "MFMA GEMM, linear loader (gemm_ring.hip asm-DMA ring / gemm.hip)",
                 ops.A_CONV3X3: "MFMA tap GEMM, 3x3 conv with the im2col tile resident in LDS (conv_halo.hip; stride-2 / odd shapes: gemm_ring.hip)",
                 ops.A_TCONV3: "MFMA tap GEMM, temporal (3,1,1) conv, tile resident in LDS (conv_halo.hip)",
                 ops.A_CONV3X3_T2: "MFMA GEMM, transposed stride-2 conv loader (gemm_ring.hip)"}
        keyname = {ops.A_PLAIN: "linear", ops.A_CONV3X3: "conv3x3", ops.A_TCONV3: "tconv3", ops.A_CONV3X3_T2: "conv3x3_t2"}
        dom = max(agg, key=lambda m: agg[m][1])
        n, secs, fl = agg[dom]
        ach = fl / secs / 1e12
        tfile = next((f for f in ("r06_gemm_traffic.json", "r05_gemm_traffic.json", "r04_gemm_traffic.json", "r03_gemm_traffic.json") if os.path.exists(os.path.join(ROOT, "profiles", f))), "r05_gemm_traffic.json")
        traffic, tnote = None, f"no profiles/{tfile} next to bench.py"
        tpath = os.path.join(ROOT, "profiles", tfile)
        tsource = f"profiles/{tfile} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate counter passes: tools/traffic_passes.sh; NOT measured in this run)"
        if os.path.exists(tpath):  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/gemm_pmc.py, rolled up by tools/pmc_traffic.py
            tj = json.load(open(tpath)).get(keyname[dom])
            if tj:
                traffic = {"hbm_side_bytes_per_launch": tj["fetch_bytes"] + tj["write_bytes"], "algorithmic_bytes_per_launch": tj["algorithmic_bytes"],
                           "ratio": round((tj["fetch_bytes"] + tj["write_bytes"]) / tj["algorithmic_bytes"], 3), "shape": tj["shape"]}
                tnote = tj.get("note", "")
        # the HBM-bound kernel north_star names: the fused guidance loss (forward + backward of the cross-attention energy), live in this run
        hbm_kernels = None
        if loss_events:
            us = sorted(a.elapsed_time(b) * 1e3 for a, b, _ in loss_events)
            med_us = us[len(us) // 2]
            nbytes = loss_events[0][2]
            hbm_kernels = {"guidance_loss": {
                "what": "csrc/guidance_loss.hip: probabilities of the object tokens + exact top-k selection + dQ for all six guidance keys, "
                        "3 launches per iteration, no attention map or autograd graph materialised",
                "us_per_iteration": round(med_us, 1), "launch_sets_timed": len(us), "algorithmic_bytes": nbytes,
                "algorithmic_bytes_is": "read Q + write dQ of the six keyed layers (BASELINE.md: ~177 MB)",
                "GB_per_s": round(nbytes / med_us / 1e3, 1), "peak_GB_per_s": 8000.0, "frac": round(nbytes / med_us / 1e3 / 8000.0, 4),
                "source": "HIP events on the launch stream around the three launches, median over the timed steps of THIS run; rocprofv3 per-kernel "
                          "durations of the same kernels: profiles/r06_guidance_loss.txt",
                "share_of_step": round(med_us * 1e-3 / ms_guided, 5)}}
        roof = {"bound": "mfma", "kernel": names[dom], "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": tsource if traffic else None, "traffic_note": tnote,
                "launches": n, "sampled_every": gt.stride,
                "class_launches_in_timed_region": gt.per_mode.get(dom, 0), "avg_launch_us": round(secs / n * 1e6, 1),
                "flops_per_launch": round(fl / n / 1e9, 2), "flops_per_launch_unit": "GFLOP",
                "hbm_kernels": hbm_kernels,
                "all_gemm": {keyname[m]: {"kernel": names[m], "launches_sampled": v[0], "launches_per_step": round(gt.per_mode.get(m, 0) / args.steps, 1),
                                          "ms_per_step": round(v[1] * 1e3 * gt.stride / args.steps, 2), "tflops": round(v[2] / v[1] / 1e12, 1),
                                          "frac": round(v[2] / v[1] / 1e12 / PEAK_BF16_TFLOPS, 4)} for m, v in agg.items()}}

    # The one collective of the product: all_gather of decoded uint8 frames over RCCL, once per finished video in a
    # harness that wants every video on rank 0.  Run it once here, untimed (not part of `value`), with a tensor of the decoded
    # size (24x320x576x3 = 13.3 MB per rank) derived from this rank's latents, so the N > 1 runs really cross xGMI.
    gather_ms = None
    rccl_seen = None
    if dist is not None:
        # proof that the collective backend saw N distinct devices: all_gather of (rank, device ordinal, device UUID hash)
        import zlib
        props = torch.cuda.get_device_properties(dev)
        ident = "|".join(str(getattr(props, k, "")) for k in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id"))
        uid = zlib.crc32(ident.encode())  # stable across processes (str hashes are salted per process): same GPU -> same id
        mine = torch.tensor([rank, local, uid, 1234 + rank, int(latents.float().sum().item() * 1e3)], dtype=torch.int64, device=dev if backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        rccl_seen = {"world_size": dist.get_world_size(), "ranks": sorted(int(t[0]) for t in allr),
                     "distinct_devices": len({(int(t[1]), int(t[2])) for t in allr}), "backend": backend,
                     # every rank works on its own (prompt, seed) sample: the generator seeds and a checksum of each rank's latents
                     "sample_seeds": [int(t[3]) for t in sorted(allr, key=lambda t: int(t[0]))],
                     "distinct_samples": len({int(t[4]) for t in allr})}
        from lvd_amd.sharding import gather_frames
        vid = (latents[0, :3].permute(1, 2, 3, 0).clamp(-1, 1).add(1).mul(127.5)).to(torch.uint8)             # (F, h, w, 3)
        vid = vid.repeat_interleave(8, 1).repeat_interleave(8, 2).contiguous()                                 # (F, 320, 576, 3)
        if backend != "nccl":
            vid = vid.cpu()
        sync()
        t0 = time.perf_counter()
        allv = gather_frames(vid)
        sync()
        gather_ms = (time.perf_counter() - t0) * 1e3
        assert len(allv) == world and allv[rank].shape == vid.shape and bool((allv[rank] == vid).all())

    cpu = None
    if sd_cpu is not None:
        cpu = cpu_baseline(sd_cpu, cfg)

    if rank == 0:
        value = world * V * FRAMES / (ms_guided * 1e-3)
        tf_cfg = 52.88 if args.gligen else TF_CFG_FWD  # CFG forward with the fusers enabled
        step_tf = V * (tf_cfg + TF_GUIDANCE_ITER)
        tf_cfg = V * tf_cfg
        skipped_tf = V * TF_CFG_SHARED_PREFIX if engine.cfg_shared_prefix else 0.0
        mean40 = (10 * ms_guided + 30 * ms_unguided) / 40
        out = {
            "metric": "denoise-step frames/sec, LVD-Zeroscope 576x320x24 w/ guidance" + (" + GLIGEN adapters" if args.gligen else "")
                      + (f" (THROUGHPUT MODE, not the headline: {V} videos per GPU, CFG batch {2 * V})" if V > 1 else ""),
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_guided, 2), "ms_per_step_stat": "median of the K per-step times (HIP events on the launch stream)",
            "mean_ms_per_step": round(ms_mean, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "lvd_zeroscope 576x320x24 (latent 40x72, 24 frames), guided step: 1 guidance iteration over 6 keys "
                                   "+ CFG UNet forward (B=2) + DPM-Solver++ update; random-init zeroscope-topology weights (1411M params)"
                                   + (", gated topology (1624M params) with the GLIGEN fusers enabled" if args.gligen else ""),
                       "videos_per_gpu": V, "parallelism": f"dp{world} (independent samples, no data-path collective)",
                       "guidance_scale": 9.0, "objects": 3},
            "unguided_ms_per_step": round(ms_unguided, 2),
            "unguided_frames_per_s": round(world * V * FRAMES / (ms_unguided * 1e-3), 2),
            "schedule40_mean_frames_per_s": round(world * V * FRAMES / (mean40 * 1e-3), 2),
            "step_algorithmic_tflop": step_tf,
            "step_mfma_frac": round(step_tf / (ms_guided * 1e-3) / PEAK_BF16_TFLOPS, 4),
            "unguided_mfma_frac": round(tf_cfg / (ms_unguided * 1e-3) / PEAK_BF16_TFLOPS, 4),
            # the same with the FLOPs the timed region really executes (the shared CFG prefix counted once per video): the figure to compare
            # between rounds; *_algorithmic keeps the reference module's count (the prefix twice)
            "step_executed_tflop": round(step_tf - skipped_tf, 2),
            "step_mfma_frac_executed": round((step_tf - skipped_tf) / (ms_guided * 1e-3) / PEAK_BF16_TFLOPS, 4),
            "unguided_mfma_frac_executed": round((tf_cfg - skipped_tf) / (ms_unguided * 1e-3) / PEAK_BF16_TFLOPS, 4),
            "loss_finite": finite,
            "guided_step_is": ("guidance.hip_latent_backward_guidance_many (the V guidance passes as ONE recorded forward / backward of batch V, per-sample layouts and losses, as pipeline.sample_many runs them)"
                               if V > 1 and not args.guidance_one_by_one else "guidance.hip_latent_backward_guidance")
                              + " (one iteration; the returned loss tensor is carried into the next step, whose entry check waits for its pinned host copy) + CFG forward + fused CFG/DPM update",
            "cfg_shared_prefix": bool(engine.cfg_shared_prefix),
            "cfg_shared_prefix_note": "the (uncond, cond) items of the CFG batch are the SAME latents (reference: torch.cat([latents] * 2)) and stay identical "
                                      "until the first text-dependent layer; that prefix (conv_in, transformer_in, first resnet / temporal conv / spatial self-"
                                      "attention) runs once per sample and is duplicated there.  step_algorithmic_tflop counts it twice, as the reference "
                                      "module does; step_executed_tflop / *_mfma_frac_executed count it once; LVD_CFG_SHARED_PREFIX=0 disables it",
            "c_abi_calls_per_guided_step": round(abi_calls_per_step, 1),
            "c_abi_calls_note": "launching entries into liblvdhip.so per guided step (one entry = one to three kernels); rocprofv3 counts 2790 dispatches per guided and "
                                "1043 per unguided step (profiles/r06_bench_summary.txt)",
            "gemm_autotune_table": table_loaded, "rccl_ranks_seen": rccl_seen,
            "frame_gather_ms_untimed": None if gather_ms is None else round(gather_ms, 2),
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
