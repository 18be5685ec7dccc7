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
    """HIP-event timing of the dominant kernel class (MFMA GEMM with the linear loader: ~750 of the ~1130 GEMM launches
    and the largest share of a guided step) over the timed region.  Events are recorded on torch's current stream, which
    is the stream the kernels are launched on.  An event pair is a barrier packet on each side of the launch (~3 us of
    idle GPU each, measured in the rocprof trace: 6 % of the step when every launch is bracketed), so every `stride`-th
    launch of the class is bracketed; a step has 753 such launches (753 % 8 == 1), so the sampled positions rotate by one
    every step and 8 timed steps cover every launch site exactly once."""

    def __init__(self, modes=(ops.A_PLAIN,), stride=8):
        self.rec = []
        self.orig = ops.gemm
        self.modes = modes
        self.stride = stride
        self.count = 0

    def __enter__(self):
        def timed(a1, w, **kw):
            mode = kw.get("mode", ops.A_PLAIN)
            if mode not in self.modes:
                return self.orig(a1, w, **kw)
            self.count += 1
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


def cpu_baseline(sd_cpu, cfg):
    """fp32 oracle (restatement of the reference) on the host cores: ONE CFG forward at BASELINE config 1
    (256x144x8, latent 18x32) = 2.87 TFLOP, extrapolated to the guided Zeroscope step by algorithmic FLOPs."""
    from oracle import unet_ref
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, 4, 8, 18, 32, generator=g)
    ehs = torch.randn(2, 77, cfg.cross_attention_dim, generator=g)
    cores = torch.get_num_threads()
    with torch.no_grad():
        t0 = time.time()
        unet_ref.unet_forward(sd_cpu, cfg, x, 500, ehs)
        dt = time.time() - t0
    tf_sample = 2.87
    tflops = tf_sample / dt
    t_guided = (TF_CFG_FWD + TF_GUIDANCE_ITER) / tflops
    return {"value": round(FRAMES / t_guided, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"one fp32 CFG UNet forward at 256x144x8 (2.87 TFLOP) took {dt:.1f}s = {tflops:.3f} TFLOP/s; "
                      f"extrapolated by FLOPs to the guided 576x320x24 step ({TF_CFG_FWD + TF_GUIDANCE_ITER:.1f} TFLOP)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unguided-steps", type=int, default=4, help="extra (untimed-for-value) unguided steps for the breakdown")
    ap.add_argument("--gligen", action="store_true", help="BASELINE config 3 instead of the default config 2: gated topology (1624M params), "
                    "GLIGEN fusers on in the CFG forward (52.88 TFLOP); not the headline metric")
    args = ap.parse_args()

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
    bboxes, positions = demo_layout()
    gligen = None
    if args.gligen:  # controllable_pipeline_text_to_video_synth.py:736-814: 30 slots per frame, [unconditional; conditional]
        gb = torch.zeros(2 * FRAMES, 30, 4)
        gm = torch.zeros(2 * FRAMES, 30)
        gb[FRAMES:, :len(bboxes)] = torch.tensor(bboxes).permute(1, 0, 2)
        gm[FRAMES:, :len(bboxes)] = 1.0
        gligen = {"boxes": gb, "masks": gm, "positive_embeddings": torch.randn(2 * FRAMES, 30, cfg.cross_attention_dim, device=dev, generator=g).cpu()}
    sched = DPMSolverPP2MSchedule()
    sched.set_timesteps(40)
    sampler = HipSampler(engine, sched, guidance_scale=9.0)
    sampler.reset(latents)
    hp = dict(loss_scale=2.5, fg_top_p=0.25, bg_top_p=0.25, fg_weight=1.0, bg_weight=2.0)  # README.md:68 weak guidance

    state = {"i": 0}

    def guided_step():
        i = state["i"] % 10  # guidance is active for step indices < max_index_step=10
        sched.step_index, sched.lower_order_nums = i, min(i, 2)
        t = int(sched.timesteps[i])
        loss, grad = guidance.guidance_loss_and_grad(engine, latents, t, text_cond, bboxes, positions, GUIDANCE_KEYS, **hp)
        ops.axpy_(latents, grad, float((1 - sched.alphas_cumprod[t]) ** 0.5))
        sampler.cfg_step(latents, i, text_cfg, gligen=gligen)
        state["i"] += 1
        return loss

    def unguided_step():
        i = 10 + state["i"] % 29
        sched.step_index, sched.lower_order_nums = i, 2
        sampler.cfg_step(latents, i, text_cfg, gligen=gligen)  # --gligen: fusers on, as in steps 10..15 of the 40 (beta = 0.4)
        state["i"] += 1

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def keep_finite():
        # random-init weights are not a denoiser: re-draw the latents so timing never runs on inf/nan (untimed)
        if not bool(torch.isfinite(latents).all()) or float(latents.abs().max()) > 50:
            latents.copy_(torch.randn(latents.shape, device=dev, generator=g))
            sampler.reset(latents)

    guided_step()  # untimed preparation, independent of --warmup: the GEMM autotuner picks a variant per shape on first use
    unguided_step()
    keep_finite()
    for _ in range(args.warmup):
        guided_step()
        keep_finite()
    # timed region: exactly K guided steps; every 8th launch of the dominant GEMM class is bracketed by HIP events on the
    # launch stream (GemmTimer; no extra synchronisation)
    gt = GemmTimer()
    sync()
    t0 = time.perf_counter()
    with gt:
        for _ in range(args.steps):
            last_loss = guided_step()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_guided = dt / args.steps * 1e3
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
        names = {ops.A_PLAIN: "MFMA GEMM, linear loader (gemm.hip / gemm_ring.hip)", ops.A_CONV3X3: "MFMA GEMM, implicit 3x3 conv loader",
                 ops.A_TCONV3: "MFMA GEMM, temporal 3-tap loader", ops.A_CONV3X3_T2: "MFMA GEMM, transposed stride-2 conv loader"}
        dom = max(agg, key=lambda m: agg[m][1])
        n, secs, fl = agg[dom]
        ach = fl / secs / 1e12
        roof = {"bound": "mfma", "kernel": names[dom], "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": None,
                "traffic_note": "not collected live (needs rocprofv3 --pmc); per-shape FETCH_SIZE/WRITE_SIZE of this kernel class: profiles/r01_gemm_pmc_traffic.txt "
                                "(e.g. M=138240 N=960 K=320: 113 MB fetched vs 89 MB algorithmic, 240 MB written vs 265 MB)",
                "launches": n, "sampled_every": gt.stride,
                "class_launches_in_timed_region": gt.count, "avg_launch_us": round(secs / n * 1e6, 1),
                "flops_per_launch": round(fl / n / 1e9, 2), "flops_per_launch_unit": "GFLOP",
                "all_gemm": {names[m]: {"launches": v[0], "ms_per_step": round(v[1] * 1e3 * gt.stride / args.steps, 2), "tflops": round(v[2] / v[1] / 1e12, 1)} for m, v in agg.items()}}

    cpu = None
    if sd_cpu is not None:
        cpu = cpu_baseline(sd_cpu, cfg)

    if rank == 0:
        value = world * FRAMES / (ms_guided * 1e-3)
        tf_cfg = 52.88 if args.gligen else TF_CFG_FWD  # CFG forward with the fusers enabled
        step_tf = tf_cfg + TF_GUIDANCE_ITER
        mean40 = (10 * ms_guided + 30 * ms_unguided) / 40
        out = {
            "metric": "denoise-step frames/sec, LVD-Zeroscope 576x320x24 w/ guidance" + (" + GLIGEN adapters" if args.gligen else ""),
            "value": round(value, 2), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_guided, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "lvd_zeroscope 576x320x24 (latent 40x72, 24 frames), guided step: 1 guidance iteration over 6 keys "
                                   "+ CFG UNet forward (B=2) + DPM-Solver++ update; random-init zeroscope-topology weights (1411M params)"
                                   + (", gated topology (1624M params) with the GLIGEN fusers enabled" if args.gligen else ""),
                       "videos_per_gpu": 1, "parallelism": f"dp{world} (independent samples, no data-path collective)",
                       "guidance_scale": 9.0, "objects": 3},
            "unguided_ms_per_step": round(ms_unguided, 2),
            "unguided_frames_per_s": round(world * FRAMES / (ms_unguided * 1e-3), 2),
            "schedule40_mean_frames_per_s": round(world * FRAMES / (mean40 * 1e-3), 2),
            "step_algorithmic_tflop": step_tf,
            "step_mfma_frac": round(step_tf / (ms_guided * 1e-3) / PEAK_BF16_TFLOPS, 4),
            "unguided_mfma_frac": round(tf_cfg / (ms_unguided * 1e-3) / PEAK_BF16_TFLOPS, 4),
            "loss_finite": finite,
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
