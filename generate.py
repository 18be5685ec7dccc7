#!/usr/bin/env python
"""Batch driver with the reference's command line (generate.py:13-108) on the MI355X sampler.

    python generate.py --model gpt-4 --run-model lvd_zeroscope --prompt-type demo --template_version v0.1 \
        --num_frames 24 --cache-dir /path/to/cache [--repeats N --seed_offset S --force_run_ind K ...]

Kept from the reference: flag names, run-model dispatch (:111-165), cache lookup per prompt (:278), run-dir / resume
logic (:225-239,288-299), the seed rule seed = prompt_index + repeat*6789 + seed_offset (:325-335), prompt sharding by
--skip_first_prompts/--num_prompts (:255-262) and error containment (:340-353).  Added: under torchrun every rank takes
the prompt indices i % WORLD_SIZE == RANK (one process per GPU, no communication while sampling), and
--synthetic-weights for runs without a checkpoint (there is no network for hub downloads)."""
import argparse
import importlib
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

RUN_MODELS = ["lvd", "lvd_zeroscope", "lvd_modelscope256", "lvd-gligen_modelscope256", "lvd-gligen_zeroscope", "lvd-plus_modelscope256",
              "lvd-plus_zeroscope", "lvd_modelscope512", "modelscope", "modelscope_256", "zeroscope"]
MODEL_NAMES = {"gpt-4": "gpt-4-1106-preview", "gpt-4-1106-preview": "gpt-4-1106-preview", "gpt-3.5": "gpt-3.5-turbo", "gpt-3.5-turbo": "gpt-3.5-turbo"}
PROMPTS_DEMO = ["A bear walks from the left to the right"]  # prompt.py:72-74


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument("--save-suffix", default=None, type=str)
    p.add_argument("--model", choices=sorted(MODEL_NAMES), required=True, help="LLM model to load the cache from")
    p.add_argument("--repeats", default=1, type=int)
    p.add_argument("--regenerate", default=1, type=int)
    p.add_argument("--force_run_ind", default=None, type=int)
    p.add_argument("--skip_first_prompts", default=0, type=int)
    p.add_argument("--seed_offset", default=0, type=int)
    p.add_argument("--num_prompts", default=None, type=int)
    p.add_argument("--run-model", default="lvd", choices=RUN_MODELS)
    p.add_argument("--no-continue-on-error", action="store_true")
    p.add_argument("--prompt-type", type=str, default="demo")
    p.add_argument("--template_version", choices=["v0.1"], required=True)
    p.add_argument("--dry-run", action="store_true", help="skip the generation")
    p.add_argument("--gemm_autotune_table", default=None, help="JSON of per-shape GEMM tile-geometry choices: loaded if it exists (every rank / run "
                   "then uses the same summation order: bit-identical videos for the same prompt and seed), written at the end otherwise (the union "
                   "of all ranks' choices).  Default: the table shipped for the zeroscope 576x320x24 topology, profiles/gemm_autotune_576x320x24.json; "
                   "shapes a table does not hold are tuned on first use")
    for a in ["fg_top_p", "bg_top_p", "fg_weight", "bg_weight", "loss_threshold", "loss_scale", "boxdiff_loss_scale", "com_loss_scale",
              "gligen_scheduled_sampling_beta"]:
        p.add_argument("--" + a, default=None, type=float)
    for a in ["num_inference_steps", "max_iter", "max_index_step", "num_frames", "use_ratio_based_loss", "boxdiff_normed"]:
        p.add_argument("--" + a, default=None, type=int)
    # additions
    p.add_argument("--cache-dir", default="cache", help="directory holding cache_{prompt_type}_{template}_{model}.json")
    p.add_argument("--prompts-file", default=None, help="one prompt per line (prompt types other than demo)")
    p.add_argument("--synthetic-weights", action="store_true", help="random-init weights of the real topology (no checkpoint on disk)")
    p.add_argument("--checkpoint", default=None, help="torch-saved reference state_dict of the UNet, or a local Hugging Face snapshot directory "
                   "(e.g. .../models--cerspense--zeroscope_v2_576w/snapshots/<rev>) holding unet/config.json + diffusion_pytorch_model.safetensors")
    p.add_argument("--img-root", default="img_generations")
    p.add_argument("--videos-per-gpu", type=int, default=1, help="throughput mode: V (prompt, seed) jobs of this rank share each classifier-free-"
                   "guidance forward (batch 2V); guidance stays one pass per sample.  Same seed rule, same files; the deep UNet levels fill the "
                   "GPU better (bench.py --videos-per-gpu: +5 %% guided / +11 %% unguided per video at V = 2)")
    return p


def main(argv=None):
    args = build_parser().parse_args(argv)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    run_model = args.run_model
    baseline = run_model in ("modelscope", "zeroscope", "modelscope_256")
    option = run_model.split("_")[1] if "_" in run_model else ""
    run = None
    dist = None
    if not args.dry_run:
        import torch
        import lvd_amd  # noqa: F401
        from lvd_amd.generation import _common
        local_rank = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local_rank)  # kernels launch on the CURRENT device's stream: one process per GPU must select its own
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group(os.environ.get("LVD_DIST_BACKEND", "nccl"))  # RCCL; only the end-of-run tally uses it
        from lvd_amd import ops
        shipped = os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "gemm_autotune_576x320x24.json")
        table_path = args.gemm_autotune_table or (shipped if os.path.exists(shipped) else None)
        if table_path and os.path.exists(table_path):
            ops.load_gemm_autotune_table(table_path)  # same tile geometry per shape in every rank / run
        elif world > 1:
            print(f"rank {rank}: no GEMM autotune table: every rank times the tile geometries itself, so this sharded run is NOT bit-reproducible "
                  "against a single-process run (pass --gemm_autotune_table: the merged table is written at the end and pins later runs)")
        if args.synthetic_weights:
            _common.configure(state_dict="synthetic")
        elif args.checkpoint and os.path.isdir(args.checkpoint):
            _common.configure(state_dict=args.checkpoint)  # local HF snapshot directory: <dir>/unet/{config.json, diffusion_pytorch_model.safetensors}
        elif args.checkpoint:
            _common.configure(state_dict=torch.load(args.checkpoint, map_location="cpu"))
        _common.configure(device=f"cuda:{local_rank}")
        modname = {"lvd-plus": "lvd_plus", "lvd-gligen": "lvd_gligen", "lvd": "lvd", "modelscope": "modelscope_dpm", "zeroscope": "zeroscope_dpm"}[run_model.split("_")[0]]
        generation = importlib.import_module(f"lvd_amd.generation.{modname}")
        H, W = generation.init(option) if baseline else generation.init(base_model=option if option else "modelscope512")
        if "zeroscope" in run_model and ((args.num_frames is not None and args.num_frames < 24) or (not baseline and args.num_frames is None)):
            raise ValueError("Running zeroscope with fewer than 24 frames. This may lead to suboptimal results.")
        assert generation.version == run_model.split("_")[0], f"{generation.version} != {run_model.split('_')[0]}"
        run = generation.run

    from lvd_amd import dsl, sharding
    model = MODEL_NAMES[args.model]
    cache = None
    if not baseline:
        cache = dsl.LayoutCache(os.path.join(args.cache_dir, f"cache_{args.prompt_type.replace('lmd_', '')}_{args.template_version}_{model}.json"))
    if args.prompt_type == "demo":
        prompts = PROMPTS_DEMO
    elif args.prompt_type.startswith("lvd") and not args.prompts_file:
        from lvd_amd.evaluation import get_prompts  # the 500-prompt benchmark (prompt.py:82-90); repeats walk the cache entries
        prompts = get_prompts(args.prompt_type)
    elif args.prompts_file:
        prompts = [l.strip() for l in open(args.prompts_file) if l.strip()]
    else:
        prompts = list(cache.data.keys()) if cache else []
    run_kwargs = {k: getattr(args, k) for k in ["fg_top_p", "bg_top_p", "fg_weight", "bg_weight", "loss_threshold", "loss_scale", "boxdiff_loss_scale",
                                                 "com_loss_scale", "gligen_scheduled_sampling_beta", "num_inference_steps", "max_iter", "max_index_step",
                                                 "num_frames", "use_ratio_based_loss", "boxdiff_normed"] if getattr(args, k) is not None}
    # same directory names as the reference (generate.py:207-208: no suffix for --model gpt-4), so its eval scripts find them
    model_in_base_save_dir = "" if args.model == "gpt-4" else f"_{model}"
    base_save = f"{args.img_root}/imgs_{args.prompt_type}_template{args.template_version}{model_in_base_save_dir}_{run_model}" + (f"_{args.save_suffix}" if args.save_suffix else "")
    if args.force_run_ind is not None:
        run_ind = args.force_run_ind
    else:
        # Reference generate.py:225-234 probes for the first free run directory.  Under N ranks only rank 0 probes and the others take its
        # answer: a rank that finishes importing late would otherwise see the run0/ its peers have already created and write to run1/.
        run_ind = 0
        if rank == 0:
            while os.path.exists(f"{base_save}/run{run_ind}"):
                run_ind += 1
        if world > 1:
            if dist is None:  # --dry-run never touched torch: a gloo group just for this broadcast and the barrier below
                import torch.distributed as dist
                dist.init_process_group("gloo")
            box = [run_ind]
            dist.broadcast_object_list(box, src=0)
            run_ind = int(box[0])
            if rank == 0:
                os.makedirs(f"{base_save}/run{run_ind}", exist_ok=True)  # claimed before any rank starts writing
    save_dir = f"{base_save}/run{run_ind}"
    print(f"Save dir: {save_dir}  (rank {rank}/{world})")

    ind, generated = 0, 0
    failure = None
    pending = []  # (layout, seed, repeat, directory) jobs of this rank waiting for a batch of --videos-per-gpu

    def flush():
        """Sample the pending jobs in one denoising loop (V = 1: exactly the reference's one `run` per repeat, generate.py:325-338)."""
        nonlocal generated
        if not pending:
            return
        jobs = list(pending)
        pending.clear()
        try:
            if len(jobs) == 1 and args.videos_per_gpu <= 1:
                j = jobs[0]
                from lvd_amd.generation import _common
                _common.configure(img_dir=j["img_dir"])
                run(j["parsed_layout"], seed=j["seed"], repeat_ind=j["repeat_ind"], **run_kwargs)
            else:
                generation.run_many(jobs, **run_kwargs)
        except Exception:
            # the whole batch was in the loop that failed: say which (prompt directory, repeat) pairs are lost with it
            print("***dropped with the failing batch: " + ", ".join(f"{j['img_dir']} repeat {j['repeat_ind']}" for j in jobs) + "***")
            raise
        generated += len(jobs)

    def drop_jobs_of(img_dir):
        """A prompt failed before / while queueing: only ITS queued jobs go; jobs of earlier prompts still waiting for a full batch stay."""
        gone = [j for j in pending if j["img_dir"] == img_dir]
        if gone:
            pending[:] = [j for j in pending if j["img_dir"] != img_dir]
            print("***dropped: " + ", ".join(f"{j['img_dir']} repeat {j['repeat_ind']}" for j in gone) + "***")

    try:
        for regenerate_ind in range(args.regenerate):
            if cache:
                cache.reset_access()
            for prompt_ind, prompt in enumerate(prompts):
                if prompt_ind < args.skip_first_prompts or (args.num_prompts is not None and prompt_ind >= args.skip_first_prompts + args.num_prompts):
                    ind += 1  # outside the requested range: the cache entry is NOT consumed (reference generate.py:255-262)
                    continue
                prompt = prompt.strip().rstrip(".")
                resp = None if baseline else cache.get(prompt)  # every rank walks the cache identically (sequential semantics)
                if not sharding.owns(ind, rank, world):
                    ind += 1
                    continue
                if not baseline and resp is None:
                    print(f"Cache miss, skipping prompt: {prompt}")
                    ind += 1
                    continue
                img_dir = f"{save_dir}/{ind}"
                done = os.path.exists(img_dir) and len([f for f in os.listdir(img_dir) if f.endswith("joblib")]) >= args.repeats
                if done:
                    print(f"Image exists at {img_dir}, skipping")
                    ind += 1
                    continue
                os.makedirs(img_dir, exist_ok=True)
                try:
                    layout = {"Prompt": prompt, "Background keyword": "", **{f"Frame {k + 1}": [] for k in range(6)}} if baseline else dsl.parse_layout_response(prompt, resp)
                    print("parsed_layout:", layout)
                    if not args.dry_run:
                        for repeat_ind in range(args.repeats):
                            pending.append(dict(parsed_layout=layout, seed=ind + repeat_ind * 6789 + args.seed_offset, repeat_ind=repeat_ind, img_dir=img_dir))
                            if len(pending) >= max(1, args.videos_per_gpu):
                                flush()
                except KeyboardInterrupt:
                    raise SystemExit(1)
                except RuntimeError:
                    print("***RuntimeError: might run out of memory, skipping the current one***")
                    print(traceback.format_exc())
                    drop_jobs_of(img_dir)
                    time.sleep(1)
                except Exception as e:  # noqa: BLE001
                    print(f"***Error: {e}***")
                    print(traceback.format_exc())
                    drop_jobs_of(img_dir)
                    if args.no_continue_on_error:
                        raise
                ind += 1
        try:
            flush()  # the last, possibly smaller, batch
        except RuntimeError:
            print("***RuntimeError: might run out of memory, skipping the current one***")
            print(traceback.format_exc())
    except BaseException as e:  # noqa: BLE001 — a rank that dies here must still meet the others in the tally below, or they hang in it
        failure = e
        print(f"rank {rank}: stopping after {generated} video(s): {type(e).__name__}: {e}")
    print(f"rank {rank}: generated {generated} video(s)")
    if not args.dry_run and args.gemm_autotune_table and not os.path.exists(args.gemm_autotune_table):
        from lvd_amd import ops
        if dist is not None:  # the union of what the ranks tuned (rank 0's choice wins where two ranks timed the same shape)
            tabs = [None] * world
            dist.all_gather_object(tabs, [[list(k), v] for k, v in ops.gemm_autotune_table().items()])
            if rank == 0:
                for tab in reversed(tabs):
                    for k, v in tab:
                        k[7] = tuple(k[7]) if k[7] is not None else None
                        ops._gemm_choice[tuple(k)] = int(v)
        if rank == 0:
            ops.save_gemm_autotune_table(args.gemm_autotune_table)
    if dist is not None:  # end-of-run tally over RCCL (the only collective: every rank wrote its own directory entries)
        import torch
        tot = torch.tensor([generated, int(failure is not None)], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(tot)
        if rank == 0:
            print(f"all ranks: generated {int(tot[0].item())} video(s)" + (f", {int(tot[1].item())} rank(s) stopped on an error" if int(tot[1].item()) else ""))
        dist.destroy_process_group()
    if failure is not None:
        raise failure
    return generated


if __name__ == "__main__":
    main()
