#!/usr/bin/env python
"""Stage-2 evaluation: detect the prompt's objects in the generated videos with OWL-ViT (HIP kernels) and score them with
the benchmark predicates.

    python scripts/eval_owl_vit.py --run_base_path img_generations/imgs_lvd_templatev0.1_lvd_zeroscope/run0 \
        --owl-vit-path /path/to/owlvit-base-patch32 [--prompt-type lvd --num_eval_frames 6 --class-aware-nms --save-eval]

Command line of /root/reference/scripts/eval_owl_vit.py:182-196.  Differences: the checkpoint is a local directory
(`--owl-vit-path`: model.safetensors or pytorch_model.bin plus the CLIP tokenizer files; there is no hub access) or
`--synthetic-weights` for a dry run, `--no-cuda` does not exist (the detector has no CPU path), and under torchrun every
rank scores the prompt indices i % WORLD_SIZE == RANK and rank 0 merges the tallies (one gather of a few hundred bytes)."""
import argparse
import glob
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load_detector(args, device):
    import torch
    from lvd_amd.evaluation.owlvit import HipOwlViTDetector, OwlViTConfig, synthetic_owlvit_state_dict
    cfg = OwlViTConfig()
    if args.synthetic_weights:
        g = torch.Generator().manual_seed(0)
        table = {}

        def tokenize(texts):  # stand-in ids: deterministic per text, CLIP layout (<bos> words <eos> padding)
            out = torch.zeros((len(texts), 16), dtype=torch.long)
            for i, t in enumerate(texts):
                if t not in table:
                    table[t] = torch.randint(1, 49405, (min(len(t.split()), 14),), generator=g)
                w = table[t]
                out[i, 0], out[i, 1:1 + len(w)], out[i, 1 + len(w)] = 49406, w, 49407
            return out
        return HipOwlViTDetector(cfg, synthetic_owlvit_state_dict(cfg), device=device, tokenize=tokenize)
    path = args.owl_vit_path
    if path is None:
        raise SystemExit("--owl-vit-path (local google/owlvit-base-patch32 directory) or --synthetic-weights is required")
    if os.path.exists(os.path.join(path, "model.safetensors")):
        from safetensors.torch import load_file
        sd = load_file(os.path.join(path, "model.safetensors"))
    else:
        sd = torch.load(os.path.join(path, "pytorch_model.bin"), map_location="cpu")
    from transformers import CLIPTokenizer
    tok = CLIPTokenizer.from_pretrained(path)
    tokenize = lambda texts: tok(texts, padding="max_length", max_length=16, truncation=True, return_tensors="pt")["input_ids"]
    return HipOwlViTDetector(cfg, sd, device=device, tokenize=tokenize)


def _local_device():
    """cuda:<LOCAL_RANK>, made the current device (kernels launch on the current device's stream: one process per GPU)."""
    import torch
    idx = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(idx)
    return f"cuda:{idx}"


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--prompt-type", type=str, default="lvd")
    ap.add_argument("--run_base_path", type=str, required=True)
    ap.add_argument("--run_start_ind", default=0, type=int)
    ap.add_argument("--num_prompts", default=None, type=int)
    ap.add_argument("--num_eval_frames", default=6, type=int)
    ap.add_argument("--skip_first_prompts", default=0, type=int)
    ap.add_argument("--detection_score_threshold", default=0.05, type=float)
    ap.add_argument("--nms_threshold", default=0.5, type=float)
    ap.add_argument("--class-aware-nms", action="store_true")
    ap.add_argument("--save-eval", action="store_true")
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--owl-vit-path", default=None)
    ap.add_argument("--synthetic-weights", action="store_true")
    args = ap.parse_args(argv)

    import joblib
    import numpy as np
    import lvd_amd  # noqa: F401
    from lvd_amd.evaluation import ScoreBoard, get_prompts, score_video
    np.set_printoptions(precision=2)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    detector = load_detector(args, _local_device())
    pairs = get_prompts(args.prompt_type, return_predicates=True)
    print(f"Number of prompts (predicates): {len(pairs)}")
    print(f"Number of evaluating frames: {args.num_eval_frames}")
    results = []  # (prompt index, task, success)
    for ind, (prompt, predicate) in enumerate(pairs):
        prompt = prompt.strip().rstrip(".")
        if ind < args.skip_first_prompts or (args.num_prompts is not None and ind >= args.skip_first_prompts + args.num_prompts):
            continue
        if ind % world != rank:
            continue
        pattern = f"{args.run_base_path}/{ind + args.run_start_ind}/video_*.joblib"
        paths = sorted(glob.glob(pattern))
        if len(paths) != 1:
            print(f"***{'No image' if not paths else 'More than one images'} matching {pattern}, skipping***")
            continue
        print(f"Video path: {paths[0]} ({paths[0].replace('.joblib', '.gif')})")
        kind, ok = score_video(prompt, predicate, joblib.load(paths[0]), detector, score_threshold=args.detection_score_threshold,
                               nms_threshold=args.nms_threshold, use_class_aware_nms=args.class_aware_nms, num_eval_frames=args.num_eval_frames,
                               verbose=args.verbose)
        print(f"Eval success ({kind}):", ok)
        results.append((ind, kind, bool(ok)))
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group("gloo")  # a few hundred bytes of python objects: host-side gather
        gathered = [None] * world if rank == 0 else None
        dist.gather_object(results, gathered, dst=0)
        if rank != 0:
            return None
        results = sorted(r for part in gathered for r in part)
    board = ScoreBoard()
    for _, kind, ok in results:
        board.add(kind, ok)
    if board.total:
        print(board.report())
    if args.save_eval:
        board.save(f"{args.run_base_path}/eval.json")
    return board


if __name__ == "__main__":
    main()
