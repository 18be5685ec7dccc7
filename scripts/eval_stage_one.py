#!/usr/bin/env python
"""Stage-1 evaluation: score the cached LLM layouts of a prompt set with the benchmark predicates (no GPU, no LLM call).

    python scripts/eval_stage_one.py --prompt-type lvd --model gpt-4 --template_version v0.1 --cache-dir /path/to/cache

Command line of /root/reference/scripts/eval_stage_one.py:21-31 (+ --cache-dir); a cache miss is an error here because
this build never queries an LLM."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lvd_amd  # noqa: E402,F401
from lvd_amd import dsl  # noqa: E402
from lvd_amd.evaluation import ScoreBoard, evaluate_with_layout, get_prompts  # noqa: E402

MODEL_NAMES = {"gpt-4": "gpt-4-1106-preview", "gpt-4-1106-preview": "gpt-4-1106-preview", "gpt-3.5": "gpt-3.5-turbo", "gpt-3.5-turbo": "gpt-3.5-turbo"}


FLAGS = [  # (flag, kwargs) — names and defaults of the reference's command line
    ("--prompt-type", dict(type=str, default="lvd")),
    ("--model", dict(choices=sorted(MODEL_NAMES), required=True)),
    ("--template_version", dict(choices=["v0.1"], required=True)),
    ("--skip_first_prompts", dict(type=int, default=0)),
    ("--num_prompts", dict(type=int, default=None)),
    ("--show-cache-access", dict(action="store_true")),
    ("--verbose", dict(action="store_true")),
    ("--cache-dir", dict(default="cache")),
]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for flag, kw in FLAGS:
        ap.add_argument(flag, **kw)
    args = ap.parse_args(argv)
    cache = dsl.LayoutCache(os.path.join(args.cache_dir, f"cache_{args.prompt_type.replace('lmd_', '')}_{args.template_version}_{MODEL_NAMES[args.model]}.json"))
    pairs = get_prompts(args.prompt_type, return_predicates=True)
    print(f"Number of prompts (predicates): {len(pairs)}")
    board = ScoreBoard()
    for ind, (prompt, predicate) in enumerate(pairs):
        prompt = prompt.strip().rstrip(".")
        if ind < args.skip_first_prompts or (args.num_prompts is not None and ind >= args.skip_first_prompts + args.num_prompts):
            continue
        response = cache.get(prompt)
        if response is None:
            raise KeyError(f"no cached layout for prompt: {prompt}")
        layout = dsl.parse_layout_response(prompt, response)
        kind, ok = evaluate_with_layout(layout, predicate, dsl.NUM_LAYOUT_FRAMES, height=dsl.LAYOUT_SIZE[0], width=dsl.LAYOUT_SIZE[1], verbose=args.verbose)
        print(f"Eval success ({kind}):", ok)
        board.add(kind, ok)
    print(board.report())
    if args.show_cache_access:
        print(json.dumps(cache._index))
        print("Number of accessed keys:", len(cache._index), "responses consumed:", cache.values_accessed())
    return board


if __name__ == "__main__":
    main()
