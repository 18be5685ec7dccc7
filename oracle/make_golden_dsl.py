"""Generate tests/golden/dsl.json by running the REFERENCE's own DSL functions (utils/llm.py, utils/parse.py,
utils/guidance.py, utils/cache.py) on the caches it ships.  Runs only where /root/reference exists."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_golden  # noqa: E402  (shim)
from fake_tokenizer import FakeTokenizer  # noqa: E402


def main():
    make_golden.install_shim()
    os.chdir(make_golden.REF)
    from utils import cache, parse
    from utils.llm import get_parsed_layout
    cases = []
    for cache_file, take in (("cache/cache_demo_v0.1_gpt-4-1106-preview.json", 1), ("cache/cache_lvd_v0.1_gpt-4-1106-preview.json", 40)):
        cache.cache_path, cache.cache_format = cache_file, "json"
        cache.init_cache(allow_nonexist=False)
        cache.reset_cache_access()
        keys = list(cache.global_cache.keys())
        # spread over the five benchmark tasks, and walk multi-response prompts twice to exercise the sequential cache
        picked = keys[:: max(1, len(keys) // take)][:take]
        for prompt in picked:
            for rep in range(min(2, len(cache.global_cache[prompt]))):
                resp = cache.get_cache(prompt)
                layout, _ = get_parsed_layout(prompt, max_partial_response_retries=1, override_response=resp, json_template=False)
                tok = FakeTokenizer()
                cond = parse.parsed_layout_to_condition(layout, height=512, width=512, num_condition_frames=24, tokenizer=tok, verbose=False)
                cond16 = parse.parsed_layout_to_condition(layout, height=512, width=512, num_condition_frames=16, tokenizer=None, verbose=False)
                cases.append({"cache": os.path.basename(cache_file), "prompt": prompt, "rep": rep, "response": resp, "layout": layout,
                              "cond_prompt": cond.prompt, "boxes": cond.boxes, "phrases": cond.phrases, "object_positions": cond.object_positions,
                              "token_map": cond.token_map, "boxes16": cond16.boxes})
    n_absent = sum(any(all(v == 0 for v in fb) for b in c["boxes"] for fb in b) for c in cases)
    print(f"{len(cases)} cases, {n_absent} with a disappearing object")
    with open(os.path.join(ROOT, "tests", "golden", "dsl.json"), "w") as f:
        json.dump(cases, f)


if __name__ == "__main__":
    main()
