"""Generate tests/golden/eval.json (+ gzip copies of the layout caches the reference ships) by running the REFERENCE's own
scoring code: utils/eval/{eval,lvd,utils}.py, scripts/eval_owl_vit.py:eval_prompt (with a stub detector so that no
OWL-ViT checkpoint is needed) and the stage-1 loop of scripts/eval_stage_one.py.  Runs only where /root/reference exists.

TEST INFRASTRUCTURE: the fixtures are data (inputs + the reference's outputs)."""
import gzip
import importlib.util
import json
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import make_golden  # noqa: E402  (shim)

CACHES = ("cache_lvd_v0.1_gpt-4-1106-preview.json", "cache_lvd_v0.1_gpt-3.5-turbo.json")


def random_boxes(rng, n, pixels):
    xy = rng.uniform(0, 0.7, (n, 2))
    wh = rng.uniform(0.05, 0.4, (n, 2))
    b = np.concatenate([xy, xy + wh], 1)
    if n > 3:
        b[1] = b[0] + 0.01  # a near-duplicate
        b[2] = [0.3, 0.3, 0.3, 0.3]  # zero area
    return (np.round(b * 512) if pixels else b).tolist()


def main():
    make_golden.install_shim()
    os.chdir(make_golden.REF)
    sys.path.insert(0, make_golden.REF)
    import torch
    import joblib
    from prompt import get_prompts
    from utils import cache, parse
    from utils.eval import class_aware_nms, evaluate_with_layout, nms, to_gen_box_format
    from utils.llm import get_parsed_layout
    spec = importlib.util.spec_from_file_location("ref_eval_owl_vit", os.path.join(make_golden.REF, "scripts", "eval_owl_vit.py"))
    owl = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(owl)
    gold = {}

    # 1. the benchmark prompt set
    pairs = get_prompts("lvd", return_predicates=True)
    gold["prompts"] = [{"prompt": p, "type": f.type, "texts": f.texts, "one_box_per_class": f.one_box_per_class} for p, f in pairs]
    gold["set_sizes"] = {t: len(get_prompts(t)) for t in ["lvd", "lvd_static", "lvd_numeracy", "lvd_attribution", "lvd_dynamic",
                                                           "lvd_dynamic_spatial", "lvd_visibility", "lvd_sequential"]}

    # 2. stage 1: the predicates over the cached LLM layouts (README.md:53-57 table)
    gold["stage1"] = {}
    for name in CACHES:
        cache.cache_path, cache.cache_format = "cache/" + name, "json"
        cache.init_cache(allow_nonexist=False)
        cache.reset_cache_access()
        outcome = []
        for prompt, pred in pairs:
            prompt = prompt.strip().rstrip(".")
            resp = cache.get_cache(prompt)
            layout, _ = get_parsed_layout(prompt, max_partial_response_retries=1, override_response=resp, json_template=False)
            kind, ok = evaluate_with_layout(layout, pred, 6, height=parse.size_h, width=parse.size_w)
            outcome.append([kind, bool(ok)])
        gold["stage1"][name] = outcome
        print(name, "overall", np.mean([o[1] for o in outcome]))
        with open(os.path.join(make_golden.REF, "cache", name), "rb") as src, gzip.GzipFile(
                os.path.join(ROOT, "tests", "golden", name + ".gz"), "wb", mtime=0) as dst:
            shutil.copyfileobj(src, dst)

    # 3. NMS known answers
    rng = np.random.RandomState(0)
    cases = []
    for n, thr, pixels in [(0, 0.5, False), (1, 0.5, False), (12, 0.5, False), (40, 0.3, False), (40, 0.7, True), (25, 0.5, True), (200, 0.5, False)]:
        boxes = random_boxes(rng, n, pixels)
        scores = np.round(rng.uniform(0, 1, n), 2).tolist()  # rounded: ties occur
        labels = rng.randint(0, 3, n).tolist()
        case = {"boxes": boxes, "scores": scores, "labels": labels, "threshold": thr, "pixels": pixels}
        for key, fn in (("nms", nms), ("class_aware_nms", class_aware_nms)):
            b, s, l = fn(boxes, scores, labels, thr, input_in_pixels=pixels)
            case[key] = {"boxes": np.asarray(b).tolist(), "scores": np.asarray(s).tolist(), "labels": np.asarray(l).tolist()}
        if n:
            b, s, l = owl.keep_one_box_per_class(np.array(boxes), np.array(scores), np.array(labels))
            case["one_per_class"] = {"boxes": b.tolist(), "scores": s.tolist(), "labels": l.tolist()}
        cases.append(case)
    gold["nms"] = cases
    gold["gen_box"] = [{"box": b, "w": w, "h": h, "rounded": to_gen_box_format(b, w, h, True), "raw": to_gen_box_format(b, w, h, False)}
                       for b, w, h in [([0.1, 0.25, 0.4, 0.755], 576, 320), ([0.0, 0.0, 1.0, 1.0], 256, 256), ([0.3333, 0.5, 0.3339, 0.9], 512, 512)]]

    # 4. eval_prompt end to end with a stub detector: detections are drawn around a scripted motion + clutter
    class Inputs(dict):
        def to(self, *_):
            return self

    class StubProcessor:
        def __init__(self):
            self.frame = 0

        def __call__(self, text, images, return_tensors):
            return Inputs(text=text)

        def post_process(self, outputs, target_sizes):
            d = self.dets[self.frame]
            self.frame += 1
            return [{"boxes": torch.tensor(d["boxes"], dtype=torch.float32).reshape(-1, 4), "scores": torch.tensor(d["scores"], dtype=torch.float32),
                     "labels": torch.tensor(d["labels"], dtype=torch.int64)}]

    videos = []
    tmp = tempfile.mkdtemp()
    H, W, F = 320, 576, 24
    picks = list(range(0, 500, 7))
    for j, idx in enumerate(picks):
        prompt, pred = pairs[idx]
        r = np.random.RandomState(1000 + idx)
        nq = len(pred.texts)
        dets = []
        for f in range(6):
            boxes, scores, labels = [], [], []
            for q in range(nq):
                for k in range(r.randint(0, 4)):
                    cx = {0: 0.15 + 0.14 * f, 1: 0.85 - 0.14 * f, 2: 0.5}[(j + q) % 3] + r.uniform(-0.05, 0.05)
                    cy = {0: 0.75 - 0.1 * f, 1: 0.3 + 0.08 * f}[(j // 3) % 2] + r.uniform(-0.05, 0.05)
                    w, h = r.uniform(0.1, 0.3, 2)
                    boxes.append([(cx - w / 2) * W, (cy - h / 2) * H, (cx + w / 2) * W, (cy + h / 2) * H])
                    scores.append(float(np.round(r.uniform(0.0, 0.6), 3)))
                    labels.append(q)
            if (j % 5 == 0 and f < 3) or (j % 5 == 1 and f >= 3):
                boxes, scores, labels = [], [], []
            dets.append({"boxes": boxes, "scores": scores, "labels": labels})
        path = os.path.join(tmp, f"video_{j}.joblib")
        joblib.dump(np.zeros((F, H, W, 3), dtype=np.uint8), path)
        entry = {"index": idx, "detections": dets, "results": {}}
        for aware in (False, True):
            proc = StubProcessor()
            proc.dets = dets
            kind, ok = owl.eval_prompt(prompt.strip().rstrip("."), pred, path, proc, lambda **kw: None, score_threshold=0.05, nms_threshold=0.5,
                                       use_class_aware_nms=aware, num_eval_frames=6, use_cuda=False)
            entry["results"]["class_aware" if aware else "plain"] = [kind, bool(ok)]
        videos.append(entry)
    shutil.rmtree(tmp)
    gold["videos"] = {"height": H, "width": W, "frames": F, "cases": videos}
    print("stub-detector videos:", len(videos), "successes:", sum(v["results"]["plain"][1] for v in videos))
    with open(os.path.join(ROOT, "tests", "golden", "eval.json"), "w") as f:
        json.dump(gold, f)


if __name__ == "__main__":
    main()
