"""Benchmark scoring against goldens produced by the reference's own scoring code
(oracle/make_golden_eval.py) and against the stage-1 table the reference publishes (README.md:53-57)."""
import gzip
import json
import os

import numpy as np
import pytest

import lvd_amd  # noqa: F401
from lvd_amd import dsl
from lvd_amd.evaluation import (ScoreBoard, class_aware_nms, get_prompts, keep_one_box_per_class, nms, score_video,
                                to_gen_box_format)
from lvd_amd.evaluation.boxes import evaluate_with_layout

GOLD = os.path.join(os.path.dirname(__file__), "golden")
with open(os.path.join(GOLD, "eval.json")) as f:
    G = json.load(f)


def test_prompt_set_matches_reference():
    pairs = get_prompts("lvd", return_predicates=True)
    assert len(pairs) == len(G["prompts"]) == 500
    for (prompt, pred), g in zip(pairs, G["prompts"]):
        assert prompt == g["prompt"]
        assert (pred.type, pred.texts, pred.one_box_per_class) == (g["type"], g["texts"], g["one_box_per_class"])
    for kind, n in G["set_sizes"].items():
        assert len(get_prompts(kind)) == n
    assert get_prompts("demo") == ["A bear walks from the left to the right"]
    with pytest.raises(ValueError):
        get_prompts("nope")
    with pytest.raises(AssertionError):
        get_prompts("demo", return_predicates=True)


def test_every_prompt_is_a_key_of_the_shipped_caches():
    for name in G["stage1"]:
        with gzip.open(os.path.join(GOLD, name + ".gz"), "rt") as f:
            keys = set(json.load(f).keys())
        prompts = {p.strip().rstrip(".") for p in get_prompts("lvd")}
        assert prompts == keys, (len(prompts), len(keys))


@pytest.mark.parametrize("name,published", [("cache_lvd_v0.1_gpt-4-1106-preview.json", (100, 100, 100, 100, 88, 98)),
                                            ("cache_lvd_v0.1_gpt-3.5-turbo.json", (100, 100, 100, 73, 15, 78))])
def test_stage_one_table(tmp_path, name, published):
    """Cached LLM layouts -> DSL parser -> predicates: per-prompt outcome equals the reference's, and the per-task rates are
    the rows of the reference's README (GPT-3.5* is the rerun whose cache is shipped)."""
    path = tmp_path / name
    with gzip.open(os.path.join(GOLD, name + ".gz"), "rb") as f:
        path.write_bytes(f.read())
    cache = dsl.LayoutCache(str(path))
    board = ScoreBoard()
    for (prompt, pred), (kind, ok) in zip(get_prompts("lvd", return_predicates=True), G["stage1"][name]):
        prompt = prompt.strip().rstrip(".")
        layout = dsl.parse_layout_response(prompt, cache.get(prompt))
        got = evaluate_with_layout(layout, pred, 6, height=dsl.LAYOUT_SIZE[0], width=dsl.LAYOUT_SIZE[1])
        assert got == (kind, ok), prompt
        board.add(*got)
    rates = board.rates()
    assert list(rates) == ["numeracy", "attribution", "visibility", "dynamic_spatial", "sequential"]
    assert tuple(round(100 * r) for r in rates.values()) + (round(100 * board.overall()),) == published
    assert "Overall: success:" in board.report() and board.report().splitlines()[-1].startswith("Summary: 1.00/1.00/1.00/")


@pytest.mark.parametrize("i", range(len(G["nms"])))
def test_nms_known_answers(i):
    c = G["nms"][i]
    for key, fn in (("nms", nms), ("class_aware_nms", class_aware_nms)):
        b, s, l = fn(c["boxes"], c["scores"], c["labels"], c["threshold"], input_in_pixels=c["pixels"])
        assert np.asarray(b).tolist() == c[key]["boxes"] and np.asarray(s).tolist() == c[key]["scores"] and np.asarray(l).tolist() == c[key]["labels"]
    if c["boxes"]:
        b, s, l = keep_one_box_per_class(np.array(c["boxes"]), np.array(c["scores"]), np.array(c["labels"]))
        assert (b.tolist(), s.tolist(), l.tolist()) == (c["one_per_class"]["boxes"], c["one_per_class"]["scores"], c["one_per_class"]["labels"])


def test_nms_properties():
    rng = np.random.RandomState(3)
    xy = rng.uniform(0, 0.6, (300, 2))
    boxes = np.concatenate([xy, xy + rng.uniform(0.05, 0.4, (300, 2))], 1)
    scores, labels = rng.uniform(0, 1, 300), rng.randint(0, 4, 300)
    b, s, l = nms(boxes, scores, labels, 0.5)
    assert np.all(np.diff(s) <= 0) and s[0] == scores.max()
    b2, s2, _ = nms(b, s, l, 0.5)  # idempotent
    assert np.array_equal(b, b2) and np.array_equal(s, s2)
    assert len(nms(boxes, scores, labels, 1.01)[0]) == 300  # nothing overlaps that much
    bc, _, lc = class_aware_nms(boxes, scores, labels, 0.5)
    assert len(bc) >= len(b) and sorted(set(lc.tolist())) == [0, 1, 2, 3]


def test_gen_box_format():
    for g in G["gen_box"]:
        assert to_gen_box_format(g["box"], g["w"], g["h"], True) == g["rounded"]
        assert to_gen_box_format(g["box"], g["w"], g["h"], False) == g["raw"]


def test_video_scoring_matches_reference_eval_prompt():
    """Stub detections -> threshold -> NMS -> one box per class -> layout -> predicate, against the outcome of the reference's
    `eval_prompt` fed the same detections."""
    pairs = get_prompts("lvd", return_predicates=True)
    v = G["videos"]
    video = np.zeros((v["frames"], v["height"], v["width"], 3), dtype=np.uint8)
    n_ok = 0
    for case in v["cases"]:
        prompt, pred = pairs[case["index"]]

        def detector(frames, texts, dets=case["detections"]):
            assert frames.shape == (6, v["height"], v["width"], 3) and texts == pred.texts
            return [(np.array(d["boxes"], dtype=np.float32).reshape(-1, 4), np.array(d["scores"], dtype=np.float32),
                     np.array(d["labels"], dtype=np.int64)) for d in dets]

        for aware, key in ((False, "plain"), (True, "class_aware")):
            got = score_video(prompt.strip().rstrip("."), pred, video, detector, score_threshold=0.05, nms_threshold=0.5,
                              use_class_aware_nms=aware, num_eval_frames=6)
            assert list(got) == case["results"][key], (case["index"], key)
        n_ok += case["results"]["plain"][1]
    assert 0 < n_ok < len(v["cases"])  # the fixture exercises both outcomes


def test_scoreboard_json(tmp_path):
    board = ScoreBoard()
    for kind, ok in [("numeracy", True), ("numeracy", False), ("sequential", True)]:
        board.add(kind, ok)
    board.save(tmp_path / "eval.json")
    d = json.load(open(tmp_path / "eval.json"))
    assert d["success_counts"] == {"numeracy": 1, "sequential": 1} and d["sample_counts_overall"] == 3
    assert d["successes"]["numeracy"] == [True, False]


def test_generate_cli_walks_the_benchmark_prompts(tmp_path, monkeypatch, capsys):
    """`generate.py --prompt-type lvd --dry-run`: the 500 benchmark prompts in order, every one served by the shipped GPT-4
    cache (repeated prompts consume successive cached responses), 250 per rank when sharded over two ranks."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("generate_cli", os.path.join(root, "generate.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    name = "cache_lvd_v0.1_gpt-4-1106-preview.json"
    with gzip.open(os.path.join(GOLD, name + ".gz"), "rb") as f:
        (tmp_path / name).write_bytes(f.read())
    base = ["--model", "gpt-4", "--template_version", "v0.1", "--prompt-type", "lvd", "--run-model", "lvd_zeroscope", "--dry-run",
            "--cache-dir", str(tmp_path), "--img-root", str(tmp_path / "img")]
    gen.main(base)
    out = capsys.readouterr().out
    assert out.count("parsed_layout:") == 500 and "Cache miss" not in out
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "2")
    gen.main(base + ["--force_run_ind", "7"])
    out = capsys.readouterr().out
    assert out.count("parsed_layout:") == 250 and "run7" in out
