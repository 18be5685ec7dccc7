"""DSL front-end against golden vectors produced by the reference's own parse/cache/token functions on its shipped caches
(tests/golden/dsl.json, generator oracle/make_golden_dsl.py): bit-exact layouts, boxes, phrases and token positions."""
import json
import os

import numpy as np
import pytest

import lvd_amd  # noqa: F401
from lvd_amd import dsl
from oracle.fake_tokenizer import FakeTokenizer

CASES = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dsl.json")))


def test_golden_has_the_demo_prompt_of_the_readme():
    demo = [c for c in CASES if c["cache"].startswith("cache_demo")][0]
    assert demo["prompt"] == "A bear walks from the left to the right"
    assert demo["cond_prompt"] == "A bear walks from the left to the right, forest background"
    assert demo["phrases"] == ["bear"] and demo["object_positions"] == [[2]]
    assert np.allclose(demo["boxes"][0][0], [0, 0.5, 0.1953125, 0.6953125]) and abs(demo["boxes"][0][-1][0] - 0.8301) < 1e-3


@pytest.mark.parametrize("i", range(len(CASES)))
def test_layout_and_condition_match_reference(i):
    c = CASES[i]
    layout = dsl.parse_layout_response(c["prompt"], c["response"])
    assert layout == c["layout"]
    cond = dsl.layout_to_condition(layout, num_condition_frames=24, tokenizer=FakeTokenizer())
    assert cond.prompt == c["cond_prompt"] and cond.phrases == c["phrases"]
    assert cond.object_positions == c["object_positions"] and cond.token_map == c["token_map"]
    assert cond.boxes == c["boxes"], "interpolated boxes must be bit-identical (same numpy ops)"
    assert dsl.layout_to_condition(layout, num_condition_frames=16).boxes == c["boxes16"]


def test_sequential_cache_and_errors(tmp_path):
    p = tmp_path / "cache.json"
    p.write_text(json.dumps({"a": ["r1", "r2"], "b": ["r3"]}))
    cache = dsl.LayoutCache(str(p))
    assert [cache.get("a"), cache.get("a"), cache.get("a"), cache.get("b"), cache.get("zzz")] == ["r1", "r2", None, "r3", None]
    assert cache.values_accessed() == 3
    cache.reset_access()
    assert cache.get("a") == "r1"
    with pytest.raises(FileNotFoundError):
        dsl.LayoutCache(str(tmp_path / "missing.json"))
    with pytest.raises(dsl.LayoutParseError):
        dsl.parse_layout_response("x", "Frame 1: []\nFrame 2: []")
    with pytest.raises(SyntaxError):
        dsl.parse_layout_response("x", "".join(f"Frame {k}: [{{'id': 0\n" for k in range(1, 7)) + "Background keyword: room")
    ok = "Reasoning: r\n" + "".join(f"Frame {k}: - [{{'id': 0, 'name': 'cat', 'box': [0, 0, 10, 10]}}] - moves\n" for k in range(1, 7)) + "Background keyword: room\n"
    lay = dsl.parse_layout_response("a cat", ok)
    assert lay["Frame 3"] == [{"id": 0, "name": "cat", "box": [0, 0, 10, 10]}] and lay["Background keyword"] == "room"


def test_draw_boxes_annotates_present_objects_only(tmp_path):
    import numpy as np
    from lvd_amd import vis
    frames = np.zeros((3, 40, 60, 3), dtype=np.uint8)
    boxes = [[[0.1, 0.2, 0.5, 0.8]] * 3, [[0.0, 0.0, 0.0, 0.0], [0.6, 0.1, 0.9, 0.4], [0.6, 0.1, 0.9, 0.4]]]
    out = vis.draw_boxes(frames, boxes, ["bear", "bird"])
    assert out.shape == frames.shape and out.dtype == np.uint8 and frames.max() == 0  # input untouched
    red = (out[..., 0] > 200) & (out[..., 1] < 50)
    assert red[0, 8, 6:30].all() and not red[0, :, 36:].any()      # frame 0: only the bear's outline (bird absent: all-zero box)
    assert red[1, 4, 36:54].all() and red[1, 8, 6:30].all()        # frame 1: both
    vis.save_frames(str(tmp_path / "v_with_box"), out, "gif")
    assert (tmp_path / "v_with_box.gif").exists()
