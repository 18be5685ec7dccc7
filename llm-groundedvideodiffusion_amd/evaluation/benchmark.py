"""The LVD benchmark: 5 tasks x 100 prompts, each paired with a predicate over a `dsl.Condition` (per-object box tracks).

Restates /root/reference/utils/eval/lvd.py:34-373 (prompt construction, counts, repeat structure, the seeded draws of the
attribution / two-object prompts) and /root/reference/utils/eval/utils.py:49-311 (predicate semantics).  The prompt strings
must equal the keys of the GPT-4 / GPT-3.5 layout caches the reference ships, and the predicates applied to those cached
layouts must reproduce the reference's stage-1 table (README.md:53-57) — both are tested (tests/test_evaluation.py).

A predicate is a callable `predicate(condition, verbose=False) -> bool` carrying `.type` (task name), `.texts` (detector
queries) and `.one_box_per_class`, the protocol scripts/eval_owl_vit.py:41-178 and scripts/eval_stage_one.py:50-73 rely on.
"""
from dataclasses import dataclass, field
from typing import Callable, List

import numpy as np

from .words import number_word, pluralize, with_article

SCENE = "A realistic lively video of a scene"
SCENE_TOP_DOWN = "A realistic lively video of a top-down viewed scene"
# (name with a motion attribute, bare noun); the bare noun is what the detector is queried with
CLASSES = (("moving car", "car"), ("lively cat", "cat"), ("flying bird", "bird"), ("moving ball", "ball"), ("walking dog", "dog"))
NOUNS = [noun for _, noun in CLASSES]
COLOURS = ["red", "orange", "yellow", "green", "blue", "purple", "pink", "brown", "black", "white", "gray"]
PROMPT_TYPES = ("lvd", "lvd_static", "lvd_numeracy", "lvd_attribution", "lvd_dynamic", "lvd_dynamic_spatial", "lvd_visibility",
                "lvd_sequential")


@dataclass
class Predicate:
    type: str
    texts: List[str]
    one_box_per_class: bool
    check: Callable = field(repr=False)

    def __call__(self, condition, verbose=False):
        ok = bool(self.check(condition))
        if verbose:
            print(f"[{self.type}] {ok} for phrases {condition.phrases}")
        return ok


# ---- box-track helpers (boxes are (x_min, y_min, x_max, y_max) fractions; an absent object is an all-zero box) ----------

def _track(condition, names):
    """Track of the first object whose phrase contains one of `names` at a word boundary ("car" must not hit "carrot")."""
    for phrase, track in zip(condition.phrases, condition.boxes):
        if any((n + " ") in phrase or phrase.endswith(n) for n in names):
            return track if len(track) else None
    return None


def _present(track):
    t = np.asarray(track, dtype=np.float64).reshape(-1, 4)
    return (t[:, 2] != 0) & (t[:, 3] != 0)  # the reference tests columns 2 and 3 (utils.py:142-149)


def _cx(b):
    return (b[0] + b[2]) / 2


def _cy(b):
    return (b[1] + b[3]) / 2


# ---- predicates --------------------------------------------------------------------------------------------------------------

def _count_is(names, wanted):
    def check(c):
        assert len(names) == 1
        if len(c.boxes) == 0:
            return wanted == 0
        per_frame = np.sum([_present(t) for t in c.boxes], axis=0).astype(np.int64)
        return int(np.bincount(per_frame).argmax()) == wanted  # the count seen in most frames
    return check


def _visible_in_half(names, second_half):
    def check(c):
        track = _track(c, names)
        if track is None:
            return False
        p = _present(track)
        mid = len(p) // 2
        early, late = p[: mid - 1].any(), p[mid + 1:].any()  # the two middle frames are ignored
        return (not early and late) if second_half else (early and not late)
    return check


def _mostly_present(*name_sets):
    def check(c):
        for names in name_sets:
            track = _track(c, names)
            if track is None or _present(track).mean() < 0.5:
                return False
        return True
    return check


def _moves(names, before):
    """`before(a, b)`: a lies before b along the direction of motion."""
    def check(c):
        track = _track(c, names)
        if track is None:
            return False
        p = _present(track)
        return bool(p[0] and p[-1]) and before(track[0], track[-1])
    return check


def _passes(names_a, names_b, before):
    def check(c):
        a, b = _track(c, names_a), _track(c, names_b)
        if a is None or b is None:
            return False
        pa, pb = _present(a), _present(b)
        if not (pa[0] and pb[0] and pa[-1] and pb[-1]):
            return False
        return before(a[0], b[0]) and before(b[-1], a[-1])
    return check


def _visits(names, corners):
    def check(c):
        track = _track(c, names)
        if track is None:
            return False
        p = _present(track)
        stops = (0, len(p) // 2, -1)
        return all(p[s] for s in stops) and all(corner(track[s]) for s, corner in zip(stops, corners))
    return check


_DIRECTIONS = (("left", "right", lambda a, b: _cx(a) < _cx(b)), ("right", "left", lambda a, b: _cx(a) > _cx(b)))
_VERTICAL = (("top", "bottom", lambda a, b: _cy(a) < _cy(b)), ("bottom", "top", lambda a, b: _cy(a) > _cy(b)))
_CORNERS = {  # image coordinates: "lower" = large y
    "lower left": lambda b: _cy(b) > 0.5 and _cx(b) < 0.5,
    "lower right": lambda b: _cy(b) > 0.5 and _cx(b) > 0.5,
    "upper left": lambda b: _cy(b) < 0.5 and _cx(b) < 0.5,
    "upper right": lambda b: _cy(b) < 0.5 and _cx(b) > 0.5,
}
_ROUTES = (("lower left", "lower right", "upper right"), ("lower left", "upper left", "upper right"),
           ("lower right", "lower left", "upper left"), ("lower right", "upper right", "upper left"))


def _photo(phrase):
    return f"a photo of {with_article(phrase)}"


# ---- task builders: lists of (prompt, Predicate), repeats adjacent as in the reference ----------------------------------

def numeracy_task(min_num=1, max_num=5, repeat=2):
    out = []
    for n in range(min_num, max_num + 1):
        for name, noun in CLASSES:
            prompt = f"{SCENE} with {number_word(n)} {pluralize(name) if n > 1 else name}"
            out += [(prompt, Predicate("numeracy", [_photo(noun)], False, _count_is((noun,), n)))] * repeat
    return out


def attribution_task(num_prompts=100, repeat=1):
    out = []
    for ind in range(num_prompts):
        rng = np.random.RandomState(ind)  # == np.random.seed(ind) + np.random.choice of the reference (lvd.py:85-91)
        colour1, colour2 = rng.choice(COLOURS, 2, replace=False)
        noun1, noun2 = rng.choice(NOUNS, 2, replace=False)
        prompt = f"{SCENE} with {with_article(colour1)} {noun1} and {with_article(colour2)} {noun2}"
        pred = Predicate("attribution", [_photo(f"{colour1} {noun1}"), _photo(f"{colour2} {noun2}")], True,
                         _mostly_present((f"{colour1} {noun1}",), (f"{colour2} {noun2}",)))
        out += [(prompt, pred)] * repeat
    return out


def visibility_task(repeat=2):
    out = []
    for name, noun in CLASSES:
        for half, second in (("second", True), ("first", False)):
            prompt = f"{SCENE} in which {with_article(name)} appears only in the {half} half of the video"
            out += [(prompt, Predicate("visibility", [_photo(noun)], True, _visible_in_half((noun,), second)))] * repeat
    return out


def one_object_motion_task(repeat=1, left_right_only=True):
    out = []
    for noun in NOUNS:
        for src, dst, before in _DIRECTIONS + (() if left_right_only else _VERTICAL):
            prompt = f"{SCENE} with {with_article(noun)} moving from the {src} to the {dst}"
            out += [(prompt, Predicate("dynamic_spatial", [_photo(noun)], True, _moves((noun,), before)))] * repeat
    return out


def two_object_motion_task(num_prompts=10, repeat=1, left_right_only=True):
    out = []
    for ind in range(num_prompts):
        rng = np.random.RandomState(ind)  # one stream per index, one draw per direction (lvd.py:258-263)
        for src, dst, before in _DIRECTIONS + (() if left_right_only else _VERTICAL):
            noun1, noun2 = rng.choice(NOUNS, 2, replace=False)
            prompt = f"{SCENE} with {with_article(noun1)} moving from the {src} of {with_article(noun2)} to its {dst}"
            pred = Predicate("dynamic_spatial", [_photo(noun1), _photo(noun2)], True, _passes((noun1,), (noun2,), before))
            out += [(prompt, pred)] * repeat
    return out


def sequential_task(repeat=1):
    out = []
    for noun in NOUNS:
        for route in _ROUTES:
            prompt = (f"{SCENE_TOP_DOWN} in which {with_article(noun)} initially on the {route[0]} of the scene. It first moves to the "
                      f"{route[1]} of the scene and then moves to the {route[2]} of the scene.")
            pred = Predicate("sequential", [_photo(noun)], True, _visits((noun,), [_CORNERS[r] for r in route]))
            out += [(prompt, pred)] * repeat
    return out


def lvd_prompt_predicates(prompt_type=None):
    """100 prompts per task (lvd.py:314-373): numeracy 1-4 x 5 classes x 5, attribution 100 draws, visibility 5 x 2 x 10,
    dynamics 5 x 2 x 5 one-object + 25 x 2 two-object, sequential 5 x 4 x 5."""
    numeracy = numeracy_task(max_num=4, repeat=5)
    attribution = attribution_task(num_prompts=100)
    visibility = visibility_task(repeat=10)
    dynamics = one_object_motion_task(repeat=5) + two_object_motion_task(num_prompts=25)
    sequential = sequential_task(repeat=5)
    sets = {
        "lvd": numeracy + attribution + visibility + dynamics + sequential,
        "lvd_static": numeracy + attribution,
        "lvd_numeracy": numeracy,
        "lvd_attribution": attribution,
        "lvd_dynamic": visibility + dynamics + sequential,
        "lvd_dynamic_spatial": dynamics,
        "lvd_visibility": visibility,
        "lvd_sequential": sequential,
    }
    return sets if prompt_type is None else sets[prompt_type]


PROMPTS_DEMO = ["A bear walks from the left to the right"]  # prompt.py:72-74


def get_prompts(prompt_type, return_predicates=False):
    """prompt.py:82-98."""
    if prompt_type.startswith("lvd"):
        pairs = lvd_prompt_predicates(prompt_type)
        return pairs if return_predicates else [p for p, _ in pairs]
    if prompt_type == "demo":
        assert not return_predicates, "Predicates are not supported for this prompt type"
        return list(PROMPTS_DEMO)
    raise ValueError(f"Unknown prompt type: {prompt_type}")
