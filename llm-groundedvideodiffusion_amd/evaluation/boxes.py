"""Detection post-processing between the detector and the predicates.

Restates /root/reference/utils/eval/eval.py:5-173 (greedy NMS across / within labels, layout evaluation, box format) and
the detection bookkeeping of /root/reference/scripts/eval_owl_vit.py:22-38,56-66,139-163.  Host-side numpy like the
reference: at most image_tokens (576) candidate boxes per frame and 6 frames per video.

NMS here builds the pairwise IoU table once and sweeps it in score order; the visiting order (ascending `np.argsort`,
consumed from the back) and the float64 IoU expression are the reference's, so picks are identical including ties.
"""
import numpy as np

from .. import dsl


def _iou_table(boxes, input_in_pixels):
    one = 1.0 if input_in_pixels else 0.0
    x0, y0, x1, y1 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    area = (x1 - x0 + one) * (y1 - y0 + one)
    w = np.maximum(0.0, np.minimum(x1[:, None], x1[None]) - np.maximum(x0[:, None], x0[None]) + one)
    h = np.maximum(0.0, np.minimum(y1[:, None], y1[None]) - np.maximum(y0[:, None], y0[None]) + one)
    inter = w * h
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (area[:, None] + area[None] - inter)


def nms(bounding_boxes, confidence_score, labels, threshold, input_in_pixels=False, return_array=True):
    """Label-agnostic greedy NMS: a kept box suppresses every lower-scored box with IoU >= threshold (a NaN IoU — two
    zero-area boxes — suppresses too, as `ratio < threshold` is False for NaN in the reference)."""
    if len(bounding_boxes) == 0:
        return np.array([]), np.array([]), np.array([])
    boxes = np.array(bounding_boxes)
    iou = _iou_table(boxes, input_in_pixels)
    alive = np.ones(len(boxes), dtype=bool)
    keep = []
    for i in np.argsort(np.array(confidence_score))[::-1]:
        if not alive[i]:
            continue
        keep.append(i)
        alive &= iou[i] < threshold
        alive[i] = False
    picked = [bounding_boxes[i] for i in keep], [confidence_score[i] for i in keep], [labels[i] for i in keep]
    return tuple(np.array(p) for p in picked) if return_array else picked


def class_aware_nms(bounding_boxes, confidence_score, labels, threshold, input_in_pixels=False):
    """NMS within each label, labels visited in sorted order."""
    if len(bounding_boxes) == 0:
        return np.array([]), np.array([]), np.array([])
    out = [], [], []
    for label in np.unique(labels):
        idx = [i for i in range(len(labels)) if labels[i] == label]
        picked = nms([bounding_boxes[i] for i in idx], [confidence_score[i] for i in idx], [label] * len(idx), threshold,
                     input_in_pixels=input_in_pixels, return_array=False)
        for acc, part in zip(out, picked):
            acc.extend(part)
    return tuple(np.array(p) for p in out)


def keep_one_box_per_class(boxes, scores, labels):
    """Highest-scoring box of every label (no tracker: avoids identity swaps between same-label boxes)."""
    boxes, scores, labels = np.asarray(boxes), np.asarray(scores), np.asarray(labels)
    uniq = np.unique(labels)
    best = [np.flatnonzero(labels == u)[scores[labels == u].argmax()] for u in uniq]
    return np.array([boxes[i] for i in best]), np.array([scores[i] for i in best]), np.array(list(uniq))


def to_gen_box_format(box, width, height, rounding):
    """(x_min, y_min, x_max, y_max) fractions -> the LLM layout format [x, y, w, h] in pixels."""
    x_min, y_min, x_max, y_max = box
    out = [x_min * width, y_min * height, (x_max - x_min) * width, (y_max - y_min) * height]
    return [round(v) for v in out] if rounding else out


def eval_frame_indices(num_frames, num_eval_frames=6):
    idx = np.round(np.linspace(0, num_frames - 1, num_eval_frames)).astype(int).tolist()
    assert len(set(idx)) == len(idx), f"Eval indices not unique: {idx}"
    return idx


def detections_to_layout(prompt, per_frame, texts, width, height):
    """per_frame: [(boxes (n,4) fractions, scores, labels)] for the evaluated frames -> a parsed layout in the DSL's
    format, so the stage-1 predicates score detections unchanged.  Object ids are label*100 + running index per label:
    boxes of different labels never share an id (eval_owl_vit.py:139-163)."""
    layout = {"Prompt": prompt, "Background keyword": None}
    for f, (boxes, scores, labels) in enumerate(per_frame):
        seen, frame = {}, []
        for box, score, label in zip(boxes, scores, labels):
            label = int(label)
            k = seen.get(label, 0)
            frame.append({"id": label * 100 + k, "name": texts[label], "box": to_gen_box_format(box, width, height, rounding=True),
                          "score": score})
            seen[label] = k + 1
        layout[f"Frame {f + 1}"] = frame
    return layout


def evaluate_with_layout(parsed_layout, predicate, num_parsed_layout_frames, height, width, verbose=False):
    """Layout -> box tracks at the layout's own frame count (no temporal resampling) -> predicate.  Returns (task, ok)."""
    condition = dsl.layout_to_condition(parsed_layout, tokenizer=None, height=height, width=width,
                                        num_parsed_layout_frames=num_parsed_layout_frames,
                                        num_condition_frames=num_parsed_layout_frames, strip_phrases=True)
    if verbose:
        print("condition:", condition)
    return predicate.type, predicate(condition, verbose=verbose)
