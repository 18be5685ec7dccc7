"""Per-video scoring and the per-task tally of scripts/eval_owl_vit.py (:41-178 eval_prompt, :262-303 summary)."""
import json

import numpy as np

from .boxes import class_aware_nms, detections_to_layout, eval_frame_indices, evaluate_with_layout, keep_one_box_per_class, nms


def score_video(prompt, predicate, video, detector, score_threshold=0.1, nms_threshold=0.5, use_class_aware_nms=False,
                num_eval_frames=6, verbose=False):
    """video: uint8 (F, H, W, 3) as saved by `vis.save_frames` (.joblib).  `detector(frames (n,H,W,3) uint8, texts)` returns,
    per frame, `(boxes_xyxy_pixels (P,4), scores (P,), labels (P,))` for every image token (the OWL-ViT post-process
    convention).  Returns (task, success)."""
    video = np.asarray(video)
    texts = predicate.texts
    height, width = video.shape[1:3]
    frames = video[eval_frame_indices(len(video), num_eval_frames)]
    per_frame = []
    for boxes, scores, labels in detector(frames, texts):
        boxes, scores, labels = (np.asarray(t.cpu() if hasattr(t, "cpu") else t) for t in (boxes, scores, labels))
        sel = scores >= score_threshold
        boxes = boxes[sel].astype(np.float64) / np.array([width, height, width, height], dtype=np.float64)
        scores, labels = scores[sel], labels[sel]
        boxes, scores, labels = (class_aware_nms if use_class_aware_nms else nms)(boxes, scores, labels, nms_threshold)
        if predicate.one_box_per_class and len(boxes):
            boxes, scores, labels = keep_one_box_per_class(boxes, scores, labels)
        if verbose:
            for b, s, l in zip(boxes, scores, labels):
                print(f"Detected {texts[int(l)]} ({l}) with confidence {round(float(s), 3)} at location {[round(float(v), 2) for v in b]}")
        per_frame.append((boxes, scores, labels))
    layout = detections_to_layout(prompt, per_frame, texts, width, height)
    return evaluate_with_layout(layout, predicate, num_parsed_layout_frames=num_eval_frames, height=height, width=width, verbose=verbose)


class ScoreBoard:
    """success / total per task in first-seen order, plus the summary line and eval.json of the reference."""

    def __init__(self):
        self.success, self.total, self.outcomes = {}, {}, {}

    def add(self, task, ok):
        self.success[task] = self.success.get(task, 0) + int(ok)
        self.total[task] = self.total.get(task, 0) + 1
        self.outcomes.setdefault(task, []).append(bool(ok))

    def rates(self):
        return {k: self.success[k] / self.total[k] for k in self.total}

    def overall(self):
        n = sum(self.total.values())
        return sum(self.success.values()) / n if n else float("nan")

    def report(self):
        lines = [f"Eval type: {k}, success: {self.success[k]}/{self.total[k]}, rate: {round(r, 2):.2f}" for k, r in self.rates().items()]
        lines.append(f"Overall: success: {sum(self.success.values())}/{sum(self.total.values())}, rate: {self.overall():.2f}")
        lines.append("Summary: " + "/".join(f"{round(r, 2):.2f}" for r in list(self.rates().values()) + [self.overall()]))
        return "\n".join(lines)

    def save(self, path):
        with open(path, "w") as f:
            json.dump({"success_counts": self.success, "sample_counts": self.total, "successes": self.outcomes,
                       "success_counts_overall": sum(self.success.values()), "sample_counts_overall": sum(self.total.values())}, f, indent=4)
