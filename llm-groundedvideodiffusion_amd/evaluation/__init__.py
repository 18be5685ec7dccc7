"""Benchmark scoring: the five-task LVD prompt set with its predicates, detection post-processing
(NMS, one-box-per-class, detections -> layout) and the OWL-ViT detector on the HIP kernels.

Reference: /root/reference/utils/eval/{eval,lvd,utils}.py, /root/reference/scripts/eval_owl_vit.py,
/root/reference/scripts/eval_stage_one.py, /root/reference/prompt.py:79-96.
"""
from .benchmark import Predicate, get_prompts, lvd_prompt_predicates, PROMPT_TYPES  # noqa: F401
from .boxes import (class_aware_nms, detections_to_layout, evaluate_with_layout, eval_frame_indices,  # noqa: F401
                    keep_one_box_per_class, nms, to_gen_box_format)
from .scoring import ScoreBoard, score_video  # noqa: F401
