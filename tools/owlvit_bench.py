"""Time the OWL-ViT scorer (base-patch32 topology, random init) on the 6 evaluated frames of one 320x576 video."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lvd_amd  # noqa: E402,F401
from lvd_amd.evaluation.owlvit import HipOwlViTDetector, OwlViTConfig, synthetic_owlvit_state_dict  # noqa: E402

cfg = OwlViTConfig()
det = HipOwlViTDetector(cfg, synthetic_owlvit_state_dict(cfg), device="cuda")
frames = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (6, 320, 576, 3)).astype(np.uint8)).cuda()
ids = torch.zeros((2, 16), dtype=torch.long)
ids[:, 0], ids[:, 1:5], ids[:, 5] = 49406, torch.randint(1, 49000, (2, 4)), 49407
q, m = det.embed_queries(ids)
for _ in range(3):
    det.detect(frames, q, m)
torch.cuda.synchronize()
t = time.perf_counter()
n = 10
for _ in range(n):
    out = det.detect(frames, q, m)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / n
flops = 6 * (2 * 577 * (12 * (4 * 768 * 768 + 2 * 768 * 3072)) + 12 * 4 * 577 * 577 * 768 + 2 * 576 * 3072 * 768)
print(f"owlvit detect, 6 frames: {dt * 1e3:.2f} ms  ({flops / dt / 1e12:.1f} TFLOP/s, {flops / 1e9:.0f} GFLOP)")
