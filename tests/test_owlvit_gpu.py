"""OWL-ViT scorer on the HIP kernels against `transformers.OwlViTForObjectDetection` — the third-party
implementation the reference calls (scripts/eval_owl_vit.py:208-212) — on random-init weights, and against Pillow for the
image preprocessing.  The CPU tests pin the host pieces (weight names, grid bias, resampling tables)."""
import numpy as np
import pytest
import torch

import lvd_amd  # noqa: F401
from lvd_amd import ops
from lvd_amd.evaluation import get_prompts, score_video
from lvd_amd.evaluation.owlvit import (CLIP_MEAN, CLIP_STD, HipOwlViTDetector, OwlViTConfig, box_bias, synthetic_owlvit_state_dict)
from lvd_amd.text_encoder import CLIPTextConfig

TINY = OwlViTConfig(image_size=224, patch_size=32, hidden_size=192, intermediate_size=384, num_hidden_layers=3, num_attention_heads=3,
                    projection_dim=128, text=CLIPTextConfig(vocab_size=1000, hidden_size=128, intermediate_size=256, num_hidden_layers=2,
                                                            num_attention_heads=2, max_position_embeddings=16, hidden_act="quick_gelu", eos_token_id=2))


def hf_model(cfg, sd):
    import transformers
    t = cfg.text
    hf_cfg = transformers.OwlViTConfig(
        text_config=dict(vocab_size=t.vocab_size, hidden_size=t.hidden_size, intermediate_size=t.intermediate_size, num_hidden_layers=t.num_hidden_layers,
                         num_attention_heads=t.num_attention_heads, max_position_embeddings=t.max_position_embeddings, hidden_act=t.hidden_act),
        vision_config=dict(hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size, num_hidden_layers=cfg.num_hidden_layers,
                           num_attention_heads=cfg.num_attention_heads, image_size=cfg.image_size, patch_size=cfg.patch_size, hidden_act=cfg.hidden_act),
        projection_dim=cfg.projection_dim)
    m = transformers.OwlViTForObjectDetection(hf_cfg).eval()
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and set(missing) <= {"owlvit.logit_scale"}, (missing, unexpected)
    return m


def pil_pixel_values(frames, size):
    from PIL import Image
    out = []
    for f in frames:
        img = np.asarray(Image.fromarray(f).resize((size, size), Image.BICUBIC)).astype(np.float32)
        out.append(((img * (1 / 255.0) - np.array(CLIP_MEAN, dtype=np.float32)) / np.array(CLIP_STD, dtype=np.float32)).transpose(2, 0, 1))
    return torch.from_numpy(np.stack(out))


def token_ids(cfg, n_queries, seed=0, pad_last=False):
    """CLIP-style ids: <bos> words <eos> then zero padding; <eos> is the highest id."""
    rng = np.random.RandomState(seed)
    L, V = cfg.text.max_position_embeddings, cfg.text.vocab_size
    ids = np.zeros((n_queries, L), dtype=np.int64)
    for q in range(n_queries):
        n = rng.randint(3, 8)
        ids[q, 0], ids[q, 1:1 + n], ids[q, 1 + n] = V - 2, rng.randint(1, V - 2, n), V - 1
    if pad_last:
        ids[-1] = 0
    return torch.from_numpy(ids)


def random_frames(n, h, w, seed=0):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (n, h // 8 + 1, w // 8 + 1, 3)).astype(np.uint8)
    return np.ascontiguousarray(np.kron(base, np.ones((1, 8, 8, 1), dtype=np.uint8))[:, :h, :w] // 2 + rng.randint(0, 128, (n, h, w, 3)).astype(np.uint8))


def hf_reference(model, frames, ids, size, target_hw):
    rep = ids.repeat(len(frames), 1)  # the model takes one query set per image; the scorer asks every frame the same queries
    with torch.no_grad():
        out = model(input_ids=rep, pixel_values=pil_pixel_values(frames, size), attention_mask=(rep > 0).long())
    logits, cxcywh = out.logits.float(), out.pred_boxes.float()
    best = logits.max(-1)
    cx, cy, w, h = cxcywh.unbind(-1)
    H, W = target_hw
    boxes = torch.stack([cx - 0.5 * w, cy - 0.5 * h, cx + 0.5 * w, cy + 0.5 * h], -1) * torch.tensor([W, H, W, H], dtype=torch.float32)
    return logits, best.values.sigmoid(), best.indices, boxes


# ---- CPU -----------------------------------------------------------------------------------------------------------------

def test_synthetic_weights_have_the_checkpoint_layout():
    m = hf_model(TINY, synthetic_owlvit_state_dict(TINY))
    assert torch.allclose(box_bias(TINY.image_size // TINY.patch_size), m.box_bias, atol=1e-6)
    names = set(synthetic_owlvit_state_dict(OwlViTConfig(num_hidden_layers=1, text=CLIPTextConfig(
        vocab_size=49408, hidden_size=512, intermediate_size=2048, num_hidden_layers=1, num_attention_heads=8, max_position_embeddings=16,
        hidden_act="quick_gelu"))))
    assert "owlvit.vision_model.embeddings.patch_embedding.weight" in names and "box_head.dense2.bias" in names


@pytest.mark.parametrize("kind", ["bicubic", "lanczos"])
@pytest.mark.parametrize("n_in,n_out", [(576, 768), (320, 768), (1000, 768), (37, 64), (576, 1024)])
def test_resampling_table_is_pillows(n_in, n_out, kind):
    """One axis at a time: a 1-pixel-high (or wide) image resized along the other axis only goes through a single pass."""
    from PIL import Image
    rng = np.random.RandomState(n_in)
    line = rng.randint(0, 256, (1, n_in, 3)).astype(np.uint8)
    bounds, coef = ops.pil_resample_table(n_in, n_out, kind)
    got = np.zeros((n_out, 3), dtype=np.int64)
    for o in range(n_out):
        lo, n = bounds[o]
        got[o] = np.clip(((line[0, lo:lo + n].astype(np.int64) * coef[o, :n, None]).sum(0) + (1 << 21)) >> 22, 0, 255)
    ref = np.asarray(Image.fromarray(line).resize((n_out, 1), Image.BICUBIC if kind == "bicubic" else Image.LANCZOS))[0]
    assert np.array_equal(got, ref)


# ---- GPU -----------------------------------------------------------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("h,w,size,patch", [(320, 576, 768, 32), (256, 256, 224, 32), (500, 900, 224, 16)])
def test_frames_to_patches_matches_pillow(h, w, size, patch):
    from PIL import Image
    frames = random_frames(2, h, w, seed=h)
    patches, resized = ops.frames_to_patches(torch.from_numpy(frames).cuda(), size, patch, CLIP_MEAN, CLIP_STD, return_resized=True)
    ref_img = np.stack([np.asarray(Image.fromarray(f).resize((size, size), Image.BICUBIC)) for f in frames])
    assert np.array_equal(resized.cpu().numpy(), ref_img), "resize is not Pillow's"
    g = size // patch
    pv = pil_pixel_values(frames, size)  # (B,3,S,S)
    ref = pv.reshape(2, 3, g, patch, g, patch).permute(0, 2, 4, 1, 3, 5).reshape(2 * g * g, 3 * patch * patch)
    assert torch.equal(patches.cpu(), ref.to(torch.bfloat16)) or (patches.cpu().float() - ref).abs().max() < 2e-2


@pytest.mark.gpu
def test_owl_detect_rows_matches_formula():
    g = torch.Generator().manual_seed(0)
    rows_per, B, D, Q = 49, 3, 128, 3
    rows = rows_per * B
    emb = torch.randn(rows, D, generator=g)
    queries = torch.nn.functional.normalize(torch.randn(Q, D, generator=g), dim=-1)
    shsc = torch.randn(rows, 4, generator=g)
    raw = torch.randn(rows, 4, generator=g)
    bias = box_bias(7)
    mask = torch.tensor([1, 1, 0], dtype=torch.int32)
    logits, scores, labels, boxes = ops.owl_detect_rows(emb.cuda(), queries.cuda(), shsc.cuda(), raw.cuda(), bias.cuda(), rows_per, 576, 320,
                                                        query_mask=mask.cuda())
    e = emb / (emb.norm(dim=-1, keepdim=True) + 1e-6)
    ref = (e @ queries.T + shsc[:, :1]) * (torch.nn.functional.elu(shsc[:, 1:2]) + 1)
    ref = torch.where(mask[None] == 0, torch.finfo(torch.float32).min, ref)
    assert torch.allclose(logits.cpu(), ref, atol=2e-5, rtol=1e-5)
    assert torch.equal(labels.cpu(), ref.argmax(-1)) and torch.allclose(scores.cpu(), ref.max(-1).values.sigmoid(), atol=1e-6)
    c = (raw + bias.repeat(B, 1)).sigmoid()
    ref_boxes = torch.stack([c[:, 0] - c[:, 2] / 2, c[:, 1] - c[:, 3] / 2, c[:, 0] + c[:, 2] / 2, c[:, 1] + c[:, 3] / 2], -1) * torch.tensor([576., 320., 576., 320.])
    assert torch.allclose(boxes.cpu(), ref_boxes, atol=1e-3)


def _compare(cfg, frames, ids, seed):
    sd = synthetic_owlvit_state_dict(cfg, seed=seed)
    det = HipOwlViTDetector(cfg, sd, device="cuda")
    H, W = frames.shape[1:3]
    ref_logits, ref_scores, ref_labels, ref_boxes = hf_reference(hf_model(cfg, sd), frames, ids, cfg.image_size, (H, W))
    queries, mask = det.embed_queries(ids)
    out = det.detect(torch.from_numpy(frames), queries, mask)
    valid = (ids[:, 0] > 0)
    lg, rl = out["logits"].cpu()[..., valid], ref_logits[..., valid]
    rel = ((lg - rl).norm() / rl.norm()).item()
    box_err = (out["boxes"].cpu() - ref_boxes).abs().max().item() / max(H, W)
    score_err = (out["scores"].cpu() - ref_scores).abs().max().item()
    # labels may differ only where the two best logits are closer than the bf16 noise
    top2 = rl.topk(min(2, rl.shape[-1]), -1).values
    decided = (top2[..., 0] - top2[..., -1]) > 2 * (lg - rl).abs().max(-1).values if rl.shape[-1] > 1 else torch.ones_like(ref_labels, dtype=torch.bool)
    label_ok = (out["labels"].cpu() == ref_labels)[decided].all().item()
    print(f"owlvit parity: logits rel-L2 {rel:.3e}, box err {box_err:.3e} of the frame, score err {score_err:.3e}, undecided labels {(~decided).sum().item()}")
    return rel, box_err, score_err, label_ok


@pytest.mark.gpu
def test_tiny_detector_matches_transformers():
    frames = random_frames(3, 96, 160, seed=1)
    rel, box_err, score_err, label_ok = _compare(TINY, frames, token_ids(TINY, 3, seed=2, pad_last=True), seed=3)
    assert rel < 2e-2 and box_err < 1e-2 and score_err < 1e-2 and label_ok  # measured 6.4e-3 / 2.3e-3 / 2.3e-3


@pytest.mark.gpu
def test_base_patch32_topology_matches_transformers():
    """google/owlvit-base-patch32 shapes (768^2 input, 577 tokens, 12+12 layers), random init, two 320x576 frames."""
    cfg = OwlViTConfig()
    frames = random_frames(2, 320, 576, seed=4)
    rel, box_err, score_err, label_ok = _compare(cfg, frames, token_ids(cfg, 2, seed=5), seed=6)
    assert rel < 2e-2 and box_err < 1e-2 and score_err < 2e-2 and label_ok  # measured 6.8e-3 / 3.6e-3 / 5.8e-3


@pytest.mark.gpu
def test_score_video_with_the_hip_detector():
    """The whole scoring chain on a 24-frame video: 6 evaluated frames -> detector -> NMS -> layout -> predicate."""
    ids_of = {}

    def tokenize(texts):
        for t in texts:
            ids_of.setdefault(t, len(ids_of))
        return torch.stack([token_ids(TINY, 1, seed=100 + ids_of[t])[0] for t in texts])

    det = HipOwlViTDetector(TINY, synthetic_owlvit_state_dict(TINY, seed=7), device="cuda", tokenize=tokenize)
    video = random_frames(24, 64, 96, seed=8)
    pairs = get_prompts("lvd", return_predicates=True)
    for idx in (0, 120, 250, 330, 470):
        prompt, pred = pairs[idx]
        kind, ok = score_video(prompt, pred, video, det, score_threshold=0.3, nms_threshold=0.5)
        assert kind == pred.type and isinstance(ok, bool)
    per_frame = det(video[:2], pairs[120][1].texts)
    boxes, scores, labels = per_frame[0]
    assert boxes.shape == (49, 4) and scores.shape == (49,) and set(np.unique(labels)) <= {0, 1} and np.isfinite(boxes).all()


@pytest.mark.gpu
def test_eval_owl_vit_cli(tmp_path):
    """scripts/eval_owl_vit.py on a run directory laid out by generate.py ({run}/{prompt index}/video_0.joblib): full-size
    detector topology with synthetic weights, two prompts, eval.json written in the reference's format."""
    import importlib.util
    import json
    import os
    import joblib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("eval_owl_vit_cli", os.path.join(root, "scripts", "eval_owl_vit.py"))
    cli = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cli)
    for ind in (0, 100):
        os.makedirs(tmp_path / str(ind))
        joblib.dump(random_frames(24, 320, 576, seed=ind), tmp_path / str(ind) / "video_0.joblib")
    os.makedirs(tmp_path / "1")
    joblib.dump(random_frames(24, 64, 64, seed=1), tmp_path / "1" / "video_0.joblib")
    joblib.dump(random_frames(24, 64, 64, seed=2), tmp_path / "1" / "video_1.joblib")  # ambiguous: skipped like the reference
    board = cli.main(["--run_base_path", str(tmp_path), "--synthetic-weights", "--save-eval", "--detection_score_threshold", "0.3"])
    assert board.total == {"numeracy": 1, "attribution": 1}
    saved = json.load(open(tmp_path / "eval.json"))
    assert saved["sample_counts_overall"] == 2 and set(saved["successes"]) == {"numeracy", "attribution"}
