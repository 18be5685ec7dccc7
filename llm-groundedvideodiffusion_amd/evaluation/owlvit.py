"""OWL-ViT open-vocabulary detector on the HIP kernels: the scorer of the benchmark.

Reference call sites: /root/reference/scripts/eval_owl_vit.py:70-96 — `processor(text=texts, images=image)`,
`OwlViTForObjectDetection(**inputs)`, `processor.post_process(outputs, target_sizes)` with `google/owlvit-base-patch32`
(:208-212).  The arithmetic is third-party (`transformers==4.36.2` modeling_owlvit / image_processing_owlvit), restated here
from its published definition and pinned by tests against `transformers.OwlViTForObjectDetection` itself (random-init
weights; the checkpoint is not on disk and there is no network):

  image  : PIL-exact bicubic resize to image_size^2, 1/255, CLIP mean/std, patchify      -> lvdhip_frames_to_patches
           patch embedding (Conv2d k = stride = patch, no bias) as one GEMM, + class token, + position embedding,
           pre-LN, N pre-LN blocks (quick-GELU), post-LN on every token                      -> shared CLIP block kernels
           tokens[1:] * token[0] (class-token merge), LayerNorm                              -> image_feats [P, C]
  text   : CLIP text tower (causal), pooled at EOS, text_projection, unit length           -> query_embeds [Q, D]
  heads  : class: dense0 -> unit length -> <.,query> -> (+shift)*(elu(scale)+1); box: MLP(gelu) + grid bias -> sigmoid
  post   : score = sigmoid(max_q logit), label = argmax_q, corners scaled to the frame size -> lvdhip_owl_detect_rows

`HipOwlViTDetector.__call__(frames, texts)` is the `detector` protocol of `evaluation.score_video`.  Tokenisation is host
text processing: pass `tokenize` (e.g. `transformers.CLIPTokenizer` of the checkpoint directory) or call with token ids.
"""
from dataclasses import dataclass, field

import torch

from .. import ops
from ..text_encoder import CLIPTextConfig, HipCLIPTextEncoder, load_clip_layers, run_clip_layers

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


@dataclass
class OwlViTConfig:
    """Defaults = google/owlvit-base-patch32."""
    image_size: int = 768
    patch_size: int = 32
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5
    projection_dim: int = 512
    text: CLIPTextConfig = field(default_factory=lambda: CLIPTextConfig(
        vocab_size=49408, hidden_size=512, intermediate_size=2048, num_hidden_layers=12, num_attention_heads=8,
        max_position_embeddings=16, hidden_act="quick_gelu", eos_token_id=2))  # eos rule: highest id (OwlViTTextTransformer)


def box_bias(grid):
    """compute_box_bias: logit of the token's grid corner for (cx, cy), logit of the cell size for (w, h)."""
    coords = torch.arange(1, grid + 1, dtype=torch.float32) / grid
    xx, yy = torch.meshgrid(coords, coords, indexing="xy")
    centre = torch.stack((xx, yy), -1).reshape(-1, 2).clip(0.0, 1.0)
    size = torch.full_like(centre, 1.0 / grid)
    logit = lambda t: torch.log(t + 1e-4) - torch.log1p(-t + 1e-4)
    return torch.cat([logit(centre), logit(size)], -1)


def _pad_rows(w, b, n):
    """GEMM needs N % 4 == 0: zero-pad an [n_out, K] head to n rows."""
    wp = torch.zeros((n, w.shape[1]), dtype=w.dtype)
    wp[: w.shape[0]] = w
    bp = torch.zeros((n,), dtype=b.dtype)
    bp[: b.shape[0]] = b
    return wp, bp


class HipOwlViTDetector:
    def __init__(self, cfg: OwlViTConfig, state_dict, device="cuda", tokenize=None):
        assert cfg.hidden_size // cfg.num_attention_heads == 64 and cfg.image_size % cfg.patch_size == 0
        assert cfg.text.hidden_size == cfg.projection_dim, "the class head maps image tokens to the text width"
        self.cfg, self.dev, self.tokenize = cfg, torch.device(device), tokenize
        sd = state_dict
        bf = lambda t: t.to(self.dev, torch.bfloat16).contiguous()
        f32 = lambda t: t.to(self.dev, torch.float32).contiguous()
        v = "owlvit.vision_model."
        C, G = cfg.hidden_size, cfg.image_size // cfg.patch_size
        self.grid, self.tokens = G, G * G
        self.patch_w = bf(sd[v + "embeddings.patch_embedding.weight"].reshape(C, -1))
        pos = sd[v + "embeddings.position_embedding.weight"].float()
        assert pos.shape[0] == self.tokens + 1
        self.cls_row = f32(sd[v + "embeddings.class_embedding"].float() + pos[0])
        self.pos_patches = pos[1:].to(self.dev, torch.bfloat16).contiguous()  # added by the patch GEMM's residual epilogue
        self.pre_ln = (f32(sd[v + "pre_layernorm.weight"]), f32(sd[v + "pre_layernorm.bias"]))
        self.post_ln = (f32(sd[v + "post_layernorm.weight"]), f32(sd[v + "post_layernorm.bias"]))
        self.layers = load_clip_layers(sd, v + "encoder.layers.", cfg.num_hidden_layers, self.dev)
        self.merge_ln = (f32(sd["layer_norm.weight"]), f32(sd["layer_norm.bias"]))
        self.class_dense = (bf(sd["class_head.dense0.weight"]), f32(sd["class_head.dense0.bias"]))
        w, b = _pad_rows(torch.cat([sd["class_head.logit_shift.weight"], sd["class_head.logit_scale.weight"]], 0),
                         torch.cat([sd["class_head.logit_shift.bias"], sd["class_head.logit_scale.bias"]], 0), 4)
        self.shift_scale = (bf(w), f32(b))
        self.box_mlp = [(bf(sd[f"box_head.dense{i}.weight"]), f32(sd[f"box_head.dense{i}.bias"])) for i in range(3)]
        self.box_bias = f32(box_bias(G))
        t = "owlvit.text_model."
        self.text = HipCLIPTextEncoder(cfg.text, {k[len(t):]: val for k, val in sd.items() if k.startswith(t)}, device=self.dev)
        self.text_proj = bf(sd["owlvit.text_projection.weight"])
        self._query_cache = {}

    # ---- text queries -----------------------------------------------------------------------------------------------------
    def embed_queries(self, input_ids):
        """ids (Q, L) -> (unit-length query embeddings fp32 [Q, D] as the class head consumes them, query_mask int32 [Q])."""
        ids = torch.as_tensor(input_ids).to(self.dev)
        pooled = self.text(ids).pooler_output.to(torch.bfloat16).contiguous()
        q = ops.gemm(pooled, self.text_proj, out_fp32=True)
        q = q / torch.linalg.norm(q, dim=-1, keepdim=True)           # OwlViTModel.forward returns text_embeds normalised ...
        q = q / (torch.linalg.norm(q, dim=-1, keepdim=True) + 1e-6)  # ... and the class head normalises once more with an eps
        return q.contiguous(), (ids[:, 0] > 0).to(torch.int32).contiguous()

    def _queries_for(self, texts):
        key = tuple(texts)
        if key not in self._query_cache:
            if self.tokenize is None:
                raise RuntimeError("HipOwlViTDetector needs `tokenize` (texts -> ids padded to 16) to accept strings; pass token ids instead")
            self._query_cache[key] = self.embed_queries(self.tokenize(list(texts)))
        return self._query_cache[key]

    # ---- image tower ------------------------------------------------------------------------------------------------------
    def image_features(self, frames):
        """uint8 (B,H,W,3) -> image_feats bf16 [B*P, C] (one row per patch token, class token merged in)."""
        cfg = self.cfg
        frames = torch.as_tensor(frames).to(self.dev).contiguous()
        B, C, P = frames.shape[0], cfg.hidden_size, self.tokens
        patches = ops.frames_to_patches(frames, cfg.image_size, cfg.patch_size, CLIP_MEAN, CLIP_STD)
        x = torch.empty((B, P + 1, C), dtype=torch.bfloat16, device=self.dev)
        emb = ops.gemm(patches, self.patch_w, res=self.pos_patches.repeat(B, 1))
        x[:, 1:] = emb.reshape(B, P, C)
        x[:, 0] = self.cls_row.to(torch.bfloat16)
        x = ops.layernorm(x.reshape(B * (P + 1), C), *self.pre_ln, eps=cfg.layer_norm_eps)
        x = run_clip_layers(x, self.layers, samples=B, seq=P + 1, heads=cfg.num_attention_heads, causal=False, act=cfg.hidden_act,
                            eps=cfg.layer_norm_eps)
        x = ops.layernorm(x, *self.post_ln, eps=cfg.layer_norm_eps).reshape(B, P + 1, C)
        merged = (x[:, 1:].float() * x[:, :1].float()).to(torch.bfloat16).reshape(B * P, C).contiguous()
        return ops.layernorm(merged, *self.merge_ln, eps=cfg.layer_norm_eps)

    # ---- detection ----------------------------------------------------------------------------------------------------------
    def detect(self, frames, queries, query_mask=None, target_size=None):
        """-> dict(logits [B,P,Q], scores [B,P], labels [B,P], boxes [B,P,4] xyxy in pixels of `target_size` (h, w), default the
        frames' own size) — `outputs.logits` and the three lists of `processor.post_process`."""
        frames = torch.as_tensor(frames)
        B, H, W, _ = frames.shape
        h, w = target_size if target_size is not None else (H, W)
        feats = self.image_features(frames)
        cls = ops.gemm(feats, self.class_dense[0], bias=self.class_dense[1], out_fp32=True)
        shsc = ops.gemm(feats, self.shift_scale[0], bias=self.shift_scale[1], out_fp32=True)
        t = ops.gelu(ops.gemm(feats, self.box_mlp[0][0], bias=self.box_mlp[0][1]), "gelu")
        t = ops.gelu(ops.gemm(t, self.box_mlp[1][0], bias=self.box_mlp[1][1]), "gelu")
        raw = ops.gemm(t, self.box_mlp[2][0], bias=self.box_mlp[2][1], out_fp32=True)
        logits, scores, labels, boxes = ops.owl_detect_rows(cls, queries, shsc, raw, self.box_bias, self.tokens, w, h, query_mask=query_mask)
        P = self.tokens
        return dict(logits=logits.reshape(B, P, -1), scores=scores.reshape(B, P), labels=labels.reshape(B, P), boxes=boxes.reshape(B, P, 4))

    def __call__(self, frames, texts):
        """`score_video` protocol: per frame (boxes, scores, labels) over all image tokens."""
        if len(texts) and isinstance(texts[0], str):
            queries, mask = self._queries_for(texts)
        else:
            queries, mask = self.embed_queries(texts)
        out = self.detect(frames, queries, mask)
        boxes, scores, labels = out["boxes"].cpu(), out["scores"].cpu(), out["labels"].cpu()
        return [(boxes[i].numpy(), scores[i].numpy(), labels[i].numpy()) for i in range(boxes.shape[0])]


def synthetic_owlvit_state_dict(cfg: OwlViTConfig, seed=0):
    """Random weights with the checkpoint's names and shapes (dry runs / benchmarks without the hub)."""
    g = torch.Generator().manual_seed(seed)
    n = lambda *s, std=0.02: torch.randn(*s, generator=g) * std
    C, I, P = cfg.hidden_size, cfg.intermediate_size, (cfg.image_size // cfg.patch_size) ** 2
    t = cfg.text
    sd = {"owlvit.vision_model.embeddings.class_embedding": n(C),
          "owlvit.vision_model.embeddings.patch_embedding.weight": n(C, 3, cfg.patch_size, cfg.patch_size),
          "owlvit.vision_model.embeddings.position_embedding.weight": n(P + 1, C),
          "owlvit.text_model.embeddings.token_embedding.weight": n(t.vocab_size, t.hidden_size),
          "owlvit.text_model.embeddings.position_embedding.weight": n(t.max_position_embeddings, t.hidden_size),
          "owlvit.text_projection.weight": n(cfg.projection_dim, t.hidden_size, std=t.hidden_size ** -0.5),
          "owlvit.visual_projection.weight": n(cfg.projection_dim, C, std=C ** -0.5),
          "class_head.dense0.weight": n(t.hidden_size, C, std=C ** -0.5), "class_head.dense0.bias": n(t.hidden_size),
          "class_head.logit_shift.weight": n(1, C), "class_head.logit_shift.bias": n(1),
          "class_head.logit_scale.weight": n(1, C), "class_head.logit_scale.bias": n(1)}
    for i, (o, k) in enumerate([(C, C), (C, C), (4, C)]):
        sd[f"box_head.dense{i}.weight"], sd[f"box_head.dense{i}.bias"] = n(o, k, std=k ** -0.5), n(o)
    ln = lambda name, c: sd.update({name + ".weight": 1 + n(c), name + ".bias": n(c)})
    for name in ("owlvit.vision_model.pre_layernorm", "owlvit.vision_model.post_layernorm", "layer_norm"):
        ln(name, C)
    ln("owlvit.text_model.final_layer_norm", t.hidden_size)
    for prefix, c, i, layers in (("owlvit.vision_model.encoder.layers.", C, I, cfg.num_hidden_layers),
                                 ("owlvit.text_model.encoder.layers.", t.hidden_size, t.intermediate_size, t.num_hidden_layers)):
        for l in range(layers):
            p = f"{prefix}{l}."
            for nm in ("q_proj", "k_proj", "v_proj", "out_proj"):
                sd[p + f"self_attn.{nm}.weight"], sd[p + f"self_attn.{nm}.bias"] = n(c, c, std=c ** -0.5), n(c)
            sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"] = n(i, c, std=c ** -0.5), n(i)
            sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"] = n(c, i, std=i ** -0.5), n(c)
            ln(p + "layer_norm1", c)
            ln(p + "layer_norm2", c)
    return sd
