"""Generate tests/golden/*.npz by importing the REAL reference (/root/reference) in this container.

Runs only where /root/reference exists (never on the GPU box, never from the product path).  The reference
needs diffusers / cv2 / skvideo / inflect / imageio / easydict / pyjson5, none of which is installed, so
infrastructure-only stand-ins are registered first: config/model mixins, logging, I/O
modules, and the six diffusers arithmetic classes that are NOT vendored in the reference (ResnetBlock2D,
TemporalConvLayer, Downsample2D, Upsample2D, Timesteps, TimestepEmbedding — restated from the public
diffusers 0.27.2 definitions; parity for those is UNPINNED).  Everything under /root/reference/models and
/root/reference/utils runs as the reference wrote it.

Usage:  python oracle/make_golden.py            (writes tests/golden/)
The fixtures are data only: seeded inputs and the reference's outputs.
"""
import functools
import inspect
import math
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def install_shim():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        if "." in name:
            parent, child = name.rsplit(".", 1)
            setattr(sys.modules[parent], child, m)
        return m

    for n in ["cv2", "skvideo", "skvideo.io", "inflect", "imageio", "easydict", "pyjson5"]:
        mod(n)

    class _Eng:
        def plural(self, x):
            return x + "s"

        def a(self, x):
            return ("an " if x[0] in "aeiou" else "a ") + x

        def number_to_words(self, n):
            return ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve",
                    "thirteen", "fourteen", "fifteen", "sixteen", "seventeen", "eighteen", "nineteen", "twenty"][n]

    sys.modules["inflect"].engine = lambda: _Eng()
    sys.modules["easydict"].EasyDict = dict
    mod("diffusers")
    for n in ["utils", "utils.torch_utils", "configuration_utils", "models", "models.modeling_utils", "models.embeddings",
              "models.resnet", "loaders", "image_processor", "schedulers", "pipelines", "pipelines.pipeline_utils",
              "pipelines.text_to_video_synthesis"]:
        mod("diffusers." + n)
    u = sys.modules["diffusers.utils"]

    class _Log:
        def get_logger(self, n):
            import logging
            return logging.getLogger(n)

    u.logging = _Log()
    u.deprecate = lambda *a, **k: None

    class BaseOutput(dict):
        def __post_init__(self):
            for k, v in self.__dict__.items():
                self[k] = v

    u.BaseOutput = BaseOutput
    u.is_accelerate_available = lambda: False
    u.is_accelerate_version = lambda *a: False
    u.replace_example_docstring = lambda s: (lambda f: f)
    tu = sys.modules["diffusers.utils.torch_utils"]
    tu.maybe_allow_in_graph = lambda c: c
    tu.randn_tensor = lambda shape, generator=None, device=None, dtype=None: torch.randn(shape, generator=generator, dtype=dtype)
    cu = sys.modules["diffusers.configuration_utils"]

    class ConfigMixin:
        pass

    def register_to_config(init):
        @functools.wraps(init)
        def inner(self, *a, **k):
            init(self, *a, **k)
            ba = inspect.signature(init).bind(self, *a, **k)
            ba.apply_defaults()
            self.config = types.SimpleNamespace(**{kk: vv for kk, vv in ba.arguments.items() if kk != "self"})
        return inner

    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config

    class ModelMixin(nn.Module):
        @property
        def dtype(self):
            return next(self.parameters()).dtype

    sys.modules["diffusers.models.modeling_utils"].ModelMixin = ModelMixin
    ld = sys.modules["diffusers.loaders"]
    for n in ["UNet2DConditionLoadersMixin", "LoraLoaderMixin", "TextualInversionLoaderMixin"]:
        setattr(ld, n, type(n, (), {}))

    # ---- un-vendored diffusers==0.27.2 arithmetic (restated; NOT pinned by the reference) ----
    e = sys.modules["diffusers.models.embeddings"]

    from oracle.diffusers_restated import Downsample2D, ResnetBlock2D, TemporalConvLayer, TimestepEmbedding, Timesteps, Upsample2D

    e.Timesteps, e.TimestepEmbedding = Timesteps, TimestepEmbedding
    for n in ["CombinedTimestepLabelEmbeddings", "ImagePositionalEmbeddings", "PatchEmbed"]:
        setattr(e, n, type(n, (nn.Module,), {}))
    r = sys.modules["diffusers.models.resnet"]

    r.ResnetBlock2D, r.TemporalConvLayer, r.Downsample2D, r.Upsample2D = ResnetBlock2D, TemporalConvLayer, Downsample2D, Upsample2D
    # CPU-only container: neutralise the reference's hard-coded .cuda() / device="cuda"
    torch.Tensor.cuda = lambda self, *a, **k: self
    _z = torch.zeros
    torch.zeros = lambda *a, **k: _z(*a, **{kk: vv for kk, vv in k.items() if kk != "device"})
    sys.path.insert(0, REF)
    return _z


def randn(*shape, seed):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed))


def build_reference_unet(cfg, sd):
    from models.unet_3d_condition import UNet3DConditionModel
    m = UNet3DConditionModel(block_out_channels=cfg.block_out_channels, layers_per_block=cfg.layers_per_block,
                             cross_attention_dim=cfg.cross_attention_dim, attention_head_dim=cfg.attention_head_dim,
                             attention_type=cfg.attention_type)
    ref_sd = m.state_dict()
    assert list(ref_sd.keys()) == list(sd.keys()), "state_dict key order/name mismatch: %s" % [(a, b) for a, b in zip(ref_sd.keys(), sd.keys()) if a != b][:3]
    for k in ref_sd:
        assert tuple(ref_sd[k].shape) == tuple(sd[k].shape), (k, ref_sd[k].shape, sd[k].shape)
    m.load_state_dict(sd)
    return m.eval()


def main():
    assert os.path.isdir(REF), "reference not present: goldens can only be generated in the build container"
    os.makedirs(OUT, exist_ok=True)
    install_shim()
    os.chdir(REF)
    sys.path.insert(0, ROOT)
    import lvd_amd  # noqa: F401
    from lvd_amd.weights import TINY, UNetConfig, synthetic_state_dict, unet_param_shapes

    torch.manual_seed(0)
    # ------------------------------------------------------------------ (a) tiny UNet forward + saved maps
    cfg = UNetConfig(**TINY)
    sd = synthetic_state_dict(cfg, seed=0)
    unet = build_reference_unet(cfg, sd)
    B, Fr, hh, ww = 2, 4, 16, 16
    sample = randn(B, 4, Fr, hh, ww, seed=11)
    ehs = randn(B, 77, cfg.cross_attention_dim, seed=12)
    keys = [("down", 1, 0, 0), ("down", 2, 0, 0), ("mid", 0, 0, 0), ("up", 1, 0, 0), ("up", 2, 1, 0)]
    saved = {}
    with torch.no_grad():
        out = unet(sample, 500, encoder_hidden_states=ehs, cross_attention_kwargs={"save_attn_to_dict": saved, "save_keys": keys}, return_dict=False)[0]
        out_fast = unet(sample, 500, encoder_hidden_states=ehs, cross_attention_kwargs={"save_attn_to_dict": {}, "save_keys": []}, return_dict=False)[0]
    assert torch.allclose(out, out_fast, atol=2e-5), "reference slow/fast attention paths disagree"
    np.savez_compressed(os.path.join(OUT, "unet_tiny.npz"), sample=sample.numpy(), ehs=ehs.numpy(), timestep=np.array(500), out=out.numpy(),
                        **{"attn_" + "_".join(map(str, k)): v.numpy().astype(np.float32) for k, v in saved.items()},
                        param_names=np.array(list(sd.keys())), param_numel=np.array([v.numel() for v in sd.values()]))
    print("unet_tiny: out", tuple(out.shape), "abs mean", out.abs().mean().item())

    # ------------------------------------------------------------------ (b) gated (GLIGEN) tiny UNet
    cfgg = UNetConfig(attention_type="gated", **TINY)
    sdg = synthetic_state_dict(cfgg, seed=1)
    unetg = build_reference_unet(cfgg, sdg)
    N = 30
    boxes = torch.zeros(B * Fr, N, 4)
    masks = torch.zeros(B * Fr, N)
    pos = torch.zeros(B * Fr, N, cfgg.cross_attention_dim)
    rb = torch.rand(B * Fr, 2, 4, generator=torch.Generator().manual_seed(5))
    boxes[:, :2] = torch.stack([rb[..., 0] * 0.5, rb[..., 1] * 0.5, 0.5 + rb[..., 2] * 0.5, 0.5 + rb[..., 3] * 0.5], -1)
    masks[: B * Fr // 2, :2] = 1  # cond half has 2 objects, uncond half masked out (as the pipeline does)
    pos[:, :2] = randn(B * Fr, 2, cfgg.cross_attention_dim, seed=6)
    with torch.no_grad():
        outg = unetg(sample, 321, encoder_hidden_states=ehs, cross_attention_kwargs={"save_attn_to_dict": {}, "save_keys": [],
                     "gligen": {"boxes": boxes, "positive_embeddings": pos, "masks": masks}}, return_dict=False)[0]
    np.savez_compressed(os.path.join(OUT, "unet_tiny_gated.npz"), sample=sample.numpy(), ehs=ehs.numpy(), timestep=np.array(321),
                        boxes=boxes.numpy(), masks=masks.numpy(), positive_embeddings=pos.numpy(), out=outg.numpy())
    print("unet_tiny_gated: abs mean", outg.abs().mean().item())

    # ------------------------------------------------------------------ (c) loss on hand-made maps (+ gradient)
    from utils import guidance
    gen = torch.Generator().manual_seed(21)
    n_f, heads, H, W = 4, 3, 8, 12
    maps = {}
    for k in [("down", 1, 0, 0), ("up", 1, 1, 0)]:
        maps[k] = torch.rand(1, n_f, heads, H * W, 10, generator=gen).softmax(-1).requires_grad_(True)
    bboxes = [[[0.1, 0.2, 0.55, 0.8], [0.15, 0.2, 0.6, 0.8], [0.2, 0.2, 0.65, 0.8], [0.25, 0.2, 0.7, 0.8]],
              [[0.5, 0.5, 0.9, 0.95], [0.0, 0.0, 0.0, 0.0], [0.4, 0.45, 0.8, 0.9], [0.35, 0.4, 0.75, 0.85]]]
    object_positions = [[2, 3], [6]]
    cases = {"topk": dict(fg_top_p=0.75, bg_top_p=0.75, fg_weight=1.0, bg_weight=4.0, com_loss_scale=0.0),
             "com": dict(fg_top_p=0.25, bg_top_p=0.25, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03)}
    store = {"bboxes": np.array(bboxes), "maps_" + "0": maps[("down", 1, 0, 0)].detach().numpy(), "maps_1": maps[("up", 1, 1, 0)].detach().numpy()}
    for name, kw in cases.items():
        loss = guidance.compute_ca_lossv3(saved_attn=maps, bboxes=bboxes, object_positions=object_positions,
                                          guidance_attn_keys=list(maps.keys()), base_attn_dim=(H, W), **kw)
        grads = torch.autograd.grad(loss, list(maps.values()))
        store[f"loss_{name}"] = np.array(loss.item())
        store[f"grad0_{name}"] = grads[0].numpy()
        store[f"grad1_{name}"] = grads[1].numpy()
        print("loss", name, loss.item())
    # optional terms the reference's command line can switch on (generate.py:78-106, generation/lvd.py:85-106): deprecated ratio
    # energy, attention sync, BoxDiff corner constraint.  Boxes without an empty frame: with an empty NEXT-frame box the reference's
    # attention-sync crop is empty and its mean is NaN.
    bboxes2 = [[[0.1, 0.2, 0.55, 0.8], [0.15, 0.2, 0.6, 0.8], [0.2, 0.2, 0.65, 0.8], [0.25, 0.2, 0.7, 0.8]],
               [[0.5, 0.5, 0.9, 0.95], [0.45, 0.5, 0.85, 0.95], [0.4, 0.45, 0.8, 0.9], [0.0, 0.3, 0.3, 1.0]]]
    store["bboxes2"] = np.array(bboxes2)
    import warnings
    cases2 = {"ratio": dict(use_ratio_based_loss=True, com_loss_scale=0.02),
              "sync": dict(fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, attn_sync_weight=3.0),
              "boxdiff": dict(fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, boxdiff_loss_scale=0.7, boxdiff_normed=True),
              "boxdiff_sum": dict(fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, boxdiff_loss_scale=0.05, boxdiff_normed=False, boxdiff_L=2),
              "all": dict(fg_top_p=0.3, bg_top_p=0.6, fg_weight=1.5, bg_weight=2.5, attn_sync_weight=1.0, boxdiff_loss_scale=0.4, com_loss_scale=0.03),
              # CE / NLL form of the top-k energy (utils/guidance.py:363-399), alone and with the centre-of-mass term
              "ce": dict(use_max_based_loss=False, use_ce_based_loss=True, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0),
              "ce_com": dict(use_max_based_loss=False, use_ce_based_loss=True, fg_top_p=0.25, bg_top_p=0.4, fg_weight=1.5, bg_weight=0.5, com_loss_scale=0.03),
              # map-level options (utils/guidance.py:209-226): 3x3 Gaussian over the (position, token) plane; second softmax over tokens 1 .. num_tokens-2
              "smooth": dict(smooth_attn=True, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0),
              "renorm": dict(attn_renorm=True, num_tokens=9, renorm_scale=2.0, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0, bg_weight=2.0, com_loss_scale=0.03),
              "smooth_renorm": dict(smooth_attn=True, attn_renorm=True, num_tokens=8, renorm_scale=3.0, fg_top_p=0.3, bg_top_p=0.4, fg_weight=1.5, bg_weight=1.0,
                                    boxdiff_loss_scale=0.4)}
    for name, kw in cases2.items():
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            loss = guidance.compute_ca_lossv3(saved_attn=maps, bboxes=bboxes2, object_positions=object_positions,
                                              guidance_attn_keys=list(maps.keys()), base_attn_dim=(H, W), **kw)
        grads = torch.autograd.grad(loss, list(maps.values()))
        store[f"loss_{name}"] = np.array(loss.item())
        store[f"grad0_{name}"] = grads[0].numpy()
        store[f"grad1_{name}"] = grads[1].numpy()
        print("loss", name, loss.item())
    np.savez_compressed(os.path.join(OUT, "guidance_loss.npz"), **store)

    # ------------------------------------------------------------------ (d) latent_backward_guidance on the tiny UNet
    from models.pipelines import latent_backward_guidance

    class Sched:
        def __init__(self):
            betas = torch.linspace(0.00085**0.5, 0.012**0.5, 1000) ** 2
            self.alphas_cumprod = torch.cumprod(1 - betas, 0)

        def scale_model_input(self, x, t):
            return x

    lat = randn(1, 4, Fr, hh, ww, seed=31)
    cond = ehs[1:2]
    gkeys = [("down", 1, 0, 0), ("down", 2, 0, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 2, 1, 0)]
    gb = [[[0.1, 0.2, 0.6, 0.8], [0.2, 0.2, 0.7, 0.8], [0.3, 0.2, 0.8, 0.8], [0.4, 0.2, 0.9, 0.8]]]
    gpos = [[2]]
    import warnings
    warnings.simplefilter("ignore")
    hp = dict(loss_scale=5.0, loss_threshold=0.01, max_iter=2, max_index_step=10, fg_top_p=0.5, bg_top_p=0.5, fg_weight=1.0,
              bg_weight=2.0, com_loss_scale=0.03, base_attn_dim=(hh, ww), upsample_scale=1, upsample_mode="bilinear",
              use_ratio_based_loss=False, exclude_bg_heads=False, attn_sync_weight=0.0, boxdiff_loss_scale=0.0, boxdiff_normed=True)
    new_lat, loss = latent_backward_guidance(Sched(), unet, cond, index=0, bboxes=gb, object_positions=gpos, t=torch.tensor(801),
                                             latents=lat.clone(), loss=torch.tensor(10000.0), guidance_attn_keys=gkeys, verbose=False, **hp)
    # single-iteration variant for gradient-level comparison
    one_lat, one_loss = latent_backward_guidance(Sched(), unet, cond, index=0, bboxes=gb, object_positions=gpos, t=torch.tensor(801),
                                                 latents=lat.clone(), loss=torch.tensor(10000.0), guidance_attn_keys=gkeys, verbose=False,
                                                 **{**hp, "max_iter": 1})
    np.savez_compressed(os.path.join(OUT, "guidance_step.npz"), latents_in=lat.numpy(), cond=cond.numpy(), t=np.array(801),
                        latents_out=new_lat.detach().numpy(), loss=np.array(float(loss)), latents_out_1=one_lat.detach().numpy(),
                        loss_1=np.array(float(one_loss)), bboxes=np.array(gb), keys=np.array(["_".join(map(str, k)) for k in gkeys]))
    print("guidance_step: loss after 2 iters", float(loss), " 1 iter", float(one_loss), " |dlat|", (new_lat - lat).abs().mean().item())


if __name__ == "__main__":
    main()
