"""`UNet3DConditionModel` with the reference's constructor, config, state_dict names and forward contract
(/root/reference/models/unet_3d_condition.py:195-859), executed by the HIP engine.

Differences that are deliberate and loud:
  * no autograd: the guidance gradient comes from `lvd_amd.guidance.hip_latent_backward_guidance` (plug it into the
    pipeline's `custom_latent_backward_guidance`); calling the reference's autograd-based loop on this model raises.
  * `from_pretrained` reads a LOCAL Hugging Face snapshot directory (config.json + safetensors / bin); hub ids are not downloaded.
"""
import types
from collections import OrderedDict
from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple, Union

import torch
import torch.nn as nn

from ..engine import HipUNet3D, TextCache
from ..weights import UNetConfig, unet_param_shapes
from .attention import GatedSelfAttentionDense
from .attention_processor import HipAttnProcessor


@dataclass
class UNet3DConditionOutput:
    sample: torch.FloatTensor


class UNet3DConditionModel(nn.Module):
    _supports_gradient_checkpointing = False

    def __init__(self, sample_size: Optional[int] = None, in_channels: int = 4, out_channels: int = 4,
                 down_block_types: Tuple[str] = ("CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "DownBlock3D"),
                 up_block_types: Tuple[str] = ("UpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D"),
                 block_out_channels: Tuple[int] = (320, 640, 1280, 1280), layers_per_block: int = 2, downsample_padding: int = 1,
                 mid_block_scale_factor: float = 1, act_fn: str = "silu", norm_num_groups: Optional[int] = 32, norm_eps: float = 1e-5,
                 cross_attention_dim: int = 1024, attention_head_dim: Union[int, Tuple[int]] = 64,
                 num_attention_heads: Optional[Union[int, Tuple[int]]] = None, attention_type: str = "default"):
        super().__init__()
        if num_attention_heads is not None:
            raise NotImplementedError("At the moment it is not possible to define the number of attention heads via `num_attention_heads` "
                                      "because of a naming issue (same restriction as the reference, unet_3d_condition.py:262-265).")
        if len(down_block_types) != len(up_block_types):
            raise ValueError(f"Must provide the same number of `down_block_types` as `up_block_types`. `down_block_types`: {down_block_types}. `up_block_types`: {up_block_types}.")
        if len(block_out_channels) != len(down_block_types):
            raise ValueError(f"Must provide the same number of `block_out_channels` as `down_block_types`. `block_out_channels`: {block_out_channels}. `down_block_types`: {down_block_types}.")
        if act_fn != "silu" or downsample_padding != 1 or mid_block_scale_factor != 1 or attention_head_dim != 64:
            raise NotImplementedError("the HIP kernels are specialised for the zeroscope/modelscope family: act_fn='silu', downsample_padding=1, "
                                      "mid_block_scale_factor=1, attention_head_dim=64")
        self.sample_size = sample_size
        self.config = types.SimpleNamespace(sample_size=sample_size, in_channels=in_channels, out_channels=out_channels,
                                            down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types),
                                            block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                            downsample_padding=downsample_padding, mid_block_scale_factor=mid_block_scale_factor, act_fn=act_fn,
                                            norm_num_groups=norm_num_groups, norm_eps=norm_eps, cross_attention_dim=cross_attention_dim,
                                            attention_head_dim=attention_head_dim, num_attention_heads=None, attention_type=attention_type)
        self.cfg = UNetConfig(in_channels=in_channels, out_channels=out_channels, block_out_channels=tuple(block_out_channels),
                              layers_per_block=layers_per_block, cross_attention_dim=cross_attention_dim, attention_head_dim=attention_head_dim,
                              norm_num_groups=norm_num_groups, norm_eps=norm_eps, attention_type=attention_type,
                              down_block_types=tuple(down_block_types), up_block_types=tuple(up_block_types), sample_size=sample_size)
        self._shapes = unet_param_shapes(self.cfg)
        self._state = None       # reference-named fp32/bf16 tensors until the engine is built
        self.engine: Optional[HipUNet3D] = None
        self._device = torch.device("cuda")
        self._processor = HipAttnProcessor()
        self._text_cache = (None, None)
        # one handle per fuser so `for m in unet.modules(): if type(m).__name__ == "GatedSelfAttentionDense"` works
        self.fusers = nn.ModuleList([GatedSelfAttentionDense(n[: -len(".alpha_attn")]) for n in self._shapes if n.endswith(".alpha_attn")])

    # ------------------------------------------------------------------ weights
    @classmethod
    def from_state_dict(cls, state_dict, device="cuda", **config):
        m = cls(**config)
        m.load_state_dict(state_dict)
        return m.to(device)

    _CONFIG_KEYS = ("sample_size", "in_channels", "out_channels", "down_block_types", "up_block_types", "block_out_channels", "layers_per_block",
                    "downsample_padding", "mid_block_scale_factor", "act_fn", "norm_num_groups", "norm_eps", "cross_attention_dim",
                    "attention_head_dim", "num_attention_heads", "attention_type")

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, device="cuda", **overrides):
        """Local Hugging Face snapshot directory (generation/lvd.py:39-44 `UNet3DConditionModel.from_pretrained(key, subfolder="unet")`):
        `<dir>[/<subfolder>]/config.json` + `diffusion_pytorch_model.safetensors` (or `.fp16.safetensors`, or `.bin`).  Hub ids are not
        resolved (no network): pass the directory of an already downloaded snapshot, e.g. .../cerspense--zeroscope_v2_576w/snapshots/<rev>."""
        import json
        import os
        root = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
        cfg_path = os.path.join(root, "config.json")
        if not os.path.isfile(cfg_path):
            raise RuntimeError(f"{cfg_path} not found: from_pretrained needs a LOCAL snapshot directory (hub ids cannot be downloaded here); "
                               "or load a state_dict and use UNet3DConditionModel.from_state_dict(state_dict, **config)")
        with open(cfg_path) as f:
            raw = json.load(f)
        config = {k: (tuple(v) if isinstance(v, list) else v) for k, v in raw.items() if k in cls._CONFIG_KEYS}
        config.update(overrides)
        sd = None
        for name in ("diffusion_pytorch_model.safetensors", "diffusion_pytorch_model.fp16.safetensors"):
            if os.path.isfile(os.path.join(root, name)):
                from safetensors.torch import load_file
                sd = load_file(os.path.join(root, name))
                break
        if sd is None:
            for name in ("diffusion_pytorch_model.bin", "diffusion_pytorch_model.fp16.bin"):
                if os.path.isfile(os.path.join(root, name)):
                    sd = torch.load(os.path.join(root, name), map_location="cpu", weights_only=True)
                    break
        if sd is None:
            raise RuntimeError(f"no diffusion_pytorch_model.safetensors / .bin under {root}")
        return cls.from_state_dict(sd, device=device, **config)

    def load_state_dict(self, state_dict, strict: bool = True):
        missing = [k for k in self._shapes if k not in state_dict]
        unexpected = [k for k in state_dict if k not in self._shapes]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for UNet3DConditionModel: missing keys {missing[:5]}… unexpected keys {unexpected[:5]}…")
        for k, shape in self._shapes.items():
            if k in state_dict and tuple(state_dict[k].shape) != tuple(shape):
                raise RuntimeError(f"size mismatch for {k}: copying a param with shape {tuple(state_dict[k].shape)}, expected {tuple(shape)}")
        self._state = OrderedDict((k, state_dict[k]) for k in self._shapes if k in state_dict)
        self.engine = None
        return types.SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def state_dict(self, *a, **k):
        if self._state is None:
            raise RuntimeError("no weights loaded")
        return OrderedDict(self._state)

    def to(self, *args, **kwargs):
        for a in list(args) + list(kwargs.values()):
            if isinstance(a, (str, torch.device)):
                self._device = torch.device(a)
        return self  # dtype requests are ignored: storage is bf16, accumulation fp32

    def eval(self):
        return self

    @property
    def dtype(self):
        return torch.bfloat16

    @property
    def device(self):
        return self._device

    def _ensure_engine(self):
        if self.engine is None:
            if self._state is None:
                raise RuntimeError("UNet3DConditionModel has no weights: call load_state_dict / from_state_dict first")
            if self._device.type != "cuda":
                raise RuntimeError("the HIP denoiser runs on a CUDA(ROCm) device only; there is no CPU fallback")
            self.engine = HipUNet3D(self.cfg, self._state, device=self._device)
        return self.engine

    # ------------------------------------------------------------------ processors (reference plug-in point)
    @property
    def attn_processors(self) -> Dict[str, Any]:
        return {n[: -len(".to_out.0.weight")] + ".processor": self._processor for n in self._shapes if n.endswith(".to_out.0.weight")}

    def set_attn_processor(self, processor):
        procs = processor.values() if isinstance(processor, dict) else [processor]
        for p in procs:
            if not isinstance(p, HipAttnProcessor):
                raise TypeError("this UNet executes attention inside fused HIP kernels; only HipAttnProcessor is accepted")
        self._processor = next(iter(procs))

    def set_default_attn_processor(self):
        self._processor = HipAttnProcessor()

    # ------------------------------------------------------------------ forward
    def encode_text(self, encoder_hidden_states) -> TextCache:
        """Project the prompt embeddings to K|V of every cross-attention layer once (cached on tensor identity)."""
        key, cache = self._text_cache
        if key is not encoder_hidden_states:
            cache = self._ensure_engine().encode_text(encoder_hidden_states)
            self._text_cache = (encoder_hidden_states, cache)
        return cache

    def forward(self, sample, timestep, encoder_hidden_states, class_labels=None, timestep_cond=None, attention_mask=None,
                cross_attention_kwargs: Optional[Dict[str, Any]] = None, down_block_additional_residuals=None,
                mid_block_additional_residual=None, return_dict: bool = True):
        if class_labels is not None or timestep_cond is not None or attention_mask is not None or \
                down_block_additional_residuals is not None or mid_block_additional_residual is not None:
            raise NotImplementedError("class_labels / timestep_cond / attention_mask / additional residuals are not on the hot path")
        if torch.is_grad_enabled() and torch.is_tensor(sample) and sample.requires_grad:
            raise RuntimeError("the HIP denoiser has no autograd graph: use lvd_amd.guidance.hip_latent_backward_guidance "
                               "(pipeline kwarg custom_latent_backward_guidance) instead of torch.autograd.grad")
        eng = self._ensure_engine()
        kw = cross_attention_kwargs if cross_attention_kwargs is not None else {}
        text = encoder_hidden_states if isinstance(encoder_hidden_states, TextCache) else self.encode_text(encoder_hidden_states)
        save_dict, save_keys = kw.get("save_attn_to_dict"), kw.get("save_keys")
        collect = None
        if save_dict is not None and save_keys:
            collect = {"keys": {tuple(k) for k in save_keys}, "q": {}}
        fuser_on = all(f.enabled for f in self.fusers) if len(self.fusers) else True
        out = eng.forward(sample, timestep, text=text, gligen=kw.get("gligen"), fuser_enabled=fuser_on, collect=collect)
        if collect is not None:
            # visualisation path only (generation/lvd.py:58-63 keeps save_keys = []): materialise the requested maps
            for key, (q, k, heads, g) in collect["q"].items():
                qf = q.float().reshape(g.B * g.F, g.HW, heads, 64).permute(0, 2, 1, 3)
                kf = k.float().reshape(text.B, text.ntext, heads, 64).permute(0, 2, 3, 1).repeat_interleave(g.F, 0)
                save_dict[key] = (qf @ kf * 0.125).softmax(-1)
        if not return_dict:
            return (out,)
        return UNet3DConditionOutput(sample=out)
