"""Shared runner of the generation methods (the five reference modules differ only in defaults and in which conditioning
they switch on).  The checkpoint and the components around the denoiser are handed over with `configure(...)` — nothing can be
downloaded here: `state_dict` = reference-named UNet weights, a LOCAL Hugging Face snapshot directory (the offline form of the
reference's `from_pretrained(key, subfolder="unet")`) or "synthetic"; `vae` / `text_encoder` = state_dicts run by the HIP VAE decoder /
CLIP text encoder (or callables); `tokenizer` = a callable (the CLIP vocabulary files are not in this image)."""
import os

import numpy as np
import torch

from .. import dsl, vis
from ..guidance import hip_latent_backward_guidance
from ..models.controllable_pipeline_text_to_video_synth import TextToVideoSDPipeline
from ..models.unet_3d_condition import UNet3DConditionModel
from ..sampler import DPMSolverPP2MSchedule
from ..weights import UNetConfig, synthetic_state_dict

BASE_MODELS = {  # generation/lvd.py:19-37
    "modelscope512": dict(base_attn_dim=(64, 64), H=512, W=512),
    "modelscope256": dict(base_attn_dim=(32, 32), H=256, W=256),
    "zeroscope": dict(base_attn_dim=(40, 72), H=320, W=576),
}
GUIDANCE_ATTN_KEYS = [("down", 1, 0, 0), ("down", 2, 0, 0), ("down", 2, 1, 0), ("up", 1, 0, 0), ("up", 1, 1, 0), ("up", 2, 2, 0)]  # lvd.py:66-73

_components = dict(state_dict=None, unet_config=None, tokenizer=None, text_encoder=None, vae=None, vae_config=None, text_encoder_config=None, device="cuda", img_dir="imgs")


def configure(**kw):
    """state_dict (reference-named UNet weights) / unet_config (ctor kwargs) / tokenizer / text_encoder / vae / device /
    img_dir.  `state_dict="synthetic"` draws seeded random weights of the requested topology (plumbing and benchmarks).
    `vae` is a callable latents -> frames, an `AutoencoderKL.state_dict()` (decoded on the HIP kernels by
    `lvd_amd.vae.HipVAEDecoder`, `vae_config` = VAEConfig kwargs) or "synthetic".  `text_encoder` is a callable
    input_ids -> (last_hidden_state, ...) or a `CLIPTextModel.state_dict()` (run by `lvd_amd.text_encoder.HipCLIPTextEncoder`,
    `text_encoder_config` = CLIPTextConfig kwargs)."""
    unknown = set(kw) - set(_components)
    if unknown:
        raise TypeError(f"unknown components {sorted(unknown)}")
    _components.update(kw)


class Method:
    def __init__(self, version, use_guidance, use_gligen):
        self.version, self.use_guidance, self.use_gligen = version, use_guidance, use_gligen
        self.pipe = None

    def init(self, base_model):
        if base_model not in BASE_MODELS:
            raise ValueError(f"Unknown base model: {base_model}")
        self.base = dict(BASE_MODELS[base_model])
        cfg_kw = dict(_components["unet_config"] or {})
        if self.use_gligen:
            cfg_kw.setdefault("attention_type", "gated")
        sd = _components["state_dict"]
        if sd is None:
            raise RuntimeError("no UNet weights: call lvd_amd.generation._common.configure(state_dict=...) first (hub downloads are unavailable)")
        if isinstance(sd, str) and sd == "synthetic":
            ucfg = UNetConfig(**{k: v for k, v in cfg_kw.items() if k in UNetConfig.__dataclass_fields__})
            sd = synthetic_state_dict(ucfg, seed=0, device=_components["device"])
        if isinstance(sd, str):  # a local Hugging Face snapshot directory, as generation/lvd.py:39-44 loads it (subfolder "unet")
            import os
            unet = UNet3DConditionModel.from_pretrained(sd, subfolder="unet" if os.path.isdir(os.path.join(sd, "unet")) else None,
                                                        device=_components["device"], **cfg_kw)
        else:
            unet = UNet3DConditionModel.from_state_dict(sd, device=_components["device"], **cfg_kw)
        vae = _components["vae"]
        if isinstance(vae, (dict, str)):
            from ..vae import HipVAEDecoder
            from ..weights import VAEConfig, synthetic_vae_state_dict
            vcfg = VAEConfig(**(_components["vae_config"] or {}))
            if isinstance(vae, str):
                if vae != "synthetic":
                    raise ValueError(f"vae={vae!r}: expected a callable, a state_dict or 'synthetic'")
                vae = synthetic_vae_state_dict(vcfg, seed=0, device=_components["device"])
            vae = HipVAEDecoder(vcfg, vae, device=_components["device"])
        text_encoder = _components["text_encoder"]
        if isinstance(text_encoder, dict):
            from ..text_encoder import CLIPTextConfig, HipCLIPTextEncoder
            text_encoder = HipCLIPTextEncoder(CLIPTextConfig(**(_components["text_encoder_config"] or {})), text_encoder, device=_components["device"])
        self.pipe = TextToVideoSDPipeline(unet=unet, scheduler=DPMSolverPP2MSchedule.from_ddim_config(), vae=vae,
                                          text_encoder=text_encoder, tokenizer=_components["tokenizer"]).to(_components["device"])
        self.pipe.guidance_models = None
        return self.base["H"], self.base["W"]

    def run(self, parsed_layout, seed, num_inference_steps=40, num_frames=16, repeat_ind=None, save_annotated_videos=False, loss_scale=5.0,
            loss_threshold=200.0, max_iter=5, max_index_step=10, fg_top_p=0.75, bg_top_p=0.75, fg_weight=1.0, bg_weight=4.0,
            attn_sync_weight=0.0, boxdiff_loss_scale=0.0, boxdiff_normed=True, com_loss_scale=0.0, use_ratio_based_loss=False,
            save_formats=("gif", "joblib"), gligen_scheduled_sampling_beta=None, prompt_embeds=None, negative_prompt_embeds=None,
            gligen_phrase_embeds=None, latents=None):
        """One video (the reference's `run`, generation/lvd.py:79-210): the V = 1 case of `run_many`."""
        job = dict(parsed_layout=parsed_layout, seed=seed, repeat_ind=repeat_ind, prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds,
                   gligen_phrase_embeds=gligen_phrase_embeds, latents=latents)
        return self.run_many([job], num_inference_steps=num_inference_steps, num_frames=num_frames, save_annotated_videos=save_annotated_videos,
                             loss_scale=loss_scale, loss_threshold=loss_threshold, max_iter=max_iter, max_index_step=max_index_step, fg_top_p=fg_top_p,
                             bg_top_p=bg_top_p, fg_weight=fg_weight, bg_weight=bg_weight, attn_sync_weight=attn_sync_weight,
                             boxdiff_loss_scale=boxdiff_loss_scale, boxdiff_normed=boxdiff_normed, com_loss_scale=com_loss_scale,
                             use_ratio_based_loss=use_ratio_based_loss, save_formats=save_formats,
                             gligen_scheduled_sampling_beta=gligen_scheduled_sampling_beta)[0]

    def run_many(self, jobs, num_inference_steps=40, num_frames=16, save_annotated_videos=False, loss_scale=5.0, loss_threshold=200.0, max_iter=5,
                 max_index_step=10, fg_top_p=0.75, bg_top_p=0.75, fg_weight=1.0, bg_weight=4.0, attn_sync_weight=0.0, boxdiff_loss_scale=0.0,
                 boxdiff_normed=True, com_loss_scale=0.0, use_ratio_based_loss=False, save_formats=("gif", "joblib"),
                 gligen_scheduled_sampling_beta=None):
        """V independent (layout, seed) samples through ONE denoising loop (pipeline.sample_many: per-sample guidance passes, one CFG forward
        of batch 2V per step) — generate.py --videos-per-gpu V.  `jobs`: dicts with parsed_layout, seed and optionally repeat_ind, img_dir
        (default: the configured one), prompt_embeds, negative_prompt_embeds, gligen_phrase_embeds, latents.  The seed rule, the file names and
        the skip-if-exists behaviour are those of `run`; returns one entry per job (frames, latents, or None for a skipped job)."""
        pipe = self.pipe
        if pipe is None:
            raise RuntimeError("call init(base_model) first")
        H, W = self.base["H"], self.base["W"]
        box_h, box_w = dsl.LAYOUT_SIZE
        want_latent = pipe.vae is None
        results, todo, samples = [None] * len(jobs), [], []
        for n, job in enumerate(jobs):
            cond = dsl.layout_to_condition(job["parsed_layout"], height=box_h, width=box_w, num_condition_frames=num_frames, tokenizer=pipe.tokenizer)
            if self.use_guidance and cond.object_positions is None:
                raise RuntimeError("attention guidance needs object token positions: inject a tokenizer (configure(tokenizer=...))")
            img_dir = job.get("img_dir") or _components["img_dir"]
            repeat_ind, seed = job.get("repeat_ind"), job["seed"]
            suffix = repeat_ind if repeat_ind is not None else f"seed{seed}"
            if os.path.exists(f"{img_dir}/video_{suffix}.gif"):
                print(f"Skipping {img_dir}/video_{suffix}.gif")
                continue
            smp = {}
            if self.use_guidance:
                smp["backward_guidance_kwargs"] = dict(
                    bboxes=cond.boxes, object_positions=cond.object_positions, loss_scale=loss_scale, loss_threshold=loss_threshold,
                    max_iter=max_iter, max_index_step=max_index_step, fg_top_p=fg_top_p, bg_top_p=bg_top_p, fg_weight=fg_weight,
                    bg_weight=bg_weight, use_ratio_based_loss=use_ratio_based_loss, guidance_attn_keys=GUIDANCE_ATTN_KEYS,
                    exclude_bg_heads=False, upsample_scale=1, upsample_mode="bilinear", base_attn_dim=self.base["base_attn_dim"],
                    attn_sync_weight=attn_sync_weight, boxdiff_loss_scale=boxdiff_loss_scale, boxdiff_normed=boxdiff_normed,
                    com_loss_scale=com_loss_scale, verbose=False)
            if self.use_gligen:
                absent = [0.0, 0.0, 0.0, 0.0]
                smp["gligen_boxes"] = [[b[i] for b in cond.boxes if b[i] != absent] for i in range(num_frames)]
                smp["gligen_phrases"] = [[p for p, b in zip(cond.phrases, cond.boxes) if b[i] != absent] for i in range(num_frames)]
                smp["gligen_phrase_embeds"] = job.get("gligen_phrase_embeds")
            pe = job.get("prompt_embeds")
            smp.update(prompt=cond.prompt if pe is None else None, negative_prompt=dsl.NEGATIVE_PROMPT if pe is None else None, prompt_embeds=pe,
                       negative_prompt_embeds=job.get("negative_prompt_embeds"), latents=job.get("latents"),
                       generator=torch.Generator(device="cpu").manual_seed(int(seed)))  # CPU generator: reproducible across devices (SURVEY §7 RNG note)
            todo.append((n, cond, img_dir, suffix))
            samples.append(smp)
        if not samples:
            return results
        kw = {}
        if self.use_guidance:
            kw.update(custom_latent_backward_guidance=hip_latent_backward_guidance, guidance_type="main")
        if self.use_gligen:
            kw["gligen_scheduled_sampling_beta"] = 1.0 if gligen_scheduled_sampling_beta is None else gligen_scheduled_sampling_beta
        outs = pipe.sample_many(samples, num_inference_steps=num_inference_steps, height=H, width=W, num_frames=num_frames,
                                cross_attention_kwargs={"save_attn_to_dict": {}, "save_keys": []}, output_type="latent" if want_latent else "np", **kw)
        for (n, cond, img_dir, suffix), out in zip(todo, outs):
            os.makedirs(img_dir, exist_ok=True)
            if want_latent:
                import joblib
                joblib.dump(out.float().cpu().numpy(), f"{img_dir}/latents_{suffix}.joblib", compress=("bz2", 3))
                results[n] = out
                continue
            frames = (np.asarray(out[0]) * 255.0).astype(np.uint8)  # uint8 (F,H,W,3), also for the gligen method
            if save_annotated_videos:  # the reference builds this path under the .gif name (generation/lvd.py:188-189); a sibling file here
                vis.save_frames(f"{img_dir}/video_{suffix}_with_box", vis.draw_boxes(frames, cond.boxes, cond.phrases), formats="gif")
            vis.save_frames(f"{img_dir}/video_{suffix}", frames, formats=list(save_formats))
            results[n] = frames
        return results
