"""`TextToVideoSDPipeline` with the reference's `__call__` contract
(/root/reference/models/controllable_pipeline_text_to_video_synth.py:541-979) driving the HIP denoiser.

In scope: timestep setup, latent preparation, GLIGEN tensor assembly (:736-814), fuser scheduled sampling
(:816-817,838-839), the denoising loop with backward guidance, CFG and the DPM-Solver++ update (:833-958).
CLIP text encoding and VAE decoding (§8f) run on the same kernels (`lvd_amd.text_encoder.HipCLIPTextEncoder`, `lvd_amd.vae.HipVAEDecoder`)
when their state_dicts are injected as `text_encoder=` / `vae=`; without them pass `prompt_embeds` / `negative_prompt_embeds` (and, for
GLIGEN, `gligen_phrase_embeds`) and use `output_type="latent"`.  The CLIP BPE tokenizer stays an injected callable (its vocabulary files
are not shipped in this image).
"""
import warnings
from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np
import torch

from .. import ops
from ..guidance import hip_latent_backward_guidance, hip_latent_backward_guidance_many
from ..sampler import DPMSolverPP2MSchedule


@dataclass
class TextToVideoSDPipelineOutput:
    frames: Any


class TextToVideoSDPipeline:
    def __init__(self, unet, scheduler=None, vae=None, text_encoder=None, tokenizer=None, vae_scale_factor=8):
        self.unet = unet
        self.scheduler = scheduler if scheduler is not None else DPMSolverPP2MSchedule.from_ddim_config()
        self.vae, self.text_encoder, self.tokenizer = vae, text_encoder, tokenizer
        self.vae_scale_factor = vae_scale_factor
        self.guidance_models = None
        self._device = torch.device("cuda")

    @classmethod
    def from_pretrained(cls, *a, unet=None, **k):
        if unet is None:
            raise RuntimeError("from_pretrained needs the HF hub; construct TextToVideoSDPipeline(unet=...) directly")
        return cls(unet=unet)

    def to(self, device):
        self._device = torch.device(device)
        self.unet.to(device)
        return self

    def enable_vae_slicing(self):
        pass

    def enable_fuser(self, enabled=True):
        for module in self.unet.modules():
            if type(module).__name__ == "GatedSelfAttentionDense":
                module.enabled = enabled

    def progress_bar(self, total):
        from contextlib import contextmanager

        @contextmanager
        def _bar():
            class _B:
                def update(self_inner, n=1):
                    pass
            yield _B()
        return _bar()

    # ------------------------------------------------------------------ input checks (controllable_pipeline…py:402-497)
    def check_inputs(self, prompt, height, width, callback_steps, gligen_phrases, gligen_boxes, negative_prompt, prompt_embeds,
                     negative_prompt_embeds, num_frames):
        if height % 8 != 0 or width % 8 != 0:
            raise ValueError(f"`height` and `width` have to be divisible by 8 but are {height} and {width}.")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps} of type {type(callback_steps)}.")
        if prompt is not None and prompt_embeds is not None:
            raise ValueError("Cannot forward both `prompt` and `prompt_embeds`. Please make sure to only forward one of the two.")
        if prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        if prompt is not None and not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        if prompt_embeds is not None and negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly")
        if gligen_boxes:
            if gligen_phrases is None or len(gligen_phrases) != len(gligen_boxes) or len(gligen_boxes) != num_frames:
                raise ValueError("length of `gligen_phrases` and `gligen_boxes` has to be same (one entry per frame)")

    def _encode_prompt(self, prompt, device, do_cfg, negative_prompt, prompt_embeds, negative_prompt_embeds):
        if prompt_embeds is None:
            if self.text_encoder is None or self.tokenizer is None:
                raise RuntimeError("text encoding is a 'next' row (SURVEY §8f): pass prompt_embeds/negative_prompt_embeds or inject "
                                   "tokenizer + text_encoder with the CLIP interfaces the reference uses")
            def enc(p):
                ids = self.tokenizer(p, padding="max_length", max_length=self.tokenizer.model_max_length, truncation=True, return_tensors="pt").input_ids
                return self.text_encoder(ids.to(device))[0]
            prompt_embeds = enc(prompt)
            if do_cfg and negative_prompt_embeds is None:
                negative_prompt_embeds = enc(negative_prompt if negative_prompt is not None else [""] * (1 if isinstance(prompt, str) else len(prompt)))
        if do_cfg:
            if negative_prompt_embeds is None:
                raise ValueError("classifier-free guidance needs negative_prompt_embeds")
            return torch.cat([negative_prompt_embeds, prompt_embeds]).to(device)
        return prompt_embeds.to(device)

    def prepare_latents(self, batch_size, num_channels, num_frames, height, width, device, generator, latents=None):
        shape = (batch_size, num_channels, num_frames, height // self.vae_scale_factor, width // self.vae_scale_factor)
        if isinstance(generator, list) and len(generator) != batch_size:
            raise ValueError(f"You have passed a list of generators of length {len(generator)}, but requested an effective batch size of {batch_size}.")
        if latents is None:
            gdev = generator.device if generator is not None else device
            latents = torch.randn(shape, generator=generator, device=gdev, dtype=torch.float32).to(device)
        else:
            latents = latents.to(device, torch.float32)
        return (latents * self.scheduler.init_noise_sigma).contiguous()

    def decode_latents(self, latents):
        if self.vae is None:
            raise RuntimeError("VAE decoding is a 'next' row (SURVEY §8f): use output_type='latent' or inject `vae`")
        return self.vae(latents)

    # ------------------------------------------------------------------ the call
    @torch.no_grad()
    def __call__(self, prompt: Union[str, List[str]] = None, height: Optional[int] = None, width: Optional[int] = None, num_frames: int = 16,
                 num_inference_steps: int = 50, guidance_scale: float = 9.0, gligen_scheduled_sampling_beta: float = 0.3,
                 gligen_phrases: List[List[str]] = None, gligen_boxes: List[List[List[float]]] = None,
                 negative_prompt: Optional[Union[str, List[str]]] = None, eta: float = 0.0, generator=None,
                 latents: Optional[torch.FloatTensor] = None, prompt_embeds: Optional[torch.FloatTensor] = None,
                 negative_prompt_embeds: Optional[torch.FloatTensor] = None, output_type: Optional[str] = "np", return_dict: bool = True,
                 callback: Optional[Callable] = None, guidance_callback: Optional[Callable] = None, callback_steps: int = 1,
                 cross_attention_kwargs: Optional[Dict[str, Any]] = None, backward_guidance_kwargs: Optional[Dict[str, Any]] = None,
                 aux_backward_guidance_kwargs: Optional[Dict[str, Any]] = None, guidance_type: Optional[str] = "main",
                 return_guidance_saved_attn: bool = False, custom_latent_backward_guidance: Callable = None,
                 backward_guidance_kwargs_custom: Optional[Dict[str, Any]] = None, both_attn_and_custom=False,
                 gligen_phrase_embeds: Optional[torch.Tensor] = None, lvd_gligen_scheduled_sampling_beta: Optional[float] = None,
                 lvd_gligen_phrases=None, lvd_gligen_boxes=None):
        # generation/lvd_gligen.py:126-128 passes lvd_gligen_* spellings: accept both
        if lvd_gligen_boxes is not None:
            gligen_boxes, gligen_phrases = lvd_gligen_boxes, lvd_gligen_phrases
        if lvd_gligen_scheduled_sampling_beta is not None:
            gligen_scheduled_sampling_beta = lvd_gligen_scheduled_sampling_beta
        sample = dict(prompt=prompt, negative_prompt=negative_prompt, generator=generator, latents=latents, prompt_embeds=prompt_embeds,
                      negative_prompt_embeds=negative_prompt_embeds, gligen_phrases=gligen_phrases, gligen_boxes=gligen_boxes,
                      gligen_phrase_embeds=gligen_phrase_embeds, backward_guidance_kwargs=backward_guidance_kwargs, callback=callback,
                      guidance_callback=guidance_callback)
        out = self.sample_many([sample], height=height, width=width, num_frames=num_frames, num_inference_steps=num_inference_steps,
                               guidance_scale=guidance_scale, gligen_scheduled_sampling_beta=gligen_scheduled_sampling_beta, output_type=output_type,
                               callback_steps=callback_steps, cross_attention_kwargs=cross_attention_kwargs, guidance_type=guidance_type,
                               return_guidance_saved_attn=return_guidance_saved_attn, custom_latent_backward_guidance=custom_latent_backward_guidance)[0]
        return TextToVideoSDPipelineOutput(frames=out) if return_dict else (out,)

    # ------------------------------------------------------------------ V samples per GPU (throughput mode)
    @torch.no_grad()
    def sample_many(self, samples: List[Dict[str, Any]], *, height=None, width=None, num_frames=16, num_inference_steps=50, guidance_scale=9.0,
                    gligen_scheduled_sampling_beta=0.3, output_type="np", callback_steps=1, cross_attention_kwargs=None, guidance_type="main",
                    return_guidance_saved_attn=False, custom_latent_backward_guidance=None):
        """The denoising loop of `__call__` over V independent (prompt, seed) samples at once (`__call__` is the V = 1 case of this function).
        Each entry of `samples` holds one sample's own arguments of `__call__` (prompt / negative_prompt or prompt_embeds /
        negative_prompt_embeds, generator, latents, gligen_*, backward_guidance_kwargs, callback, guidance_callback).  What is shared is the
        schedule: every step runs the V guidance passes one after the other (each a batch-1 recorded forward + backward on its own latents,
        loss and layout — exactly the launches of a single-sample run) and then ONE classifier-free-guidance forward of batch 2V
        [uncond_0, cond_0, uncond_1, cond_1, ...], followed by the V fused CFG / DPM-Solver++ updates.  The deep UNet levels, whose grids
        do not fill 256 CUs at batch 2, are what gains (bench.py --videos-per-gpu: +5 % guided, +11 % unguided at V = 2).  Every sample keeps
        its own generator, so its initial noise — and, up to the bf16 rounding of tile geometries chosen for another M — its video is the one
        a V = 1 run of the same (prompt, seed) produces.  Returns one entry per sample: latents (output_type="latent") or decoded frames."""
        sample_size = self.unet.config.sample_size
        height = height or sample_size * self.vae_scale_factor
        width = width or sample_size * self.vae_scale_factor
        device = self._device
        do_cfg = guidance_scale > 1.0
        cross_attention_kwargs = dict(cross_attention_kwargs) if cross_attention_kwargs is not None else {}
        for smp in samples:  # every argument check before anything touches the GPU
            prompt = smp.get("prompt")
            self.check_inputs(prompt, height, width, callback_steps, smp.get("gligen_phrases"), smp.get("gligen_boxes"), smp.get("negative_prompt"),
                              smp.get("prompt_embeds"), smp.get("negative_prompt_embeds"), num_frames)
            batch_size = 1 if isinstance(prompt, str) else len(prompt) if prompt is not None else smp["prompt_embeds"].shape[0]
            if batch_size != 1:
                raise NotImplementedError("one video per sample entry (the reference's generation modules never batch prompts); pass V entries instead")
        if not do_cfg:
            raise NotImplementedError("guidance_scale <= 1 (no classifier-free guidance) is not on the measured path")
        engine = self.unet._ensure_engine()
        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = self.scheduler.timesteps
        states = []
        for smp in samples:
            prompt = smp.get("prompt")
            batch_size = 1
            pe = self._encode_prompt(prompt, device, do_cfg, smp.get("negative_prompt"), smp.get("prompt_embeds"), smp.get("negative_prompt_embeds"))
            st = dict(prompt_embeds=pe, text_cond=engine.encode_text(pe[1:2]),
                      latents=self.prepare_latents(batch_size, self.unet.config.in_channels, num_frames, height, width, device, smp.get("generator"),
                                                   smp.get("latents")),
                      gligen=self._gligen_tensors(smp.get("gligen_phrases"), smp.get("gligen_boxes"), smp.get("gligen_phrase_embeds"), device),
                      bg_kwargs=smp.get("backward_guidance_kwargs"), callback=smp.get("callback"), guidance_callback=smp.get("guidance_callback"),
                      loss_attn=torch.tensor(10000.0))
            st["x0_prev"] = torch.zeros_like(st["latents"])
            states.append(st)
        V = len(states)
        text_cfg = engine.encode_text(torch.cat([st["prompt_embeds"] for st in states]))  # [uncond_0, cond_0, uncond_1, cond_1, ...]
        with_gl = [st["gligen"] is not None for st in states]
        if any(with_gl) and not all(with_gl):
            raise ValueError("either every sample of a batch carries GLIGEN boxes or none does")
        gligen = {k: torch.cat([st["gligen"][k] for st in states]) for k in states[0]["gligen"]} if all(with_gl) and V else None
        num_grounding_steps = int(gligen_scheduled_sampling_beta * len(timesteps))
        self.enable_fuser(True)
        backward_guidance = custom_latent_backward_guidance if custom_latent_backward_guidance else hip_latent_backward_guidance
        # V > 1 with the stock guidance function and the same guidance hyper-parameters for every sample (what run_many builds): the V guidance
        # passes of a step are ONE recorded forward / backward of batch V (guidance.hip_latent_backward_guidance_many) — per-sample layouts,
        # losses, thresholds and iteration counts as in the one-by-one loop.  Anything else keeps the one-by-one loop below.
        shared_hp = None
        if V > 1 and backward_guidance is hip_latent_backward_guidance and not return_guidance_saved_attn and guidance_type == "main" \
                and all(st["bg_kwargs"] is not None and st["guidance_callback"] is None for st in states):
            hps = [{k: v for k, v in st["bg_kwargs"].items() if k not in ("bboxes", "object_positions")} for st in states]
            try:  # a tensor / ndarray among the values makes `==` ambiguous: such batches keep the one-by-one loop
                same = all(bool(h == hps[0]) for h in hps[1:])
            except (ValueError, RuntimeError, TypeError):
                same = False
            if same:
                shared_hp = hps[0]
                text_cond_all = engine.encode_text(torch.cat([st["prompt_embeds"][1:2] for st in states]))
        for i, t in enumerate(timesteps):
            t = int(t)
            if i == num_grounding_steps:
                self.enable_fuser(False)
            if shared_hp is not None:
                for st in states:
                    assert st["latents"].shape[1] == 4, f"latent channel mismatch: {st['latents'].shape}"
                lats, losses = hip_latent_backward_guidance_many(self.scheduler, self.unet, text_cond_all, i, [st["bg_kwargs"]["bboxes"] for st in states],
                                                                 [st["bg_kwargs"]["object_positions"] for st in states], t,
                                                                 [st["latents"] for st in states], [st["loss_attn"] for st in states], **shared_hp)
                for st, lat, ls in zip(states, lats, losses):
                    st["latents"], st["loss_attn"] = lat, ls
            for st in (states if shared_hp is None else ()):
                assert st["latents"].shape[1] == 4, f"latent channel mismatch: {st['latents'].shape}"
                if st["bg_kwargs"] is not None:
                    if guidance_type != "main":
                        raise ValueError(f"Unsupported guidance type: {guidance_type}")
                    ret = backward_guidance(self.scheduler, self.unet, st["text_cond"], latents=st["latents"], index=i, t=t, loss=st["loss_attn"],
                                            return_saved_attn=return_guidance_saved_attn, **st["bg_kwargs"])
                    st["latents"], st["loss_attn"] = ret[0], ret[1]
                    if st["guidance_callback"] is not None and i % callback_steps == 0:
                        st["guidance_callback"](i, t, st["latents"], None, ret[2] if return_guidance_saved_attn else None)
            fuser_on = all(m.enabled for m in self.unet.modules() if type(m).__name__ == "GatedSelfAttentionDense")
            xs = torch.cat([st["latents"] for st in states]).contiguous()
            eps = engine.forward_cfg(xs, t, text=text_cfg, gligen=gligen, fuser_enabled=fuser_on)  # [uncond_0, cond_0, uncond_1, ...]
            a_t, s_t, c_x, c_0, c_1 = self.scheduler.coefficients(i)
            for v, st in enumerate(states):
                st["latents"] = st["latents"].contiguous()
                ops.cfg_dpm_step(eps[2 * v:2 * v + 1], eps[2 * v + 1:2 * v + 2], guidance_scale, st["latents"], st["x0_prev"], a_t, s_t, c_x, c_0, c_1)
            self.scheduler.advance()
            for st in states:
                if st["callback"] is not None and i % callback_steps == 0:
                    st["callback"](i, t, st["latents"])
        outs = []
        for st in states:
            if output_type == "latent":
                outs.append(st["latents"])
                continue
            video = self.decode_latents(st["latents"])  # the reference decodes twice (:963,969); once is enough
            if isinstance(video, torch.Tensor):
                video = video.float().cpu().numpy()
            outs.append(np.asarray(video))
        return outs

    def _gligen_tensors(self, gligen_phrases, gligen_boxes, gligen_phrase_embeds, device):
        """GLIGEN tensors of one sample (controllable_pipeline…py:736-814): 30 slots per frame, cond half masked in, uncond half masked out."""
        if not gligen_boxes:
            return None
        max_objs, dc = 30, self.unet.config.cross_attention_dim
        boxes_all, emb_all, masks_all = [], [], []
        for f, (phr, bxs) in enumerate(zip(gligen_phrases, gligen_boxes)):
            if len(bxs) > max_objs:
                warnings.warn(f"More than {max_objs} objects found. Only first {max_objs} objects will be processed.", FutureWarning)
                phr, bxs = phr[:max_objs], bxs[:max_objs]
            n = len(bxs)
            boxes = torch.zeros(max_objs, 4)
            emb = torch.zeros(max_objs, dc)
            masks = torch.zeros(max_objs)
            if n:
                boxes[:n] = torch.tensor(bxs)
                if gligen_phrase_embeds is not None:
                    emb[:n] = gligen_phrase_embeds[f][:n].float().cpu()
                else:
                    if self.text_encoder is None:
                        raise RuntimeError("GLIGEN phrase embeddings: pass gligen_phrase_embeds [frames, n_obj, cross_dim] or inject text_encoder")
                    tok = self.tokenizer(phr, padding=True, return_tensors="pt").to(device)
                    emb[:n] = self.text_encoder(**tok).pooler_output.float().cpu()
                masks[:n] = 1
            boxes_all.append(torch.stack([boxes, boxes]))
            emb_all.append(torch.stack([emb, emb]))
            masks_all.append(torch.stack([torch.zeros_like(masks), masks]))
        return {"boxes": torch.stack(boxes_all, 1).flatten(0, 1), "positive_embeddings": torch.stack(emb_all, 1).flatten(0, 1),
                "masks": torch.stack(masks_all, 1).flatten(0, 1)}
