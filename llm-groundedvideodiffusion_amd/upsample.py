"""Video-to-video upsampling with a zeroscope-XL-topology denoiser on the HIP kernels (optional part).

Reference: /root/reference/scripts/upsample.py:49-101 (`upsample_zsxl`): Lanczos-resize the generated video to 1024x576 (or
1024^2), then diffusers' `VideoToVideoSDPipeline(prompt, video=..., strength=0.35, negative_prompt=...)` with
`cerspense/zeroscope_v2_XL` under DPM-Solver++.  zeroscope_v2_XL is the same UNet3DConditionModel topology as the 576w model, so
the denoiser is `HipUNet3D`; the pipeline around it, restated from diffusers 0.27.2 (third-party, parity unpinned):

  video uint8 --Lanczos, 2v/255-1--> VAE encode --sample, x scaling_factor--> z0          (HipVAEEncoder)
  timesteps = schedule[t_start:],  t_start = steps - min(int(steps*strength), steps);  z = alpha z0 + sigma noise
  per step: CFG UNet forward (scale 15) + DPM-Solver++ update                              (HipSampler.cfg_step)
  VAE decode + tensor2vid                                                                  (HipVAEDecoder)

Random numbers are drawn in the reference's order from one generator: posterior epsilon first, then the noise, both (F,4,h,w).
The SDXL-refiner variants of the script (`--use_sdxl`, `--use_zssdxl`) run a different UNet family and are not provided.
"""
import numpy as np
import torch

from .sampler import DPMSolverPP2MSchedule, HipSampler


class HipVideoToVideo:
    def __init__(self, unet, vae_encoder, vae_decoder, schedule=None, encode_prompt=None):
        """`encode_prompt(list_of_str) -> (n, 77, cross_dim)` embeddings (e.g. tokenizer + HipCLIPTextEncoder); only needed when
        the call passes strings instead of `prompt_embeds`."""
        self.unet, self.enc, self.dec = unet, vae_encoder, vae_decoder
        self.schedule = schedule or DPMSolverPP2MSchedule.from_ddim_config()
        self.encode_prompt = encode_prompt

    @staticmethod
    def get_timesteps(num_inference_steps, strength):
        init = min(int(num_inference_steps * strength), num_inference_steps)
        return max(num_inference_steps - init, 0)

    def __call__(self, prompt=None, video=None, strength=0.6, num_inference_steps=50, guidance_scale=15.0, negative_prompt=None,
                 generator=None, prompt_embeds=None, negative_prompt_embeds=None, size=None, output_type="np"):
        if not 0 <= strength <= 1:
            raise ValueError(f"The value of strength should in [0.0, 1.0] but is {strength}")
        if prompt_embeds is None:
            if self.encode_prompt is None:
                raise ValueError("pass prompt_embeds / negative_prompt_embeds, or construct with encode_prompt")
            prompt_embeds = self.encode_prompt([prompt])
            negative_prompt_embeds = self.encode_prompt([negative_prompt or ""])
        dev = self.unet.dev
        video = torch.as_tensor(np.asarray(video))
        frames = video.shape[0]
        sch = self.schedule
        sch.set_timesteps(num_inference_steps)
        t_start = self.get_timesteps(num_inference_steps, strength)
        text = self.unet.encode_text(torch.cat([negative_prompt_embeds, prompt_embeds]).to(dev))
        L = self.enc.cfg.latent_channels
        SH, SW = size if size is not None else video.shape[1:3]
        shape = (frames, L, SH // 8, SW // 8)
        gdev = generator.device if generator is not None else "cpu"
        eps = torch.randn(shape, generator=generator, device=gdev)
        noise = torch.randn(shape, generator=generator, device=gdev)
        z0 = self.enc.encode(video, eps=eps, size=size)
        if t_start >= num_inference_steps:  # strength 0: nothing to denoise
            latents = z0
        else:
            latents = sch.add_noise(z0, noise.to(dev, torch.float32).permute(1, 0, 2, 3).unsqueeze(0), t_start).contiguous()
            sampler = HipSampler(self.unet, sch, guidance_scale=guidance_scale)
            sampler.reset(latents)
            for i in range(t_start, num_inference_steps):
                sampler.cfg_step(latents, i, text)
        if output_type == "latent":
            return latents
        return self.dec(latents)[0]  # (F, H, W, 3) fp32 in [0, 1], like `.frames[0]`
