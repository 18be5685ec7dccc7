"""Denoising-loop pieces around the HIP UNet: DPM-Solver++(2M) schedule (host scalars) + fused CFG/solver update.

Mirrors what models/controllable_pipeline_text_to_video_synth.py:833-958 does per step with the third-party
``DPMSolverMultistepScheduler`` (generation/lvd.py:46): the per-step scalar coefficients are computed on the host in
float64 (they depend only on the step index), and one elementwise kernel applies
    eps = eps_u + s (eps_c - eps_u);  x0 = (x - sigma_t eps) / alpha_t;  x <- c_x x + c_0 x0 + c_1 x0_prev
on the fp32 latents (csrc/elementwise.hip).  The (B,C,F,h,w) <-> (B·F,C,h,w) reshapes of :933-950 vanish because the
update is elementwise.  diffusers is not vendored in the reference: defaults restated from diffusers 0.27.2
 — parity for the schedule itself is unpinned.
"""
import math

import numpy as np
import torch

from . import ops


class DPMSolverPP2MSchedule:
    """Constructor defaults = the diffusers class defaults (linspace spacing).  The reference never uses those: it builds
    `DPMSolverMultistepScheduler.from_config(pipe.scheduler.config)` (generation/lvd.py:46) from the checkpoint's DDIMScheduler,
    whose registered config carries DDIM's `timestep_spacing="leading"` and the checkpoint's `steps_offset=1`; that is
    `from_ddim_config()` below, and what every generation method here uses: 40 steps -> t = 961, 937, ..., 25."""
    order = 1  # pipeline bookkeeping (`scheduler.order`): one model evaluation per step
    init_noise_sigma = 1.0

    @classmethod
    def from_ddim_config(cls, **kw):
        return cls(**{"timestep_spacing": "leading", "steps_offset": 1, **kw})

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, timestep_spacing="linspace", steps_offset=0):
        betas = np.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=np.float32).astype(np.float64) ** 2
        self.alphas_cumprod = torch.from_numpy(np.cumprod(1.0 - betas).astype(np.float32))
        self.num_train_timesteps = num_train_timesteps
        self.timestep_spacing = timestep_spacing
        self.steps_offset = steps_offset
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        T = self.num_train_timesteps
        if self.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif self.timestep_spacing == "leading":
            ts = (np.arange(0, n + 1) * (T // (n + 1))).round()[::-1][:-1].copy().astype(np.int64) + self.steps_offset
        else:
            raise ValueError(f"unsupported timestep_spacing {self.timestep_spacing!r}")
        ac = self.alphas_cumprod.numpy().astype(np.float64)
        sig = ((1 - ac) / ac) ** 0.5
        self.sigmas = np.concatenate([np.interp(ts, np.arange(len(sig)), sig), [0.0]])
        self.timesteps = ts
        self.step_index = 0
        self.lower_order_nums = 0

    def scale_model_input(self, x, t=None):
        return x

    @staticmethod
    def _alpha_sigma(s):
        a = 1.0 / math.sqrt(s * s + 1.0)
        return a, s * a

    def coefficients(self, i):
        """(alpha_t, sigma_t, c_x, c_0, c_1) of step i so that x' = c_x x + c_0 x0 + c_1 x0_prev."""
        n = len(self.timesteps)
        a_s, s_s = self._alpha_sigma(float(self.sigmas[i]))
        a_t, s_t = self._alpha_sigma(float(self.sigmas[i + 1]))
        lam = lambda a, s: math.log(a) - math.log(s) if s > 0 else math.inf
        h = lam(a_t, s_t) - lam(a_s, s_s)
        e = math.exp(-h) - 1.0
        c_x = s_t / s_s
        first_order = self.lower_order_nums < 1 or i == n - 1  # lower_order_final with final_sigmas_type="zero"
        if first_order:
            return a_s, s_s, c_x, -a_t * e, 0.0
        a_p, s_p = self._alpha_sigma(float(self.sigmas[i - 1]))
        r0 = (lam(a_s, s_s) - lam(a_p, s_p)) / h
        k = 0.5 * a_t * e / r0
        return a_s, s_s, c_x, -a_t * e - k, k

    def add_noise(self, x, noise, i):
        """DPMSolverMultistepScheduler.add_noise at schedule position i: alpha_i x + sigma_i noise (video-to-video start)."""
        a, s = self._alpha_sigma(float(self.sigmas[i]))
        return a * x + s * noise

    def advance(self):
        self.lower_order_nums = min(self.lower_order_nums + 1, 2)
        self.step_index += 1


class HipSampler:
    """One video: CFG forward (batch 2: [uncond, cond]) + fused CFG/DPM update on fp32 latents (1,4,F,h,w)."""

    def __init__(self, engine, schedule: DPMSolverPP2MSchedule, guidance_scale=9.0):
        self.engine = engine
        self.schedule = schedule
        self.guidance_scale = guidance_scale
        self.x0_prev = None

    def reset(self, latents):
        self.x0_prev = torch.zeros_like(latents)

    def cfg_step(self, latents, i, text_cfg, gligen=None, fuser_enabled=True):
        """latents (1,4,F,h,w) fp32 is updated in place; text_cfg = TextCache of [negative; positive] prompts."""
        sch = self.schedule
        t = int(sch.timesteps[i])
        eps = self.engine.forward_cfg(latents, t, text=text_cfg, gligen=gligen, fuser_enabled=fuser_enabled)  # [uncond, cond]
        a_t, s_t, c_x, c_0, c_1 = sch.coefficients(i)
        ops.cfg_dpm_step(eps[0:1], eps[1:2], self.guidance_scale, latents, self.x0_prev, a_t, s_t, c_x, c_0, c_1)
        sch.advance()
        return latents
