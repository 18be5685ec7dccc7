"""ORACLE (test infrastructure only) — DPM-Solver++(2M) as diffusers==0.27.2 DPMSolverMultistepScheduler runs it
for the reference (generation/lvd.py:46: DPMSolverMultistepScheduler.from_config(pipe.scheduler.config)).

PARITY UNPINNED: diffusers is not vendored under /root/reference and the checkpoint's scheduler_config.json is
not available offline.  Restated from the public 0.27.2 definitions with the SD-family config the zeroscope /
modelscope checkpoints ship (1000 train steps, scaled_linear betas 0.00085..0.012, epsilon prediction) and the
class defaults solver_order=2, algorithm_type="dpmsolver++", solver_type="midpoint", lower_order_final=True,
final_sigmas_type="zero", timestep_spacing="linspace" (SURVEY Appendix D).
"""
import numpy as np
import torch


class DPMSolverPP2M:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, timestep_spacing="linspace", steps_offset=0):
        betas = torch.linspace(beta_start**0.5, beta_end**0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.num_train_timesteps = num_train_timesteps
        self.timestep_spacing = timestep_spacing
        self.steps_offset = steps_offset
        self.init_noise_sigma = 1.0

    def set_timesteps(self, n):
        T = self.num_train_timesteps
        if self.timestep_spacing == "linspace":
            ts = np.linspace(0, T - 1, n + 1).round()[::-1][:-1].copy().astype(np.int64)
        elif self.timestep_spacing == "leading":
            ts = (np.arange(0, n + 1) * (T // (n + 1))).round()[::-1][:-1].copy().astype(np.int64) + self.steps_offset
        else:
            raise ValueError(self.timestep_spacing)
        ac = self.alphas_cumprod.numpy().astype(np.float64)
        sig = ((1 - ac) / ac) ** 0.5
        sig = np.interp(ts, np.arange(0, len(sig)), sig)
        self.sigmas = np.concatenate([sig, [0.0]]).astype(np.float32)
        self.timesteps = ts
        self.x0_prev = None
        self.step_index = 0
        self.lower_order_nums = 0

    @staticmethod
    def _alpha_sigma(s):
        a = 1.0 / (s * s + 1.0) ** 0.5
        return a, s * a

    def scale_model_input(self, x, t=None):
        return x

    def step(self, eps, sample):
        i = self.step_index
        n = len(self.timesteps)
        a_s, s_s = self._alpha_sigma(float(self.sigmas[i]))
        x0 = (sample - s_s * eps) / a_s
        a_t, s_t = self._alpha_sigma(float(self.sigmas[i + 1]))
        lam = lambda a, s: np.log(a) - np.log(s) if s > 0 else np.inf
        h = lam(a_t, s_t) - lam(a_s, s_s)
        e = float(np.exp(-h)) - 1.0
        final = i == n - 1
        if self.lower_order_nums < 1 or final:
            out = (s_t / s_s) * sample - a_t * e * x0
        else:
            a_p, s_p = self._alpha_sigma(float(self.sigmas[i - 1]))
            h0 = lam(a_s, s_s) - lam(a_p, s_p)
            r0 = h0 / h
            d1 = (x0 - self.x0_prev) / r0
            out = (s_t / s_s) * sample - a_t * e * x0 - 0.5 * a_t * e * d1
        self.x0_prev = x0
        self.lower_order_nums = min(self.lower_order_nums + 1, 2)
        self.step_index += 1
        return out

    def add_noise(self, x, noise, i):
        """alpha_i x + sigma_i noise at schedule position i (DPMSolverMultistepScheduler.add_noise looks the sigma up by timestep)."""
        a, s = self._alpha_sigma(float(self.sigmas[i]))
        return a * x + s * noise
