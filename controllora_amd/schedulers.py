"""Noise schedules used on the path: DDPM ``add_noise`` for training
(reference train_text_to_image_control_lora.py:399, 765) and the DDIM(eta=0) update of the BASELINE
inference configuration.  Scalar/per-sample coefficient math only -- tiny torch ops, no kernels."""
from __future__ import annotations

import torch


class DDPMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, prediction_type="epsilon"):
        self.num_train_timesteps = num_train_timesteps
        self.prediction_type = prediction_type
        betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        self.alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
        self.init_noise_sigma = 1.0

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        a = ac[timesteps].sqrt().reshape(-1, 1, 1, 1)
        s = (1 - ac[timesteps]).sqrt().reshape(-1, 1, 1, 1)
        return a * original_samples + s * noise


    def get_velocity(self, sample, noise, timesteps):
        """v-prediction target (reference train...:776-777): sqrt(a_t) * noise - sqrt(1 - a_t) * sample"""
        ac = self.alphas_cumprod.to(device=sample.device, dtype=sample.dtype)
        a = ac[timesteps].sqrt().reshape(-1, 1, 1, 1)
        s = (1 - ac[timesteps]).sqrt().reshape(-1, 1, 1, 1)
        return a * noise - s * sample


class DDIMScheduler(DDPMScheduler):
    """eta = 0, steps_offset = 1, set_alpha_to_one = False (the SD-1.5 scheduler config)."""

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ratio = self.num_train_timesteps // n
        self.timesteps = [int(i * ratio) + 1 for i in reversed(range(n))]

    def step(self, eps, t, sample):
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[prev] if prev >= 0 else self.alphas_cumprod[0])
        x0 = (sample.float() - (1 - a_t) ** 0.5 * eps.float()) / a_t ** 0.5
        return (a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps.float()).to(sample.dtype)


class DPMSolverMultistepScheduler(DDPMScheduler):
    """DPM-Solver++(2M), epsilon prediction, `lower_order_final` -- the sampler the reference apps switch to
    (apps/gradio_canny2image.py:34, train...:817: `DPMSolverMultistepScheduler.from_config(...)`, upstream defaults:
    algorithm_type "dpmsolver++", solver_order 2, solver_type "midpoint").  Restated from the DPM-Solver++ paper
    (Lu et al. 2022, eqs. for the data-prediction multistep solver); upstream diffusers is absent, so this is PARITY
    UNPINNED against it -- tests pin the identities that must hold (first order == DDIM(eta=0); an exact-x0 model is
    integrated exactly by both orders).  Scalar coefficient math on the host; the latents stay on the device."""

    def __init__(self, *a, solver_order=2, lower_order_final=True, **kw):
        super().__init__(*a, **kw)
        self.solver_order, self.lower_order_final = solver_order, lower_order_final
        acp = self.alphas_cumprod.double()
        self._alpha, self._sigma = acp.sqrt(), (1 - acp).sqrt()
        self._lambda = torch.log(self._alpha) - torch.log(self._sigma)

    def set_timesteps(self, n):
        self.num_inference_steps = n
        ts = torch.linspace(0, self.num_train_timesteps - 1, n + 1).round().long().flip(0)[:-1]
        self.timesteps = [int(t) for t in ts]
        self._x0_hist, self._t_hist, self._lower = [], [], 0

    def _coef(self, t):
        if t < 0:                      # "final" step integrates to clean data: alpha = 1, sigma = 0
            return 1.0, 0.0, float("inf")
        return float(self._alpha[t]), float(self._sigma[t]), float(self._lambda[t])

    def step(self, eps, t, sample):
        i = self.timesteps.index(int(t))
        prev = self.timesteps[i + 1] if i + 1 < len(self.timesteps) else 0
        a_s, s_s, l_s = self._coef(int(t))
        a_t, s_t, l_t = self._coef(prev)
        x = sample.float()
        x0 = (x - s_s * eps.float()) / a_s
        self._x0_hist = (self._x0_hist + [x0])[-self.solver_order:]
        self._t_hist = (self._t_hist + [int(t)])[-self.solver_order:]
        h = l_t - l_s
        import math
        em1 = math.expm1(-h)                                       # e^{-h} - 1
        last = i == len(self.timesteps) - 1
        first_order = self.solver_order == 1 or self._lower < 1 or (self.lower_order_final and last and len(self.timesteps) < 15)
        if first_order:
            out = (s_t / s_s) * x - a_t * em1 * x0
        else:
            m0, m1 = self._x0_hist[-1], self._x0_hist[-2]
            l_s1 = float(self._lambda[self._t_hist[-2]])
            r0 = (l_s - l_s1) / h
            d1 = (m0 - m1) / r0
            out = (s_t / s_s) * x - a_t * em1 * m0 - 0.5 * a_t * em1 * d1
        if self._lower < self.solver_order:
            self._lower += 1
        return out.to(sample.dtype)
