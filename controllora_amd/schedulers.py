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
