"""Inference call pattern of the path (reference apps/gradio_canny2image.py:66-92):
hint-encode the guide once (the processors keep the control states across scheduler steps), then a
scheduler loop with classifier-free guidance (UNet batch 2x).  VAE decode / CLIP are outside the hot path
 -- this returns denoised latents."""
from __future__ import annotations

import torch

from .schedulers import DDIMScheduler


@torch.no_grad()
def ddim_sample(unet, control_lora, guide, cond_emb, uncond_emb, steps=50, guidance_scale=9.0, latents=None,
                generator=None):
    """guide [Bc,3,H,W] (control batch 1 broadcasts over the CFG batch, quirk C6); cond/uncond [B,77,768]."""
    B = cond_emb.shape[0]
    dev = cond_emb.device
    H, W = guide.shape[2] // 8, guide.shape[3] // 8
    sched = DDIMScheduler()
    sched.set_timesteps(steps)
    if latents is None:
        latents = torch.randn((B, 4, H, W), device=dev, dtype=torch.float16, generator=generator) * sched.init_noise_sigma
    if control_lora is not None:
        control_lora(guide)
    ehs = torch.cat([uncond_emb, cond_emb], 0)
    for t in sched.timesteps:
        eps = unet(torch.cat([latents, latents], 0), t, ehs).sample
        eps_u, eps_c = eps.float().chunk(2)
        eps = eps_u + guidance_scale * (eps_c - eps_u)
        latents = sched.step(eps, t, latents)
    return latents
