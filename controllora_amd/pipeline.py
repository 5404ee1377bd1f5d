"""Inference call pattern of the path (reference apps/gradio_canny2image.py:66-92):
hint-encode the guide once (the processors keep the control states across scheduler steps), then a
scheduler loop with classifier-free guidance (UNet batch 2x).  VAE decode / CLIP are outside the hot path
 -- this returns denoised latents."""
from __future__ import annotations

import torch

from .schedulers import DDIMScheduler, DPMSolverMultistepScheduler


@torch.no_grad()
def ddim_sample(unet, control_lora, guide, cond_emb, uncond_emb, steps=50, guidance_scale=9.0, latents=None,
                generator=None, sampler="ddim", graph=None, cache_text_kv=True, callback=None):
    """guide [Bc,3,H,W] (control batch 1 broadcasts over the CFG batch, quirk C6); cond/uncond [B,77,768].
    sampler: "ddim" (BASELINE inference config) or "dpm" (DPM-Solver++(2M), what the reference apps select).

    What is invariant over the scheduler loop is evaluated ONCE: the hint encoder and the 32 control terms (reference
    apps/gradio_canny2image.py:84 -- the processors keep the control states), and with `cache_text_kv` the cross-attention
    K/V projections of the text embedding with their adapters (16 sites; they depend on neither the latents nor the
    timestep).  graph: replay ONE captured hipGraph of the UNet forward per step (default: on a GPU); the timestep lives
    in a device tensor, so all steps replay the same graph (default: on a GPU when steps >= 8).
    callback(i, latents, eps): called after scheduler step i = 1..steps (tests record the trajectory with it).
    Returns the denoised latents in **fp32** (the scheduler state is kept in fp32 between steps)."""
    from . import models
    B = cond_emb.shape[0]
    dev = cond_emb.device
    H, W = guide.shape[2] // 8, guide.shape[3] // 8
    sched = DDIMScheduler() if sampler == "ddim" else DPMSolverMultistepScheduler()
    sched.set_timesteps(steps)
    if latents is None:
        latents = torch.randn((B, 4, H, W), device=dev, dtype=torch.float16, generator=generator) * sched.init_noise_sigma
    # the scheduler state stays fp32 between steps (16 KB per image): only the UNet input is rounded to fp16, so the
    # 50 updates do not each add an fp16 rounding of the latents (denoised-latent parity, tests/full_cases.py)
    latents = latents.float()
    if control_lora is not None:
        # validated ONCE here (the per-site check in models._control_tokens is bypassed by precomputed control terms): a
        # control batch of 1 broadcasts; one guide per image is tiled the way the CFG batch is (uncond..., cond...), which
        # makes the batches equal; anything else has no defined pairing (reference quirk C6) and is rejected
        if guide.shape[0] == B and B > 1:
            guide = torch.cat([guide, guide], 0)
        elif guide.shape[0] not in (1, 2 * B):
            raise ValueError(f"guide batch {guide.shape[0]}: expected 1 (broadcast) or one guide per image ({B})")
        control_lora(guide)
    ehs = torch.cat([uncond_emb, cond_emb], 0).half().contiguous()
    if graph is None:
        # capture costs one extra warm-up forward + the capture itself: only worth it for real sampling runs
        graph = dev.type == "cuda" and steps >= 8
    with models.text_kv_cache(enabled=cache_text_kv):
        if not graph:
            for i, t in enumerate(sched.timesteps, 1):
                eps = unet(torch.cat([latents, latents], 0).half(), t, ehs).sample
                eps_u, eps_c = eps.float().chunk(2)
                latents = sched.step(eps_u + guidance_scale * (eps_c - eps_u), t, latents)
                if callback is not None:
                    callback(i, latents, eps)
            return latents
        x_in = torch.empty((2 * B, 4, H, W), device=dev, dtype=torch.float16)
        t_in = torch.zeros(1, device=dev, dtype=torch.long)
        x_in.copy_(torch.cat([latents, latents], 0))
        t_in.fill_(int(sched.timesteps[0]))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                # warm-up outside the capture: lazy weight packs, K/V cache, allocator
            unet(x_in, t_in, ehs)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            eps_out = unet(x_in, t_in, ehs).sample
        for i, t in enumerate(sched.timesteps, 1):
            x_in.copy_(torch.cat([latents, latents], 0))
            t_in.fill_(int(t))
            g.replay()
            eps_u, eps_c = eps_out.float().chunk(2)
            latents = sched.step(eps_u + guidance_scale * (eps_c - eps_u), t, latents)
            if callback is not None:
                callback(i, latents, eps_out)
        return latents


class ControlLoRAPipeline:
    """The `process()` body of the reference apps (apps/gradio_canny2image.py:66-92) without the UI and the annotator:
    prompt(s) + one guide image -> images.  Text encode (cond / negative), hint-encode the guide ONCE (control batch 1 is
    broadcast over the CFG batch, quirk C6), DDIM loop with classifier-free guidance on the HIP UNet, VAE decode.

        pipe = ControlLoRAPipeline.from_pretrained("random:sd15", "path/to/control_lora")
        images = pipe("a red circle", guide, num_samples=4, ddim_steps=50, scale=9.0, seed=1)   # uint8 [N, H, W, 3]
    """

    def __init__(self, unet, control_lora, vae, text_encoder, tokenizer):
        self.unet, self.control_lora, self.vae, self.text_encoder, self.tokenizer = unet, control_lora, vae, text_encoder, tokenizer

    @classmethod
    def from_pretrained(cls, base: str, control_lora, device="cuda"):
        from . import loading, models, text
        dev = torch.device(device)
        unet = loading.load_unet(base, dev)
        if isinstance(control_lora, str):
            control_lora = models.ControlLoRA.from_pretrained(control_lora)
        control_lora = control_lora.to(dev)
        unet.set_attn_processor(models.map_processors_to_unet(unet, control_lora))
        return cls(unet, control_lora, loading.load_vae(base, dev), text.load_text_encoder(base, dev, small=base.endswith("small")),
                   text.load_tokenizer(base))

    @torch.no_grad()
    def encode_prompt(self, prompts):
        dev = next(self.text_encoder.parameters()).device
        return self.text_encoder(self.tokenizer(list(prompts)).to(dev))[0].half()

    @torch.no_grad()
    def __call__(self, prompt, guide, a_prompt="", n_prompt="", num_samples=1, ddim_steps=50, scale=9.0, seed=None,
                 output_type="uint8", sampler="ddim"):
        """guide: float tensor [1, 3, H, W] (broadcast over the samples) or [num_samples, 3, H, W] (one guide per image), in
        [-1, 1]; H, W multiples of 64"""
        dev = next(self.text_encoder.parameters()).device
        gen = torch.Generator(device=dev)
        if seed is not None:
            gen.manual_seed(int(seed))
        cond = self.encode_prompt([prompt + (", " + a_prompt if a_prompt else "")] * num_samples)
        uncond = self.encode_prompt([n_prompt] * num_samples)
        lat = ddim_sample(self.unet, self.control_lora, guide.to(dev).half(), cond, uncond, steps=ddim_steps,
                          guidance_scale=scale, generator=gen, sampler=sampler)
        img = self.vae.decode(lat.half() / self.vae.scaling_factor).sample.float().clamp(-1, 1)
        if output_type == "uint8":
            return ((img.permute(0, 2, 3, 1) + 1.0) * 127.5).round().to(torch.uint8).cpu()
        return img
