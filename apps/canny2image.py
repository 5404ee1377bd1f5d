#!/usr/bin/env python
"""Command-line form of the reference app's `process()` (reference apps/gradio_canny2image.py:66-92) on the MI355X path:
resize the input image, Canny edge map (numpy implementation below: gradio / OpenCV are not in this image), control
tensor = edges / 127.5 - 1, hint-encode once per sample, CFG sampling (DPM-Solver++ like the app, or DDIM), VAE decode.

    python apps/canny2image.py --base /path/to/stable-diffusion-v1-5 --control_lora /path/to/control-lora \\
        --input photo.png --prompt "a cute dog" --out out.png --num_samples 2 --image_resolution 512

`process()` keeps the reference's argument list and return convention ([255 - edge map] + generated images, uint8 HWC).
"""
from __future__ import annotations

import argparse
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))


def hwc3(x: np.ndarray) -> np.ndarray:
    """uint8 image -> 3 channels (grey replicated, alpha composited on white)"""
    assert x.dtype == np.uint8
    if x.ndim == 2:
        x = x[:, :, None]
    if x.shape[2] == 1:
        return np.concatenate([x, x, x], axis=2)
    if x.shape[2] == 3:
        return x
    color, alpha = x[:, :, :3].astype(np.float32), x[:, :, 3:4].astype(np.float32) / 255.0
    return (color * alpha + 255.0 * (1.0 - alpha)).clip(0, 255).astype(np.uint8)


def resize_image(img: np.ndarray, resolution: int) -> np.ndarray:
    """short side -> `resolution`, both sides rounded to multiples of 64 (what the app's annotator util does)"""
    from PIL import Image
    H, W, _ = img.shape
    k = float(resolution) / min(H, W)
    H2, W2 = int(np.round(H * k / 64.0)) * 64, int(np.round(W * k / 64.0)) * 64
    return np.asarray(Image.fromarray(img).resize((W2, H2), Image.LANCZOS if k > 1 else Image.BOX))


from controllora_amd.process import canny          # noqa: E402  (numpy Canny: shared with the process/diffusiondb_canny data set)


def process(pipe, input_image, prompt, a_prompt, n_prompt, num_samples, image_resolution, sample_steps, scale, seed, eta,
            low_threshold, high_threshold, sampler="dpm"):
    """reference argument order; `eta` is accepted for compatibility (both samplers here are deterministic, eta = 0)"""
    import torch
    img = resize_image(hwc3(input_image), image_resolution)
    detected_map = hwc3(canny(img, low_threshold, high_threshold))
    control = torch.from_numpy(detected_map[..., ::-1].copy().transpose(2, 0, 1)).float()[None] / 127.5 - 1.0
    if seed == -1:
        seed = random.randint(0, 65535)
    images = pipe(prompt, control, a_prompt=a_prompt, n_prompt=n_prompt, num_samples=num_samples, ddim_steps=sample_steps,
                  scale=scale, seed=seed, sampler=sampler)
    return [255 - detected_map] + [im.numpy() for im in images]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--base", required=True, help="SD-1.5 checkpoint directory (diffusers layout) or random:sd15")
    ap.add_argument("--control_lora", required=True, help="directory written by ControlLoRA.save_pretrained")
    ap.add_argument("--input", required=True)
    ap.add_argument("--prompt", required=True)
    ap.add_argument("--a_prompt", default="best quality, extremely detailed")
    ap.add_argument("--n_prompt", default="longbody, lowres, bad anatomy, bad hands, missing fingers, extra digit, fewer digits, cropped, worst quality, low quality")
    ap.add_argument("--num_samples", type=int, default=1)
    ap.add_argument("--image_resolution", type=int, default=512)
    ap.add_argument("--sample_steps", type=int, default=20)
    ap.add_argument("--scale", type=float, default=9.0)
    ap.add_argument("--seed", type=int, default=-1)
    ap.add_argument("--eta", type=float, default=0.0)
    ap.add_argument("--low_threshold", type=int, default=100)
    ap.add_argument("--high_threshold", type=int, default=200)
    ap.add_argument("--sampler", default="dpm", choices=["dpm", "ddim"])
    ap.add_argument("--out", default="canny2image.png")
    a = ap.parse_args(argv)
    from PIL import Image
    from controllora_amd.pipeline import ControlLoRAPipeline
    pipe = ControlLoRAPipeline.from_pretrained(a.base, a.control_lora)
    results = process(pipe, np.asarray(Image.open(a.input).convert("RGB")), a.prompt, a.a_prompt, a.n_prompt, a.num_samples,
                      a.image_resolution, a.sample_steps, a.scale, a.seed, a.eta, a.low_threshold, a.high_threshold, a.sampler)
    Image.fromarray(np.concatenate(results, axis=1)).save(a.out)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
