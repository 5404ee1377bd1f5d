#!/usr/bin/env python
"""Command-line form of the pose app's `process()` (reference apps/gradio_pose2image.py:68-96) on the MI355X path.

The reference runs an OpenPose annotator on the input photo (out of scope here: SURVEY.md section 2 marks `annotator/` OOS);
what it does WITH the detected map is the call pattern this file keeps: the pose map (a pre-rendered skeleton image -- the
`--pose` argument, e.g. the first image the reference app returns) is resized to the generation resolution with NEAREST
interpolation (`cv2.resize(..., INTER_NEAREST)`, line 77), turned into the control tensor `map[..., ::-1] / 127.5 - 1`
(line 79), hint-encoded once per sample, and sampled with CFG -- 30 DPM-Solver++ steps by default (slider default, line 112).

    python apps/pose2image.py --base /path/to/stable-diffusion-v1-5 --control_lora /path/to/sd-mpii-pose-model-control-lora \\
        --pose skeleton.png --prompt "a man dancing" --out out.png --num_samples 2 --image_resolution 512

`process()` keeps the reference's argument list (the `input_image` slot carries the photo that only fixes the output size; pass
the pose map itself when there is no photo) and its return convention ([detected map] + generated images, uint8 HWC).
"""
from __future__ import annotations

import argparse
import os
import random
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
sys.path.insert(0, os.path.abspath(os.path.dirname(__file__)))

from canny2image import hwc3, resize_image     # noqa: E402  (the annotator utilities both apps share)


def nearest_resize(img: np.ndarray, W: int, H: int) -> np.ndarray:
    """`cv2.resize(img, (W, H), interpolation=cv2.INTER_NEAREST)`: source index = floor(dst * src / dst_size)"""
    h, w = img.shape[:2]
    ys = np.minimum((np.arange(H) * (h / H)).astype(np.int64), h - 1)
    xs = np.minimum((np.arange(W) * (w / W)).astype(np.int64), w - 1)
    return img[ys][:, xs]


def process(pipe, input_image, prompt, a_prompt, n_prompt, num_samples, image_resolution, detect_resolution, sample_steps, scale,
            seed, eta, pose_map=None, sampler="dpm"):
    """reference argument order (apps/gradio_pose2image.py:68) + `pose_map`: the annotator's output, supplied instead of computed
    (None: `input_image` IS the pose map).  `detect_resolution` only sized the annotator's input in the reference and is accepted
    for compatibility; `eta` likewise (both samplers here are deterministic)."""
    import torch
    input_image = hwc3(input_image)
    detected_map = hwc3(pose_map if pose_map is not None else resize_image(input_image, detect_resolution))
    img = resize_image(input_image, image_resolution)
    H, W, _ = img.shape
    detected_map = nearest_resize(detected_map, W, H)
    control = torch.from_numpy(detected_map[..., ::-1].copy().transpose(2, 0, 1)).float()[None] / 127.5 - 1.0
    if seed == -1:
        seed = random.randint(0, 65535)
    images = pipe(prompt, control, a_prompt=a_prompt, n_prompt=n_prompt, num_samples=num_samples, ddim_steps=sample_steps,
                  scale=scale, seed=seed, sampler=sampler)
    return [detected_map] + [im.numpy() for im in images]


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--base", required=True, help="SD-1.5 checkpoint directory (diffusers layout) or random:sd15")
    ap.add_argument("--control_lora", required=True, help="directory written by ControlLoRA.save_pretrained")
    ap.add_argument("--pose", required=True, help="pre-rendered pose map (the OpenPose annotator is out of scope)")
    ap.add_argument("--input", default=None, help="optional photo: only fixes the aspect ratio of the output, like the reference")
    ap.add_argument("--prompt", required=True)
    ap.add_argument("--a_prompt", default="best quality, extremely detailed")
    ap.add_argument("--n_prompt", default="longbody, lowres, bad anatomy, bad hands, missing fingers, extra digit, fewer digits, cropped, worst quality, low quality")
    ap.add_argument("--num_samples", type=int, default=1)
    ap.add_argument("--image_resolution", type=int, default=512)
    ap.add_argument("--detect_resolution", type=int, default=512)
    ap.add_argument("--sample_steps", type=int, default=30)
    ap.add_argument("--scale", type=float, default=9.0)
    ap.add_argument("--seed", type=int, default=-1)
    ap.add_argument("--eta", type=float, default=0.0)
    ap.add_argument("--sampler", default="dpm", choices=["dpm", "ddim"])
    ap.add_argument("--out", default="pose2image.png")
    a = ap.parse_args(argv)
    from PIL import Image
    from controllora_amd.pipeline import ControlLoRAPipeline
    pipe = ControlLoRAPipeline.from_pretrained(a.base, a.control_lora)
    pose = np.asarray(Image.open(a.pose).convert("RGB"))
    photo = np.asarray(Image.open(a.input).convert("RGB")) if a.input else pose
    results = process(pipe, photo, a.prompt, a.a_prompt, a.n_prompt, a.num_samples, a.image_resolution, a.detect_resolution,
                      a.sample_steps, a.scale, a.seed, a.eta, pose_map=pose, sampler=a.sampler)
    Image.fromarray(np.concatenate(results, axis=1)).save(a.out)
    print("wrote", a.out)


if __name__ == "__main__":
    main()
