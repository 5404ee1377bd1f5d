"""Host-side data path of the training entry point (reference train_text_to_image_control_lora.py:520-648):
image / guide / caption triples -> `pixel_values`, `guide_values` in [-1, 1] and `input_ids`.

* `SyntheticFill50k`  -- fill50k-like generator (reference tasks/make_dataset_fill50k.py:12-18 semantics: a filled
  disc of random colour on a random background, the guide is the disc outline, the caption names the colours); no
  dataset is reachable offline, so this is what bench/tests/smoke train on.
* `ImageGuideDataset` -- wraps a HuggingFace `datasets` split or an image folder with the reference's transform:
  bilinear resize of the short side to `resolution`, ToTensor, Normalize(0.5, 0.5), one shared random crop.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

_COLOURS = {"red": (230, 25, 25), "green": (25, 200, 25), "blue": (25, 25, 230), "yellow": (240, 240, 30), "white": (245, 245, 245),
            "black": (10, 10, 10), "purple": (150, 30, 200), "orange": (250, 150, 20), "pink": (250, 150, 200), "cyan": (20, 220, 220)}


class SyntheticFill50k(torch.utils.data.Dataset):
    def __init__(self, resolution: int = 512, length: int = 50000, seed: int = 42, tokenizer: Optional[Callable] = None):
        self.res, self.length, self.seed, self.tokenizer = resolution, length, seed, tokenizer
        yy, xx = np.mgrid[0:resolution, 0:resolution]
        self._yy, self._xx = yy.astype(np.float32), xx.astype(np.float32)

    def __len__(self):
        return self.length

    def __getitem__(self, i: int) -> Dict[str, torch.Tensor]:
        rng = np.random.default_rng(self.seed * 1000003 + i)
        names = list(_COLOURS)
        fg, bg = rng.choice(len(names), size=2, replace=False)
        r = rng.uniform(0.08, 0.3) * self.res
        cx, cy = rng.uniform(r, self.res - r, size=2)
        d = np.sqrt((self._xx - cx) ** 2 + (self._yy - cy) ** 2)
        inside = (d <= r)[..., None]
        img = np.where(inside, np.array(_COLOURS[names[fg]], np.float32), np.array(_COLOURS[names[bg]], np.float32))
        ring = (np.abs(d - r) <= max(1.0, self.res / 256.0))[..., None]
        guide = np.where(ring, 255.0, 0.0).astype(np.float32).repeat(3, axis=2)
        caption = f"{names[fg]} circle with {names[bg]} background"
        out = {"pixel_values": torch.from_numpy(img / 127.5 - 1.0).permute(2, 0, 1).contiguous(),
               "guide_values": torch.from_numpy(guide / 127.5 - 1.0).permute(2, 0, 1).contiguous(), "caption": caption}
        if self.tokenizer is not None:
            out["input_ids"] = self.tokenizer([caption])[0]
        return out


def _to_tensor_resized(img, resolution: int) -> torch.Tensor:
    """PIL image -> float [3,h,w] in [-1,1], short side bilinearly resized to `resolution`"""
    from PIL import Image
    img = img.convert("RGB")
    w, h = img.size
    s = resolution / min(w, h)
    img = img.resize((max(resolution, round(w * s)), max(resolution, round(h * s))), Image.BILINEAR)
    t = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0).permute(2, 0, 1)
    return (t - 0.5) / 0.5


class ImageGuideDataset(torch.utils.data.Dataset):
    def __init__(self, rows, image_column: str, guide_column: str, caption_column: str, resolution: int, tokenizer: Callable):
        self.rows, self.cols, self.res, self.tokenizer = rows, (image_column, guide_column, caption_column), resolution, tokenizer

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, i):
        ex = self.rows[int(i)]
        image, guide = _to_tensor_resized(ex[self.cols[0]], self.res), _to_tensor_resized(ex[self.cols[1]], self.res)
        _, h, w = image.shape
        y1 = int(torch.randint(0, h - self.res, (1,))) if h != self.res else 0     # one crop shared by image and guide
        x1 = int(torch.randint(0, w - self.res, (1,))) if (h == self.res and w != self.res) else 0
        image = image[:, y1:y1 + self.res, x1:x1 + self.res]
        guide = guide[:, y1:y1 + self.res, x1:x1 + self.res]
        cap = ex[self.cols[2]]
        if isinstance(cap, (list, tuple, np.ndarray)):
            cap = cap[int(torch.randint(0, len(cap), (1,)))]
        return {"pixel_values": image.contiguous(), "guide_values": guide.contiguous(), "caption": cap,
                "input_ids": self.tokenizer([cap])[0]}


def collate(examples: List[dict]) -> Dict[str, torch.Tensor]:
    out = {"pixel_values": torch.stack([e["pixel_values"] for e in examples]).float(),
           "guide_values": torch.stack([e["guide_values"] for e in examples]).float()}
    if "input_ids" in examples[0]:
        out["input_ids"] = torch.stack([e["input_ids"] for e in examples])
    return out


# ---- LR schedules (diffusers.optimization.get_scheduler names accepted by --lr_scheduler) as multipliers of the base LR
def lr_lambda(name: str, warmup: int, total: int, cycles: float = 0.5, power: float = 1.0) -> Callable[[int], float]:
    warm = lambda s: float(s) / float(max(1, warmup))
    if name == "constant":
        return lambda s: 1.0
    if name == "constant_with_warmup":
        return lambda s: warm(s) if s < warmup else 1.0
    if name == "linear":
        return lambda s: warm(s) if s < warmup else max(0.0, float(total - s) / float(max(1, total - warmup)))
    if name == "cosine":
        return lambda s: warm(s) if s < warmup else max(
            0.0, 0.5 * (1.0 + math.cos(math.pi * cycles * 2.0 * float(s - warmup) / float(max(1, total - warmup)))))
    if name == "cosine_with_restarts":
        def f(s, n=1):
            if s < warmup:
                return warm(s)
            p = float(s - warmup) / float(max(1, total - warmup))
            return 0.0 if p >= 1.0 else max(0.0, 0.5 * (1.0 + math.cos(math.pi * ((float(n) * p) % 1.0))))
        return f
    if name == "polynomial":
        def g(s, lr_end=1e-7, lr_init=1.0):
            if s < warmup:
                return warm(s)
            if s > total:
                return lr_end / lr_init
            decay = (1 - (s - warmup) / (total - warmup)) ** power
            return ((lr_init - lr_end) * decay + lr_end) / lr_init
        return g
    raise ValueError(f"unknown --lr_scheduler {name!r}")
