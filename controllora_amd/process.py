"""The `--dataset_name process/<name>` data sets of the training entry point (reference process/base.py:8-38 registry,
process/diffusiondb_canny.py:11-48, process/mpii_pose.py:10-46, process/danbooru_sketch.py:9-76; selected at
train_text_to_image_control_lora.py:546-550).  SURVEY.md section 8 (f)4, CPU-side data path.

One item = {"pixel_values" [3,h,w] in [-1,1], "guide_values" [3,h,w] in [-1,1], "input_ids"}; the random numbers are drawn from
torch's global generator in the reference's order (style -> crop x -> crop y -> Canny low -> Canny high), so a seeded run picks the
crops / thresholds the reference would.  What differs, by necessity of this image:

* OpenCV is absent: the edge map comes from `canny()` below (grey-level Sobel / L1 magnitude / NMS / hysteresis; checked against a
  scipy construction in tests/test_canny_cpu.py).  `cv2.Canny` on a colour image takes the per-pixel strongest channel gradient
  instead of the grey level, and swaps the thresholds when low > high -- the swap is kept here.
* the DiffusionDB split and the MPII / Danbooru folders are not reachable offline: every class takes its records through a
  keyword (`rows=` / `root=`), defaulting to the reference's source (`datasets.load_dataset("poloclub/diffusiondb", "2m_random_1k")`,
  `data/mpii/prompt.jsonl`, `data/danbooru-2020-512-prompt.jsonl`).
* the tokenizer is called as `tokenizer([caption])[0]` (controllora_amd/text.py), not with an examples dict.
"""
from __future__ import annotations

import json
import os
from typing import Callable, Dict, Optional, Tuple

import numpy as np
import torch


def _conv2_same(a: np.ndarray, k: np.ndarray) -> np.ndarray:
    ph, pw = k.shape[0] // 2, k.shape[1] // 2
    p = np.pad(a, ((ph, ph), (pw, pw)), mode="edge")
    out = np.zeros_like(a, dtype=np.float32)
    for i in range(k.shape[0]):
        for j in range(k.shape[1]):
            out += k[i, j] * p[i:i + a.shape[0], j:j + a.shape[1]]
    return out


def canny(img: np.ndarray, low: float, high: float) -> np.ndarray:
    """Canny edge detector (grey -> Sobel gradient, L1 magnitude like OpenCV's default, non-maximum suppression along the
    quantised gradient direction, double threshold with 8-connected hysteresis) -> uint8 {0, 255} map [H, W]"""
    g = img.astype(np.float32)
    if g.ndim == 3:
        g = 0.299 * g[..., 0] + 0.587 * g[..., 1] + 0.114 * g[..., 2]
    kx = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], np.float32)
    gx, gy = _conv2_same(g, kx), _conv2_same(g, kx.T)
    mag = np.abs(gx) + np.abs(gy)
    ang = (np.rad2deg(np.arctan2(gy, gx)) + 180.0) % 180.0
    sector = ((ang + 22.5) // 45).astype(np.int32) % 4                 # 0: E-W, 1: NE-SW, 2: N-S, 3: NW-SE
    p = np.pad(mag, 1)
    H, W = mag.shape
    nb = {0: (p[1:H + 1, 2:], p[1:H + 1, :W]), 1: (p[2:, 2:], p[:H, :W]), 2: (p[2:, 1:W + 1], p[:H, 1:W + 1]),
          3: (p[2:, :W], p[:H, 2:])}
    keep = np.zeros_like(mag, dtype=bool)
    for s_, (a, b) in nb.items():
        keep |= (sector == s_) & (mag >= a) & (mag >= b)
    strong = keep & (mag >= high)
    weak = keep & (mag >= low)
    out = strong.copy()
    while True:                                                        # hysteresis: grow strong edges through weak pixels
        q = np.pad(out, 1)
        grown = weak & (q[:-2, :-2] | q[:-2, 1:-1] | q[:-2, 2:] | q[1:-1, :-2] | q[1:-1, 2:] | q[2:, :-2] | q[2:, 1:-1] | q[2:, 2:] | out)
        if (grown == out).all():
            break
        out = grown
    return out.astype(np.uint8) * 255


def _draw(lo: int, hi: int) -> int:
    """one `torch.randint(lo, hi, (1,)).item()` of the reference (global generator: `--seed` governs it)"""
    return int(torch.randint(lo, hi, (1,)).item())


def _read_jsonl(path: str):
    with open(path, "r") as f:
        return [json.loads(line) for line in f if line.strip()]


class Dataset(torch.utils.data.Dataset):
    """Base + registry (reference process/base.py:8-38): `Dataset.from_name("process/<name>")` returns the class."""
    DATASET_TYPE_DICT: Dict[str, type] = {}

    def __init__(self, tokenizer: Optional[Callable], resolution: int = 512, use_crop: bool = True, **kwargs):
        self.tokenizer, self.size, self.use_crop = tokenizer, int(resolution), bool(use_crop)

    @classmethod
    def register_cls(cls, name: str):
        Dataset.DATASET_TYPE_DICT["process/" + name] = cls

    @staticmethod
    def from_name(name: str):
        try:
            return Dataset.DATASET_TYPE_DICT[name]
        except KeyError:
            raise KeyError(f"unknown data set {name!r}; registered: {sorted(Dataset.DATASET_TYPE_DICT)}") from None

    @staticmethod
    def control_channel() -> int:
        return 3

    @staticmethod
    def cat_input(image, target: torch.Tensor, guide: torch.Tensor):
        """validation strip (reference process/base.py:26-38): [training target | guide | generated image], each at the
        generated image's size; `target` / `guide` are [1,3,h,w] in [-1,1]"""
        from PIL import Image

        def panel(t):
            a = ((t[0].detach().float().cpu() + 1.0) * 127.5).permute(1, 2, 0).numpy().clip(0, 255).astype(np.uint8)
            return Image.fromarray(a).convert("RGB").resize(image.size)

        W, H = image.size
        strip = Image.new("RGB", (3 * W, H), (0, 0, 0))
        for i, im in enumerate((panel(target), panel(guide), image)):
            strip.paste(im, (i * W, 0))
        return strip

    # ---- shared item pipeline; subclasses provide `_record` (and may override `_pre_draw` / `_guide_array`)
    def _record(self, index: int) -> Tuple["Image.Image", Optional["Image.Image"], str]:      # noqa: F821
        raise NotImplementedError

    def _pre_draw(self, index: int):
        return None

    def _guide_array(self, img_u8: np.ndarray, guide_img) -> np.ndarray:
        """-> uint8 [h,w] or [h,w,3]"""
        return np.asarray(guide_img)

    def _crop_box(self, w: int, h: int):
        x0 = _draw(0, w - self.size) if w > self.size else 0
        y0 = _draw(0, h - self.size) if h > self.size else 0
        return (x0, y0, x0 + self.size, y0 + self.size)

    def __getitem__(self, index: int):
        self._state = self._pre_draw(index)
        img, guide_img, text = self._record(index)
        if self.use_crop:
            box = self._crop_box(*img.size)
            img = img.crop(box)
            if guide_img is not None:
                guide_img = guide_img.crop(box)
        img_u8 = np.asarray(img)
        g = self._guide_array(img_u8, guide_img).astype(np.float32) / 127.5 - 1.0
        g = torch.from_numpy(np.ascontiguousarray(g))
        g = g[None].repeat(3, 1, 1) if g.dim() == 2 else g.permute(2, 0, 1)
        out = {"pixel_values": torch.from_numpy(img_u8.astype(np.float32)).permute(2, 0, 1) / 127.5 - 1.0, "guide_values": g.float().contiguous()}
        if self.tokenizer is not None:
            out["input_ids"] = self.tokenizer([text])[0]
        return out


class DiffusionDBCanny(Dataset):
    """reference process/diffusiondb_canny.py:11-48: random crop, Canny with two thresholds drawn in [1, 255) per item"""

    def __init__(self, tokenizer, resolution=512, use_crop=True, rows=None, **kwargs):
        super().__init__(tokenizer, resolution, use_crop)
        if rows is None:
            from datasets import load_dataset
            rows = load_dataset("poloclub/diffusiondb", "2m_random_1k")["train"]
        self.rows = rows

    def __len__(self):
        return len(self.rows)

    def _record(self, index):
        item = self.rows[int(index)]
        return item["image"].convert("RGB"), None, item["prompt"]

    def _guide_array(self, img_u8, guide_img):
        low, high = _draw(1, 255), _draw(1, 255)                         # low first
        return canny(img_u8, min(low, high), max(low, high))


class _FolderPairs(Dataset):
    """records = JSON lines {"image", "guide"?, "text"} with paths relative to `root`"""
    DEFAULT_ROOT, DEFAULT_INDEX = ".", "prompt.jsonl"

    def __init__(self, tokenizer, resolution=512, use_crop=True, root=None, index=None, **kwargs):
        super().__init__(tokenizer, resolution, use_crop)
        self.root = root if root is not None else self.DEFAULT_ROOT
        self.items = _read_jsonl(index if index is not None else os.path.join(self.root, self.DEFAULT_INDEX))

    def __len__(self):
        return len(self.items)


class MPIIPose(_FolderPairs):
    """reference process/mpii_pose.py:10-46: colour pose rendering as the guide, same crop for both"""
    DEFAULT_ROOT, DEFAULT_INDEX = "./data/mpii", "prompt.jsonl"

    def _record(self, index):
        from PIL import Image
        item = self.items[int(index)]
        return (Image.open(os.path.join(self.root, item["image"])).convert("RGB"),
                Image.open(os.path.join(self.root, item["guide"])).convert("RGB"), item["text"])


class DanbooruSketch(_FolderPairs):
    """reference process/danbooru_sketch.py:9-76: the guide is the grey sketch of one of three styles, drawn BEFORE the crop; the
    sketch lives beside the image under `danbooru-2020-512-<style>`"""
    DEFAULT_ROOT, DEFAULT_INDEX = "./data", "danbooru-2020-512-prompt.jsonl"
    SKETCH_STYLES = ("illyasviel", "erika", "infor")

    def _pre_draw(self, index):
        return self.SKETCH_STYLES[_draw(0, len(self.SKETCH_STYLES))]

    def _record(self, index):
        from PIL import Image
        item = self.items[int(index)]
        path = os.path.join(self.root, item["image"])
        return (Image.open(path).convert("RGB"),
                Image.open(path.replace("danbooru-2020-512", "danbooru-2020-512-" + self._state)).convert("L"), item["text"])


DiffusionDBCanny.register_cls("diffusiondb_canny")
MPIIPose.register_cls("mpii_pose")
DanbooruSketch.register_cls("danbooru_sketch")
