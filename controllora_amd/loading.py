"""Frozen base-model loading for the entry points: a checkpoint directory in the diffusers SD-1.5 layout
(`unet/`, `vae/`, `text_encoder/`, `tokenizer/` with `diffusion_pytorch_model.{safetensors,bin}`), as the reference
loads at train_text_to_image_control_lora.py:395-409, or -- because no weights are reachable offline -- the
pseudo-names `random:sd15` / `random:small` = seeded random weights at the SD-1.5 (or the tests' small) shapes."""
from __future__ import annotations

import json
import os

import torch

from . import unet as U
from . import vae as V

# newer diffusers releases renamed the VAE attention keys; accept both spellings
_VAE_RENAMES = {".to_q.": ".query.", ".to_k.": ".key.", ".to_v.": ".value.", ".to_out.0.": ".proj_attn."}

SMALL_UNET = dict(in_channels=4, out_channels=4, block_out_channels=(32, 64, 128, 128), layers_per_block=2,
                  attention_head_dim=4, cross_attention_dim=64, norm_num_groups=8, norm_eps=1e-5)
SMALL_VAE = dict(in_channels=3, out_channels=3, latent_channels=4, block_out_channels=(16, 32, 32, 64),
                 layers_per_block=1, norm_num_groups=8)


def _read_state_dict(folder: str):
    safe, pt = os.path.join(folder, "diffusion_pytorch_model.safetensors"), os.path.join(folder, "diffusion_pytorch_model.bin")
    if os.path.exists(safe):
        from safetensors.torch import load_file
        return load_file(safe)
    if os.path.exists(pt):
        return torch.load(pt, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no diffusion_pytorch_model.safetensors / .bin under {folder}")


def _load_into(module, sd, renames=None):
    own = module.state_dict()
    if renames:
        sd = {_rename(k, renames): v for k, v in sd.items()}
    missing, extra = sorted(set(own) - set(sd)), sorted(set(sd) - set(own))
    if missing or extra:
        raise ValueError(f"checkpoint does not match the model: missing {missing[:4]} unexpected {extra[:4]}")
    with torch.no_grad():
        for k, v in own.items():
            v.copy_(sd[k].reshape(v.shape).to(v.dtype))


def _rename(k, renames):
    for a, b in renames.items():
        k = k.replace(a, b)
    return k


def is_random(name: str) -> bool:
    return name.startswith("random:")


def load_unet(name: str, device, seed: int = 0) -> U.UNet2DConditionModel:
    if is_random(name):
        unet = U.UNet2DConditionModel(**(SMALL_UNET if name.endswith("small") else {}))
        unet.to(device)
        U.init_random_(unet, seed=seed)
        return unet
    folder = os.path.join(name, "unet")
    cfg = json.load(open(os.path.join(folder, "config.json")))
    keep = ("in_channels", "out_channels", "block_out_channels", "layers_per_block", "down_block_types", "up_block_types",
            "attention_head_dim", "cross_attention_dim", "norm_num_groups", "norm_eps")
    unet = U.UNet2DConditionModel(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if k in keep})
    _load_into(unet, {k: v for k, v in _read_state_dict(folder).items()})
    return unet.to(device)


def load_vae(name: str, device, seed: int = 1) -> V.AutoencoderKL:
    if is_random(name):
        vae = V.AutoencoderKL(**(SMALL_VAE if name.endswith("small") else V.SD15_VAE))
        V.init_random_(vae, seed=seed)
        return vae.to(device)
    folder = os.path.join(name, "vae")
    cfg = json.load(open(os.path.join(folder, "config.json")))
    keep = ("in_channels", "out_channels", "latent_channels", "block_out_channels", "layers_per_block", "norm_num_groups",
            "scaling_factor")
    vae = V.AutoencoderKL(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in cfg.items() if k in keep})
    _load_into(vae, _read_state_dict(folder), _VAE_RENAMES)
    return vae.to(device)
