"""Frozen CLIP text encoder + tokenizer glue (reference train_text_to_image_control_lora.py:395-402, 768;
SURVEY.md section 8f rank 4: 13.3 GFLOP, forward-only, outside the hot path).  The encoder itself runs on the gfx950
kernels (controllora_amd/clip.py); a checkpoint directory in the SD-1.5 layout (`text_encoder/`, `tokenizer/`) loads as
is (the tokenizer through `transformers.CLIPTokenizer`); offline there are no weights / vocab files, so the stand-ins
are a seeded random-init model of the SD-1.5 shape and a deterministic hashing tokenizer (start token, hashed word
ids, end-token padding to 77)."""
from __future__ import annotations

import os
import zlib

import torch

CTX_LEN, VOCAB, BOS, EOS = 77, 49408, 49406, 49407


class HashTokenizer:
    model_max_length = CTX_LEN

    def __call__(self, captions, **unused):
        ids = torch.full((len(captions), CTX_LEN), EOS, dtype=torch.int64)
        for i, c in enumerate(captions):
            words = [BOS] + [zlib.crc32(w.encode()) % (VOCAB - 2) for w in str(c).lower().split()][:CTX_LEN - 2]
            ids[i, :len(words)] = torch.tensor(words)
        return ids


def load_tokenizer(root: str):
    path = os.path.join(root, "tokenizer")
    if os.path.isdir(path):
        from transformers import CLIPTokenizer
        tok = CLIPTokenizer.from_pretrained(path)
        return lambda caps: tok(list(caps), max_length=tok.model_max_length, padding="max_length", truncation=True,
                                return_tensors="pt").input_ids
    return HashTokenizer()


def _read_text_encoder_weights(folder: str):
    """transformers file names of a `text_encoder/` folder (model.safetensors / pytorch_model.bin)"""
    safe, pt = os.path.join(folder, "model.safetensors"), os.path.join(folder, "pytorch_model.bin")
    if os.path.exists(safe):
        from safetensors.torch import load_file
        return load_file(safe)
    if os.path.exists(pt):
        return torch.load(pt, map_location="cpu", weights_only=True)
    raise FileNotFoundError(f"no model.safetensors / pytorch_model.bin under {folder}")


def load_text_encoder(root: str, device, seed: int = 0, small: bool = False):
    """The frozen CLIP text encoder on the gfx950 kernels (controllora_amd/clip.py: same parameters / key names as
    `transformers.CLIPTextModel`).  A checkpoint directory in the SD-1.5 layout (`text_encoder/config.json` + weights) loads as is;
    offline the stand-in is a seeded random-init model of the SD-1.5 shape (`small`: a 2-layer test shape)."""
    import json
    from . import clip
    path = os.path.join(root, "text_encoder")
    if os.path.isdir(path):
        cfg = json.load(open(os.path.join(path, "config.json")))
        keep = ("vocab_size", "hidden_size", "intermediate_size", "num_hidden_layers", "num_attention_heads",
                "max_position_embeddings", "hidden_act", "layer_norm_eps")
        model = clip.CLIPTextModel(**{k: v for k, v in cfg.items() if k in keep})
        model.load_state_dict(_read_text_encoder_weights(path), strict=True)
    else:
        model = clip.CLIPTextModel(vocab_size=VOCAB, hidden_size=64 if small else 768, intermediate_size=128 if small else 3072,
                                   num_hidden_layers=2 if small else 12, num_attention_heads=4 if small else 12,
                                   max_position_embeddings=CTX_LEN)
        clip.init_random_(model, seed)
    return model.to(device).eval().requires_grad_(False)
