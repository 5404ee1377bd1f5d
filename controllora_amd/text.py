"""Frozen CLIP text encoder + tokenizer glue (reference train_text_to_image_control_lora.py:395-402, 768;
SURVEY.md section 8f rank 4: 13.3 GFLOP, forward-only, outside the hot path).  `transformers` ships in the
image, so a checkpoint directory in the SD-1.5 layout (`text_encoder/`, `tokenizer/`) loads as is; offline there
are no weights / vocab files, so the fallbacks are a seeded random-init CLIPTextModel of the SD-1.5 shape and a
deterministic hashing tokenizer (start token, hashed word ids, end-token padding to 77)."""
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


def load_text_encoder(root: str, device, seed: int = 0, small: bool = False):
    from transformers import CLIPTextConfig, CLIPTextModel
    path = os.path.join(root, "text_encoder")
    if os.path.isdir(path):
        model = CLIPTextModel.from_pretrained(path)
    else:
        cfg = CLIPTextConfig(vocab_size=VOCAB, hidden_size=64 if small else 768, intermediate_size=128 if small else 3072,
                             num_hidden_layers=2 if small else 12, num_attention_heads=4 if small else 12,
                             max_position_embeddings=CTX_LEN, hidden_act="quick_gelu", projection_dim=64 if small else 768)
        torch.manual_seed(seed)
        model = CLIPTextModel(cfg)
    return model.to(device=device, dtype=torch.float16).eval().requires_grad_(False)
