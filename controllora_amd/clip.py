"""Frozen CLIP text encoder on the gfx950 kernels: what the reference calls every train step as
`encoder_hidden_states = text_encoder(batch["input_ids"])[0]` (train_text_to_image_control_lora.py:395-402, 768) and the apps
call through the pipeline's prompt encoding (apps/gradio_canny2image.py:83-89).

Same parameters and key names as `transformers.CLIPTextModel` (with or without the `text_model.` prefix that older
checkpoints / the SD-1.5 `text_encoder/` folder carry), forward only, fp16 activations:

  token + position embedding (a gather: device memory plumbing) ->
  12 x [ LayerNorm -> ONE fused q|k|v GEMM (bias in the epilogue) -> causal flash attention (clora_attn_fwd_causal_f16) ->
         out-projection GEMM (+bias +residual) -> LayerNorm -> fc1 GEMM (+bias) -> quick_gelu -> fc2 GEMM (+bias +residual) ] ->
  final LayerNorm.

Oracle for the parity tests: the stock `transformers.CLIPTextModel` in fp32 on the CPU with the same weights
(tests/clip_cases.py) -- the real upstream implementation, importable in this image, so this boundary is pinned.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
import torch.nn as nn

from . import kernels as K
from . import ops
from .unet import LayerNorm, Linear, _frozen

f16, f32 = torch.float16, torch.float32

SD15_CLIP = dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                 max_position_embeddings=77)


class _Embeddings(nn.Module):
    def __init__(self, vocab, hidden, max_pos):
        super().__init__()
        self.token_embedding = nn.Embedding(vocab, hidden).to(f16).requires_grad_(False)
        self.position_embedding = nn.Embedding(max_pos, hidden).to(f16).requires_grad_(False)


class _SelfAttention(nn.Module):
    def __init__(self, hidden, heads):
        super().__init__()
        self.heads, self.dim_head = heads, hidden // heads
        self.q_proj, self.k_proj, self.v_proj, self.out_proj = (Linear(hidden, hidden) for _ in range(4))
        self._fused, self._fused_key = None, None

    def fused_pack(self):
        """one [3C, C] operand (and bias) for q | k | v, rebuilt when a weight changes"""
        mods = (self.q_proj, self.k_proj, self.v_proj)
        key = tuple(k for m in mods for k in m._key())
        if self._fused is None or self._fused_key != key:
            self._fused = ops.LinearPack(torch.cat([m.weight for m in mods], 0), torch.cat([m.bias for m in mods], 0))
            self._fused_key = key
        return self._fused

    def forward(self, h, B, N, residual):
        C_ = self.heads * self.dim_head
        qkv = ops.frozen_linear(h, self.fused_pack())                              # [B*N, 3C]
        a = K.attn_fwd_causal(qkv[:, :C_], qkv[:, C_:2 * C_], qkv[:, 2 * C_:], B, self.heads, N, self.dim_head,
                              self.dim_head ** -0.5)
        return self.out_proj(a, residual)


class _MLP(nn.Module):
    def __init__(self, hidden, inter):
        super().__init__()
        self.fc1, self.fc2 = Linear(hidden, inter), Linear(inter, hidden)

    def forward(self, h, residual):
        return self.fc2(K.quick_gelu(self.fc1(h)), residual)


class _Layer(nn.Module):
    def __init__(self, hidden, inter, heads):
        super().__init__()
        self.self_attn = _SelfAttention(hidden, heads)
        self.layer_norm1 = LayerNorm(hidden)
        self.mlp = _MLP(hidden, inter)
        self.layer_norm2 = LayerNorm(hidden)

    def forward(self, x, B, N):
        x = self.self_attn(self.layer_norm1(x), B, N, x)
        return self.mlp(self.layer_norm2(x), x)


class _Encoder(nn.Module):
    def __init__(self, hidden, inter, heads, layers):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(hidden, inter, heads) for _ in range(layers)])


class CLIPTextModel(nn.Module):
    """`model(input_ids)[0]` -> last hidden state [B, N, hidden] fp16, like `transformers.CLIPTextModel` (the pooled output
    is not produced: nothing on this path reads it)."""

    def __init__(self, vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                 max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, **unused):
        super().__init__()
        if hidden_act != "quick_gelu":
            raise NotImplementedError(f"hidden_act {hidden_act!r}: the SD-1.5 text encoder uses quick_gelu")
        if abs(layer_norm_eps - 1e-5) > 1e-12:
            raise NotImplementedError("layer_norm_eps other than 1e-5")
        if (hidden_size // num_attention_heads) > 64 or (hidden_size // num_attention_heads) % 8:
            raise NotImplementedError("head dim must be a multiple of 8 and <= 64 (clora_attn_fwd_causal_f16)")
        self.config = SimpleNamespace(vocab_size=vocab_size, hidden_size=hidden_size, intermediate_size=intermediate_size,
                                      num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads,
                                      max_position_embeddings=max_position_embeddings)
        self.embeddings = _Embeddings(vocab_size, hidden_size, max_position_embeddings)
        self.encoder = _Encoder(hidden_size, intermediate_size, num_attention_heads, num_hidden_layers)
        self.final_layer_norm = LayerNorm(hidden_size)

    @torch.no_grad()
    def forward(self, input_ids, **unused):
        B, N = input_ids.shape
        C_ = self.config.hidden_size
        tok = self.embeddings.token_embedding.weight[input_ids.reshape(-1)]                      # gather [B*N, C]
        pos = self.embeddings.position_embedding.weight[:N].repeat(B, 1)                         # [B*N, C]
        x = K.add(tok.contiguous(), pos)
        for layer in self.encoder.layers:
            x = layer(x, B, N)
        return (self.final_layer_norm(x).reshape(B, N, C_),)

    def load_state_dict(self, state_dict, strict=True, **kw):
        """accepts the keys of either transformers layout (`text_model.` prefix or not; `position_ids` buffers are ignored)"""
        sd = {}
        for k, v in state_dict.items():
            k = k[len("text_model."):] if k.startswith("text_model.") else k
            if k.endswith("position_ids"):
                continue
            sd[k] = v
        return super().load_state_dict(sd, strict=strict, **kw)


def init_random_(model: CLIPTextModel, seed: int = 0) -> None:
    """seeded synthetic weights (no checkpoint is available offline): N(0, 0.02) matrices / embeddings, zero biases, unit norms"""
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if p.ndim >= 2:
                p.copy_((torch.randn(p.shape, generator=g, dtype=f32) * 0.02).to(p.dtype))
            elif name.endswith("bias"):
                p.zero_()
            else:
                p.fill_(1.0)
