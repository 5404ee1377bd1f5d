"""Host-side SD-1.5 ``UNet2DConditionModel`` for the MI355X path.

Mirrors the module tree / state-dict keys / attention-processor protocol of the diffusers UNet the
reference loads (``train_text_to_image_control_lora.py:407-409``), so diffusers
format checkpoints load unchanged and ``unet.set_attn_processor({...})`` works exactly like upstream --
but the forward/backward runs on the hand-written gfx950 kernels (``controllora_amd.ops``):

* activations are fp16 NHWC ("tokens x channels") end to end: the NCHW<->token permutes of
  Transformer2DModel and head_to_batch_dim disappear, 3x3 convs are implicit GEMMs;
* weights are frozen fp16; each layer keeps a K-contiguous packed copy for the forward GEMM and a
  second one for the dgrad GEMM (no wgrad is ever computed for the 860 M frozen parameters);
* GroupNorm+SiLU, bias, time-embedding add, residual adds and the rank-r adapter updates are fused
  into the producing kernels' epilogues.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import kernels as K
from . import ops

f16, f32 = torch.float16, torch.float32

import os as _os
CAT_IN_PLACE = _os.environ.get("CLORA_CAT_IN_PLACE", "1") != "0"      # "0": materialise cat([x, skip]) with copy launches (round-5 path, A/B)

SD15_CONFIG = dict(
    in_channels=4, out_channels=4, block_out_channels=(320, 640, 1280, 1280), layers_per_block=2,
    down_block_types=("CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "CrossAttnDownBlock2D", "DownBlock2D"),
    up_block_types=("UpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D", "CrossAttnUpBlock2D"),
    attention_head_dim=8, cross_attention_dim=768, norm_num_groups=32, norm_eps=1e-5)


def _frozen(t: torch.Tensor) -> nn.Parameter:
    return nn.Parameter(t, requires_grad=False)


class _Packed(nn.Module):
    """A frozen layer whose kernel-ready operands are built lazily and rebuilt if the weights change."""

    def __init__(self):
        super().__init__()
        self._pack = None
        self._pack_key = None

    def _key(self):
        return tuple((p.data_ptr(), p._version, str(p.device)) for p in self.parameters(recurse=False))

    def pack(self):
        k = self._key()
        if self._pack is None or self._pack_key != k:
            self._pack, self._pack_key = self._build(), k
        return self._pack


class Conv3x3(_Packed):
    def __init__(self, cin, cout, stride=1, pad=1, upsample=False, need_dgrad=True, asym_pad=False):
        super().__init__()
        self.weight = _frozen(torch.empty(cout, cin, 3, 3, dtype=f16))
        self.bias = _frozen(torch.empty(cout, dtype=f16))
        self.cfg = dict(stride=stride, pad=pad, upsample=upsample, need_dgrad=need_dgrad, asym_pad=asym_pad)

    def _build(self):
        return ops.ConvPack(self.weight, self.bias, **self.cfg)

    def forward(self, x, B, H, W, residual=None, rowadd=None, defer_out=False, defer_dx=False):
        return ops.frozen_conv3x3(x, self.pack(), B, H, W, residual, rowadd, defer_out, defer_dx)


class Conv1x1(_Packed):
    """state-dict layout of a Conv2d(k=1) ([Co,Ci,1,1]); runs as a plain GEMM on NHWC tokens."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = _frozen(torch.empty(cout, cin, 1, 1, dtype=f16))
        self.bias = _frozen(torch.empty(cout, dtype=f16))

    def _build(self):
        return ops.LinearPack(self.weight, self.bias)

    def forward(self, x, residual=None, defer_out=False, defer_dx=False, ln=None, trunk=False):
        return ops.frozen_linear(x, self.pack(), residual, defer_out=defer_out, defer_dx=defer_dx, ln=ln, trunk=trunk)


class Linear(_Packed):
    def __init__(self, cin, cout, bias=True):
        super().__init__()
        self.weight = _frozen(torch.empty(cout, cin, dtype=f16))
        self.bias = _frozen(torch.empty(cout, dtype=f16)) if bias else None

    def _build(self):
        return ops.LinearPack(self.weight, self.bias)

    def forward(self, x, residual=None):
        return ops.frozen_linear(x, self.pack(), residual)


class _Norm(_Packed):
    def __init__(self, c):
        super().__init__()
        self.weight = _frozen(torch.empty(c, dtype=f16))
        self.bias = _frozen(torch.empty(c, dtype=f16))

    def _build(self):
        return self.weight.detach().to(f32).contiguous(), self.bias.detach().to(f32).contiguous()


class GroupNorm(_Norm):
    def __init__(self, groups, c, eps):
        super().__init__(c)
        self.groups, self.eps = groups, eps

    def forward(self, x, silu):
        g, b = self.pack()
        return ops.group_norm(x, g, b, self.groups, self.eps, silu)

    def fork(self, x, silu):
        """-> (norm(x), x'): route the branch that bypasses the norm through x' (gradient add fused into the backward)"""
        g, b = self.pack()
        return ops.group_norm_fork(x, g, b, self.groups, self.eps, silu)

    def fork_cat(self, x, skip, silu):
        """-> (norm(cat(x, skip)), cat(x, skip)) with the concatenation read in place by the norm kernels (ops._GroupNormCatFn)"""
        g, b = self.pack()
        return ops.group_norm_cat(x, skip, g, b, self.groups, self.eps, silu)


class LayerNorm(_Norm):
    def forward(self, x, pre=None):
        g, b = self.pack()
        return ops.layer_norm(x, g, b, 1e-5, pre)

    def fork(self, x, pre=None):
        g, b = self.pack()
        return ops.layer_norm_fork(x, g, b, 1e-5, pre)

    def slot(self):
        """this norm offered to the GEMM that produces its input (kernels.LayerNormSlot)"""
        g, b = self.pack()
        return K.LayerNormSlot(g, b, 1e-5)


# ------------------------------------------------------------------------------------------------ attention
class CrossAttnProcessor:
    """Plain attention without adapters (what a bare UNet runs)."""

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, residual=None):
        return attn.plain_attention(hidden_states, encoder_hidden_states, residual)


class CrossAttention(nn.Module):
    """Same attribute surface the reference processors use: to_q/to_k/to_v/to_out,
    heads, scale, processor, set_processor -- plus packed fused operands for the kernels."""

    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64):
        super().__init__()
        inner = heads * dim_head
        self.is_cross = cross_attention_dim is not None
        ctx = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.heads, self.dim_head, self.scale, self.inner_dim, self.query_dim = heads, dim_head, dim_head ** -0.5, inner, query_dim
        self.to_q = Linear(query_dim, inner, bias=False)
        self.to_k = Linear(ctx, inner, bias=False)
        self.to_v = Linear(ctx, inner, bias=False)
        self.to_out = nn.ModuleList([Linear(inner, query_dim), nn.Dropout(0.0)])
        self.processor = CrossAttnProcessor()
        self._fused = None
        self._fused_key = None

    def set_processor(self, processor):
        if isinstance(getattr(self, "processor", None), nn.Module) and not isinstance(processor, nn.Module):
            self._modules.pop("processor")
        self.processor = processor

    def prepare_attention_mask(self, attention_mask, target_length, batch_size=None):
        if attention_mask is not None:
            raise NotImplementedError("attention masks are never used on this path (SURVEY.md A2)")
        return None

    def fused_packs(self):
        """self-attn: one [3C, C] operand for q|k|v; cross-attn: [C, C] for q and [2C, ctx] for k|v."""
        key = tuple((m.weight.data_ptr(), m.weight._version) for m in (self.to_q, self.to_k, self.to_v))
        if self._fused is None or self._fused_key != key:
            if self.is_cross:
                self._fused = (ops.LinearPack(self.to_q.weight, None),
                               ops.LinearPack(torch.cat([self.to_k.weight, self.to_v.weight], 0), None))
            else:
                self._fused = (ops.LinearPack(torch.cat([self.to_q.weight, self.to_k.weight, self.to_v.weight], 0), None),)
            self._fused_key = key
        return self._fused

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, residual=None, **kw):
        if residual is not None and getattr(self.processor, "fuses_residual", True) is False:
            out = self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                                 attention_mask=attention_mask, **kw)
            return ops.add(out.reshape(residual.shape), residual)
        return self.processor(self, hidden_states, encoder_hidden_states=encoder_hidden_states,
                              attention_mask=attention_mask, residual=residual, **kw)

    def attend(self, q_or_qkv, kv, B, N, Nk):
        if kv is None:
            return ops.attention_self(q_or_qkv, B, self.heads, N, self.dim_head, self.scale)
        return ops.attention_cross(q_or_qkv, kv, B, self.heads, N, Nk, self.dim_head, self.scale)

    def plain_attention(self, hidden_states, encoder_hidden_states=None, residual=None):
        B, N, C_ = hidden_states.shape
        h2 = hidden_states.reshape(B * N, C_)
        packs = self.fused_packs()
        if self.is_cross:
            e2 = encoder_hidden_states.reshape(-1, encoder_hidden_states.shape[-1])
            q = ops.frozen_linear(h2, packs[0])
            kv = ops.frozen_linear(e2, packs[1])
            a = self.attend(q, kv, B, N, encoder_hidden_states.shape[1])
        else:
            a = self.attend(ops.frozen_linear(h2, packs[0]), None, B, N, N)
        res2 = residual.reshape(B * N, C_) if residual is not None else None
        return self.to_out[0](a, res2).reshape(B, N, C_)


class _GegluLinear(Linear):
    """`proj` of upstream GEGLU: same parameters / state-dict keys as a Linear, packed for the fused activation epilogue"""

    def _build(self):
        return ops.GegluPack(self.weight, self.bias)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out):
        super().__init__()
        self.proj = _GegluLinear(dim_in, dim_out * 2)


class FeedForward(nn.Module):
    def __init__(self, dim, mult=4):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * mult), nn.Dropout(0.0), Linear(dim * mult, dim)])

    def forward(self, x, residual, defer_dx=False):
        return ops.feed_forward(x, self.net[0].proj.pack(), self.net[2].pack(), residual, defer_dx)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, cross_attention_dim):
        super().__init__()
        self.attn1 = CrossAttention(dim, None, heads, dim_head)
        self.ff = FeedForward(dim)
        self.attn2 = CrossAttention(dim, cross_attention_dim, heads, dim_head)
        self.norm1, self.norm2, self.norm3 = LayerNorm(dim), LayerNorm(dim), LayerNorm(dim)

    def forward(self, x, ehs, kw, pre_ln1=None, last=True):
        """pre_ln1: norm1(x) as the launch that produced x already wrote it (Transformer2DModel.proj_in), or None; last: the block's output
        feeds proj_out (an operand, not a residual add: no rounding remainder is kept for it, kernels.TrunkNoOut)"""
        B, N, C_ = x.shape
        grad = torch.is_grad_enabled() and x.requires_grad
        # the attention / feed-forward inputs below are LayerNorm outputs with no other consumer: their projections' dgrad GEMMs may
        # leave a split-K finish to the LayerNorm backward (ops.input_from_norm)
        # norm2 / norm3 are offered to the out-projection of the attention call in front of them (ops.next_layernorm): where one GEMM tile
        # spans the row (C = 320) that launch writes the normalised rows too and the norm's own forward launch disappears
        n, xr = self.norm1.fork(x, pre_ln1) if grad else (self.norm1(x, pre_ln1), x)
        with ops.input_from_norm(), ops.next_layernorm(self.norm2.slot()) as s2:
            x = self.attn1(n, residual=xr, **kw)
        n, xr = self.norm2.fork(x, s2.out) if grad else (self.norm2(x, s2.out), x)
        with ops.input_from_norm(), ops.next_layernorm(self.norm3.slot()) as s3:
            x = self.attn2(n, encoder_hidden_states=ehs, residual=xr, **kw)
        n, xr = self.norm3.fork(x, s3.out) if grad else (self.norm3(x, s3.out), x)
        with K.TrunkNoOut(last):
            return self.ff(n.reshape(B * N, C_), xr.reshape(B * N, C_), defer_dx=grad).reshape(B, N, C_)


class Transformer2DModel(nn.Module):
    def __init__(self, heads, dim_head, in_channels, cross_attention_dim, groups):
        super().__init__()
        inner = heads * dim_head
        self.norm = GroupNorm(groups, in_channels, 1e-6)
        self.proj_in = Conv1x1(in_channels, inner)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(inner, heads, dim_head, cross_attention_dim)])
        self.proj_out = Conv1x1(inner, in_channels)

    def forward(self, x, ehs, kw, out_to_norm=False, trunk_out=True):
        """out_to_norm: the caller feeds the result straight to a GroupNorm (see ops._FrozenLinearFn defer_out); trunk_out = False: no
        residual add will read the result (it goes into a channel concatenation / the output norm): kernels.TrunkNoOut"""
        B, N, C_ = x.shape
        grad = torch.is_grad_enabled() and x.requires_grad
        n, xr = self.norm.fork(x, False) if grad else (self.norm(x, False), x)
        s1 = self.transformer_blocks[0].norm1.slot()          # the first block's norm1, offered to proj_in's launch
        h = self.proj_in(n.reshape(B * N, C_), defer_dx=grad, ln=s1, trunk=True).reshape(B, N, -1)
        for i, blk in enumerate(self.transformer_blocks):
            h = blk(h, ehs, kw, pre_ln1=s1.out if i == 0 else None, last=i == len(self.transformer_blocks) - 1)
        with K.TrunkNoOut(not trunk_out):
            return self.proj_out(h.reshape(B * N, -1), xr.reshape(B * N, C_), defer_out=out_to_norm).reshape(B, N, C_)


# ------------------------------------------------------------------------------------------------ resnet / samplers
class ResnetBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_c, groups, eps):
        super().__init__()
        self.norm1 = GroupNorm(groups, cin, eps)
        self.conv1 = Conv3x3(cin, cout)
        self.time_emb_proj = Linear(temb_c, cout)
        self.norm2 = GroupNorm(groups, cout, eps)
        self.conv2 = Conv3x3(cout, cout)
        self.conv_shortcut = Conv1x1(cin, cout) if cin != cout else None

    def forward(self, x, temb_act, H, W, temb_proj=None, skip=None, out_to_norm=False, trunk_out=True):
        """skip: the block input is cat([x, skip], channels) (up blocks), read in place by norm1; out_to_norm: the caller feeds the
        result straight to a GroupNorm (a split-K conv2 then leaves its finish pass to that norm, ops._FrozenConvFn defer_out)"""
        B, N, _ = x.shape
        if temb_proj is not None:          # this block's column slice of the UNet's batched time-embedding projections
            t = temb_proj
        else:
            with torch.no_grad():          # the time embedding has no trainable ancestor
                t = self.time_emb_proj(temb_act)
        grad = torch.is_grad_enabled() and (x.requires_grad or (skip is not None and skip.requires_grad))
        if skip is not None:
            n, xr = self.norm1.fork_cat(x, skip, True)
        else:
            n, xr = self.norm1.fork(x, True) if grad else (self.norm1(x, True), x)
        Cin = xr.shape[-1]
        # conv1 / conv2 read GroupNorm outputs that have no other consumer (defer_dx) and conv1 feeds norm2 (defer_out)
        h = self.conv1(n.reshape(B * N, Cin), B, H, W, rowadd=t, defer_out=True, defer_dx=grad)
        Cout = h.shape[1]
        h = self.norm2(h.reshape(B, N, Cout), True).reshape(B * N, Cout)
        x2 = xr.reshape(B * N, Cin)
        sc = self.conv_shortcut(x2, trunk=True) if self.conv_shortcut is not None else x2
        with K.TrunkNoOut(not trunk_out):              # (trunk_out = False: the result goes into a channel concatenation, see Transformer2DModel)
            return self.conv2(h, B, H, W, residual=sc, defer_out=out_to_norm, defer_dx=grad).reshape(B, N, Cout)


class Downsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = Conv3x3(c, c, stride=2, pad=1)

    def forward(self, x, H, W, out_to_norm=False):
        B, N, C_ = x.shape
        return self.conv(x.reshape(B * N, C_), B, H, W, defer_out=out_to_norm).reshape(B, (H // 2) * (W // 2), C_)


class Upsample2D(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = Conv3x3(c, c, upsample=True)

    def forward(self, x, H, W, out_to_norm=False):
        B, N, C_ = x.shape
        return self.conv(x.reshape(B * N, C_), B, H, W, defer_out=out_to_norm).reshape(B, 4 * N, C_)


class CrossAttnDownBlock2D(nn.Module):
    has_attn = True

    def __init__(self, cin, cout, temb_c, layers, heads, ctx, groups, eps, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_c, groups, eps) for i in range(layers)])
        self.attentions = nn.ModuleList([Transformer2DModel(heads, cout // heads, cout, ctx, groups) for _ in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None


class DownBlock2D(nn.Module):
    has_attn = False

    def __init__(self, cin, cout, temb_c, layers, groups, eps, add_down):
        super().__init__()
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_c, groups, eps) for i in range(layers)])
        self.downsamplers = nn.ModuleList([Downsample2D(cout)]) if add_down else None


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, c, temb_c, heads, ctx, groups, eps):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(heads, c // heads, c, ctx, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_c, groups, eps), ResnetBlock2D(c, c, temb_c, groups, eps)])


class _UpBase(nn.Module):
    def __init__(self, cin, prev, cout, temb_c, layers, groups, eps, add_up):
        super().__init__()
        rs = []
        for i in range(layers):
            skip = cin if i == layers - 1 else cout
            rin = prev if i == 0 else cout
            rs.append(ResnetBlock2D(rin + skip, cout, temb_c, groups, eps))
        self.resnets = nn.ModuleList(rs)
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None


class UpBlock2D(_UpBase):
    has_attn = False


class CrossAttnUpBlock2D(_UpBase):
    has_attn = True

    def __init__(self, cin, prev, cout, temb_c, layers, heads, ctx, groups, eps, add_up):
        super().__init__(cin, prev, cout, temb_c, layers, groups, eps, add_up)
        self.attentions = nn.ModuleList([Transformer2DModel(heads, cout // heads, cout, ctx, groups) for _ in range(layers)])


class TimestepEmbedding(nn.Module):
    def __init__(self, cin, ct):
        super().__init__()
        self.linear_1, self.linear_2 = Linear(cin, ct), Linear(ct, ct)


class UNetOutput:
    def __init__(self, sample):
        self.sample = sample


class UNet2DConditionModel(nn.Module):
    def __init__(self, **cfg):
        super().__init__()
        c = dict(SD15_CONFIG)
        c.update(cfg)
        self.config = SimpleNamespace(**c)
        boc = tuple(c["block_out_channels"])
        heads, ctx, groups, eps, layers = (c["attention_head_dim"], c["cross_attention_dim"], c["norm_num_groups"],
                                           c["norm_eps"], c["layers_per_block"])
        temb_c = boc[0] * 4
        self.conv_in = Conv3x3(c["in_channels"], boc[0], need_dgrad=False)
        self.time_embedding = TimestepEmbedding(boc[0], temb_c)
        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i, t in enumerate(c["down_block_types"]):
            in_c, out_c = out_c, boc[i]
            last = i == len(boc) - 1
            if t == "CrossAttnDownBlock2D":
                self.down_blocks.append(CrossAttnDownBlock2D(in_c, out_c, temb_c, layers, heads, ctx, groups, eps, not last))
            else:
                self.down_blocks.append(DownBlock2D(in_c, out_c, temb_c, layers, groups, eps, not last))
        self.mid_block = UNetMidBlock2DCrossAttn(boc[-1], temb_c, heads, ctx, groups, eps)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_c = rev[0]
        for i, t in enumerate(c["up_block_types"]):
            prev, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            last = i == len(boc) - 1
            if t == "CrossAttnUpBlock2D":
                self.up_blocks.append(CrossAttnUpBlock2D(in_c, prev, out_c, temb_c, layers + 1, heads, ctx, groups, eps, not last))
            else:
                self.up_blocks.append(UpBlock2D(in_c, prev, out_c, temb_c, layers + 1, groups, eps, not last))
        self.conv_norm_out = GroupNorm(groups, boc[0], eps)
        self.conv_out = Conv3x3(boc[0], c["out_channels"])

    # ---- attention-processor protocol (identical naming to upstream)
    def _walk_attn(self, fn):
        def walk(name, module):
            if hasattr(module, "set_processor"):
                fn(f"{name}.processor", module)
            for sub, child in module.named_children():
                if sub != "processor":
                    walk(f"{name}.{sub}", child)
        for name, module in self.named_children():
            walk(name, module)

    @property
    def attn_processors(self) -> Dict[str, object]:
        out = {}
        self._walk_attn(lambda n, m: out.__setitem__(n, m.processor))
        return out

    def set_attn_processor(self, processor):
        count = len(self.attn_processors)
        if isinstance(processor, dict) and len(processor) != count:
            raise ValueError(f"A dict of processors was passed, but the number of processors {len(processor)} does "
                             f"not match the number of attention layers: {count}.")
        self._walk_attn(lambda n, m: m.set_processor(processor[n] if isinstance(processor, dict) else processor))

    # ---- forward
    def time_embed(self, timestep, batch, device):
        if not torch.is_tensor(timestep):
            timestep = torch.tensor([timestep], dtype=torch.long, device=device)
        timestep = timestep.reshape(-1).to(device)
        half = self.config.block_out_channels[0] // 2
        freq = getattr(self, "_temb_freq", None)
        if freq is None or freq.device != timestep.device or freq.numel() != half:
            # the frequencies never change: evaluated once with the ops the embedding used to be made of on every call
            exponent = -math.log(10000) * torch.arange(half, dtype=f32, device=device) / half
            freq = self._temb_freq = torch.exp(exponent).contiguous()
        if timestep.numel() not in (1, batch):
            timestep = timestep.expand(batch)
        t_emb = K.timestep_embedding(timestep, batch, freq)      # cat(cos, sin): flip_sin_to_cos, shift 0 -- one launch (round 6)
        with torch.no_grad():
            e = self.time_embedding.linear_1(t_emb)
            e = self.time_embedding.linear_2(K.silu(e))
            return K.silu(e)   # every resnet applies SiLU to emb before time_emb_proj

    def _temb_projections(self, temb_act):
        """All 22 `time_emb_proj` linears (M = batch rows each) as ONE GEMM against the row-concatenated weights; every
        ResnetBlock2D then reads its column slice as the conv1 row-add.  The concatenation is cached on the weights'
        versions like every other frozen pack."""
        layers = getattr(self, "_temb_layers", None)
        if layers is None:
            layers = self._temb_layers = [m.time_emb_proj for m in self.modules() if isinstance(m, ResnetBlock2D)]
        key = tuple(k for l in layers for k in l._key())
        if getattr(self, "_temb_cat_key", None) != key:
            w = torch.cat([l.weight.detach() for l in layers], 0)
            b = torch.cat([l.bias.detach() for l in layers], 0)
            self._temb_cat, self._temb_cat_key = ops.LinearPack(w, b), key
            offs, o = {}, 0
            for l in layers:
                offs[id(l)] = (o, l.weight.shape[0])
                o += l.weight.shape[0]
            self._temb_offs = offs
        with torch.no_grad():
            allp = ops.frozen_linear(temb_act, self._temb_cat)
        return {k: allp[:, o:o + n] for k, (o, n) in self._temb_offs.items()}

    def forward(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None, return_dict=True):
        kw = cross_attention_kwargs or {}
        B, Cin, H, W = sample.shape
        dev = sample.device
        temb_act = self.time_embed(timestep, B, dev)
        tp = self._temb_projections(temb_act)
        ehs = encoder_hidden_states.to(f16).contiguous()
        from . import models as _models                  # (models imports this module: resolved at call time)
        # K.TrunkLo: every residual sum of this forward continues from the un-rounded previous sum (compensated trunk, kernels.gemm;
        # by default for forwards without autograd -- the samplers --, CLORA_TRUNK_LO = always / off for A/B)
        with _models.grouped_text_kv(self, ehs, kw), K.TrunkLo(K.trunk_lo_wanted()):     # training: the text K|V projections of all sites, one launch per width
            return self._forward_blocks(sample, temb_act, tp, ehs, kw, return_dict)

    def _forward_blocks(self, sample, temb_act, tp, ehs, kw, return_dict):
        B, Cin, H, W = sample.shape
        tproj = lambda r: tp[id(r.time_emb_proj)]
        x = sample.new_zeros((B, H, W, self.conv_in.pack().Cip), dtype=f16)
        x[..., :Cin] = sample.permute(0, 2, 3, 1)
        x = self.conv_in(x.reshape(B * H * W, -1), B, H, W).reshape(B, H * W, -1)
        # out_to_norm below: the NEXT reader of the tensor is a GroupNorm kernel call (a resnet's norm1 -- also through the in-place
        # concatenation of the up path --, a transformer's norm, conv_norm_out), never a torch-native op: a split-K producer may then
        # leave its finish pass to that norm (kernels._PENDING).  The tensors in front of a downsampler conv are the exception.
        skips = [(x, H, W)]
        for blk in self.down_blocks:
            nres = len(blk.resnets)
            for j, r in enumerate(blk.resnets):
                last = j == nres - 1
                to_norm = True if blk.has_attn else not (last and blk.downsamplers is not None)
                x = r(x, temb_act, H, W, tproj(r), out_to_norm=to_norm)
                if blk.has_attn:
                    x = blk.attentions[j](x, ehs, kw, out_to_norm=not (last and blk.downsamplers is not None))
                skips.append((x, H, W))
            if blk.downsamplers is not None:
                x = blk.downsamplers[0](x, H, W, out_to_norm=True)
                H, W = H // 2, W // 2
                skips.append((x, H, W))
        x = self.mid_block.resnets[0](x, temb_act, H, W, tproj(self.mid_block.resnets[0]), out_to_norm=True)
        x = self.mid_block.attentions[0](x, ehs, kw, out_to_norm=True)
        # (trunk_out=False from here on: every block output of the up path is concatenated with a skip tensor, or normed for conv_out --
        # only a resnet in front of its own transformer is read by a residual add)
        x = self.mid_block.resnets[1](x, temb_act, H, W, tproj(self.mid_block.resnets[1]), out_to_norm=CAT_IN_PLACE, trunk_out=False)
        for blk in self.up_blocks:
            for j, r in enumerate(blk.resnets):
                s, _, _ = skips.pop()
                if CAT_IN_PLACE:
                    nxt_norm = blk.has_attn or j < len(blk.resnets) - 1 or blk.upsamplers is None
                    x = r(x, temb_act, H, W, tproj(r), skip=s, out_to_norm=nxt_norm, trunk_out=blk.has_attn)
                else:
                    x = r(ops.concat_channels(x, s), temb_act, H, W, tproj(r), out_to_norm=blk.has_attn, trunk_out=blk.has_attn)
                if blk.has_attn:
                    last = j == len(blk.resnets) - 1     # then: the upsampler conv, or conv_norm_out after the last block
                    x = blk.attentions[j](x, ehs, kw, out_to_norm=(CAT_IN_PLACE and not last) or (last and blk.upsamplers is None), trunk_out=False)
            if blk.upsamplers is not None:
                x = blk.upsamplers[0](x, H, W, out_to_norm=CAT_IN_PLACE)
                H, W = 2 * H, 2 * W
        x = self.conv_norm_out(x, True)
        y = self.conv_out(x.reshape(B * H * W, -1), B, H, W)
        Co = self.config.out_channels
        out = y.reshape(B, H, W, -1)[..., :Co].permute(0, 3, 1, 2)
        return UNetOutput(out) if return_dict else (out,)


def load_from_oracle_(unet: UNet2DConditionModel, oracle_unet: nn.Module) -> None:
    """Copy weights from the fp32 oracle UNet (same key names) -- used by tests / smoke / bench init."""
    sd = {k: v for k, v in oracle_unet.state_dict().items() if ".processor." not in k}
    own = unet.state_dict()
    missing = set(own) - set(sd)
    extra = set(sd) - set(own)
    assert not missing and not extra, (sorted(missing)[:5], sorted(extra)[:5])
    with torch.no_grad():
        for k, v in own.items():
            v.copy_(sd[k].to(v.dtype))


def init_random_(unet: UNet2DConditionModel, seed: int = 0, chunk: int = 1 << 24) -> None:
    """Seeded synthetic weights at the configured shapes, directly on the module's device (fan-in scaled
    normal for matrices/filters, zeros for biases, ones for norm scales) -- bench.py uses this because no
    SD-1.5 checkpoint is available offline."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        for name, p in unet.named_parameters():
            if ".processor." in name:
                continue
            if p.ndim >= 2:
                fan_in = p[0].numel()
                w = torch.randn(p.shape, generator=g, dtype=f32) / math.sqrt(fan_in)
                p.copy_(w.to(p.dtype))
            elif name.endswith("bias"):
                p.zero_()
            else:
                p.fill_(1.0)
