"""ControlLoRA for MI355X: same public surface as the reference ``models.py`` (classes, constructor
arguments, attribute / state-dict names, ``inject_*`` / ``skip_*`` methods, config schema -- SURVEY.md
section 8b), with the math running on the gfx950 kernels.

Reference map (file:line in /root/reference/models.py):
  LoRACrossAttnProcessor            :72-152    -> LoRACrossAttnProcessor
  ControlLoRACrossAttnProcessor     :155-287   -> ControlLoRACrossAttnProcessor
  ControlLoRACrossAttnProcessorV2   :292-431   -> ControlLoRACrossAttnProcessorV2
  ConvBlock2D / SimpleDownEncoderBlock2D :434-610 -> ConvBlock2D / SimpleDownEncoderBlock2D
  ControlLoRA                       :618-835   -> ControlLoRA

Per attention site the reference launches ~25-30 micro-kernels for the adapters and materialises the
score matrix; here a site is: (control add) -> one fused q|k|v GEMM with the rank-r updates in its epilogue
-> flash attention -> one out-projection GEMM (+ adapter + bias + residual).  Quirks C1-C9 of SURVEY.md
Appendix C are kept.  ``post_add=True`` and ``pre_loras`` / ``post_loras`` chains run the
same kernels unfused in the reference's exact order (`_generic_call`).
"""
from __future__ import annotations

import inspect
import json
import os
from dataclasses import dataclass
from typing import List, Optional, Tuple, Union

import torch
import torch.nn as nn

from . import kernels as K
from . import ops

f16, f32 = torch.float16, torch.float32


class LoRALinearLayer(nn.Module):
    """Parameter holder with upstream naming/initialisation: down ~ N(0, 1/rank), up = 0."""

    def __init__(self, in_features, out_features, rank=4):
        super().__init__()
        if rank > min(in_features, out_features):
            raise ValueError(f"LoRA rank {rank} must be less or equal than {min(in_features, out_features)}")
        self.down = nn.Linear(in_features, rank, bias=False)
        self.up = nn.Linear(rank, out_features, bias=False)
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)


def _flat2(t):
    return t.reshape(-1, t.shape[-1])


class _StockLinear:
    """`to_out[0]` of a foreign attention module seen through the packed-operand interface of unet.Linear"""

    def __init__(self, lin):
        self.lin, self._pack, self._key = lin, None, None

    def pack(self):
        b = getattr(self.lin, "bias", None)
        key = (self.lin.weight.data_ptr(), self.lin.weight._version, None if b is None else (b.data_ptr(), b._version))
        if self._pack is None or self._key != key:
            self._pack, self._key = ops.LinearPack(self.lin.weight, b), key
        return self._pack

    def __call__(self, x, residual=None):
        return ops.frozen_linear(x, self.pack(), residual)


class StockAttentionHost:
    """What the processors below need from `attn` -- packed q|k|v operands, the flash-attention entry, `to_out[0].pack()` -- built on
    top of a module that only has the stock diffusers `CrossAttention` surface the reference's processors use (`to_q / to_k / to_v /
    to_out[0] / heads / scale / prepare_attention_mask`, reference models.py:122-150): the frozen weights are packed once per
    (storage, version) and the call runs on the same kernels as with this repository's own `unet.CrossAttention`.  Self- vs
    cross-attention is what the CALL says (`encoder_hidden_states is None`), as in the reference (models.py:128-129)."""

    def __init__(self, attn):
        self.attn = attn
        self.heads = int(attn.heads)
        self.inner_dim = attn.to_q.weight.shape[0]
        self.dim_head = self.inner_dim // self.heads
        self.scale = float(getattr(attn, "scale", self.dim_head ** -0.5))
        self.to_out = [_StockLinear(attn.to_out[0])]
        self.is_cross = False
        self._fused = {}

    def prepare_attention_mask(self, attention_mask, target_length, batch_size=None):
        if attention_mask is not None:
            raise NotImplementedError("attention masks are never used on this path (SURVEY.md A2)")
        return None

    def fused_packs(self):
        a = self.attn
        key = (self.is_cross,) + tuple((m.weight.data_ptr(), m.weight._version) for m in (a.to_q, a.to_k, a.to_v))
        hit = self._fused.get(self.is_cross)
        if hit is None or hit[0] != key:
            if self.is_cross:
                packs = (ops.LinearPack(a.to_q.weight, None), ops.LinearPack(torch.cat([a.to_k.weight, a.to_v.weight], 0), None))
            else:
                packs = (ops.LinearPack(torch.cat([a.to_q.weight, a.to_k.weight, a.to_v.weight], 0), None),)
            hit = self._fused[self.is_cross] = (key, packs)
        return hit[1]

    def attend(self, q_or_qkv, kv, B, N, Nk):
        if kv is None:
            return ops.attention_self(q_or_qkv, B, self.heads, N, self.dim_head, self.scale)
        return ops.attention_cross(q_or_qkv, kv, B, self.heads, N, Nk, self.dim_head, self.scale)


def _host(attn, encoder_hidden_states):
    """this repository's `unet.CrossAttention` as it is; anything else through a StockAttentionHost kept on the module"""
    if hasattr(attn, "fused_packs"):
        return attn
    host = attn.__dict__.get("_clora_host")
    if host is None:
        host = attn.__dict__["_clora_host"] = StockAttentionHost(attn)
    host.is_cross = encoder_hidden_states is not None
    return host


# ---- inference-only cache of the cross-attention K/V projections.  k = Wk e + s Lk(e), v = Wv e + s Lv(e) depend on the
# text embedding and the (frozen, at inference) adapters only -- not on the latents or the timestep -- so a scheduler loop
# (apps/gradio_canny2image.py:85-88: 50 UNet calls per image) needs them once.  Active only inside `text_kv_cache()` and
# only with autograd off; keyed per processor on the embedding's storage / version / shape and the call's scale.
_TEXT_KV = None


class text_kv_cache:
    def __init__(self, enabled=True):
        self.enabled = enabled

    def __enter__(self):
        global _TEXT_KV
        self._prev = _TEXT_KV
        _TEXT_KV = {} if self.enabled else None
        return self

    def __exit__(self, *exc):
        global _TEXT_KV
        _TEXT_KV = self._prev
        return False


# ---- training: the text K|V projections of all cross-attention sites of one width as ONE launch (ops.TextKVGroup).  The UNet opens
# `grouped_text_kv` around its forward; `_attend` below picks its site's column block up from `_TEXT_GROUPS`.
_TEXT_GROUPS = None


class grouped_text_kv:
    """Context manager used by `unet.UNet2DConditionModel.forward`: with autograd on and a text embedding that needs no gradient
    (the reference's frozen text encoder, train...:441-445, 766-768), evaluate k | v of every cross-attention site whose processor
    runs the fused path -- grouped by hidden size -- before the first block runs.  Sites it does not cover run as before."""

    def __init__(self, unet, ehs, kw):
        self.unet, self.ehs, self.scale = unet, ehs, float((kw or {}).get("scale", 1.0))

    def __enter__(self):
        global _TEXT_GROUPS
        self._prev = _TEXT_GROUPS
        _TEXT_GROUPS = None
        if not (ops.GROUP_TEXT_KV and torch.is_grad_enabled()) or self.ehs.requires_grad:
            return self
        sites = getattr(self.unet, "_cross_sites", None)
        if sites is None:
            sites = self.unet._cross_sites = [m for m in self.unet.modules() if getattr(m, "is_cross", False) and hasattr(m, "fused_packs")]
        buckets = {}
        for attn in sites:
            p = attn.processor
            if isinstance(p, LoRACrossAttnProcessor) and not p._needs_generic_path():
                buckets.setdefault((attn.inner_dim, attn.to_k.weight.shape[1]), []).append(attn)
        e2 = _flat2(self.ehs)
        table = {}
        for (C_, ctx_dim), group in buckets.items():
            if len(group) < 2 or e2.shape[1] != ctx_dim:
                continue
            segs = []
            for attn in group:
                p = attn.processor
                for name, skipped in (("to_k_lora", p.key_states_skipped), ("to_v_lora", p.value_states_skipped)):
                    sg = p._seg(name, e2, self.scale, skipped)
                    segs.append(None if sg is None else (sg[1], sg[2], float(sg[3])))
            ranks = {sg[0].shape[0] for sg in segs if sg is not None}
            if len(ranks) > 1 or any(r > 16 for r in ranks):
                continue
            key = tuple((a.to_k.weight.data_ptr(), a.to_k.weight._version, a.to_v.weight.data_ptr(), a.to_v.weight._version) for a in group)
            cache = self.unet.__dict__.setdefault("_text_kv_packs", {})
            hit = cache.get((C_, ctx_dim))
            if hit is None or hit[0] != key:
                w = torch.cat([w_ for a in group for w_ in (a.to_k.weight.detach(), a.to_v.weight.detach())], 0)
                hit = cache[(C_, ctx_dim)] = (key, ops.LinearPack(w, None))
            g = ops.TextKVGroup(e2, hit[1], C_, segs, len(group))
            for i, attn in enumerate(group):
                table[id(attn)] = (g, i, attn.processor, self.scale)
        _TEXT_GROUPS = table or None
        return self

    def __exit__(self, *exc):
        global _TEXT_GROUPS
        _TEXT_GROUPS = self._prev
        return False


class LoRACrossAttnProcessor(nn.Module):
    fuses_residual = True
    version = 0

    def __init__(self, hidden_size, cross_attention_dim=None, rank=4, post_add=False, key_states_skipped=False,
                 value_states_skipped=False, output_states_skipped=False):
        super().__init__()
        self.hidden_size, self.cross_attention_dim, self.rank, self.post_add = hidden_size, cross_attention_dim, rank, post_add
        kv_in = hidden_size if post_add else (cross_attention_dim or hidden_size)
        self.to_q_lora = LoRALinearLayer(hidden_size, hidden_size, rank)
        if not key_states_skipped:
            self.to_k_lora = LoRALinearLayer(kv_in, hidden_size, rank)
        if not value_states_skipped:
            self.to_v_lora = LoRALinearLayer(kv_in, hidden_size, rank)
        if not output_states_skipped:
            self.to_out_lora = LoRALinearLayer(hidden_size, hidden_size, rank)
        self.key_states_skipped: bool = key_states_skipped
        self.value_states_skipped: bool = value_states_skipped
        self.output_states_skipped: bool = output_states_skipped

    def skip_key_states(self, is_skipped: bool = True):
        if is_skipped == False:  # noqa: E712  (same truthiness test as the reference)
            assert hasattr(self, "to_k_lora")
        self.key_states_skipped = is_skipped

    def skip_value_states(self, is_skipped: bool = True):
        if is_skipped == False:  # noqa: E712
            assert hasattr(self, "to_q_lora")       # quirk C4 kept
        self.value_states_skipped = is_skipped

    def skip_output_states(self, is_skipped: bool = True):
        if is_skipped == False:  # noqa: E712
            assert hasattr(self, "to_out_lora")
        self.output_states_skipped = is_skipped

    # -- shared fused pipeline ------------------------------------------------------------------
    def _seg(self, name, xa, scale, skipped=False):
        if skipped or not hasattr(self, name):
            return None
        layer = getattr(self, name)
        return (xa, layer.down.weight, layer.up.weight, scale)

    def _needs_generic_path(self):
        """post_add adapters read the projection they are added to, and chained (pre/post) adapters read earlier
        adapter outputs: those cannot ride in one GEMM epilogue.  They run the same kernels, unfused, in exactly
        the reference's order (`_generic_call`)."""
        chain = list(getattr(self, "pre_loras", [])) + list(getattr(self, "post_loras", []))
        return bool(self.post_add or chain)

    def _generic_call(self, attn, hidden_states, encoder_hidden_states, scale, residual):
        """Reference order, one kernel group per adapter (models.py:118-152, 222-287, 357-431 incl. quirks C2/C3)."""
        version = getattr(self, "version", 0)
        B, N, C_ = hidden_states.shape
        chain = list(getattr(self, "pre_loras", [])) + [self] + list(getattr(self, "post_loras", []))
        h = _flat2(hidden_states)
        if version == 2:
            for p in chain:
                if isinstance(p, ControlLoRACrossAttnProcessorV2):
                    h = p.process_control_states(h.reshape(B, N, C_), scale)
        h3 = h.reshape(B, N, C_)
        packs = attn.fused_packs()
        if attn.is_cross:
            e = _flat2(encoder_hidden_states)
            Nk = encoder_hidden_states.shape[1]
            q = ops.frozen_linear(h, packs[0])
            kv = ops.frozen_linear(e, packs[1])
            k, v = ops.split_channels(kv, C_)
        else:
            e, Nk = h, N
            q, k, v = ops.split3_channels(ops.frozen_linear(h, packs[0]), C_)
        for p in chain:
            src = q if p.post_add else h
            if version == 1 and isinstance(p, ControlLoRACrossAttnProcessor):
                if p.post_add and p.concat_hidden:
                    # reference models.py:208-218 with :236-238: the concatenation reads hidden_states, the sum reads the query:
                    # lora_in = query + scale * to_control(cat(hidden_states, control))
                    src = ops.add(src, p.process_control_states(h3, scale, term_only=True))
                else:
                    src = p.process_control_states(src.reshape(B, N, C_), scale)     # src + scale*to_control(control)
            q = ops.lora_apply(q, src, p.to_q_lora.down.weight, p.to_q_lora.up.weight, scale)
        for p in chain:
            if not p.key_states_skipped:
                k = ops.lora_apply(k, k if p.post_add else e, p.to_k_lora.down.weight, p.to_k_lora.up.weight, scale)
        for p in chain:
            if not p.value_states_skipped:
                vs = scale if (p is self or version == 0) else 1.0          # quirk C3
                v = ops.lora_apply(v, v if p.post_add else e, p.to_v_lora.down.weight, p.to_v_lora.up.weight, vs)
        a = attn.attend(q, ops.concat_channels(k, v), B, N, Nk)
        if version == 2:
            for p in chain:
                if isinstance(p, ControlLoRACrossAttnProcessorV2):
                    a = p.process_control_states(a.reshape(B, N, C_), scale, is_out=True)
        out = attn.to_out[0](a)
        for p in chain:
            if (p is self and version) or not p.output_states_skipped:       # quirk C2
                out = ops.lora_apply(out, out if p.post_add else a, p.to_out_lora.down.weight, p.to_out_lora.up.weight, scale)
        if residual is not None:
            out = ops.add(out, _flat2(residual))
        return out.reshape(B, N, C_)

    def _attend(self, attn, h2, q_in, e2, B, N, Nk, scale, out_in_fn, residual, own_out_always, t_pre=None):
        packs = attn.fused_packs()
        # t_pre: the control term's share of the q adapter's down-projection -- in rank space (ops.control_terms_rank: q_in is
        # the hidden states alone then), or, with the materialised term (q_in = (h, c)), c . D_q^T evaluated for the whole level
        # in one launch (ops.control_q_parts; only meaningful with that very c)
        if t_pre is None and isinstance(q_in, tuple) and len(q_in) == 2 and q_in[1] is getattr(self, "_control_term", None):
            t_pre = getattr(self, "_control_T", None)
        if attn.is_cross:
            q = ops.lora_proj(h2, packs[0], [self._seg("to_q_lora", q_in, scale)], t_pre=t_pre[:, :4] if t_pre is not None else None)
            grp = _TEXT_GROUPS.get(id(attn)) if (_TEXT_GROUPS is not None and torch.is_grad_enabled()) else None
            if grp is not None and grp[2] is self and grp[3] == float(scale) and grp[0].e2.data_ptr() == e2.data_ptr():
                # k | v of this site were evaluated with every other site of its width at the head of the UNet forward
                a = ops.attention_cross_grouped(q, grp[0], grp[1], B, attn.heads, N, Nk, attn.dim_head, attn.scale)
            else:
                cache = _TEXT_KV if not torch.is_grad_enabled() else None
                key = (id(self), id(attn), e2.data_ptr(), e2._version, tuple(e2.shape), float(scale)) if cache is not None else None
                kv = cache.get(key) if cache is not None else None
                if kv is None:
                    kv = ops.lora_proj(e2, packs[1], [self._seg("to_k_lora", e2, scale, self.key_states_skipped),
                                                     self._seg("to_v_lora", e2, scale, self.value_states_skipped)])
                    if cache is not None:
                        cache[key] = kv
                a = attn.attend(q, kv, B, N, Nk)
        else:
            qkv = ops.lora_proj(h2, packs[0], [self._seg("to_q_lora", q_in, scale),
                                               self._seg("to_k_lora", h2, scale, self.key_states_skipped),
                                               self._seg("to_v_lora", h2, scale, self.value_states_skipped)], t_pre=t_pre)
            a = attn.attend(qkv, None, B, N, N)
        a = out_in_fn(a)
        out_seg = self._seg("to_out_lora", a, scale, (not own_out_always) and self.output_states_skipped)
        res2 = _flat2(residual) if residual is not None else None
        return ops.lora_proj(a, attn.to_out[0].pack(), [out_seg], residual=res2)

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, scale=1.0, residual=None):
        attn = _host(attn, encoder_hidden_states)
        attn.prepare_attention_mask(attention_mask, hidden_states.shape[1])
        if self._needs_generic_path():
            return self._generic_call(attn, hidden_states, encoder_hidden_states, scale, residual)
        B, N, C_ = hidden_states.shape
        h2 = _flat2(hidden_states)
        e2 = _flat2(encoder_hidden_states) if attn.is_cross else None
        Nk = encoder_hidden_states.shape[1] if attn.is_cross else N
        out = self._attend(attn, h2, h2, e2, B, N, Nk, scale, lambda a: a, residual, own_out_always=False)
        return out.reshape(B, N, C_)


class _ControlMixin:
    def inject_pre_lora(self, lora_layer):
        self.pre_loras.append(lora_layer)

    def inject_post_lora(self, lora_layer):
        self.post_loras.append(lora_layer)

    def inject_control_states(self, control_states):
        self.control_states = control_states
        self._control_term = None          # a term precomputed for the previous control states is stale now
        self._control_T = None
        self._control_parts = None

    def inject_control_term(self, term, scale=1.0, t_q=None):
        """ControlLoRA.forward evaluates `scale * to_control(control)` for all sites of a level in one batched launch
        pair (ops.control_terms) and hands each site its slice; used when the call's `scale` matches.  t_q (fp32 [Mc, 12], columns
        0..3 valid): the term's share of the q adapter's down-projection, `term . D_q^T` (see ops.lora_proj `t_pre`)."""
        self._control_term, self._control_term_scale, self._control_T = term, float(scale), t_q

    def _control_tokens(self, hidden_states):
        """models.py:202-206 / :337-341: flatten NCHW -> [B, HW, C] once and cache on self (quirk C5).  The hint
        encoder hands out NCHW *views* of NHWC memory, so this is a zero-copy reshape here."""
        ctrl = self.control_states.to(hidden_states.dtype)
        if hidden_states.ndim == 3 and ctrl.ndim == 4:
            b, _, hh, ww = ctrl.shape
            ctrl = ctrl.permute(0, 2, 3, 1).reshape(b, hh * ww, -1)
            self.control_states = ctrl
        b1, b2 = ctrl.shape[0], hidden_states.shape[0]
        if b1 != b2 and b1 != 1:
            # 1 < control batch < UNet batch.  The reference repeat-interleaves the control batch in the concat path
            # (models.py:209-213, 343-347: c0,c0,c1,c1,...) and cannot broadcast at all in the plain path (`hidden + control`
            # raises there).  The kernels broadcast by TILING, which is the same thing only for a control batch of 1 (quirk C6,
            # the inference call pattern), so here the repeat-interleaved tensor is materialised -- concat path only, like the
            # reference; autograd sums the copies' gradients back.
            if not self.concat_hidden or b2 % b1:
                raise ValueError(f"control batch {b1} vs UNet batch {b2}: the reference (models.py:237-238 `hidden_states + "
                                 "process_control_states(...)`) broadcasts only equal batches or a control batch of 1 here")
            ctrl = ctrl.repeat_interleave(b2 // b1, dim=0)
        return ctrl.contiguous()

    def process_control_states(self, hidden_states, scale=1.0, is_out=False, term_only=False):
        """Returns hidden_states + scale * to_control[_out](control | cat(hidden, control)) -- i.e. the sum the
        reference forms right after calling its process_control_states (control_self_add is always False, C1); term_only:
        the second summand alone (the reference's own return value), for a caller that adds it to another tensor."""
        ctrl = self._control_tokens(hidden_states)
        layer = self.to_control_out if is_out else self.to_control
        # the control map's share of a concat adapter's down-projection, evaluated for the whole level by ControlLoRA.forward
        # (_batched_control_terms); valid while the map is used as it was injected (equal batches, or one map for the whole batch)
        parts = getattr(self, "_control_parts", None)
        t_ctrl = parts.get(id(layer)) if (parts and self.concat_hidden and ctrl.shape[0] in (1, hidden_states.shape[0])
                                          and ctrl.shape[0] * ctrl.shape[1] == parts["rows"]) else None
        return ops.control_add(_flat2(hidden_states), _flat2(ctrl), layer.down.weight, layer.up.weight, scale,
                               self.concat_hidden, t_ctrl=t_ctrl, term_only=term_only)


class ControlLoRACrossAttnProcessor(_ControlMixin, LoRACrossAttnProcessor):
    version = 1

    def __init__(self, hidden_size, cross_attention_dim=None, rank=4, control_rank=None, post_add=False,
                 concat_hidden=False, control_channels=None, control_self_add=True, key_states_skipped=False,
                 value_states_skipped=False, output_states_skipped=False, **kwargs):
        super().__init__(hidden_size, cross_attention_dim, rank, post_add=post_add, key_states_skipped=key_states_skipped,
                         value_states_skipped=value_states_skipped, output_states_skipped=output_states_skipped)
        control_rank = rank if control_rank is None else control_rank
        control_channels = hidden_size if control_channels is None else control_channels
        self.concat_hidden = concat_hidden
        self.control_self_add = False      # quirk C1: the reference's conditional can only yield False
        self.control_states: Optional[torch.Tensor] = None
        self.to_control = LoRALinearLayer(control_channels + (hidden_size if concat_hidden else 0), hidden_size, control_rank)
        self.pre_loras: List[LoRACrossAttnProcessor] = []
        self.post_loras: List[LoRACrossAttnProcessor] = []

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, scale=1.0, residual=None):
        assert self.control_states is not None
        attn = _host(attn, encoder_hidden_states)
        attn.prepare_attention_mask(attention_mask, hidden_states.shape[1])
        if self._needs_generic_path():
            return self._generic_call(attn, hidden_states, encoder_hidden_states, scale, residual)
        B, N, C_ = hidden_states.shape
        h2 = _flat2(hidden_states)
        e2 = _flat2(encoder_hidden_states) if attn.is_cross else None
        Nk = encoder_hidden_states.shape[1] if attn.is_cross else N
        if self.concat_hidden:
            q_in = self.process_control_states(hidden_states, scale)     # h + to_control(cat(h, ctrl)): depends on h
        else:
            # hidden + control term feeds ONLY the q adapter (models.py:237-238); Lq(h + c) = Lq(h) + Lq(c), so the sum is
            # never formed: h keeps one consumer (no autograd gradient add) and the q/k/v downs share their read of h
            b1 = self.control_states.shape[0]
            if b1 not in (1, B):               # the precomputed term would be TILED over the batch: only right for 1 (quirk C6)
                raise ValueError(f"control batch {b1} vs UNet batch {B}: the reference (models.py:237-238 `hidden_states + "
                                 "process_control_states(...)`) broadcasts only equal batches or a control batch of 1 here")
            c = getattr(self, "_control_term", None)
            tq = getattr(self, "_control_T", None)
            if c is None and tq is not None and self._control_term_scale == float(scale):
                # rank-space control term (ops.control_terms_rank): the q adapter reads h alone and gets the term's share as t_pre
                out = self._attend(attn, h2, h2, e2, B, N, Nk, scale, lambda a: a, residual, own_out_always=True, t_pre=tq)
                return out.reshape(B, N, C_)
            if c is None or self._control_term_scale != float(scale):
                ctrl = self._control_tokens(hidden_states)
                c = ops.control_term(_flat2(ctrl), self.to_control.down.weight, self.to_control.up.weight, scale, B * N)
            q_in = (h2, c)
        out = self._attend(attn, h2, q_in, e2, B, N, Nk, scale, lambda a: a, residual, own_out_always=True)  # quirk C2
        return out.reshape(B, N, C_)


class ControlLoRACrossAttnProcessorV2(_ControlMixin, LoRACrossAttnProcessor):
    version = 2

    def __init__(self, hidden_size, cross_attention_dim=None, rank=4, control_rank=None, control_channels=None, **kwargs):
        super().__init__(hidden_size, cross_attention_dim, rank, post_add=False, key_states_skipped=True,
                         value_states_skipped=True, output_states_skipped=False)
        control_rank = rank if control_rank is None else control_rank
        control_channels = hidden_size if control_channels is None else control_channels
        self.concat_hidden = True
        self.control_self_add = False
        self.control_states: Optional[torch.Tensor] = None
        self.to_control = LoRALinearLayer(hidden_size + control_channels, hidden_size, control_rank)
        self.to_control_out = LoRALinearLayer(hidden_size + control_channels, hidden_size, control_rank)
        self.pre_loras: List[LoRACrossAttnProcessor] = []
        self.post_loras: List[LoRACrossAttnProcessor] = []

    def __call__(self, attn, hidden_states, encoder_hidden_states=None, attention_mask=None, scale=1.0, residual=None):
        assert self.control_states is not None
        attn = _host(attn, encoder_hidden_states)
        attn.prepare_attention_mask(attention_mask, hidden_states.shape[1])
        if self._needs_generic_path():
            return self._generic_call(attn, hidden_states, encoder_hidden_states, scale, residual)
        B, N, C_ = hidden_states.shape
        e2 = _flat2(encoder_hidden_states) if attn.is_cross else None
        Nk = encoder_hidden_states.shape[1] if attn.is_cross else N
        hp = self.process_control_states(hidden_states, scale)       # h' replaces h everywhere (models.py:369)
        shape3 = hidden_states.shape

        def post(a):                                                  # a' = a + control_out term (models.py:415)
            return self.process_control_states(a.reshape(shape3), scale, is_out=True)

        out = self._attend(attn, hp, hp, e2, B, N, Nk, scale, post, residual, own_out_always=True)
        return out.reshape(B, N, C_)


# ------------------------------------------------------------------------------------------------ hint encoder
class _Conv2dParams(nn.Module):
    """Parameter holder with nn.Conv2d naming (weight [Co,Ci,k,k], bias) and PyTorch's default init."""

    def __init__(self, cin, cout, k):
        super().__init__()
        ref = nn.Conv2d(cin, cout, k)
        self.weight, self.bias = ref.weight, ref.bias
        self.k = k


class ConvBlock2D(nn.Module):
    """H1: SiLU(GN2(conv_k(SiLU(GN1(x)))))  -- reference models.py:512-547 with temb=None (non residual, C7)."""

    def __init__(self, *, in_channels, out_channels=None, conv_kernel_size=3, groups=32, eps=1e-6, **unused):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels, self.groups, self.eps = in_channels, out_channels, groups, eps
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = _Conv2dParams(in_channels, out_channels, conv_kernel_size)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)

    def forward(self, x, B, H, W):
        h = ops.group_norm(x, self.norm1.weight, self.norm1.bias, self.groups, self.eps, True)
        h = ops.train_conv(h.reshape(B * H * W, -1), self.conv1.weight, self.conv1.bias, B, H, W)
        return ops.group_norm(h.reshape(B, H * W, -1), self.norm2.weight, self.norm2.bias, self.groups, self.eps, True)


class _Down(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv = _Conv2dParams(c, c, 3)


class SimpleDownEncoderBlock2D(nn.Module):
    """H2: reference models.py:550-610; the downsampler is diffusers' Downsample2D(padding=0): pad (0,1,0,1) + stride 2."""

    def __init__(self, in_channels, out_channels, num_layers=1, convnet_eps=1e-6, convnet_groups=32,
                 convnet_kernel_size=3, add_downsample=True, downsample_padding=0, **unused):
        super().__init__()
        assert downsample_padding == 0
        self.convnets = nn.ModuleList([
            ConvBlock2D(in_channels=in_channels if i == 0 else out_channels, out_channels=out_channels,
                        conv_kernel_size=convnet_kernel_size, groups=convnet_groups, eps=convnet_eps)
            for i in range(num_layers)])
        self.downsamplers = nn.ModuleList([_Down(out_channels)]) if add_downsample else None

    def forward(self, x, B, H, W):
        for c in self.convnets:
            x = c(x, B, H, W)
        if self.downsamplers is not None:
            d = self.downsamplers[0].conv
            x = ops.train_conv(x.reshape(B * H * W, -1), d.weight, d.bias, B, H, W, stride=2, asym_pad=True)
            H, W = (H + 1 - 3) // 2 + 1, (W + 1 - 3) // 2 + 1
            x = x.reshape(B, H * W, -1)
        return x, H, W


@dataclass
class ControlLoRAOutput:
    control_states: Tuple[torch.Tensor, ...]

    def __getitem__(self, k):
        return self.control_states if k in (0, "control_states") else (_ for _ in ()).throw(IndexError(k))


DEFAULT_CROSS_DIMS = ([None, 768] * 5, [None, 768] * 5, [None, 768] * 5, [None, 768])
CONFIG_NAME, WEIGHTS_NAME, SAFE_WEIGHTS_NAME = "config.json", "diffusion_pytorch_model.bin", "diffusion_pytorch_model.safetensors"


class ControlLoRA(nn.Module):
    def __init__(
        self,
        in_channels: int = 3,
        down_block_types: Tuple[str] = ("SimpleDownEncoderBlock2D",) * 4,
        block_out_channels: Tuple[int] = (32, 64, 128, 256),
        layers_per_block: int = 1,
        act_fn: str = "silu",
        norm_num_groups: int = 32,
        lora_pre_down_block_types: Tuple[str] = (None,) + ("SimpleDownEncoderBlock2D",) * 3,
        lora_pre_down_layers_per_block: int = 1,
        lora_pre_conv_skipped: bool = False,
        lora_pre_conv_types: Tuple[str] = ("SimpleDownEncoderBlock2D",) * 4,
        lora_pre_conv_layers_per_block: int = 1,
        lora_pre_conv_layers_kernel_size: int = 1,
        lora_block_in_channels: Tuple[int] = (256, 256, 256, 256),
        lora_block_out_channels: Tuple[int] = (320, 640, 1280, 1280),
        lora_cross_attention_dims: Tuple[List[int]] = DEFAULT_CROSS_DIMS,
        lora_rank: int = 4,
        lora_control_rank: int = None,
        lora_post_add: bool = False,
        lora_concat_hidden: bool = False,
        lora_control_channels: Tuple[int] = (None, None, None, None),
        lora_control_self_add: bool = True,
        lora_key_states_skipped: bool = False,
        lora_value_states_skipped: bool = False,
        lora_output_states_skipped: bool = False,
        lora_control_version: int = 1,
    ):
        super().__init__()
        cfg = {k: v for k, v in locals().items() if k not in ("self", "__class__")}
        self.config = cfg                                                # register_to_config equivalent
        if act_fn not in ("silu", "swish"):
            raise ValueError(f"unsupported act_fn {act_fn}: every shipped config uses silu")
        cls = ControlLoRACrossAttnProcessorV2 if lora_control_version == 2 else ControlLoRACrossAttnProcessor
        assert lora_block_in_channels[0] == block_out_channels[-1]
        if lora_pre_conv_skipped:
            lora_control_channels = lora_block_in_channels
            lora_control_self_add = False
        g = norm_num_groups
        self.layers_per_block = layers_per_block
        self.lora_pre_down_layers_per_block = lora_pre_down_layers_per_block
        self.lora_pre_conv_layers_per_block = lora_pre_conv_layers_per_block
        self.conv_in = _Conv2dParams(in_channels, block_out_channels[0], 3)
        self.down_blocks = nn.ModuleList([])
        self.pre_lora_layers = nn.ModuleList([])
        self.lora_layers = nn.ModuleList([])
        stages, c = [], block_out_channels[0]
        for i, co in enumerate(block_out_channels):
            stages.append(SimpleDownEncoderBlock2D(c, co, num_layers=layers_per_block, convnet_groups=g,
                                                   add_downsample=i != len(block_out_channels) - 1))
            c = co
        for i in range(len(lora_pre_down_block_types)):
            if i == 0:
                self.down_blocks.append(nn.Sequential(*stages))
                cin = lora_block_in_channels[0]
            else:
                cin_prev, cin = lora_block_in_channels[i - 1], lora_block_in_channels[i]
                self.down_blocks.append(SimpleDownEncoderBlock2D(cin_prev, cin, num_layers=lora_pre_down_layers_per_block,
                                                                 convnet_groups=g, add_downsample=True))
            cc = lora_control_channels[i]
            if lora_pre_conv_skipped:
                self.pre_lora_layers.append(nn.Identity())
            else:
                self.pre_lora_layers.append(SimpleDownEncoderBlock2D(
                    cin, lora_block_out_channels[i] if cc is None else cc, num_layers=lora_pre_conv_layers_per_block,
                    convnet_groups=g, convnet_kernel_size=lora_pre_conv_layers_kernel_size, add_downsample=False))
            self.lora_layers.append(nn.ModuleList([
                cls(lora_block_out_channels[i], cross_attention_dim=cad, rank=lora_rank, control_rank=lora_control_rank,
                    post_add=lora_post_add, concat_hidden=lora_concat_hidden, control_channels=cc,
                    control_self_add=lora_control_self_add, key_states_skipped=lora_key_states_skipped,
                    value_states_skipped=lora_value_states_skipped, output_states_skipped=lora_output_states_skipped)
                for cad in lora_cross_attention_dims[i]]))

    # ---- config / checkpoint surface (diffusers ConfigMixin / ModelMixin subset)
    @classmethod
    def load_config(cls, path_or_dict, subfolder=None):
        if isinstance(path_or_dict, dict):
            return dict(path_or_dict)
        path = path_or_dict
        if subfolder:
            path = os.path.join(path, subfolder)
        if os.path.isdir(path):
            path = os.path.join(path, CONFIG_NAME)
        with open(path) as f:
            return json.load(f)

    @classmethod
    def from_config(cls, config, **kwargs):
        cfg = cls.load_config(config)
        accepted = set(inspect.signature(cls.__init__).parameters) - {"self"}
        init = {k: v for k, v in cfg.items() if k in accepted}      # tolerates _class_name / _diffusers_version
        init.update(kwargs)
        return cls(**init)

    def save_config(self, save_directory):
        os.makedirs(save_directory, exist_ok=True)
        d = {k: (list(v) if isinstance(v, tuple) else v) for k, v in self.config.items()}
        d["_class_name"], d["_diffusers_version"] = "ControlLoRA", "0.13.0.dev0"
        with open(os.path.join(save_directory, CONFIG_NAME), "w") as f:
            json.dump(d, f, indent=2, sort_keys=True)

    def save_pretrained(self, save_directory, safe_serialization=False):
        self.save_config(save_directory)
        sd = {k: v.detach().cpu().contiguous() for k, v in self.state_dict().items()}
        if safe_serialization:
            from safetensors.torch import save_file
            save_file(sd, os.path.join(save_directory, SAFE_WEIGHTS_NAME))
        else:
            torch.save(sd, os.path.join(save_directory, WEIGHTS_NAME))

    @classmethod
    def from_pretrained(cls, path, subfolder=None, **kwargs):
        root = os.path.join(path, subfolder) if subfolder else path
        model = cls.from_config(os.path.join(root, CONFIG_NAME), **kwargs)
        safe = os.path.join(root, SAFE_WEIGHTS_NAME)
        if os.path.exists(safe):
            from safetensors.torch import load_file
            sd = load_file(safe)
        else:
            sd = torch.load(os.path.join(root, WEIGHTS_NAME), map_location="cpu")
        model.load_state_dict(sd)
        return model

    # ---- forward (H4, reference models.py:810-835)
    @staticmethod
    def _inject_control_terms(procs, c):
        _batched_control_terms(procs, c)

    def forward(self, x: torch.Tensor, return_dict: bool = True) -> Union[ControlLoRAOutput, Tuple]:
        orig_dtype = x.dtype
        B, Cin, H, W = x.shape
        xin = x.new_zeros((B, H, W, (Cin + 7) // 8 * 8), dtype=f16)
        xin[..., :Cin] = x.permute(0, 2, 3, 1)
        h = ops.train_conv(xin.reshape(B * H * W, -1), self.conv_in.weight, self.conv_in.bias, B, H, W, need_dx=False)
        h = h.reshape(B, H * W, -1)
        outs = []
        for down, pre, procs in zip(self.down_blocks, self.pre_lora_layers, self.lora_layers):
            for stage in (down if isinstance(down, nn.Sequential) else [down]):
                h, H, W = stage(h, B, H, W)
            c = h
            if not isinstance(pre, nn.Identity):
                c, _, _ = pre(h, B, H, W)
            # The processors get the [B, HW, C] token tensor directly (the reference's process_control_states accepts
            # 3-D control states as they are, models.py:203): gradients from the 10 sites of a level then accumulate
            # as contiguous fp16 adds.  The RETURNED tuple keeps the reference's NCHW shape as a zero-copy view.
            for p in procs:
                p.inject_control_states(c)
            self._inject_control_terms(procs, c)
            ctrl = c.reshape(B, H, W, -1).permute(0, 3, 1, 2)
            if orig_dtype != f16:
                ctrl = ctrl.to(orig_dtype)
            outs.append(ctrl)
        if not return_dict:
            return tuple(outs)
        return ControlLoRAOutput(control_states=tuple(outs))


def _batched_control_terms(procs, c):
    """v1 sites whose control term depends on the control map only (no concat_hidden, no post_add / chained adapters,
    rank <= 16): evaluate `to_control(control)` for all of them at once (reference models.py:214-218, 10 sites/level)."""
    if ops.CONTROL_PARTS:
        # concat adapters (v2: to_control + to_control_out; v1 with concat_hidden): the control map's share of every down-projection
        # of the level in one launch (reference models.py:209-214, 343-349; 20 layers per level under mpii-pose-v2.json)
        csites = [p for p in procs if getattr(p, "concat_hidden", False) and hasattr(p, "to_control") and not p._needs_generic_path()]
        layers = [l for p in csites for l in ([p.to_control] + ([p.to_control_out] if hasattr(p, "to_control_out") else []))]
        Cc = c.shape[-1]
        if len(layers) >= 2 and all(l.down.weight.shape[0] <= 16 and l.down.weight.shape[1] > Cc for l in layers) and \
                len({l.down.weight.shape[0] for l in layers}) == 1:
            parts = ops.control_down_parts(c.reshape(-1, Cc), [l.down.weight for l in layers])
            by_layer = {id(l): t for l, t in zip(layers, parts)}
            for p in csites:
                p._control_parts = dict(by_layer, rows=c.shape[0] * c.shape[1] if c.dim() == 3 else c.shape[0])
    sites = [p for p in procs if isinstance(p, ControlLoRACrossAttnProcessor) and not p.concat_hidden
             and not p._needs_generic_path() and p.to_control.down.weight.shape[0] <= 16]
    if len(sites) < 2 or len({tuple(p.to_control.down.weight.shape) + tuple(p.to_control.up.weight.shape) for p in sites}) != 1:
        return
    tri = [(p.to_control.down.weight, p.to_control.up.weight, p.to_q_lora.down.weight) for p in sites]
    if ops.RANK_CONTROL and ops.rank_control_ok(tri):
        # the terms stay in rank space: no [M, C] tensor per site, forward or backward (ops._ControlTermsRankFn)
        for p, tq in zip(sites, ops.control_terms_rank(c.reshape(-1, c.shape[-1]), tri, 1.0)):
            p.inject_control_term(None, 1.0, tq)
        return
    terms = ops.control_terms(c.reshape(-1, c.shape[-1]), [(p.to_control.down.weight, p.to_control.up.weight) for p in sites], 1.0)
    # rank-4 q adapters ride in their projection GEMM (ops._fusable); the GEMM streams h only, so the control term's share
    # L_q(c) = c . D_q^T of every site of the level is evaluated here, in one more launch for the whole level
    tq = ops.control_q_parts(terms, [p.to_q_lora.down.weight for p in sites]) if ops.FUSE_DOWN else [None] * len(sites)
    for p, t, q in zip(sites, terms, tq):
        p.inject_control_term(t, 1.0, q)


def map_processors_to_unet(unet, control_lora) -> dict:
    """M1: the name -> processor assignment of train_text_to_image_control_lora.py:469-487
    (also apps/gradio_canny2image.py:43-63)."""
    n = len(unet.config.block_out_channels)
    pools = [list(l) for l in control_lora.lora_layers]
    procs = {}
    for name in unet.attn_processors.keys():
        if name.startswith("mid_block"):
            cid = n - 1
        elif name.startswith("up_blocks"):
            cid = n - 1 - int(name[len("up_blocks."):].split(".")[0])
        else:
            cid = int(name[len("down_blocks."):].split(".")[0])
        if pools[cid]:
            procs[name] = pools[cid].pop(0)
    return procs
