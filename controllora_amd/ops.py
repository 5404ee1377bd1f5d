"""torch.autograd.Function wrappers: how the HIP kernels plug into ``loss.backward()``.

Every Function calls ``controllora_amd.kernels`` (the C ABI) in forward AND backward; frozen weights
only ever get a dgrad (no wgrad is computed for the 860 M UNet parameters), adapter / hint-encoder
parameters get fp32 gradients.  Activations are fp16, tokens x channels ([M, C], NHWC for images).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import kernels as K

f16, f32 = torch.float16, torch.float32


# ------------------------------------------------------------------------------------------------ packs
class LinearPack:
    """Frozen Linear / 1x1-conv weight prepared for the kernels: W [N,K] (forward, K-contiguous),
    Wt [K,N] (dgrad presented as a second K-contiguous operand), bias fp32."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], pad_n_to: int = 8):
        w = weight.detach().reshape(weight.shape[0], -1).to(f16)
        N, Kd = w.shape
        if Kd % 8:                                   # e.g. the VAE's 4-channel post_quant_conv: zero-padded input channels
            w = torch.cat([w, w.new_zeros(N, 8 - Kd % 8)], 1)
            Kd = w.shape[1]
        Np = (N + pad_n_to - 1) // pad_n_to * pad_n_to
        if Np != N:
            w = torch.cat([w, w.new_zeros(Np - N, Kd)], 0)
        self.N, self.K, self.N_logical = Np, Kd, N
        self.w = w.contiguous()
        self.wt = w.t().contiguous()
        self.bias = None
        if bias is not None:
            b = bias.detach().to(f32)
            self.bias = torch.cat([b, b.new_zeros(Np - N)]) if Np != N else b.contiguous()


class GegluPack:
    """`FeedForward.net[0].proj` (Linear(C, 8C), upstream GEGLU) packed for the fused activation epilogue: the forward
    operand's rows (and the bias) are regrouped as [64 a-columns | the 64 matching g-columns] per 128 (include/clora.h
    `geglu`); the dgrad operand stays in the standard layout because the epilogues read / write h = (a | g) that way."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor]):
        w = weight.detach().to(f16)
        N, Kd = w.shape
        F_ = N // 2
        assert N % 128 == 0 and Kd % 8 == 0
        idx = torch.arange(N, device=w.device).reshape(N // 128, 2, 64)
        src = torch.where(torch.arange(2, device=w.device).reshape(1, 2, 1) == 0,
                          (idx[:, :1] // 128) * 64 + idx[:, :1] % 64,
                          F_ + (idx[:, 1:] // 128) * 64 + idx[:, 1:] % 64).reshape(-1)
        self.N, self.K, self.F = N, Kd, F_
        self.w = w[src].contiguous()                 # forward operand, grouped rows
        self.wt = w.t().contiguous()                 # dgrad operand [K, 2F], standard column order
        self.bias = bias.detach().to(f32)[src].contiguous() if bias is not None else None


import os as _os
# channel-chunk-major K order of the frozen 3x3 convs (clora_conv_t.kchunk) when the channel count allows; 0 = tap-major (A/B runs)
KCHUNK = int(_os.environ.get("CLORA_KCHUNK", "64"))


def conv_k_order(w_taps_ci: torch.Tensor, kchunk: int) -> torch.Tensor:
    """[rows, 9, C] (tap-major) -> [rows, 9*C] in the K order the gather walks: tap-major for kchunk == 0, else
    ((ci / kchunk) * 9 + tap) * kchunk + ci % kchunk."""
    rows, taps, Cc = w_taps_ci.shape
    if kchunk == 0:
        return w_taps_ci.reshape(rows, taps * Cc).contiguous()
    return w_taps_ci.reshape(rows, taps, Cc // kchunk, kchunk).permute(0, 2, 1, 3).reshape(rows, taps * Cc).contiguous()


class ConvPack:
    """Frozen 3x3 conv: forward operand [Co, 9*Ci] in (ky,kx,ci) order, dgrad operand [Ci, 9*Co] in
    (ky,kx,co) order (the gather descriptor walks the taps with kmul = -1, so no flip is needed); with the channel
    count a multiple of 64 both are stored channel-chunk-major instead (conv_k_order): the 9 shifted re-reads of an input
    slab then sit next to each other in the k loop."""

    def __init__(self, weight: torch.Tensor, bias: Optional[torch.Tensor], stride=1, pad=1, upsample=False,
                 asym_pad=False, need_dgrad=True):
        w = weight.detach().to(f16)
        Co, Ci, kh, kw = w.shape
        assert kh == 3 and kw == 3
        Cip, Cop = (Ci + 7) // 8 * 8, (Co + 7) // 8 * 8
        wp = w.new_zeros(Cop, 3, 3, Cip)
        wp[:Co, :, :, :Ci] = w.permute(0, 2, 3, 1)
        self.Ci, self.Co, self.Cip, self.Cop = Ci, Co, Cip, Cop
        self.kchunk = KCHUNK if (KCHUNK and Cip % KCHUNK == 0) else 0
        self.kchunk_d = KCHUNK if (KCHUNK and Cop % KCHUNK == 0) else 0
        self.w = conv_k_order(wp.reshape(Cop, 9, Cip), self.kchunk)
        self.wd = None
        if need_dgrad:
            wd = w.new_zeros(Cip, 3, 3, Cop)
            wd[:Ci, :, :, :Co] = w.permute(1, 2, 3, 0)
            self.wd = conv_k_order(wd.reshape(Cip, 9, Cop), self.kchunk_d)
        self.bias = None
        if bias is not None:
            b = bias.detach().to(f32)
            self.bias = torch.cat([b, b.new_zeros(Cop - Co)]) if Cop != Co else b.contiguous()
        self.stride, self.pad, self.upsample, self.asym_pad = stride, pad, upsample, asym_pad


# ------------------------------------------------------------------------------------------------ frozen layers
# defer_out / defer_dx (round 6, K._PENDING): the caller states that the NEXT reader of the output (forward) / of the input gradient
# (backward: x is the output of a norm with no other consumer) is a GroupNorm / LayerNorm kernel call -- a split-K launch then leaves its
# finish pass to that call.  Never set it where a torch-native op may read the tensor first.
class _FrozenLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pack: LinearPack, residual, rowadd, rows_per_batch, defer_out=False, defer_dx=False, ln=None, trunk=False):
        M = x.shape[0]
        y = K.gemm(x, pack.w, M, pack.N, pack.K, bias=pack.bias, residual=residual, rowadd=rowadd,
                   rows_per_batch=rows_per_batch, defer=defer_out and ln is None, ln=ln, trunk=trunk)
        ctx.pack, ctx.has_res, ctx.defer_dx = pack, residual is not None, defer_dx
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        p = ctx.pack
        dx = K.gemm(dy, p.wt, dy.shape[0], p.K, p.N, defer=ctx.defer_dx) if ctx.needs_input_grad[0] else None
        return dx, None, (dy if ctx.has_res and ctx.needs_input_grad[2] else None), None, None, None, None, None, None


def frozen_linear(x, pack: LinearPack, residual=None, rowadd=None, rows_per_batch=0, defer_out=False, defer_dx=False, ln=None, trunk=False):
    """y = x W^T + b (+ rowadd[batch]) (+ residual); x [M,K] fp16.  ln (kernels.LayerNormSlot): the LayerNorm that follows y, offered to
    the launch (filled in ln.out where one tile spans the row, see kernels.gemm).  trunk: y starts a stretch of the residual trunk
    (proj_in, a resnet's shortcut conv): inside a kernels.TrunkLo window its rounding remainder is kept for the next residual add."""
    return _FrozenLinearFn.apply(x, pack, residual, rowadd, rows_per_batch, defer_out, defer_dx, ln, trunk)


class _FrozenConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, pack: ConvPack, B, H, W, residual, rowadd, defer_out=False, defer_dx=False):
        cd, Ho, Wo = K.conv_fwd_desc(H, W, pack.Cip, 3, pack.stride, pack.pad, pack.upsample, pack.asym_pad, pack.kchunk)
        M = B * Ho * Wo
        y = K.gemm(x, pack.w, M, pack.Cop, 9 * pack.Cip, conv=cd, bias=pack.bias, residual=residual, rowadd=rowadd,
                   rows_per_batch=Ho * Wo, defer=defer_out)
        ctx.pack, ctx.dims, ctx.has_res, ctx.defer_dx = pack, (B, H, W, Ho, Wo), residual is not None, defer_dx
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        p = ctx.pack
        B, H, W, Ho, Wo = ctx.dims
        dx = None
        if ctx.needs_input_grad[0]:
            Hi, Wi = (2 * H, 2 * W) if p.upsample else (H, W)
            cdd = K.conv_dgrad_desc(Ho, Wo, p.Cop, Hi, Wi, 3, p.stride, p.pad, p.asym_pad, p.kchunk_d)
            dx = K.gemm(dy, p.wd, B * Hi * Wi, p.Cip, 9 * p.Cop, conv=cdd, defer=ctx.defer_dx and not p.upsample)
            if p.upsample:
                dx = K.pool2x2_sum(dx, B, H, W, p.Cip).reshape(B * H * W, p.Cip)
        return dx, None, None, None, None, (dy if ctx.has_res and ctx.needs_input_grad[5] else None), None, None, None


def frozen_conv3x3(x, pack: ConvPack, B, H, W, residual=None, rowadd=None, defer_out=False, defer_dx=False):
    """x [B*H*W, Cip] NHWC -> [B*Ho*Wo, Cop]."""
    return _FrozenConvFn.apply(x, pack, B, H, W, residual, rowadd, defer_out, defer_dx)


# ------------------------------------------------------------------------------------------------ norms / activations
class _GroupNormFn(torch.autograd.Function):
    """fork=True additionally returns x itself (an alias): callers route the branch that bypasses the norm (residual /
    shortcut) through it, so x has this ONE consumer in the autograd graph and the two gradients are summed inside
    the backward kernel instead of by a separate autograd add."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, silu, fork):
        y, stats = K.groupnorm_fwd(x, gamma, beta, G, eps, silu)
        ctx.save_for_backward(x, gamma, beta, stats)
        ctx.cfg = (G, silu)
        ctx.affine = (gamma, beta)              # the Parameter objects (leaf tensors) when the norm is trainable
        return (y, x) if fork else y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma, beta, stats = ctx.saved_tensors
        G, silu = ctx.cfg
        into = None
        if ctx.needs_input_grad[1]:             # trainable affine (hint encoder): accumulate into .grad, no AccumulateGrad add
            g_, b_ = ctx.affine
            into = (_grad_buffer(g_), _grad_buffer(b_))
        dx, _, _ = K.groupnorm_bwd(x, dy.contiguous(), gamma, beta, stats, G, silu, grads_into=into,
                                   dres=dres.contiguous().view_as(x) if dres is not None else None)
        return dx, None, None, None, None, None, None


def group_norm(x, gamma, beta, G, eps, silu):
    """x [B, HW, C] fp16, gamma/beta fp32 (frozen buffers or trainable parameters)."""
    return _GroupNormFn.apply(x, gamma, beta, G, eps, silu, False)


class _GroupNormCatFn(torch.autograd.Function):
    """GroupNorm(cat([a, b], channels)) without the concatenation launches (upstream up blocks: `torch.cat([hidden_states,
    res_hidden_states], dim=1)` in front of every ResnetBlock2D): the kernels read the two tensors in place and write
    the concatenated input ONCE as a by-product (the block's 1x1 shortcut reads it, the backward re-reads it); the backward writes the
    two input gradients as two tensors.  Returns (norm(cat), cat): route the shortcut through the second output (see _GroupNormFn).
    Until round 6: 2 copy launches forward + 2 backward per up-block resnet (48 per train step)."""

    @staticmethod
    def forward(ctx, a, b, gamma, beta, G, eps, silu):
        y, stats, xcat = K.groupnorm_fwd(a, gamma, beta, G, eps, silu, x2=b)
        ctx.save_for_backward(xcat, gamma, beta, stats)
        ctx.cfg = (G, silu, a.shape[-1])
        return y, xcat

    @staticmethod
    def backward(ctx, dy, dres=None):
        xcat, gamma, beta, stats = ctx.saved_tensors
        G, silu, Ca = ctx.cfg
        (da, db), _, _ = K.groupnorm_bwd(xcat, dy.contiguous(), gamma, beta, stats, G, silu,
                                         dres=dres.contiguous().view_as(xcat) if dres is not None else None, split_at=Ca)
        return da, db, None, None, None, None, None


def group_norm_cat(a, b, gamma, beta, G, eps, silu):
    """-> (GroupNorm(cat(a, b)), cat(a, b)); a [B, HW, Ca], b [B, HW, Cb] fp16"""
    return _GroupNormCatFn.apply(a, b, gamma, beta, G, eps, silu)


def group_norm_fork(x, gamma, beta, G, eps, silu):
    """-> (GroupNorm(x), x): use the second output for the residual / shortcut branch (see _GroupNormFn)."""
    return _GroupNormFn.apply(x, gamma, beta, G, eps, silu, True)


class _LayerNormFn(torch.autograd.Function):
    """fork=True: also returns x (alias) for the residual branch, whose gradient the backward kernel adds into dx
    (x + f(LN(x)) is every BasicTransformerBlock sub-layer; see _GroupNormFn)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps, fork, pre=None):
        # pre: LayerNorm(x) as the launch that produced x already wrote it (kernels.LayerNormSlot): nothing to launch here, the
        # backward is this function's as always
        ctx.save_for_backward(x, gamma)
        ctx.eps = eps
        y = pre.view_as(x) if pre is not None else K.layernorm_fwd(x, gamma, beta, eps)
        return (y, x) if fork else y

    @staticmethod
    def backward(ctx, dy, dres=None):
        x, gamma = ctx.saved_tensors
        dx = K.layernorm_bwd(x, dy.contiguous(), gamma, ctx.eps, dres=dres.contiguous().view_as(x) if dres is not None else None)
        return dx, None, None, None, None, None


def layer_norm(x, gamma, beta, eps=1e-5, pre=None):
    return _LayerNormFn.apply(x, gamma, beta, eps, False, pre)


def layer_norm_fork(x, gamma, beta, eps=1e-5, pre=None):
    return _LayerNormFn.apply(x, gamma, beta, eps, True, pre)


class _GegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        ctx.save_for_backward(h)
        return K.geglu_fwd(h)

    @staticmethod
    def backward(ctx, dy):
        (h,) = ctx.saved_tensors
        return K.geglu_bwd(h, dy.contiguous())


def geglu(h):
    return _GegluFn.apply(h)


class _FeedForwardFn(torch.autograd.Function):
    """out = W2 . (a * gelu_erf(g)) + b2 (+ residual), (a | g) = W1 x + b1 -- upstream FeedForward(GEGLU)
    as TWO launches each way: the activation rides in the `proj` GEMM's epilogue and its derivative in the epilogue of the
    dgrad GEMM of `out`; h = (a | g) is what is kept for the backward (the round-1 path ran separate geglu kernels that
    re-read h: 69 + 46 MB per launch at level 0)."""

    @staticmethod
    def forward(ctx, x, pack1: "GegluPack", pack2: LinearPack, residual, defer_dx=False):
        ctx.defer_dx = defer_dx
        M = x.shape[0]
        need = ctx.needs_input_grad[0]            # inference: h = (a | g) is never written
        y, h = K.gemm(x, pack1.w, M, pack1.N, pack1.K, bias=pack1.bias, geglu=1, geglu_keep_h=need)
        out = K.gemm(y, pack2.w, M, pack2.N, pack2.K, bias=pack2.bias, residual=residual)
        ctx.packs, ctx.has_res = (pack1, pack2), residual is not None
        if need:
            ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        (h,) = ctx.saved_tensors
        p1, p2 = ctx.packs
        M = dout.shape[0]
        dx = None
        if ctx.needs_input_grad[0]:
            dh = K.gemm(dout, p2.wt, M, p2.K, p2.N, geglu=2, geglu_h=h)          # [M, 2F] = d(a | g)
            dx = K.gemm(dh, p1.wt, M, p1.K, p1.N, defer=ctx.defer_dx)            # -> LayerNorm backward (norm3)
        return dx, None, None, (dout if ctx.has_res and ctx.needs_input_grad[3] else None), None


def feed_forward(x, pack1: "GegluPack", pack2: LinearPack, residual=None, defer_dx=False):
    """defer_dx: x is a LayerNorm output with no other consumer (see _FrozenLinearFn)"""
    return _FeedForwardFn.apply(x, pack1, pack2, residual, defer_dx)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return K.add(a, b)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a, b):
    return _AddFn.apply(a, b)


class _ConcatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        ctx.ca = a.shape[-1]
        return K.concat_channels(a, b)

    @staticmethod
    def backward(ctx, dy):
        return K.split_channels(dy.contiguous(), ctx.ca)


def concat_channels(a, b):
    return _ConcatFn.apply(a, b)


class _SplitFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, ca):
        ctx.ca = ca
        return K.split_channels(y, ca)

    @staticmethod
    def backward(ctx, da, db):
        return K.concat_channels(da.contiguous(), db.contiguous()), None


def split_channels(y, ca):
    return _SplitFn.apply(y, ca)


def split3_channels(y, c):
    a, bc = split_channels(y, c)
    b, cc = split_channels(bc, c)
    return a, b, cc


# ------------------------------------------------------------------------------------------------ attention
class _AttnSelfFn(torch.autograd.Function):
    """qkv: [B*N, 3*H*D] (q | k | v column blocks, as written by the fused projection)."""

    @staticmethod
    def forward(ctx, qkv, B, H, N, D, scale):
        C_ = H * D
        o, lse = K.attn_fwd(qkv[:, :C_], qkv[:, C_:2 * C_], qkv[:, 2 * C_:], B, H, N, N, D, scale)
        ctx.save_for_backward(qkv, o, lse)
        ctx.cfg = (B, H, N, D, scale)
        return o

    @staticmethod
    def backward(ctx, dO):
        qkv, o, lse = ctx.saved_tensors
        B, H, N, D, scale = ctx.cfg
        C_ = H * D
        d = torch.empty_like(qkv)
        K.attn_bwd(qkv[:, :C_], qkv[:, C_:2 * C_], qkv[:, 2 * C_:], o, dO.contiguous(), lse, B, H, N, N, D, scale,
                   d[:, :C_], d[:, C_:2 * C_], d[:, 2 * C_:])
        return d, None, None, None, None, None


class _AttnCrossFn(torch.autograd.Function):
    """q: [B*N, H*D]; kv: [B*Nk, 2*H*D] (k | v)."""

    @staticmethod
    def forward(ctx, q, kv, B, H, N, Nk, D, scale):
        C_ = H * D
        o, lse = K.attn_fwd(q, kv[:, :C_], kv[:, C_:], B, H, N, Nk, D, scale)
        ctx.save_for_backward(q, kv, o, lse)
        ctx.cfg = (B, H, N, Nk, D, scale)
        return o

    @staticmethod
    def backward(ctx, dO):
        q, kv, o, lse = ctx.saved_tensors
        B, H, N, Nk, D, scale = ctx.cfg
        C_ = H * D
        dq, dkv = torch.empty_like(q), torch.empty_like(kv)
        K.attn_bwd(q, kv[:, :C_], kv[:, C_:], o, dO.contiguous(), lse, B, H, N, Nk, D, scale, dq, dkv[:, :C_], dkv[:, C_:])
        return dq, dkv, None, None, None, None, None, None


def attention_self(qkv, B, H, N, D, scale):
    return _AttnSelfFn.apply(qkv, B, H, N, D, scale)


def attention_cross(q, kv, B, H, N, Nk, D, scale):
    return _AttnCrossFn.apply(q, kv, B, H, N, Nk, D, scale)


# ---- grouped text K|V projections ---------------------------------------------------------------------------------------------
# The cross-attention sites' k = W_k e + s L_k(e), v = W_v e + s L_v(e) (reference models.py:128-136, 249-265) read ONLY the text
# embedding: the 16 projections of a train step (M = B * 77 rows each) do not depend on anything the UNet computes.  All sites of one
# width run as ONE GEMM over the row-concatenated [W_k1; W_v1; W_k2; ...] with one adapter per column segment (+ one multi-job
# down-projection launch), at the head of the UNet forward; every site's attention reads its column block in place.  Backward: each
# site's attention writes dK | dV into its block of one gradient buffer; when the last site of the group has run, dT of all adapters
# is one multi-job launch and the weight gradients join the deferred queue (the text embedding itself needs no gradient: the text
# encoder is frozen, reference train...:441-445).  16 GEMM + 16 down + 16 dT launches -> 3 + 3 + 3.
GROUP_TEXT_KV = _os.environ.get("CLORA_GROUP_TEXT_KV", "1") != "0"        # "0": one projection per site (round-5 path, A/B runs)


class TextKVGroup:
    def __init__(self, e2: torch.Tensor, pack: LinearPack, seg_w: int, segs, n_sites: int):
        """segs[s] = None or (down_weight [r, K], up_weight [seg_w, r], scale) for column segment s (site i owns 2i, 2i + 1)"""
        M, S = e2.shape[0], len(segs)
        assert pack.N == S * seg_w and S == 2 * n_sites
        self.e2, self.pack, self.seg_w, self.segs, self.n_sites = e2, pack, seg_w, segs, n_sites
        live = [sg for sg in segs if sg is not None]
        self.r = live[0][0].shape[0] if live else 0
        assert all(sg[0].shape[0] == self.r and sg[0].shape[1] == pack.K for sg in live) and self.r <= 16
        self.T = None
        if live:
            r = self.r
            self.T = (torch.empty if len(live) == S else torch.zeros)((M, S * r), dtype=f32, device=e2.device)
            specs = [dict(X=e2, D=sg[0].detach(), toff=s * r, R=r, X2=None, r2=0) for s, sg in enumerate(segs) if sg is not None]
            K.lora_down_multi([K.down_job(j["X"], j["D"], self.T, j["toff"], M, j["D"].shape[1], R=j["R"]) for j in _merge_down_jobs(specs)])
            pieces = [_zero_const((seg_w, r), e2.device) if sg is None else (sg[1].detach() if sg[2] == 1.0 else sg[1].detach() * sg[2])
                      for sg in segs]
            self.out = K.gemm(e2, pack.w, M, pack.N, pack.K, bias=pack.bias, lora_t=self.T, lora_u=_stack_rows(pieces), lora_seg=seg_w,
                              lora_scale=1.0)
        else:
            self.out = K.gemm(e2, pack.w, M, pack.N, pack.K, bias=pack.bias)
        self.dout, self.pending, self.done = None, set(range(n_sites)), False

    def site_kv(self, i):
        w = self.seg_w
        return self.out[:, 2 * i * w:(2 * i + 2) * w]

    def sink(self, i):
        """dK, dV views of site i inside the group's gradient buffer"""
        if self.dout is None:
            self.dout = torch.empty_like(self.out)
            # safety net: a site whose output never reached the loss leaves the group open -- close it when the pass ends
            torch.autograd.Variable._execution_engine.queue_callback(lambda: self.finalize(late=True))
        w = self.seg_w
        return self.dout[:, 2 * i * w:(2 * i + 1) * w], self.dout[:, (2 * i + 1) * w:(2 * i + 2) * w]

    def site_done(self, i):
        self.pending.discard(i)
        if not self.pending:
            self.finalize()

    def finalize(self, late=False):
        if self.done:
            return
        self.done = True
        if self.T is None or self.dout is None:
            return
        M, r, w, N = self.e2.shape[0], self.r, self.seg_w, self.pack.N
        written = [s for s, sg in enumerate(self.segs) if sg is not None and (s // 2) not in self.pending]
        if not written:
            return
        dT = (torch.empty if len(written) == len(self.segs) else torch.zeros)((M, len(self.segs) * r), dtype=f32, device=self.e2.device)
        djobs, wjobs = [], []
        for s in written:
            D, U, sc = self.segs[s]
            dys = self.dout[:, s * w:(s + 1) * w]
            djobs.append(K.down_job(dys, U.detach(), dT, s * r, M, w, ldx=N, kmajor=True, R=r, d_scale=sc))
            if U.requires_grad:
                wjobs.append(K.wgrad_job(dys, self.T, s * r, _grad_buffer(U), U.shape[1], 1, M, w, r, scale=sc, lda=N))
            if D.requires_grad:
                wjobs.append(K.wgrad_job(self.e2, dT, s * r, _grad_buffer(D), 1, D.shape[1], M, D.shape[1], r))
        K.lora_down_multi(djobs)
        if wjobs:
            if late:                                     # the end-of-backward flush may already have run
                K.lora_wgrad_multi(wjobs, self.e2.device)
            else:
                K.lora_wgrad_defer(wjobs, self.e2.device, self.dout, self.T, dT, self.e2)


class _AttnCrossGroupedFn(torch.autograd.Function):
    """_AttnCrossFn for a site of a TextKVGroup: k | v are the site's column block of the grouped projection, dK | dV go to the
    group's gradient buffer (no autograd edge to the group: its backward is the group's own, run after its last site)"""

    @staticmethod
    def forward(ctx, q, group: TextKVGroup, i, B, H, N, Nk, D, scale):
        C_ = H * D
        kv = group.site_kv(i)
        o, lse = K.attn_fwd(q, kv[:, :C_], kv[:, C_:], B, H, N, Nk, D, scale)
        ctx.save_for_backward(q, o, lse)
        ctx.cfg = (group, i, B, H, N, Nk, D, scale)
        return o

    @staticmethod
    def backward(ctx, dO):
        q, o, lse = ctx.saved_tensors
        group, i, B, H, N, Nk, D, scale = ctx.cfg
        C_ = H * D
        kv = group.site_kv(i)
        dk, dv = group.sink(i)
        dq = torch.empty_like(q)
        K.attn_bwd(q, kv[:, :C_], kv[:, C_:], o, dO.contiguous(), lse, B, H, N, Nk, D, scale, dq, dk, dv)
        group.site_done(i)
        return dq, None, None, None, None, None, None, None, None


def attention_cross_grouped(q, group: TextKVGroup, i, B, H, N, Nk, D, scale):
    return _AttnCrossGroupedFn.apply(q, group, i, B, H, N, Nk, D, scale)


# ------------------------------------------------------------------------------------------------ adapters
def _grad_buffer(p: torch.Tensor) -> torch.Tensor:
    """Adapter weight gradients are accumulated by the kernels straight into ``param.grad`` (fp32 atomics;
    with ControlLoRATrainer that is a view of the flat all-reduce buffer) instead of being returned to
    autograd: no zero-filled temporaries, no AccumulateGrad add per parameter."""
    if p.grad is None:
        p.grad = torch.zeros_like(p)
    return p.grad


_ZERO_CONST = {}


def _zero_const(shape, device):
    """a read-only all-zero fp32 matrix (the up matrix of a skipped adapter segment: v2 has no k / v adapters), made once"""
    key = (tuple(shape), str(device))
    z = _ZERO_CONST.get(key)
    if z is None:
        z = _ZERO_CONST[key] = torch.zeros(shape, dtype=f32, device=device)
    return z


def _stack_rows(ts):
    """Row-concatenate [n_i, r] fp32 matrices.  ControlLoRATrainer lays the adapter weights of a processor out
    back to back in its flat parameter buffer, so the concatenation is usually a zero-copy strided view."""
    if len(ts) == 1:
        return ts[0]
    t0 = ts[0]
    ptr, ok = t0.data_ptr(), True
    for t in ts:
        if not (t.is_contiguous() and t.dim() == 2 and t.shape[1] == t0.shape[1] and t.dtype == t0.dtype and t.data_ptr() == ptr
                and t.untyped_storage().data_ptr() == t0.untyped_storage().data_ptr()):
            ok = False
            break
        ptr += t.numel() * t.element_size()
    if ok:
        return t0.as_strided((sum(t.shape[0] for t in ts), t0.shape[1]), (t0.shape[1], 1))
    return torch.cat(ts, 0)


MERGE_DOWNS = _os.environ.get("CLORA_MERGE_DOWNS", "1") != "0"      # "0": one job per adapter (A/B runs)
_STACKED_D = {}     # inference-only cache of row-stacked down matrices (the trainer's flat buffer makes them adjacent views)


def _merge_down_jobs(specs):
    """Adapters of one projection that read the SAME input (q|k|v of a self-attention site, k|v of the text projection) become
    one job over the row-stacked down matrix [D_q; D_k; D_v] (<= 16 rows: one MFMA column tile): x is streamed once instead
    of once per adapter.  Only the first adapter of a group may carry a second input (the control term of the q adapter,
    reference models.py:237-238): the job's r2 limits that pass to its rows."""
    out = []
    for j in specs:
        a = out[-1] if (out and MERGE_DOWNS) else None
        if (a is not None and j["X2"] is None and j["X"] is a["X"] and a["toff"] + a["R"] == j["toff"] and a["R"] + j["R"] <= 16
                and j["D"].shape[1] == a["D"].shape[1] and j["D"].is_contiguous() and a["D"].is_contiguous()):
            parts = a.setdefault("parts", [a["D"]]) + [j["D"]]
            stacked = None
            if _adjacent(parts):                                       # the trainer's flat buffer: a zero-copy view
                stacked = _stack_rows(parts)
            elif not torch.is_grad_enabled():                          # inference on separate tensors: stack once, keep
                key = tuple((t.data_ptr(), t._version) for t in parts)
                hit = _STACKED_D.get(key)
                if hit is None:
                    if len(_STACKED_D) > 4096:
                        _STACKED_D.clear()
                    # the entry keeps the source tensors alive: their addresses cannot be handed to another model's
                    # adapters while the key is cached; in-place updates bump _version and miss
                    hit = _STACKED_D[key] = (torch.cat(parts, 0), tuple(parts))
                stacked = hit[0]
            if stacked is not None:
                if a["X2"] is not None and a["r2"] == 0:
                    a["r2"] = a["R"]
                a["parts"], a["D"], a["R"] = parts, stacked, a["R"] + j["R"]
                continue
        out.append(dict(j))
    return out


def _adjacent(ts):
    ptr = ts[0].data_ptr()
    for t in ts:
        if t.data_ptr() != ptr or t.untyped_storage().data_ptr() != ts[0].untyped_storage().data_ptr():
            return False
        ptr += t.numel() * t.element_size()
    return True


# ---- adapter operand packs for the in-launch down-projection (include/clora.h clora_epilogue_t.lora_dpack) -----------------
# A rank-4 adapter whose input is the projection's own input is evaluated INSIDE the projection GEMM (8 extra operand rows per
# ring stage): the separate lora_down launch and its second pass over the activation disappear.  The GEMM needs the down matrix
# (forward) / the scaled up matrix (backward: dT = dy . (s U)) as an 8-row fp16 block (hi | lo split, fp32-equivalent); the blocks
# of all adapters are refreshed by ONE launch over a device-resident job table (`repack_adapters`, called by the trainer after
# every optimizer step -- its flat AdamW kernel updates the weights behind torch's back); an in-place torch update of a
# parameter is caught by its _version at the point of use.
FUSE_DOWN = _os.environ.get("CLORA_FUSE_DOWN", "1") != "0"        # "0": separate lora_down launches everywhere (A/B runs)
FUSE_TILE_N = 320                                                 # column width of the 8-wave tiles that carry the extra rows


class _AdapterPacks:
    def __init__(self):
        self.groups = {}          # key -> dict(out, jobs, versions, srcs)
        self.table = None         # device uint8 tensor of clora_lora_pack_job_t, all groups
        self.njobs = 0
        self.max_k = 0
        self.retired = []         # tables baked into captured graphs stay alive
        self.epoch = 0            # bumped whenever a group is registered: a captured optimizer graph replays the job table of ITS epoch

    def _jobs_of(self, srcs, kmajor, scales, out, K):
        import ctypes as C
        from . import capi
        jobs = []
        for i, (D, sc) in enumerate(zip(srcs, scales)):
            if D is None:
                continue
            R = D.shape[1] if kmajor else D.shape[0]
            assert R <= 4
            jobs.append(capi.LoraPackJob(D.data_ptr(), out[8 * i:].data_ptr(), D.stride(0), R, K, int(kmajor), float(sc), 0))
        return jobs

    def _launch(self, jobs, K, device):
        import ctypes as C
        from . import capi
        arr = (capi.LoraPackJob * len(jobs))(*jobs)
        tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
        K_.lora_pack(tab, len(jobs), K)
        return tab

    def get(self, srcs, kmajor=False, scales=None):
        """srcs: the adapter matrices of the column segments of one GEMM (None = no adapter on that segment: zero block);
        -> [8 * len(srcs), K] fp16, refreshed if a source was updated in place since the last pack"""
        scales = tuple(1.0 if sc is None else float(sc) for sc in (scales or [None] * len(srcs)))
        key = (tuple((D.data_ptr(), tuple(D.shape), D.stride(0)) if D is not None else None for D in srcs), bool(kmajor), scales)
        g = self.groups.get(key)
        live = [D for D in srcs if D is not None]
        if g is None:
            D0 = live[0]
            assert all(D.dtype == f32 and D.stride(1) == 1 for D in live)
            K = D0.shape[0] if kmajor else D0.shape[1]
            out = torch.zeros((8 * len(srcs), K), dtype=f16, device=D0.device)
            jobs = self._jobs_of(srcs, kmajor, scales, out, K)
            g = self.groups[key] = dict(out=out, jobs=jobs, K=K, srcs=tuple(srcs), versions=None, tab=None)
            if self.table is not None:
                self.retired.append(self.table)                 # a captured hipGraph may still replay a launch over the old table
            self.table = None                                   # rebuilt (with the new group) at the next repack
            self.epoch += 1
        vers = tuple(D._version for D in live)
        if g["versions"] != vers:
            g["tab"] = self._launch(g["jobs"], g["K"], g["out"].device)    # first use / in-place update: this group alone
            g["versions"] = vers
        return g["out"]

    def repack_all(self):
        """one launch over every registered group (the weights changed behind torch's back: flat optimizer step)"""
        if not self.groups:
            return
        if self.table is None:
            jobs = [j for g in self.groups.values() for j in g["jobs"]]
            import ctypes as C
            from . import capi
            arr = (capi.LoraPackJob * len(jobs))(*jobs)
            dev = next(iter(self.groups.values()))["out"].device
            self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
            self.njobs, self.max_k = len(jobs), max(g["K"] for g in self.groups.values())
        K_.lora_pack(self.table, self.njobs, self.max_k)


K_ = K
ADAPTER_PACKS = _AdapterPacks()


def repack_adapters():
    """refresh every fp16 operand derived from the trainable fp32 master weights after they changed behind torch's back (the flat
    AdamW kernel, a flat-buffer checkpoint load): the adapters' in-GEMM operand blocks and the hint encoder's conv operands"""
    ADAPTER_PACKS.repack_all()
    TRAIN_CONV_PACKS.repack_all()


# ---- fp16 GEMM operands of the trainable hint-encoder convolutions (reference models.py:470, 529, 594-597, 684) ---------------
# accelerate's autocast casts every conv weight once per forward; here the cast is fused with the layout change the implicit GEMM
# wants (K.conv_weight_pack) and -- since round 6 -- done for ALL convolutions by one launch per optimizer step instead of one
# launch per convolution per forward (18 launches of 5-10 us under fill50k.json): the operands are persistent buffers hung on the
# parameter itself, refreshed (a) by `repack_adapters()` after the flat optimizer step / a flat checkpoint load, (b) at the point of
# use when torch saw an in-place update (the parameter's _version moved: torch optimizers, load_state_dict, tests).
PERSISTENT_CONV_PACKS = _os.environ.get("CLORA_PERSISTENT_CONV_PACKS", "1") != "0"     # "0": pack per call (round-5 path, A/B runs)


class _TrainConvPacks:
    def __init__(self):
        self.live = []            # weakrefs of the parameters that carry a pack

    def get(self, weight: torch.Tensor, Cip: int, need_dx: bool):
        import weakref
        key = (weight.data_ptr(), str(weight.device), int(Cip), bool(need_dx))
        st = getattr(weight, "_clora_pack", None)
        if st is None or st["key"] != key:
            Co, Ci, k, _ = weight.shape
            Cop = (Co + 7) // 8 * 8
            fwd = torch.empty((Co, k * k * Cip), dtype=f16, device=weight.device)
            dgrad = torch.empty((Cip, k * k * Cop), dtype=f16, device=weight.device) if need_dx else None
            fresh = st is None
            st = dict(key=key, fwd=fwd, dgrad=dgrad, Cip=int(Cip), version=None)
            weight._clora_pack = st
            if fresh:
                self.live.append(weakref.ref(weight))
            ADAPTER_PACKS.epoch += 1                       # a captured optimizer graph does not repack this one yet (train.step_graphed)
        if st["version"] != weight._version:
            K.conv_weight_pack(weight.detach(), st["Cip"], st["dgrad"] is not None, out=(st["fwd"], st["dgrad"]))
            st["version"] = weight._version
        return st["fwd"], st["dgrad"]

    def repack_all(self):
        jobs, alive = [], []
        for ref in self.live:
            w = ref()
            st = getattr(w, "_clora_pack", None) if w is not None else None
            if st is None:
                continue
            alive.append(ref)
            if st["key"][0] != w.data_ptr() or not w.is_cuda and capi_requires_device():
                continue                                   # moved since (module.to()): re-packed at its next use
            jobs.append(K.conv_pack_job(w.detach(), st["Cip"], st["fwd"], st["dgrad"]))
            st["version"] = w._version
        self.live = alive
        if jobs:
            K.conv_weight_pack_multi(jobs)


def capi_requires_device():
    from . import capi
    return capi.lib().require_device


TRAIN_CONV_PACKS = _TrainConvPacks()


FUSE_MIN_BLOCKS = int(_os.environ.get("CLORA_FUSE_MIN_BLOCKS", "192"))


def _fills_the_chip(M, N):
    """the 8-wave 320-column tiles run one block per CU: below ~3/4 of the 256 CUs the narrow 64x64 tiles (3 blocks per CU, 5x the
    blocks) win by more than the separate down-projection launch costs (measured r04: the 16x16 / 32x32 levels lost 0.6 ms/step)"""
    bm = 128 if M >= 32768 else 64
    return ((M + bm - 1) // bm) * (N // FUSE_TILE_N) >= FUSE_MIN_BLOCKS


SMALL_FUSE_TILES = {21: 128, 41: 128, 22: 64, 42: 64, 26: 64, 23: 64, 43: 64}     # 4-wave BK = 64 tiles that can carry the extra rows: columns
FUSE_SMALL = _os.environ.get("CLORA_FUSE_SMALL", "1") != "0"                        # "0": only the 8-wave 320-column tiles (A/B runs)


_FUSE_TILE_COLS = {**SMALL_FUSE_TILES, 51: 320, 52: 320, 54: 320, 55: 320}      # tile_cfg -> columns a segment must be a multiple of


def _fuse_plan(M, N, Kd, seg_w):
    """None: keep the separate down-projection launch; 0: fuse on the 320-column tiles, the library picks which; > 0: fuse on this
    tile -- the shape's ":x" entry of the launch table (timed WITH the extra operand rows, tools/tune_fused.py), else the plain entry
    when that is a 4-wave tile that can carry them (the deeper UNet levels, where the 320-column tiles would leave most CUs idle)"""
    if not FUSE_DOWN or Kd % 64 or seg_w % 64:
        return None
    hx = K._tuned_fused(M, N, Kd)
    if hx is not None and seg_w % _FUSE_TILE_COLS.get(hx[0], 1 << 30) == 0 and (FUSE_SMALL or hx[0] not in SMALL_FUSE_TILES):
        return hx[0]
    if seg_w % FUSE_TILE_N == 0 and _fills_the_chip(M, N):
        return 0
    hit = K._tuned(M, N, Kd, None) if FUSE_SMALL else None
    if hit is not None and hit[1] == 1 and hit[0] in SMALL_FUSE_TILES and seg_w % SMALL_FUSE_TILES[hit[0]] == 0:
        return hit[0]
    return None


def _fusable(pack, meta, ranks, seg_w, M):
    """may the adapters of this projection ride in its GEMM?  rank 4 everywhere, whole 64-deep k-steps, every adapter fed by x
    itself (index 0) plus at most one more input, and a tile that can take it (_fuse_plan) -> the plan, or None"""
    if not ranks or any(rk != 4 for rk in ranks):
        return None
    if not all(m is None or (m[0][0] == 0 and len(m[0]) <= 2) for m in meta):
        return None
    return _fuse_plan(M, pack.N, pack.K, seg_w)


# Dynamic scope set by the UNet's transformer block around an attention call whose hidden_states IS a LayerNorm output with no
# other consumer: the projection's dgrad GEMM may then leave a split-K finish to the LayerNorm backward (see _FrozenLinearFn).
_FROM_NORM = [0]


class input_from_norm:
    def __enter__(self):
        _FROM_NORM[0] += 1
        return self

    def __exit__(self, *exc):
        _FROM_NORM[0] -= 1
        return False


# Dynamic scope set by the transformer block around an attention call: the LayerNorm that FOLLOWS the call's result (norm2 after
# attn1, norm3 after attn2), offered to the out-projection GEMM -- the one lora_proj call of the processor that carries the residual.
_NEXT_LN = [None]


class next_layernorm:
    def __init__(self, slot):
        self.slot = slot

    def __enter__(self):
        self.prev, _NEXT_LN[0] = _NEXT_LN[0], self.slot
        return self.slot

    def __exit__(self, *exc):
        _NEXT_LN[0] = self.prev
        return False


class _Meta(tuple):
    """the per-segment adapter description of _LoraProjFn plus call flags (a tuple subclass: autograd passes it through untouched)"""
    defer_dx = False
    ln = None


class _LoraProjFn(torch.autograd.Function):
    """y = x W^T (+b) (+residual) + scale_s * up_s(down_s(xa_s)) on column segment s.

    One frozen GEMM over x with the rank-r updates applied in its epilogue.
    ``meta[s]`` is None (no adapter on that segment) or (xa_indices, scale); the indices select the adapter input(s)
    among ``xas`` (0 = x itself).  Several indices mean the adapter acts on the SUM of those tensors, evaluated by
    linearity as down(a) + down(b) (SURVEY.md section 8c pin 5: Lq(h + c) = Lq(h) + Lq(c)) so the sum is never
    materialised and h keeps a single consumer.  Tensor arguments are (x, residual, *xas[1:], D_0, U_0, D_1, U_1, ...)."""

    @staticmethod
    def forward(ctx, pack: LinearPack, meta, n_xa, t_pre, x, residual, *rest):
        xas = [x] + list(rest[:n_xa - 1])
        params = rest[n_xa - 1:]
        S = len(meta)
        seg_w = pack.N // S
        M = x.shape[0]
        ranks = [params[2 * i].shape[0] for i in range(len(params) // 2)]
        r = max(ranks + [1])
        full = all(m is not None for m in meta) and all(rk == r for rk in ranks)
        plan = _fusable(pack, meta, ranks, seg_w, M)
        fused = plan is not None
        T = (torch.empty if (full or fused) else torch.zeros)((M, S * r), dtype=f32, device=x.device)
        pieces, pi, info, dspecs = [], 0, [], []
        f_srcs, f_in_mask, f_in = [], 0, []           # fused: down matrices per segment, segments with a second (precomputed) part
        for s, m in enumerate(meta):
            if m is None:
                pieces.append(_zero_const((seg_w, r), x.device))
                info.append(None)
                f_srcs.append(None)
                continue
            xis, sc = m
            D, Uw = params[2 * pi], params[2 * pi + 1]
            pi += 1
            rs = D.shape[0]
            assert len(xis) <= 2
            tin = t_pre.detach() if (t_pre is not None and s == 0) else None     # rank-space share of segment 0 (the q adapter)
            if fused:
                f_srcs.append(D.detach())
                if len(xis) > 1:
                    f_in_mask |= 1 << s
                    f_in.append((s, xas[xis[1]], D.detach()))
            elif rs <= 16:                            # both inputs of a summed adapter input go through ONE job
                x2 = xas[xis[1]] if len(xis) > 1 else None          # may hold fewer rows (control batch 1 broadcast, quirk C6)
                if tin is not None:
                    x2 = None                                       # t_pre IS the second input's share (c . D_q^T): count it once
                dspecs.append(dict(X=xas[xis[0]], D=D.detach(), toff=s * r, R=rs, X2=x2, r2=0,
                                   x2_rows=x2.shape[0] if (x2 is not None and x2.shape[0] != M) else 0,
                                   T_in=tin, t_in_r=rs if tin is not None else 0))
            else:
                for n_in, xi in enumerate(xis):       # separate launches: stream order makes the accumulation safe
                    K.lora_down(xas[xi], D.detach(), T, s * r, M, D.shape[1], accumulate=n_in > 0)
            u = Uw.detach() if sc == 1.0 else Uw.detach() * sc
            if rs != r:
                u = torch.cat([u, u.new_zeros(seg_w, r - rs)], 1)
            pieces.append(u)
            info.append((xis, sc, rs))
        if dspecs:                                    # every adapter down-projection of this GEMM in one launch,
            K.lora_down_multi([K.down_job(j["X"], j["D"], T, j["toff"], M, j["D"].shape[1], X2=j["X2"], x2_rows=j["x2_rows"],
                                          R=j["R"], r2=j["r2"], T_in=j.get("T_in"), t_in_r=j.get("t_in_r", 0))
                               for j in _merge_down_jobs(dspecs)])   # adapters sharing x in one pass
        U = _stack_rows(pieces)
        if fused:
            # the part of T that does not come from x (the control term's share of the q adapter, L_q(h + c) = L_q(h) + L_q(c)):
            # handed in precomputed for the whole level (models._batched_control_terms) or evaluated here on the small input
            t_in, t_in_rows = None, 0
            if t_pre is not None and not f_in:                       # rank-space control term (ops.control_terms_rank)
                t_in, t_in_rows, f_in_mask = t_pre.detach(), (t_pre.shape[0] if t_pre.shape[0] != M else 0), 1
            elif f_in:
                rows = f_in[0][1].shape[0]
                pre = t_pre if (t_pre is not None and t_pre.shape == (rows, S * r)) else None
                if pre is None:
                    pre = torch.empty((rows, S * r), dtype=f32, device=x.device)
                    K.lora_down_multi([K.down_job(xi, Dd, pre, s_ * r, rows, Dd.shape[1]) for s_, xi, Dd in f_in])
                t_in, t_in_rows = pre, (rows if rows != M else 0)
            y = K.gemm(x, pack.w, M, pack.N, pack.K, bias=pack.bias, residual=residual, lora_t=T, lora_u=U,
                       lora_seg=seg_w, lora_scale=1.0, lora_dpack=ADAPTER_PACKS.get(f_srcs), lora_t_in=t_in,
                       lora_t_in_mask=f_in_mask, lora_t_in_rows=t_in_rows, tile_cfg=plan, ln=getattr(meta, "ln", None))
        else:
            y = K.gemm(x, pack.w, M, pack.N, pack.K, bias=pack.bias, residual=residual, lora_t=T, lora_u=U,
                       lora_seg=seg_w, lora_scale=1.0, ln=getattr(meta, "ln", None))
        ctx.pack, ctx.info, ctx.n_xa, ctx.r, ctx.has_res = pack, info, n_xa, r, residual is not None
        ctx.defer_dx = bool(getattr(meta, "defer_dx", False))
        ctx.t_pre_rows = t_pre.shape[0] if t_pre is not None else 0
        ctx.params = params                       # the Parameter objects themselves (leaf tensors)
        ctx.save_for_backward(T, *xas)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        pack, info, n_xa, r, params = ctx.pack, ctx.info, ctx.n_xa, ctx.r, ctx.params
        saved = ctx.saved_tensors
        T, xas = saved[0], saved[1:1 + n_xa]
        S = len(info)
        seg_w = pack.N // S
        M = dy.shape[0]
        dT = (torch.empty if all(i is not None and i[2] == r for i in info) else torch.zeros)((M, S * r), dtype=f32, device=dy.device)
        # single-adapter projections (out / cross-attention q): dT = dy . (s U) rides in the dgrad GEMM that streams dy anyway
        bplan = _fuse_plan(M, pack.K, pack.N, pack.K) if (S == 1 and info[0] is not None and info[0][2] == 4 and 0 in info[0][0]
                                                          and ctx.needs_input_grad[4]) else None
        fused_bwd = bplan is not None
        d_xas: List[Optional[torch.Tensor]] = [None] * n_xa
        own = []                                  # (segment, D) of adapters fed by x itself -> dgrad GEMM epilogue
        djobs, wjobs, later, keep = [], [], [], []
        pi = 0
        for s, m in enumerate(info):
            if m is None:
                continue
            xis, sc, rs = m
            D, Uw = params[2 * pi], params[2 * pi + 1]
            pi += 1
            dys = dy[:, s * seg_w:(s + 1) * seg_w]
            big = rs > 16
            # dT_s = sc * dy_s . U_s : a "down" projection of dy with U (k-major) as the matrix
            if fused_bwd:
                pass
            elif big:
                K.lora_down(dys, Uw.detach(), dT, s * r, M, seg_w, ldx=pack.N, kmajor=True, R=rs, d_scale=sc)
            else:
                djobs.append(K.down_job(dys, Uw.detach(), dT, s * r, M, seg_w, ldx=pack.N, kmajor=True, R=rs, d_scale=sc))
            if Uw.requires_grad:
                if big:
                    later.append((dys, T, s * r, _grad_buffer(Uw), Uw.shape[1], 1, seg_w, rs, sc, pack.N))
                else:
                    wjobs.append(K.wgrad_job(dys, T, s * r, _grad_buffer(Uw), Uw.shape[1], 1, M, seg_w, rs, scale=sc, lda=pack.N))
            if D.requires_grad:
                if big:                               # separate launches in stream order: accumulation is safe
                    for xi in xis:
                        later.append((xas[xi], dT, s * r, _grad_buffer(D), 1, D.shape[1], D.shape[1], rs, 1.0, None))
                else:                                 # ONE job over fp16(xa_0 + xa_1): jobs of a launch must not share G
                    a2 = xas[xis[1]] if len(xis) > 1 else None
                    if a2 is not None and a2.shape[0] != M:        # broadcast second input (control batch < UNet batch)
                        a2 = a2.repeat(M // a2.shape[0], 1)
                        keep.append(a2)                               # the temporary must outlive the deferred launch
                    wjobs.append(K.wgrad_job(xas[xis[0]], dT, s * r, _grad_buffer(D), 1, D.shape[1], M, D.shape[1], rs, A2=a2))
            for xi in xis:
                if xi == 0:
                    own.append((s, D))
                else:
                    later.append(("dx", xi, s, D))
        dx = None
        if fused_bwd:
            D0, U0, sc0 = params[0], params[1], info[0][1]
            dx = K.gemm(dy, pack.wt, M, pack.K, pack.N, lora_t=dT, lora_u=D0.detach(), lora_seg=pack.K, lora_u_tr=True, lora_r=4,
                        lora_dpack=ADAPTER_PACKS.get([U0.detach()], kmajor=True, scales=[sc0]), tile_cfg=bplan)
        if djobs:
            K.lora_down_multi(djobs)              # dT of every adapter of this GEMM: one launch
        if wjobs:
            K.lora_wgrad_defer(wjobs, dy.device, dy, T, dT, *xas, *keep)   # dU / dD: queued, flushed once at the end of backward
        for item in later:
            if item[0] == "dx":
                _, xi, s, D = item
                if ctx.needs_input_grad[5 + xi]:
                    g = K.lora_up(None, dT, s * r, D.detach(), M, D.shape[1], 1.0, u_tr=True)
                    rows = xas[xi].shape[0]
                    if rows != M:                                   # the input was broadcast over the batch: sum it back
                        g = g.reshape(M // rows, rows, -1).float().sum(0).to(f16)
                    d_xas[xi] = g if d_xas[xi] is None else K.add(d_xas[xi], g)
            else:
                A_, T_, to_, G_, gsn, gsj, N_, rs_, sc_, lda_ = item
                K.lora_wgrad(A_, T_, to_, G_, gsn, gsj, M, N_, rs_, scale=sc_, lda=lda_)
        if ctx.needs_input_grad[4] and not fused_bwd:
            dfr = ctx.defer_dx                      # the dgrad is this backward's last launch: its finish may ride in the norm's backward
            if not own:
                dx = K.gemm(dy, pack.wt, M, pack.K, pack.N, defer=dfr)
            else:
                segs = [s for s, _ in own]
                contiguous = segs == list(range(segs[0], segs[0] + len(segs))) and all(D.shape[0] == r for _, D in own)
                if contiguous:                                   # rank-r part of dx rides in the GEMM epilogue
                    Dcat = _stack_rows([D.detach() for _, D in own])
                    dx = K.gemm(dy, pack.wt, M, pack.K, pack.N, lora_t=dT[:, segs[0] * r:], lora_u=Dcat,
                                lora_seg=pack.K, lora_u_tr=True, lora_r=Dcat.shape[0], defer=dfr)
                else:
                    Dx = torch.zeros((S * r, pack.K), dtype=f32, device=dy.device)
                    for s, D in own:
                        Dx[s * r:s * r + D.shape[0]] = D.detach()
                    dx = K.gemm(dy, pack.wt, M, pack.K, pack.N, lora_t=dT, lora_u=Dx, lora_seg=pack.K, lora_u_tr=True, defer=dfr)
        dres = dy if ctx.has_res and ctx.needs_input_grad[5] else None
        d_tpre = None
        if ctx.t_pre_rows and ctx.needs_input_grad[3]:           # d(rank-space share of the q adapter) = dT of segment 0
            d_tpre = dT[:, :info[0][2]]
            if ctx.t_pre_rows != M:                               # it was broadcast over the batch: sum the copies
                d_tpre = d_tpre.reshape(M // ctx.t_pre_rows, ctx.t_pre_rows, -1).sum(0)
        return (None, None, None, d_tpre, dx, dres, *d_xas[1:], *([None] * len(params)))


def lora_proj(x, pack: LinearPack, segs: Sequence[Optional[Tuple[torch.Tensor, torch.Tensor, torch.Tensor, float]]],
              residual=None, t_pre=None):
    """segs[s] = None or (xa, down_weight [r,K], up_weight [seg,r], scale); xa is a tensor, or a tuple of tensors
    whose sum is the adapter input.  t_pre (optional, fp32 [rows of the second input, S * r]): the second input's share of the
    down-projections, already evaluated (models._batched_control_terms); a plain buffer, gradients flow through `xa`."""
    xas = [x]
    meta, params = [], []
    for sg in segs:
        if sg is None:
            meta.append(None)
            continue
        xa, D, U, sc = sg
        idxs = []
        for one in (xa if isinstance(xa, (tuple, list)) else (xa,)):
            idx = next((i for i, t in enumerate(xas) if t is one), None)
            if idx is None:
                xas.append(one)
                idx = len(xas) - 1
            idxs.append(idx)
        meta.append((tuple(idxs), float(sc)))
        params += [D, U]
    meta = _Meta(meta)
    meta.defer_dx = _FROM_NORM[0] > 0
    if residual is not None and _NEXT_LN[0] is not None:      # the out-projection of an attention call whose result feeds a LayerNorm
        meta.ln = _NEXT_LN[0]
    return _LoraProjFn.apply(pack, meta, len(xas), t_pre, x, residual, *xas[1:], *params)


class _ControlAddFn(torch.autograd.Function):
    """y = fp16(h + scale * up(down(ctrl)))                  (v1, reference models.py:214-218, 237-238)
       y = fp16(h + scale * up(down(cat(h, ctrl))))          (concat_hidden / V2, models.py:209-214, 343-349)
    ctrl [Mc, Cc] may hold fewer batch elements than h (control batch 1 broadcast, quirk C6)."""

    @staticmethod
    def forward(ctx, h, ctrl, D, U, scale, concat, t_ctrl=None, term_only=False):
        # term_only: return the control term alone, fp16(scale * fp16(up(down(...)))) -- for a caller that adds it to ANOTHER tensor
        # (post_add together with concat_hidden: `query + to_control(cat(hidden_states, control))`, reference models.py:208-218, 236-238)
        M, C_ = h.shape
        Mc, Cc = ctrl.shape
        R = D.shape[0]
        T = torch.empty((M, R), dtype=f32, device=h.device)
        xr = Mc if Mc != M else 0
        Dd, Ud = D.detach(), U.detach()
        if concat and t_ctrl is not None:
            # the control map's share ctrl . D[:, C:]^T was evaluated for the whole level (control_down_parts): one launch here
            K.lora_down_multi([K.down_job(h, Dd, T, 0, M, C_, T_in=t_ctrl, t_in_r=R)])
        elif concat:
            K.lora_down(h, Dd, T, 0, M, C_)                          # D[:, :C] acts on h (row pitch ldd = C + Cc)
            K.lora_down(ctrl, Dd[:, C_:], T, 0, M, Cc, accumulate=True, x_rows=xr)
        else:
            K.lora_down(ctrl, Dd, T, 0, M, Cc, x_rows=xr)
        y = K.lora_up(None if term_only else h, T, 0, Ud, M, C_, scale)
        ctx.save_for_backward(h, ctrl, T)
        ctx.params = (D, U)
        ctx.cfg = (scale, concat, xr, t_ctrl is not None)
        ctx.term_only = term_only
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        h, ctrl, T = ctx.saved_tensors
        D, U = ctx.params
        scale, concat, xr, parts = ctx.cfg
        M, C_ = h.shape
        Mc, Cc = ctrl.shape
        R = D.shape[0]
        Dd, Ud = D.detach(), U.detach()
        dT = torch.empty((M, R), dtype=f32, device=dy.device)
        K.lora_down(dy, Ud, dT, 0, M, C_, kmajor=True, R=R, d_scale=scale)
        wj = []       # (A, T, toff, G, gs_n, gs_j, N, scale, a_rows): batched below when the rank allows
        if U.requires_grad:
            wj.append((dy, T, 0, _grad_buffer(U), R, 1, C_, scale, 0))
        dh = None if ctx.term_only else dy
        dctrl = None
        if concat:
            if D.requires_grad:
                gD = _grad_buffer(D)
                wj.append((h, dT, 0, gD, 1, D.shape[1], C_, 1.0, 0))
                if not parts:
                    wj.append((ctrl, dT, 0, gD[:, C_:], 1, D.shape[1], Cc, 1.0, xr))
            dh = K.lora_up(None if ctx.term_only else dy, dT, 0, Dd[:, :C_], M, C_, 1.0, u_tr=True)
            Dc = Dd[:, C_:]
        else:
            if D.requires_grad:
                wj.append((ctrl, dT, 0, _grad_buffer(D), 1, D.shape[1], Cc, 1.0, xr))
            Dc = Dd
        if R <= 16:
            K.lora_wgrad_defer([K.wgrad_job(a_, t_, to_, g_, gn, gj, M, n_, R, scale=sc_, a_rows=ar_)
                                for a_, t_, to_, g_, gn, gj, n_, sc_, ar_ in wj], dy.device, dy, T, dT, h, ctrl)
        else:
            for a_, t_, to_, g_, gn, gj, n_, sc_, ar_ in wj:
                K.lora_wgrad(a_, t_, to_, g_, gn, gj, M, n_, R, scale=sc_, a_rows=ar_)
        if parts:
            # d(ctrl) and the control columns of dD are formed once per level from every site's dT (_ControlDownPartsFn.backward)
            dt_ctrl = dT.reshape(M // Mc, Mc, R).sum(0) if xr else dT
            return dh, None, None, None, None, None, dt_ctrl, None
        if ctx.needs_input_grad[1]:
            dctrl = K.lora_up(None, dT, 0, Dc, M, Cc, 1.0, u_tr=True)
            if xr:
                dctrl = dctrl.reshape(M // Mc, Mc, Cc).float().sum(0).to(f16)
        return dh, dctrl, None, None, None, None, None, None


class _ControlAddWideFn(torch.autograd.Function):
    """_ControlAddFn for WIDE ranks (> 16: configs/danbooru-sketch.json trains rank-256 control adapters on cat(h, ctrl), reference
    models.py:209-218): at that size `to_control` is two ordinary Linear layers ([M, C + Cc] x [C + Cc, 256] and [M, 256] x [256, C]),
    so every contraction runs on the MFMA GEMM / weight-gradient kernels with fp16 operands cast from the fp32 master weights each
    call (the rank-r kernels walk the rank 16 columns at a time: 528 us per expand and 96 weight-gradient launches per site at
    level 0 -- 138.7 ms per train step under danbooru-sketch.json, profiles/r05_bench_sketch.json).  Arithmetic = the reference's
    fp16 path: cat -> Linear (fp16 T) -> Linear -> x scale -> + h; the scale is folded into the up operand (exact for scale = 1)."""

    @staticmethod
    def forward(ctx, h, ctrl, D, U, scale, concat):
        M, C_ = h.shape
        Mc, Cc = ctrl.shape
        R = D.shape[0]
        cm = ctrl if Mc == M else ctrl.repeat(M // Mc, 1)                # control batch 1 broadcast over the batch (quirk C6)
        X = K.concat_channels(h, cm) if concat else cm.contiguous()
        Kin = X.shape[1]
        D16 = K.to_f16(D.detach().contiguous())                           # [R, Kin]
        Us = U.detach() if scale == 1.0 else U.detach() * scale
        U16 = K.to_f16(Us.contiguous())                                   # [C, R]
        T = K.gemm(X, D16, M, R, Kin)                                     # [M, R] fp16
        y = K.gemm(T, U16, M, C_, R, residual=h)
        ctx.save_for_backward(h, ctrl, T)
        ctx.params, ctx.cfg = (D, U), (scale, concat)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        h, ctrl, T = ctx.saved_tensors
        D, U = ctx.params
        scale, concat = ctx.cfg
        M, C_ = h.shape
        Mc, Cc = ctrl.shape
        R = D.shape[0]
        Us = U.detach() if scale == 1.0 else U.detach() * scale
        UsT16 = K.to_f16(Us.t().contiguous())                             # [R, C]: dT = dy . (s U)
        dT = K.gemm(dy, UsT16, M, R, C_)                                  # [M, R] fp16
        if U.requires_grad:
            if scale == 1.0:
                K.wgrad_accumulate(dy, T, M, C_, R, _grad_buffer(U))      # dU += dy^T . T
            else:
                g = torch.zeros_like(U, dtype=f32)
                K.wgrad_accumulate(dy, T, M, C_, R, g)
                _grad_buffer(U).add_(g, alpha=scale)
        cm = ctrl if Mc == M else ctrl.repeat(M // Mc, 1)
        X = K.concat_channels(h, cm) if concat else cm.contiguous()
        Kin = X.shape[1]
        if D.requires_grad:
            K.wgrad_accumulate(dT, X, M, R, Kin, _grad_buffer(D))         # dD += dT^T . cat(h, ctrl)
        Dt16 = K.to_f16(D.detach().t().contiguous())                      # [Kin, R]
        dh, dctrl = dy, None
        if concat:
            dh = K.gemm(dT, Dt16[:C_], M, C_, R, residual=dy)             # dh = dy + dT . D[:, :C]
        if ctx.needs_input_grad[1]:
            dctrl = K.gemm(dT, Dt16[C_:] if concat else Dt16, M, Cc, R)
            if Mc != M:
                dctrl = dctrl.reshape(M // Mc, Mc, Cc).float().sum(0).to(f16)
        return dh, dctrl, None, None, None, None


WIDE_RANK = _os.environ.get("CLORA_WIDE_RANK", "1") != "0"      # "0": rank > 16 adapters stay on the rank-r kernels (A/B)


def control_add(h, ctrl, D, U, scale, concat, t_ctrl=None, term_only=False):
    if WIDE_RANK and not term_only and D.shape[0] > 16 and t_ctrl is None and D.shape[0] % 8 == 0 and D.shape[1] % 8 == 0 and h.shape[0] % ctrl.shape[0] == 0:
        return _ControlAddWideFn.apply(h, ctrl, D, U, float(scale), bool(concat))
    return _ControlAddFn.apply(h, ctrl, D, U, float(scale), bool(concat), t_ctrl, bool(term_only))


CONTROL_PARTS = _os.environ.get("CLORA_CONTROL_PARTS", "1") != "0"      # "0": every concat adapter projects the control map itself (A/B)


class _ControlDownPartsFn(torch.autograd.Function):
    """The control map's share of the CONCAT adapters' down-projections -- `to_control(cat(h, ctrl))` of the v2 processors and of
    v1 with `concat_hidden` (reference models.py:209-214, 343-349, 366-372, 412-418) -- for every such layer of one UNet level at once:
    T_l = ctrl . D_l[:, C:]^T  [Mc, r], one multi-job launch (the layers share the level's hint-encoder map); each site then adds its
    h share in ONE launch (`_ControlAddFn` with t_ctrl).  Independent of the call's `scale`.  The backward receives every site's dT_l
    after the whole UNet backward: the control columns of dD_l join the deferred weight-gradient queue and
    d(ctrl) = [dT_1 | ... | dT_n] . [D_1[:, C:]; ...; D_n[:, C:]] is ONE rank-(n r) expand instead of one expand + one gradient
    accumulation into the shared map per layer.   params = the full down matrices D_l [r, C + Cc]."""

    @staticmethod
    def forward(ctx, ctrl, n, *Ds):
        Mc, Cc = ctrl.shape
        r = Ds[0].shape[0]
        assert all(D.shape[0] == r and D.shape[1] > Cc for D in Ds) and r <= 16
        Ts = [torch.empty((Mc, r), dtype=f32, device=ctrl.device) for _ in range(n)]
        K.lora_down_multi([K.down_job(ctrl, D.detach()[:, D.shape[1] - Cc:], Ts[l], 0, Mc, Cc) for l, D in enumerate(Ds)])
        ctx.save_for_backward(ctrl)
        ctx.params, ctx.cfg = Ds, (n, r)
        return tuple(Ts)

    @staticmethod
    def backward(ctx, *dTs):
        ctrl, = ctx.saved_tensors
        n, r = ctx.cfg
        Ds = ctx.params
        Mc, Cc = ctrl.shape
        live = [l for l in range(n) if dTs[l] is not None]
        if not live:
            return (None, None) + (None,) * n
        gs = [dTs[l] if (dTs[l].stride(1) == 1 and dTs[l].dtype == f32) else dTs[l].float().contiguous() for l in live]
        wj = []
        for g, l in zip(gs, live):
            if Ds[l].requires_grad:
                C_ = Ds[l].shape[1] - Cc
                wj.append(K.wgrad_job(ctrl, g, 0, _grad_buffer(Ds[l])[:, C_:], 1, Ds[l].shape[1], Mc, Cc, r))
        dctrl = None
        keep = list(gs)
        if ctx.needs_input_grad[0]:
            dTcat = torch.cat(gs, 1) if len(gs) > 1 else gs[0]
            Dcat = torch.cat([Ds[l].detach()[:, Ds[l].shape[1] - Cc:] for l in live], 0)
            dctrl = K.lora_up(None, dTcat, 0, Dcat, Mc, Cc, 1.0, u_tr=True)
            keep += [dTcat, Dcat]
        if wj:
            K.lora_wgrad_defer(wj, ctrl.device, ctrl, *keep)
        return (dctrl, None) + (None,) * n


def control_down_parts(ctrl, downs):
    """downs: the down matrices [r, C + Cc] of the concat adapters sharing `ctrl` [Mc, Cc] -> tuple of T_l [Mc, r] fp32"""
    return _ControlDownPartsFn.apply(ctrl, len(downs), *downs)


class _ControlTermFn(torch.autograd.Function):
    """c = fp16(scale * fp16(up(down(ctrl))))  [M, C]: the v1 control term on its own (reference models.py:214-218),
    for callers that feed `h + c` to an adapter by linearity instead of materialising the sum.  ctrl [Mc, Cc] may hold
    fewer batch elements than M rows (control batch 1 broadcast, quirk C6)."""

    @staticmethod
    def forward(ctx, ctrl, D, U, scale, M):
        Mc, Cc = ctrl.shape
        R, C_ = D.shape[0], U.shape[0]
        xr = Mc if Mc != M else 0
        T = torch.empty((M, R), dtype=f32, device=ctrl.device)
        K.lora_down(ctrl, D.detach(), T, 0, M, Cc, x_rows=xr)
        c = K.lora_up(None, T, 0, U.detach(), M, C_, scale)
        ctx.save_for_backward(ctrl, T)
        ctx.params, ctx.cfg = (D, U), (scale, xr, M)
        return c

    @staticmethod
    def backward(ctx, dc):
        dc = dc.contiguous()
        ctrl, T = ctx.saved_tensors
        D, U = ctx.params
        scale, xr, M = ctx.cfg
        Mc, Cc = ctrl.shape
        R, C_ = D.shape[0], U.shape[0]
        dT = torch.empty((M, R), dtype=f32, device=dc.device)
        K.lora_down(dc, U.detach(), dT, 0, M, C_, kmajor=True, R=R, d_scale=scale)
        wj = []
        if U.requires_grad:
            wj.append((dc, T, _grad_buffer(U), R, 1, C_, scale, 0))
        if D.requires_grad:
            wj.append((ctrl, dT, _grad_buffer(D), 1, D.shape[1], Cc, 1.0, xr))
        if R <= 16 and wj:
            K.lora_wgrad_defer([K.wgrad_job(a_, t_, 0, g_, gn, gj, M, n_, R, scale=sc_, a_rows=ar_)
                                for a_, t_, g_, gn, gj, n_, sc_, ar_ in wj], dc.device, dc, T, dT, ctrl)
        else:
            for a_, t_, g_, gn, gj, n_, sc_, ar_ in wj:
                K.lora_wgrad(a_, t_, 0, g_, gn, gj, M, n_, R, scale=sc_, a_rows=ar_)
        dctrl = None
        if ctx.needs_input_grad[0]:
            dctrl = K.lora_up(None, dT, 0, D.detach(), M, Cc, 1.0, u_tr=True)
            if xr:
                dctrl = dctrl.reshape(M // Mc, Mc, Cc).float().sum(0).to(f16)
        return dctrl, None, None, None, None


def control_term(ctrl, D, U, scale, M):
    return _ControlTermFn.apply(ctrl, D, U, float(scale), int(M))


class _ControlTermsFn(torch.autograd.Function):
    """The control terms c_l = fp16(scale * fp16(U_l (D_l ctrl))) of ALL attention sites of one UNet level in one go
    (they share the level's hint-encoder feature map `ctrl` [Mc, Cc]): one multi-job down launch + one multi-job up
    launch instead of two per site.  The backward receives every dc_l at once -- autograd runs it after the whole UNet
    backward, it only feeds the hint encoder -- so dT is one multi-job launch, the weight gradients join the deferred
    queue, and d(ctrl) = [dT_1 | ... | dT_n] . [D_1; ...; D_n] is ONE rank-(n*r) expand: the n-way gradient
    accumulation into the shared control map (n-1 autograd adds per level) disappears."""

    @staticmethod
    def forward(ctx, ctrl, scale, n, *params):
        Mc, Cc = ctrl.shape
        Ds, Us = params[0::2], params[1::2]
        r, C_ = Ds[0].shape[0], Us[0].shape[0]
        assert all(D.shape == (r, Cc) for D in Ds) and all(U.shape == (C_, r) for U in Us) and r <= 16
        T = torch.empty((Mc, n * r), dtype=f32, device=ctrl.device)
        K.lora_down_multi([K.down_job(ctrl, D.detach(), T, l * r, Mc, Cc) for l, D in enumerate(Ds)])
        cs = [torch.empty((Mc, C_), dtype=f16, device=ctrl.device) for _ in range(n)]
        K.lora_up_multi([K.up_job(None, T, l * r, U.detach(), cs[l], Mc, C_, scale) for l, U in enumerate(Us)])
        ctx.save_for_backward(ctrl, T)
        ctx.params, ctx.cfg = params, (scale, n, r, C_)
        return tuple(cs)

    @staticmethod
    def backward(ctx, *dcs):
        ctrl, T = ctx.saved_tensors
        scale, n, r, C_ = ctx.cfg
        Ds, Us = ctx.params[0::2], ctx.params[1::2]
        Mc, Cc = ctrl.shape
        live = [l for l in range(n) if dcs[l] is not None]
        dcs = [d.contiguous() if d is not None else None for d in dcs]
        dT = (torch.empty if len(live) == n else torch.zeros)((Mc, n * r), dtype=f32, device=ctrl.device)
        K.lora_down_multi([K.down_job(dcs[l], Us[l].detach(), dT, l * r, Mc, C_, kmajor=True, R=r, d_scale=scale) for l in live])
        wj = []
        for l in live:
            if Us[l].requires_grad:
                wj.append(K.wgrad_job(dcs[l], T, l * r, _grad_buffer(Us[l]), r, 1, Mc, C_, r, scale=scale))
            if Ds[l].requires_grad:
                wj.append(K.wgrad_job(ctrl, dT, l * r, _grad_buffer(Ds[l]), 1, Cc, Mc, Cc, r))
        if wj:
            K.lora_wgrad_defer(wj, ctrl.device, ctrl, T, dT, *[d for d in dcs if d is not None])
        dctrl = None
        if ctx.needs_input_grad[0]:
            Dcat = _stack_rows([D.detach() for D in Ds])
            dctrl = K.lora_up(None, dT, 0, Dcat, Mc, Cc, 1.0, u_tr=True)
        return (dctrl, None, None) + (None,) * (2 * n)


def control_terms(ctrl, layers, scale=1.0):
    """layers: [(down_weight, up_weight), ...] of the sites sharing `ctrl` -> tuple of control terms [Mc, C]"""
    flat = [w for D, U in layers for w in (D, U)]
    return _ControlTermsFn.apply(ctrl, float(scale), len(layers), *flat)


RANK_CONTROL = _os.environ.get("CLORA_RANK_CONTROL", "1") != "0"      # "0": materialise the control terms c_l [M, C] (round-3 path, A/B)


class _ControlTermsRankFn(torch.autograd.Function):
    """The v1 control terms of all sites of one UNet level in RANK SPACE (include/clora.h clora_rank_*; reference
    models.py:214-218, 237-238).  c_l = s U_c,l (D_c,l ctrl) only ever meets the q adapter's down matrix, D_q,l (h + c_l) =
    D_q,l h + (s D_q,l U_c,l)(D_c,l ctrl): the function returns Tq_l = (D_c,l ctrl) M_l^T  [Mc, 4] with M_l = s D_q,l U_c,l (4 x r_c),
    which ops.lora_proj adds to the q adapter's T (`t_pre`) -- the [Mc, C] tensors c_l are never formed, forward or backward.
    Backward gets dTq_l (= dT of each site's q adapter): dTc_l = dTq_l M_l, G_l = dTq_l^T Tc_l (4 x r_c Gram matrix, deterministic
    two-stage reduction), dU_c,l += s D_q,l^T G_l, dD_q,l += s G_l U_c,l^T (the control share of the q adapter's down gradient),
    dD_c,l = dTc_l^T ctrl (deferred adapter weight-gradient job), d ctrl = [dTc_1 | ...] [D_c,1; ...].
    params = (D_c,1, U_c,1, D_q,1, D_c,2, ...)."""

    @staticmethod
    def forward(ctx, ctrl, scale, n, *params):
        Mc, Cc = ctrl.shape
        Dcs, Ucs, Dqs = params[0::3], params[1::3], params[2::3]
        rc = Dcs[0].shape[0]
        dev = ctrl.device
        Tc = torch.empty((Mc, n * rc), dtype=f32, device=dev)
        K.lora_down_multi([K.down_job(ctrl, D.detach(), Tc, l * rc, Mc, Cc) for l, D in enumerate(Dcs)])
        Mbuf = torch.empty((n, 4 * rc), dtype=f32, device=dev)
        Tq = [torch.empty((Mc, 4), dtype=f32, device=dev) for _ in range(n)]
        sites = [K.rank_site(Dqs[l].detach(), Ucs[l].detach(), Mbuf[l], Tc, l * rc, Tq[l], Mc, scale) for l in range(n)]
        K.rank_compose(sites)
        K.rank_mix(sites)
        ctx.save_for_backward(ctrl, Tc, Mbuf)
        ctx.params, ctx.cfg = params, (scale, n, rc)
        return tuple(Tq)

    @staticmethod
    def backward(ctx, *dTq):
        ctrl, Tc, Mbuf = ctx.saved_tensors
        scale, n, rc = ctx.cfg
        Dcs, Ucs, Dqs = ctx.params[0::3], ctx.params[1::3], ctx.params[2::3]
        Mc, Cc = ctrl.shape
        live = [l for l in range(n) if dTq[l] is not None]
        dTc = (torch.empty if len(live) == n else torch.zeros)((Mc, n * rc), dtype=f32, device=ctrl.device)
        keep = []
        if live:
            sites = []
            for l in live:
                g = dTq[l] if dTq[l].stride(1) == 1 else dTq[l].contiguous()
                keep.append(g)
                sites.append(K.rank_site(Dqs[l].detach(), Ucs[l].detach(), Mbuf[l], Tc, l * rc, g, Mc, scale, dTc=dTc,
                                         gDq=_grad_buffer(Dqs[l]) if Dqs[l].requires_grad else None,
                                         gUc=_grad_buffer(Ucs[l]) if Ucs[l].requires_grad else None))
            ws = K.rank_mix(sites, backward=True, device=ctrl.device)
            K.rank_compose_bwd(sites, ws)
            keep.append(ws)
        wj = [K.wgrad_job(ctrl, dTc, l * rc, _grad_buffer(Dcs[l]), 1, Cc, Mc, Cc, rc) for l in live if Dcs[l].requires_grad]
        if wj:
            K.lora_wgrad_defer(wj, ctrl.device, ctrl, dTc, *keep)
        dctrl = None
        if ctx.needs_input_grad[0]:
            dctrl = K.lora_up(None, dTc, 0, _stack_rows([D.detach() for D in Dcs]), Mc, Cc, 1.0, u_tr=True)
        return (dctrl, None, None) + (None,) * (3 * n)


def control_terms_rank(ctrl, layers, scale=1.0):
    """layers: [(to_control.down, to_control.up, to_q_lora.down), ...] of the sites sharing `ctrl` -> tuple of Tq_l [Mc, 4] (fp32),
    the control terms' shares of the sites' q down-projections (pass each to ops.lora_proj as `t_pre`)"""
    flat = [w for tri in layers for w in tri]
    return _ControlTermsRankFn.apply(ctrl, float(scale), len(layers), *flat)


def rank_control_ok(layers):
    """shapes the rank-space kernels take: control rank <= 8, q adapter rank 4, one shape for all sites of the level"""
    shapes = {(tuple(Dc.shape), tuple(Uc.shape), tuple(Dq.shape)) for Dc, Uc, Dq in layers}
    if len(shapes) != 1:
        return False
    Dc, Uc, Dq = layers[0]
    return Dc.shape[0] <= 8 and Dq.shape[0] == 4 and Uc.shape[1] == Dc.shape[0] and Dq.shape[1] == Uc.shape[0] and len(layers) <= 16


@torch.no_grad()
def control_q_parts(terms, q_downs):
    """T_l = c_l . D_q,l^T for the control terms c_l [Mc, C] of the sites of one level (one multi-job launch): the share of
    each site's q down-projection that does not come from the hidden states (reference models.py:237-238, by linearity).
    Plain fp32 buffers [Mc, 12] (columns 0..3 written; the q | k | v column layout of the fused self-attention projection);
    None where the adapter cannot ride in the GEMM anyway.  Gradients flow through the terms themselves (ops.lora_proj)."""
    outs, jobs = [], []
    for c, D in zip(terms, q_downs):
        if D.shape[0] != 4 or D.shape[1] % 64 or D.shape[1] % FUSE_TILE_N:
            outs.append(None)
            continue
        T = torch.empty((c.shape[0], 12), dtype=f32, device=c.device)
        jobs.append(K.down_job(c.detach(), D.detach(), T, 0, c.shape[0], D.shape[1]))
        outs.append(T)
    if jobs:
        K.lora_down_multi(jobs)
    return outs


class _LoraApplyFn(torch.autograd.Function):
    """y = fp16(base + scale * up(down(x)))  -- one LoRALinearLayer applied where the reference writes it
    (`t = t + scale * lora(x)`, models.py:125-147, 232-282).  Building block of the generic (unfused) processor
    path: post_add=True and pre_loras / post_loras chains, where adapter inputs depend on earlier adapter outputs."""

    @staticmethod
    def forward(ctx, base, x, D, U, scale):
        M, N = base.shape
        T = torch.empty((M, D.shape[0]), dtype=f32, device=base.device)
        K.lora_down(x, D.detach(), T, 0, M, D.shape[1])
        y = K.lora_up(base, T, 0, U.detach(), M, N, scale)
        ctx.save_for_backward(x, T)
        ctx.params, ctx.scale, ctx.same = (D, U), scale, x is base
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        x, T = ctx.saved_tensors
        D, U = ctx.params
        scale = ctx.scale
        M, N = dy.shape
        R = D.shape[0]
        dT = torch.empty((M, R), dtype=f32, device=dy.device)
        K.lora_down(dy, U.detach(), dT, 0, M, N, kmajor=True, R=R, d_scale=scale)
        wj = []
        if U.requires_grad:
            wj.append((dy, T, _grad_buffer(U), R, 1, N, scale))
        if D.requires_grad:
            wj.append((x, dT, _grad_buffer(D), 1, D.shape[1], D.shape[1], 1.0))
        if R <= 16 and wj:
            K.lora_wgrad_defer([K.wgrad_job(a_, t_, 0, g_, gn, gj, M, n_, R, scale=sc_) for a_, t_, g_, gn, gj, n_, sc_ in wj],
                               dy.device, dy, T, dT, x)
        else:
            for a_, t_, g_, gn, gj, n_, sc_ in wj:
                K.lora_wgrad(a_, t_, 0, g_, gn, gj, M, n_, R, scale=sc_)
        dbase, dx = dy, None
        if ctx.same:                       # post_add: the adapter reads the tensor it is added to
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                dbase = K.lora_up(dy, dT, 0, D.detach(), M, D.shape[1], 1.0, u_tr=True)
            return dbase, None, None, None, None
        if ctx.needs_input_grad[1]:
            dx = K.lora_up(None, dT, 0, D.detach(), M, D.shape[1], 1.0, u_tr=True)
        return dbase, dx, None, None, None


def lora_apply(base, x, down_weight, up_weight, scale):
    """base [M,N] + scale * LoRA(x [M,K]); pass x = base for post_add adapters."""
    return _LoraApplyFn.apply(base, x, down_weight, up_weight, float(scale))


# ------------------------------------------------------------------------------------------------ trainable conv (hint encoder)
class _TrainConvFn(torch.autograd.Function):
    """Conv2d of the trainable hint encoder (reference models.py:470, 529, 594-597, 684): fp32 master
    weight, fp16 compute (what accelerate's autocast does around control_lora.forward).
    One pack launch makes both fp16 GEMM operands of the step; the weight / bias gradients are accumulated by the
    wgrad kernel straight into the parameters' .grad (OIHW), so autograd sees no gradient for them."""

    @staticmethod
    def forward(ctx, x, weight, bias, B, H, W, stride, asym_pad, need_dx):
        Co, Ci, k, _ = weight.shape
        Cip = x.shape[1]
        if PERSISTENT_CONV_PACKS:
            wp, wd = TRAIN_CONV_PACKS.get(weight, Cip, need_dx)
        else:
            wp, wd = K.conv_weight_pack(weight.detach(), Cip, need_dx)
        if k == 3:
            cd, Ho, Wo = K.conv_fwd_desc(H, W, Cip, 3, stride, 0 if asym_pad else 1, False, asym_pad)
            y = K.gemm(x, wp, B * Ho * Wo, Co, 9 * Cip, conv=cd, bias=bias.detach())
        else:
            cd, Ho, Wo = None, H, W
            y = K.gemm(x, wp, B * H * W, Co, Cip, bias=bias.detach())
        ctx.save_for_backward(x)
        ctx.params, ctx.wd = (weight, bias), wd
        ctx.cfg = (B, H, W, Ho, Wo, stride, asym_pad, need_dx, cd)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        (x,) = ctx.saved_tensors
        weight, bias = ctx.params
        B, H, W, Ho, Wo, stride, asym_pad, need_dx, cd = ctx.cfg
        Co, Ci, k, _ = weight.shape
        Cip = x.shape[1]
        Cop = (Co + 7) // 8 * 8
        M = B * Ho * Wo
        dx = None
        if need_dx:
            if k == 3:
                cdd = K.conv_dgrad_desc(Ho, Wo, Cop, H, W, 3, stride, 1, asym_pad)
                dx = K.gemm(dy, ctx.wd, B * H * W, Cip, 9 * Cop, conv=cdd)
            else:
                dx = K.gemm(dy, ctx.wd, M, Cip, Cop)
        if weight.requires_grad:
            gb = _grad_buffer(bias) if bias.requires_grad else None
            if k == 1:      # [Co][Ci] IS the parameter layout: accumulate straight into .grad
                K.conv_wgrad_into(dy, x, M, Co, Cip, cd, _grad_buffer(weight), gb)
            else:           # gather-ordered persistent staging (coalesced atomics) + one unpack launch into OIHW .grad
                stage = getattr(weight, "_clora_wgrad_stage", None)
                if stage is None or stage.numel() != Co * 9 * Cip + Co or stage.device != dy.device:
                    stage = torch.zeros(Co * 9 * Cip + Co, dtype=f32, device=dy.device)
                    weight._clora_wgrad_stage = stage
                K.conv_wgrad_staged(dy, x, M, Co, 9 * Cip, cd, stage, _grad_buffer(weight), gb, Cip)
        return dx, None, None, None, None, None, None, None, None


def train_conv(x, weight, bias, B, H, W, stride=1, asym_pad=False, need_dx=True):
    """x [B*H*W, Cip] fp16 NHWC (Cip = channels padded to 8), weight [Co,Ci,k,k] fp32 parameter."""
    return _TrainConvFn.apply(x, weight, bias, B, H, W, stride, asym_pad, need_dx)
