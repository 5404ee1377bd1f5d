"""Tensor-level wrappers over the C ABI (one Python function per entry point of include/clora.h).

No math happens here: the wrappers validate shapes/dtypes, allocate outputs / workspaces with torch
(device memory plumbing) and enqueue the HIP kernel on torch's current stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import capi
from .capi import ConvDesc, Epilogue, ptr

f16, f32 = torch.float16, torch.float32


class KernelProfiler:
    """Optional per-launch timing with HIP events recorded on the launch stream (bench.py's roofline leg).
    Off by default; never active inside a timed region."""

    def __init__(self, detail=False):
        self.detail = detail       # per-shape records (tools / debugging)
        self.records = []          # (entry point, start event, end event, algorithmic flops, algorithmic bytes)

    def summary(self):
        torch.cuda.synchronize()
        agg = {}
        for name, e0, e1, fl, by in self.records:
            a = agg.setdefault(name, dict(calls=0, ms=0.0, flops=0.0, bytes=0.0))
            a["calls"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += fl
            a["bytes"] += by
        return agg


PROFILER: Optional[KernelProfiler] = None


# ---- deferred split-K finishes (include/clora.h clora_deferred_t).  A GEMM launched with defer=True that turned out split-K leaves
# its slabs in the workspace and registers here under its OUTPUT's address; the GroupNorm / LayerNorm wrapper that reads that tensor
# next takes the entry and folds the slabs while it loads (71 of the 108 finish launches of a train step).  Safety net: ANY other
# kernel call first finishes whatever is still pending (the workspace may be reused, the tensor may be read) -- a caller may only ask
# for deferral where the output's next reader goes through this module (torch-native reads would see an unwritten buffer).
_PENDING = {}
_pending_task = [None]
FUSE_LN = os.environ.get("CLORA_FUSE_LN", "1") != "0"                 # "0": every LayerNorm is its own launch (round-5 path, A/B runs)


class LayerNormSlot:
    """A LayerNorm that FOLLOWS a projection (upstream BasicTransformerBlock: x = attn(...) + x; n = norm(x)), offered to the GEMM that
    produces x: `params` = (gamma fp32, beta fp32, eps); `out` is filled with LayerNorm(x) by a launch that could fuse it (gemm `ln=`),
    else stays None and the caller launches the norm itself."""

    def __init__(self, gamma, beta, eps):
        self.params, self.out = (gamma, beta, eps), None
DEFER_FINISH = os.environ.get("CLORA_DEFER_FINISH", "1") != "0"       # "0": every split-K GEMM runs its own finish pass (A/B runs)


def flush_pending():
    while _PENDING:
        _, (d, keep) = _PENDING.popitem()
        capi.lib().call("clora_finish_deferred", C.byref(d), capi.stream())


def take_pending(t: torch.Tensor):
    """-> (clora_deferred_t of the deferred producer of tensor `t`, its operand tensors) or (None, None); anything else pending is
    finished now.  The caller keeps the second value referenced until its own launch has been issued: the producer's epilogue operands
    (residual, bias, T ...) may have no other owner left."""
    hit = _PENDING.pop(t.data_ptr(), None) if _PENDING else None
    if _PENDING:
        flush_pending()
    return hit if hit is not None else (None, None)


def _call(name, *args, flops=0.0, nbytes=0.0, tag=None):
    if _PENDING:
        flush_pending()
    if PROFILER is None:
        capi.lib().call(name, *args, capi.stream())
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    capi.lib().call(name, *args, capi.stream())
    e1.record()
    PROFILER.records.append((name if not (tag and PROFILER.detail) else f"{name}[{tag}]", e0, e1, float(flops), float(nbytes)))


def _c2(t: torch.Tensor) -> torch.Tensor:
    """view a [..., C] contiguous tensor as 2-D [M, C]"""
    assert t.is_contiguous(), "kernel operands must be contiguous"
    return t.reshape(-1, t.shape[-1])


# ------------------------------------------------------------------ conv descriptors (see include/clora.h)
def conv_fwd_desc(Hin, Win, Cin, ksize=3, stride=1, pad=1, upsample=False, asym_pad=False, kchunk=0) -> Tuple[ConvDesc, int, int]:
    """Forward gather.  asym_pad = diffusers Downsample2D(padding=0): F.pad (0,1,0,1) then stride 2 (A9).
    kchunk: K order of the weight operand (include/clora.h clora_conv_t.kchunk; ops.conv_k_order)."""
    if upsample:
        Hout, Wout = 2 * Hin, 2 * Win
        return ConvDesc(1, Hin, Win, Cin, Hout, Wout, ksize, 1, 1, -pad, 2 * Hin, 2 * Win, 1, 0, kchunk), Hout, Wout
    if asym_pad:
        Hout, Wout = (Hin + 1 - ksize) // stride + 1, (Win + 1 - ksize) // stride + 1
        return ConvDesc(1, Hin, Win, Cin, Hout, Wout, ksize, stride, 1, 0, Hin, Win, 0, 0, kchunk), Hout, Wout
    Hout, Wout = (Hin + 2 * pad - ksize) // stride + 1, (Win + 2 * pad - ksize) // stride + 1
    return ConvDesc(1, Hin, Win, Cin, Hout, Wout, ksize, stride, 1, -pad, Hin, Win, 0, 0, kchunk), Hout, Wout


def conv_dgrad_desc(Hout, Wout, Cout, Hin, Win, ksize=3, stride=1, pad=1, asym_pad=False, kchunk=0) -> ConvDesc:
    """Gather for dX (rows enumerate the INPUT pixels of the forward conv, A operand is dY).  For an
    upsampled forward pass (Hin, Win) are the upsampled dims and the result is 2x2 sum-pooled afterwards."""
    p = 0 if asym_pad else pad
    if stride == 1:
        return ConvDesc(1, Hout, Wout, Cout, Hin, Win, ksize, 1, -1, p, Hout, Wout, 0, 0, kchunk)
    assert stride == 2
    return ConvDesc(1, Hout, Wout, Cout, Hin, Win, ksize, 1, -1, p, 2 * Hout, 2 * Wout, 1, 1, kchunk)


_ws_cache = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only per-device scratch (split-K slabs, GroupNorm partials). Reused across calls on one stream.
    A captured hipGraph bakes the buffer's address into its kernel nodes, so a buffer that has been handed out is NEVER
    freed (growing keeps the old one alive) and on a GPU the first allocation already covers the largest request the
    library's planner can make -- a later, larger problem (batch-32 inference after a captured train step) must not
    move the scratch under the graph."""
    key = str(device)
    w = _ws_cache.get(key)
    if w is None or w.numel() < nbytes:
        floor = GEMM_WS_BYTES if torch.device(device).type == "cuda" else (1 << 20)
        if w is not None:
            _ws_retired.append(w)
        w = torch.empty(max(nbytes, floor), dtype=torch.uint8, device=device)
        _ws_cache[key] = w
    return w


_ws_retired = []
CONV_STRIP = os.environ.get("CLORA_CONV_STRIP", "1") != "0"       # "0": the large-map hint-encoder convolutions stay on the implicit GEMM (A/B)
WIDE_TILE_CFGS = (1, 4, 7, 8, 9, 21, 31, 41, 53, 56, 58, 59)      # tile_cfg values whose tiles are >= 128 columns wide (GEGLU-forward epilogue)
GEMM_WS_BYTES = 256 << 20   # split-K slab budget handed to the library's launch planner

# Autotuned launch configurations (tools/tune_gemm.py on an MI355X): exact-shape lookups for the GEMMs of the
# SD-1.5 step; any other shape falls back to the library's latency model (split_k = 0, tile_cfg = 0).
_TUNING = None


def tuning_key(M, N, K, conv) -> str:
    if conv is None:
        return f"{M}x{N}x{K}"
    return f"{M}x{N}x{K}:c{conv.ksize}m{conv.mul}k{conv.kmul}s{conv.shift}e{conv.need_even}h{conv.Hin}"


def _tuned(M, N, K, conv):
    global _TUNING
    if _TUNING is None:
        import json
        import os
        path = os.environ.get("CLORA_GEMM_TUNING_FILE") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "gemm_tuning_gfx950.json")
        use = os.path.exists(path) and os.environ.get("CLORA_GEMM_TUNING", "1") != "0"   # "0": latency model only (A/B runs)
        _TUNING = json.load(open(path))["table"] if use else {}
    return _TUNING.get(tuning_key(M, N, K, conv))


def _tuned_fused(M, N, K):
    """table entry of a projection launch that carries its adapter's down-projection (key suffix ":x"), or None"""
    _tuned(M, N, K, None)                                   # loads the table
    return _TUNING.get(f"{M}x{N}x{K}:x")


def gemm(A: torch.Tensor, Bw: torch.Tensor, M: int, N: int, K: int, *, lda: Optional[int] = None,
         conv: Optional[ConvDesc] = None, bias=None, rowadd=None, rows_per_batch=0, residual=None,
         lora_t=None, lora_u=None, lora_seg=0, lora_scale=1.0, lora_u_tr=False, lora_r=None,
         out: Optional[torch.Tensor] = None,
         split_k: int = 0, tile_cfg: int = 0, _tuned: bool = True,
         geglu: int = 0, geglu_h: Optional[torch.Tensor] = None, geglu_y: Optional[torch.Tensor] = None,
         geglu_keep_h: bool = True, lora_dpack: Optional[torch.Tensor] = None, lora_t_in: Optional[torch.Tensor] = None,
         lora_t_in_mask: int = 0, lora_t_in_rows: int = 0, defer: bool = False, ln=None, trunk: bool = False) -> torch.Tensor:
    """C[M,N] = A . Bw^T with the fused epilogue of clora_epilogue_t.
    geglu=1 (Bw / bias packed by ops.GegluPack, N = 2F): returns (y [M,F], h [M,2F] or None when not geglu_keep_h);
    geglu=2 (N = F, geglu_h = the saved h): returns dh [M, 2F].
    lora_dpack ([N / lora_seg * 8, K] fp16 from lora_pack): the launch also computes T = A . D^T per column segment, WRITES it to
    lora_t and uses it in its epilogue (clora_epilogue_t.lora_dpack); lora_t_in / lora_t_in_mask: precomputed part added to the
    segments of the mask (row m reads row m % lora_t_in_rows when that is > 0)."""
    assert A.dtype == f16 and Bw.dtype == f16 and Bw.shape == (N, K) and Bw.is_contiguous()
    e = Epilogue()
    if geglu == 1:
        F_ = N // 2
        C_ = torch.empty((M, N), dtype=f16, device=A.device) if geglu_keep_h else None
        y = geglu_y if geglu_y is not None else torch.empty((M, F_), dtype=f16, device=A.device)
        e.geglu, e.geglu_f, e.geglu_y = 1, F_, ptr(y, f16)
    elif geglu == 2:
        assert geglu_h is not None and geglu_h.shape == (M, 2 * N) and geglu_h.is_contiguous()
        C_ = out if out is not None else torch.empty((M, 2 * N), dtype=f16, device=A.device)
        e.geglu, e.geglu_f, e.geglu_h = 2, N, ptr(geglu_h, f16)
    else:
        C_ = out if out is not None else torch.empty((M, N), dtype=f16, device=A.device)
    assert C_ is None or (C_.dtype == f16 and C_.stride(-1) == 1)
    ldc = (C_.stride(0) if C_.dim() == 2 else N) if C_ is not None else N
    if bias is not None:
        assert bias.dtype == f32 and bias.numel() == N
        e.bias = ptr(bias)
    if rowadd is not None:
        assert rowadd.dtype == f16 and rowadd.dim() == 2 and rowadd.shape[1] == N and rowadd.stride(1) == 1 and rowadd.stride(0) % 8 == 0
        e.rowadd, e.rows_per_batch, e.ld_rowadd = ptr(rowadd), rows_per_batch, rowadd.stride(0)
    if residual is not None:
        assert residual.dtype == f16 and residual.stride(-1) == 1
        e.residual, e.ldr = ptr(residual), (residual.stride(0) if residual.dim() == 2 else N)
    # compensated residual trunk (clora_epilogue_t.residual_lo / c_lo): inside a TrunkLo window every launch that
    # adds a residual -- the x + f(x) sums of the UNet -- and every launch flagged `trunk` (proj_in, the shortcut convs: where a trunk
    # segment starts) also writes the rounding remainder of its output; the launch whose residual is that output picks it up
    c_lo = None
    want_lo_out = not _TRUNK_NO_OUT[0]
    if _TRUNK_LO_ON[0] and not geglu and C_ is not None and C_.dim() == 2 and (residual is not None or (trunk and want_lo_out)):
        if residual is not None:
            hit = _TRUNK_LO.pop(residual.data_ptr(), None)
            if hit is not None and hit[0].numel() == residual.numel() and (residual.stride(0) if residual.dim() == 2 else N) == e.ldr:
                e.residual_lo = ptr(hit[1], f16)
                _trunk_keep[:] = [hit]                       # alive across this launch call (afterwards the allocator's stream order protects it)
        if want_lo_out:
            c_lo = torch.empty_strided(C_.shape, C_.stride(), dtype=f16, device=A.device)
            e.c_lo = ptr(c_lo)
            defer = False
    if lora_t is not None:
        assert lora_t.dtype == f32 and lora_u.dtype == f32 and lora_t.stride(1) == 1 and lora_u.stride(1) == 1
        e.lora_t, e.ldt, e.lora_u, e.ldu, e.lora_u_tr = ptr(lora_t), lora_t.stride(0), ptr(lora_u), lora_u.stride(0), int(lora_u_tr)
        e.lora_r = lora_r if lora_r is not None else (lora_u.shape[0] if lora_u_tr else lora_u.shape[1])
        e.lora_seg, e.lora_scale = (lora_seg or N), float(lora_scale)
    if lora_dpack is not None:
        assert lora_t is not None and lora_dpack.dtype == f16 and lora_dpack.is_contiguous() and lora_dpack.shape == (N // e.lora_seg * 8, K)
        e.lora_dpack = ptr(lora_dpack)
        if lora_t_in is not None:
            assert lora_t_in.dtype == f32 and lora_t_in.stride(1) == 1
            e.lora_t_in, e.ldt_in, e.lora_t_in_rows, e.lora_t_in_mask = ptr(lora_t_in), lora_t_in.stride(0), int(lora_t_in_rows), int(lora_t_in_mask)
        split_k = 1
    if (CONV_STRIP and conv is not None and split_k == 0 and tile_cfg == 0 and not geglu and rowadd is None and residual is None
            and lora_t is None and not defer and ln is None and capi.lib().cdll.clora_conv_strip_eligible(M, N, C.byref(conv))):
        tile_cfg, split_k = 61, 1            # the hint encoder's large-map 3x3 convolutions: strip kernel (include/clora.h)
    if _tuned and split_k == 0 and tile_cfg == 0:
        hit = globals()["_tuned"](M, N, K, conv)
        if hit is not None:
            tile_cfg, split_k = hit
    if geglu:
        split_k = 1
        if geglu == 1 and tile_cfg not in WIDE_TILE_CFGS:
            tile_cfg = 0                       # the library picks a >= 128-column tile itself
    ws = workspace(GEMM_WS_BYTES if split_k == 0 else max(split_k, 1) * M * N * 4, A.device) if split_k != 1 else None
    # ln (a LayerNormSlot): where one tile spans the output row (N = 320 on the 8-wave 320-column tiles) the launch also writes
    # LayerNorm(C) -- the norm that follows an attention out-projection / proj_in in every BasicTransformerBlock -- into ln.out
    if ln is not None and ln.out is None and FUSE_LN and not geglu and conv is None and C_ is not None and C_.is_contiguous():
        tc = tile_cfg
        if lora_dpack is not None and tc not in (51, 52, 54, 55) and e.lora_seg % 320 == 0:
            tc = 54 if M >= 32768 else 55                    # what the library would pick (clora_gemm_f16_ex): made explicit
        if capi.lib().cdll.clora_gemm_ln_fusable(M, N, K, tc, split_k):
            g_, b_, eps_ = ln.params
            assert g_.dtype == f32 and b_.dtype == f32 and g_.numel() == N and b_.numel() == N
            tile_cfg = tc
            ln.out = torch.empty((M, N), dtype=f16, device=A.device)
            e.ln_gamma, e.ln_beta, e.ln_out, e.ln_eps = ptr(g_), ptr(b_), ptr(ln.out), float(eps_)
    # defer=True: if this launch is split-K, leave its finish pass to the GroupNorm / LayerNorm call that reads C next (_PENDING)
    d = None
    if defer and DEFER_FINISH and split_k != 1 and not geglu and PROFILER is None and C_ is not None and C_.is_contiguous():
        d = capi.Deferred()
        e.defer = C.addressof(d)
    _call("clora_gemm_f16_ex", ptr(A), lda if lda is not None else K, ptr(Bw), ptr(C_) if C_ is not None else None, ldc, M, N, K,
          C.byref(conv) if conv is not None else None, C.byref(e), split_k, tile_cfg,
          ptr(ws) if ws is not None else None, ws.numel() if ws is not None else 0,
          flops=2.0 * M * N * K, nbytes=2.0 * (A.numel() + N * K + M * N),
          tag=f"{M}x{N}x{K}{'conv' if conv is not None else ''}")
    if d is not None and d.splits > 1:
        _PENDING[C_.data_ptr()] = (d, (C_, bias, rowadd, residual, lora_t, lora_u, ws))     # operands stay alive until it is consumed
        task = _graph_task_id()
        if task >= 0 and _pending_task[0] != task:       # inside a backward pass: whatever nobody consumed is finished when the pass ends
            _pending_task[0] = task
            torch.autograd.Variable._execution_engine.queue_callback(flush_pending)
    if c_lo is not None:
        _TRUNK_LO[C_.data_ptr()] = (C_, c_lo)                # the strong reference to C_ keeps its address from being re-used
    if geglu == 1:
        return y, C_
    return C_


# hi tensor's data_ptr -> (hi, lo): the rounding remainders of the trunk tensors written inside the current TrunkLo window
_TRUNK_LO = {}
_trunk_keep = []
_TRUNK_LO_ON = [False]
TRUNK_LO_MODE = os.environ.get("CLORA_TRUNK_LO", "infer")      # "infer": forwards without autograd; "always"; "off"


class TrunkLo:
    """`with TrunkLo(enabled):` around ONE UNet forward: residual sums continue from their un-rounded values (see gemm).  The
    remainders live until the window closes (a captured hipGraph keeps the kernels, not these tensors)."""

    def __init__(self, enabled: bool):
        self.enabled = bool(enabled)

    def __enter__(self):
        self.prev = _TRUNK_LO_ON[0]
        if self.enabled and not self.prev:
            _TRUNK_LO.clear(); _trunk_keep.clear()
        _TRUNK_LO_ON[0] = self.enabled or self.prev
        return self

    def __exit__(self, *exc):
        _TRUNK_LO_ON[0] = self.prev
        if not self.prev:
            _TRUNK_LO.clear(); _trunk_keep.clear()
        return False


_TRUNK_NO_OUT = [False]


class TrunkNoOut:
    """`with TrunkNoOut(cond):` the residual launches inside still continue from the incoming remainder but do not write their own: the
    caller knows that no residual add reads this sum (a transformer's last FeedForward sum feeds proj_out as an operand; up-path block
    outputs go into a channel concatenation)."""

    def __init__(self, cond: bool = True):
        self.cond = bool(cond)

    def __enter__(self):
        self.prev = _TRUNK_NO_OUT[0]
        _TRUNK_NO_OUT[0] = self.cond or self.prev
        return self

    def __exit__(self, *exc):
        _TRUNK_NO_OUT[0] = self.prev
        return False


def trunk_lo_wanted() -> bool:
    return TRUNK_LO_MODE == "always" or (TRUNK_LO_MODE == "infer" and not torch.is_grad_enabled())


PATCH_TILE_CFGS = (71, 72, 73, 74, 75, 76, 77, 78, 79)     # conv3x3_patch_kernel variants of clora_gemm_f16_ex (77, 78: 392-pixel patch, rows >= 128 wide; 79: 256x160)


def conv_patch_eligible(M: int, conv: ConvDesc, tile_cfg: int) -> bool:
    """would `gemm(..., conv=conv, tile_cfg=tile_cfg)` run on the patch-staged 3x3 kernel (True) or fall back (False)?"""
    return bool(capi.lib().cdll.clora_conv_patch_eligible(M, C.byref(conv), tile_cfg))


TILE_ORDERS = {"m": 0, "n": 1, "auto": 2, "grid": 3}
DEFAULT_TILE_ORDER = "grid"      # the library's default (clora_set_option "tile_order" = 3): "auto" was -0.16 ms/step over "m" (same-box A/B r03),
#                                  "grid" adds per-XCD rectangles: fabric traffic 1.69x -> 1.59x of the algorithmic bytes, bit-identical (r04)


def set_option(name: str, value: int) -> None:
    """a tuning knob of the kernel library (clora_set_option): results never depend on it"""
    capi.lib().call("clora_set_option", name.encode(), int(value))


def set_tile_order(mode: str) -> None:
    """tile / attention-block -> XCD assignment of the launches that follow: 'm', 'n', 'auto' or 'grid' (default)"""
    set_option("tile_order", TILE_ORDERS[mode])


def conv_wgrad(dY: torch.Tensor, X: torch.Tensor, M: int, N: int, K: int, conv: Optional[ConvDesc],
               ldx: Optional[int] = None, with_bias: bool = False):
    """dW [N, K] (and the bias gradient [N] when with_bias) of a trainable conv / linear, one pass over dY and X."""
    buf = torch.zeros(N * K + (N if with_bias else 0), dtype=f32, device=dY.device)
    dW, db = buf[:N * K].view(N, K), (buf[N * K:] if with_bias else None)
    _call("clora_conv_wgrad_f16", ptr(dY, f16), N, ptr(X, f16), ldx if ldx is not None else K, ptr(dW), ptr(db), M, N, K,
          C.byref(conv) if conv is not None else None, 0, flops=2.0 * M * N * K)
    return (dW, db) if with_bias else dW


def conv_wgrad_into(dY: torch.Tensor, X: torch.Tensor, M: int, N: int, K: int, conv: Optional[ConvDesc],
                    weight_grad: torch.Tensor, bias_grad: Optional[torch.Tensor]):
    """accumulate the weight (OIHW, fp32) and bias gradients of a trainable conv straight into the parameters' .grad"""
    assert weight_grad.dtype == f32 and weight_grad.is_contiguous() and weight_grad.shape[0] == N
    _call("clora_conv_wgrad_f16", ptr(dY, f16), N, ptr(X, f16), K, ptr(weight_grad), ptr(bias_grad, f32) if bias_grad is not None else None,
          M, N, K, C.byref(conv) if conv is not None else None, weight_grad.shape[1], flops=2.0 * M * N * K)


def conv_wgrad_staged(dY, X, M, N, K, conv, stage: torch.Tensor, weight_grad: torch.Tensor, bias_grad: Optional[torch.Tensor], Cip: int):
    """3x3 trainable conv: coalesced atomics into the persistent, gather-ordered staging buffer `stage` (fp32
    [N*K + N], zero on entry and left zero), then the unpack adds it into the OIHW weight / bias gradients -- inside a backward
    pass as ONE multi-job launch for all convolutions at the end of the pass (conv_unpack_defer), else right away."""
    assert stage.dtype == f32 and stage.numel() == N * K + N and weight_grad.dtype == f32 and weight_grad.is_contiguous()
    sw, sb = stage[:N * K], stage[N * K:]
    _call("clora_conv_wgrad_f16", ptr(dY, f16), N, ptr(X, f16), K, ptr(sw), ptr(sb), M, N, K,
          C.byref(conv) if conv is not None else None, 0, flops=2.0 * M * N * K)
    if bias_grad is not None and conv_unpack_defer(conv_unpack_job(stage, weight_grad, bias_grad, Cip), stage, weight_grad, bias_grad):
        return
    Co, Ci, k, _ = weight_grad.shape
    _call("clora_conv_wgrad_unpack_f32", ptr(sw), ptr(sb) if bias_grad is not None else None, ptr(weight_grad),
          ptr(bias_grad, f32) if bias_grad is not None else None, Co, Ci, k, Cip)
    if bias_grad is None:
        sb.zero_()


def conv_weight_pack(weight: torch.Tensor, Cip: int, need_dgrad: bool, out=None):
    """fp32 [Co,Ci,k,k] -> fp16 forward operand [Co, k*k*Cip] and (optionally) dgrad operand [Cip, k*k*Cop], one launch;
    out = (fwd, dgrad) reuses persistent operand buffers (ops.TRAIN_CONV_PACKS)"""
    Co, Ci, k, _ = weight.shape
    Cop = (Co + 7) // 8 * 8
    if out is not None:
        fwd, dgrad = out
    else:
        fwd = torch.empty((Co, k * k * Cip), dtype=f16, device=weight.device)
        dgrad = torch.empty((Cip, k * k * Cop), dtype=f16, device=weight.device) if need_dgrad else None
    _call("clora_conv_weight_pack_f32", ptr(weight, f32), Co, Ci, k, Cip, Cop, ptr(fwd), ptr(dgrad) if dgrad is not None else None)
    return fwd, dgrad


def conv_pack_job(weight: torch.Tensor, Cip: int, fwd: torch.Tensor, dgrad: Optional[torch.Tensor]):
    """one problem of conv_weight_pack_multi (clora_conv_pack_job_t)"""
    Co, Ci, k, _ = weight.shape
    assert weight.dtype == f32 and weight.is_contiguous() and fwd.dtype == f16 and fwd.numel() == Co * k * k * Cip
    return capi.ConvPackJob(ptr(weight, f32), ptr(fwd), ptr(dgrad) if dgrad is not None else None, Co, Ci, k, Cip, (Co + 7) // 8 * 8, 0)


def conv_weight_pack_multi(jobs):
    """the fp16 operands of every trainable convolution in one launch per 32 jobs (clora_conv_weight_pack_multi_f32)"""
    for i in range(0, len(jobs), capi.CONV_MAX_JOBS):
        chunk = jobs[i:i + capi.CONV_MAX_JOBS]
        arr = (capi.ConvPackJob * len(chunk))(*chunk)
        _call("clora_conv_weight_pack_multi_f32", arr, len(chunk))


def conv_unpack_job(stage: torch.Tensor, weight_grad: torch.Tensor, bias_grad: Optional[torch.Tensor], Cip: int):
    """one problem of conv_wgrad_unpack_multi: `stage` = the persistent fp32 [Co * k*k*Cip + Co] staging buffer of conv_wgrad_staged"""
    Co, Ci, k, _ = weight_grad.shape
    nw = Co * k * k * Cip
    assert stage.dtype == f32 and stage.numel() == nw + Co and weight_grad.dtype == f32 and weight_grad.is_contiguous()
    sw, sb = stage[:nw], stage[nw:]
    return capi.ConvUnpackJob(ptr(sw), ptr(sb) if bias_grad is not None else None, ptr(weight_grad),
                              ptr(bias_grad, f32) if bias_grad is not None else None, Co, Ci, k, Cip)


def conv_wgrad_unpack_multi(jobs):
    for i in range(0, len(jobs), capi.CONV_MAX_JOBS):
        chunk = jobs[i:i + capi.CONV_MAX_JOBS]
        arr = (capi.ConvUnpackJob * len(chunk))(*chunk)
        _call("clora_conv_wgrad_unpack_multi_f32", arr, len(chunk))


# ------------------------------------------------------------------ attention
def softmax_rows(x: torch.Tensor, scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """row softmax of a 2-D fp16 matrix (fp32 statistics); `out` may alias `x`"""
    assert x.dim() == 2 and x.stride(1) == 1
    y = out if out is not None else torch.empty_like(x)
    assert y.stride(0) == x.stride(0)
    _call("clora_softmax_rows_f16", ptr(x, f16), ptr(y, f16), x.shape[0], x.shape[1], x.stride(0), float(scale),
          nbytes=4.0 * x.numel())
    return y


def attn_fwd(q, k, v, B, H, Nq, Nk, D, scale, out=None):
    """q/k/v: 2-D row-strided views [B*N, >=H*D] (stride(0) is the row pitch)."""
    o = out if out is not None else torch.empty((B * Nq, H * D), dtype=f16, device=q.device)
    lse = torch.empty((B, H, Nq), dtype=f32, device=q.device)
    _call("clora_attn_fwd_f16", ptr(q, f16), q.stride(0), ptr(k, f16), k.stride(0), ptr(v, f16), v.stride(0),
          ptr(o), o.stride(0), ptr(lse), B, H, Nq, Nk, D, float(scale), flops=4.0 * B * H * Nq * Nk * D)
    return o, lse


def attn_fwd_causal(q, k, v, B, H, N, D, scale, out=None):
    """causal self-attention, forward only (CLIP text encoder); q/k/v: 2-D row-strided views [B*N, >= H*D]"""
    o = out if out is not None else torch.empty((B * N, H * D), dtype=f16, device=q.device)
    _call("clora_attn_fwd_causal_f16", ptr(q, f16), q.stride(0), ptr(k, f16), k.stride(0), ptr(v, f16), v.stride(0),
          ptr(o), o.stride(0), B, H, N, D, float(scale), flops=2.0 * B * H * N * N * D)
    return o


def attn_bwd(q, k, v, o, dO, lse, B, H, Nq, Nk, D, scale, dq, dk, dv):
    delta = torch.empty((B, H, Nq), dtype=f32, device=q.device)
    ws = workspace(16 * 2 * B * Nk * H * D * 4, q.device) if Nk <= 1024 else None     # up to 16 query splits, one fp32 slab pair each
    _call("clora_attn_bwd_f16", ptr(q, f16), q.stride(0), ptr(k, f16), k.stride(0), ptr(v, f16), v.stride(0),
          ptr(o, f16), o.stride(0), ptr(dO, f16), dO.stride(0), ptr(lse, f32), ptr(delta),
          ptr(dq, f16), dq.stride(0), ptr(dk, f16), dk.stride(0), ptr(dv, f16), dv.stride(0),
          B, H, Nq, Nk, D, float(scale), ptr(ws) if ws is not None else None, ws.numel() if ws is not None else 0,
          flops=10.0 * B * H * Nq * Nk * D)
    return dq, dk, dv


# ------------------------------------------------------------------ norms
_gn_team = {}


def _dev_key(device) -> str:
    d = torch.device(device)
    if d.type == "cuda" and d.index is None:
        d = torch.device("cuda", torch.cuda.current_device())
    return str(d)


def gn_team_state(device) -> torch.Tensor:
    """the persistent exchange state of the one-launch GroupNorm of the large maps (clora_groupnorm_*_team, include/clora.h): one per
    device (= per stream here), zeroed once, never freed (captured graphs hold its address).  Word 0 is the kernels' sticky error flag
    (gn_team_errors)."""
    key = _dev_key(device)
    st = _gn_team.get(key)
    if st is None:
        st = torch.zeros(capi.lib().cdll.clora_groupnorm_team_state_bytes(), dtype=torch.uint8, device=device)
        _gn_team[key] = st
    return st


def gn_team_errors(device) -> int:
    """non-zero if an in-launch exchange of a team GroupNorm kernel ever gave up on this device (its output was invalid)"""
    st = _gn_team.get(_dev_key(device))
    return 0 if st is None else int(st[:4].view(torch.int32).item())


def _gn_ws(B, HW, Cc, G, device, bwd, params):
    n = capi.lib().cdll.clora_groupnorm_workspace_bytes(B, HW, Cc, G, int(bwd), int(params))
    if n == 0:
        raise capi.CloraError(f"unsupported GroupNorm shape B={B} HW={HW} C={Cc} G={G}")
    return workspace(n, device)


def groupnorm_fwd(x, gamma, beta, G, eps, silu, x2=None):
    """x2: the input is the channel concatenation cat(x, x2) read in place -> (y, stats, xcat) with xcat the concatenated
    tensor (written once by the kernel for the shortcut / backward).  If x is the still-unfinished output of a deferred split-K
    GEMM (take_pending) the kernel folds the slabs while it loads and stores the finished x."""
    B, HW, Ca = x.shape
    Cc = Ca + (x2.shape[-1] if x2 is not None else 0)
    src, _keep = take_pending(x)
    y = torch.empty((B, HW, Cc), dtype=f16, device=x.device)
    stats = torch.empty((B, G, 2), dtype=f32, device=x.device)
    xcat = torch.empty((B, HW, Cc), dtype=f16, device=x.device) if x2 is not None else None
    if src is not None and x2 is not None:          # the kernels take one or the other
        capi.lib().call("clora_finish_deferred", C.byref(src), capi.stream())
        src = None
    ws = _gn_ws(B, HW, Cc, G, x.device, False, False)
    team = gn_team_state(x.device)
    _call("clora_groupnorm_fwd_f16_team", ptr(x, f16), ptr(x2, f16) if x2 is not None else None, Ca if x2 is not None else 0,
          C.byref(src) if src is not None else None, ptr(xcat) if xcat is not None else None, ptr(y), ptr(gamma, f32), ptr(beta, f32),
          ptr(stats), B, HW, Cc, G, float(eps), int(silu), ptr(team), team.numel(), ptr(ws), ws.numel())
    return (y, stats, xcat) if x2 is not None else (y, stats)


def groupnorm_bwd(x, dy, gamma, beta, stats, G, silu, want_param_grads=False, grads_into=None, dres=None, split_at=0):
    """grads_into = (dgamma_buffer, dbeta_buffer): accumulate the affine gradients there instead of returning them;
    dres: gradient of x from the branch that bypasses the norm (added into dx by the kernel);
    split_at = Ca > 0: dx is returned as the pair (dx[..., :Ca], dx[..., Ca:]) of contiguous tensors (the gradients of the two
    inputs of a concatenating forward).  dy may be the unfinished output of a deferred split-K GEMM (take_pending)."""
    B, HW, Cc = x.shape
    dy_src, _keep = take_pending(dy)
    if split_at:
        dx = torch.empty((B, HW, split_at), dtype=f16, device=x.device)
        dx2 = torch.empty((B, HW, Cc - split_at), dtype=f16, device=x.device)
    else:
        dx, dx2 = torch.empty_like(x), None
    dg = db = None
    if grads_into is not None:
        dg, db = grads_into
        want_param_grads = True
    elif want_param_grads:
        dg = torch.empty(Cc, dtype=f32, device=x.device)
        db = torch.empty(Cc, dtype=f32, device=x.device)
    ws = _gn_ws(B, HW, Cc, G, x.device, True, want_param_grads)
    assert dres is None or (dres.shape == x.shape and dres.is_contiguous())
    team = gn_team_state(x.device)
    _call("clora_groupnorm_bwd_f16_team", ptr(x, f16), ptr(dy, f16), C.byref(dy_src) if dy_src is not None else None,
          ptr(dres, f16) if dres is not None else None, ptr(dx), ptr(dx2) if dx2 is not None else None, int(split_at),
          ptr(gamma, f32), ptr(beta, f32), ptr(stats, f32),
          ptr(dg, f32) if dg is not None else None, ptr(db, f32) if db is not None else None, B, HW, Cc, G, int(silu),
          int(grads_into is not None), ptr(team), team.numel(), ptr(ws), ws.numel())
    if split_at:
        dx = (dx, dx2)
    return (dx, None, None) if grads_into is not None else (dx, dg, db)


def layernorm_fwd(x, gamma, beta, eps):
    x2 = _c2(x)
    y = torch.empty_like(x)
    _call("clora_layernorm_fwd_f16", ptr(x2, f16), ptr(y), ptr(gamma, f32), ptr(beta, f32), x2.shape[0], x2.shape[1], float(eps))
    return y


def layernorm_bwd(x, dy, gamma, eps, dres=None):
    """dres: gradient of x from the branch that bypasses the norm (added into dx by the kernel)"""
    x2 = _c2(x)
    dx = torch.empty_like(x)
    dy_src, _keep = take_pending(dy)     # dy may be the unfinished output of a deferred split-K dgrad GEMM: folded while loading
    _call("clora_layernorm_bwd_f16_ex", ptr(x2, f16), ptr(_c2(dy), f16), C.byref(dy_src) if dy_src is not None else None,
          ptr(_c2(dres), f16) if dres is not None else None, ptr(dx), ptr(gamma, f32), x2.shape[0], x2.shape[1], float(eps))
    return dx


def geglu_fwd(h):
    h2 = _c2(h)
    M, F2 = h2.shape
    y = torch.empty(h.shape[:-1] + (F2 // 2,), dtype=f16, device=h.device)
    _call("clora_geglu_fwd_f16", ptr(h2, f16), ptr(y), M, F2 // 2)
    return y


def geglu_bwd(h, dy):
    h2 = _c2(h)
    M, F2 = h2.shape
    dh = torch.empty_like(h)
    _call("clora_geglu_bwd_f16", ptr(h2, f16), ptr(_c2(dy), f16), ptr(dh), M, F2 // 2)
    return dh


# ------------------------------------------------------------------ adapters
def lora_pack(table: torch.Tensor, njobs: int, max_k: int):
    """one launch over a device-resident clora_lora_pack_job_t table (ops.AdapterPacks builds it)"""
    assert table.dtype == torch.uint8 and table.numel() >= njobs * C.sizeof(capi.LoraPackJob)
    _call("clora_lora_pack_f16", ptr(table), njobs, max_k)


def rank_site(Dq, Uc, Mbuf, Tc, toff, Tq, rows, scale, dTc=None, gDq=None, gUc=None):
    """one site of the rank-space control term (clora_rank_site_t)"""
    assert Dq.dtype == f32 and Uc.dtype == f32 and Dq.shape[0] == 4 and Dq.stride(1) == 1 and Uc.stride(1) == 1 and Tq.stride(1) == 1
    return capi.RankSite(ptr(Dq), Dq.stride(0), ptr(Uc), Uc.stride(0), ptr(Mbuf), ptr(Tc), Tc.stride(0), toff, ptr(Tq), Tq.stride(0),
                         ptr(dTc) if dTc is not None else None, dTc.stride(0) if dTc is not None else 0,
                         ptr(gDq) if gDq is not None else None, ptr(gUc) if gUc is not None else None,
                         rows, Dq.shape[1], Uc.shape[1], float(scale))


def rank_compose(sites):
    arr = (capi.RankSite * len(sites))(*sites)
    _call("clora_rank_compose_f32", arr, len(sites))


def rank_mix(sites, backward=False, device=None):
    """forward: Tq = Tc_l . M_l^T; backward: dTc_l = dTq . M_l plus the partial Gram sums -> returns the workspace holding them"""
    arr = (capi.RankSite * len(sites))(*sites)
    ws = None
    if backward:
        rows = max(s_.rows for s_ in sites)
        ws = torch.empty(capi.lib().cdll.clora_rank_gram_ws_bytes(rows, len(sites)) // 4, dtype=f32, device=device)
    _call("clora_rank_mix_f32", arr, len(sites), int(backward), ptr(ws) if ws is not None else None, ws.numel() * 4 if ws is not None else 0)
    return ws


def rank_compose_bwd(sites, gram_ws):
    arr = (capi.RankSite * len(sites))(*sites)
    _call("clora_rank_compose_bwd_f32", arr, len(sites), ptr(gram_ws))


def lora_down(X, D, T, toff, M, K, accumulate=False, x_rows=0, ldx=None, kmajor=False, R=None, d_scale=1.0):
    """T[:, toff:toff+R] (+)= X . (d_scale * D)^T ; X [rows, K] fp16 (row pitch ldx), T [M, ldt] fp32.
    D is [R, K] fp32, or with kmajor=True an up-projection matrix [K, R] used as its own transpose."""
    assert D.dtype == f32 and D.stride(1) == 1 and T.dtype == f32 and T.stride(1) == 1
    rank = R if R is not None else (D.shape[1] if kmajor else D.shape[0])
    _call("clora_lora_down_f16", ptr(X, f16), ldx if ldx is not None else X.stride(0), ptr(D), D.stride(0), ptr(T),
          T.stride(0), toff, M, K, rank, int(accumulate), x_rows, int(kmajor), float(d_scale), nbytes=2.0 * M * K)
    return T


def down_job(X, D, T, toff, M, K, accumulate=False, x_rows=0, ldx=None, kmajor=False, R=None, d_scale=1.0, X2=None, x2_rows=0, r2=0,
             T_in=None, t_in_r=0):
    """one problem of lora_down_multi (same arguments as lora_down; rank <= 16); X2: second input, T = (X + X2) . D^T;
    r2 > 0: X2 feeds only the first r2 rows of D (several adapters sharing X stacked into this one job);
    T_in (fp32 [rows, >= t_in_r]): added to the first t_in_r output columns (row m reads m % rows when rows != M)"""
    assert D.dtype == f32 and D.stride(1) == 1 and T.dtype == f32 and T.stride(1) == 1
    rank = R if R is not None else (D.shape[1] if kmajor else D.shape[0])
    tin_rows = 0
    if T_in is not None:
        assert T_in.dtype == f32 and T_in.stride(1) == 1 and t_in_r > 0
        tin_rows = T_in.shape[0] if T_in.shape[0] != M else 0
    return capi.LoraDownJob(ptr(X, f16), ldx if ldx is not None else X.stride(0), ptr(D), D.stride(0), ptr(T), T.stride(0), toff,
                            M, K, rank, int(accumulate), x_rows, int(kmajor), float(d_scale),
                            ptr(X2, f16) if X2 is not None else None, X2.stride(0) if X2 is not None else 0, x2_rows, int(r2),
                            ptr(T_in) if T_in is not None else None, T_in.stride(0) if T_in is not None else 0, tin_rows, int(t_in_r))


def up_job(base, T, toff, U, Y, M, N, scale, u_tr=False):
    """one problem of lora_up_multi: Y = fp16(base + scale * T[:, toff:toff+R] . U^T), one rounding (base None: fp16(scale * fp16(T . U^T)))"""
    assert U.dtype == f32 and U.stride(1) == 1 and T.dtype == f32 and Y.dtype == f16
    return capi.LoraUpJob(ptr(base, f16) if base is not None else None, base.stride(0) if base is not None else 0, ptr(T), T.stride(0),
                          toff, ptr(U), U.stride(0), int(u_tr), ptr(Y), Y.stride(0), M, N, U.shape[0] if u_tr else U.shape[1], float(scale))


def lora_up_multi(jobs):
    for i in range(0, len(jobs), capi.LORA_MAX_JOBS):
        chunk = jobs[i:i + capi.LORA_MAX_JOBS]
        arr = (capi.LoraUpJob * len(chunk))(*chunk)
        _call("clora_lora_up_multi_f16", arr, len(chunk), nbytes=sum(2.0 * j.M * j.N for j in chunk))


def lora_down_multi(jobs):
    """several adapter down-projections (possibly over different inputs) in one launch"""
    for i in range(0, len(jobs), capi.LORA_MAX_JOBS):
        chunk = jobs[i:i + capi.LORA_MAX_JOBS]
        arr = (capi.LoraDownJob * len(chunk))(*chunk)
        _call("clora_lora_down_multi_f16", arr, len(chunk), nbytes=sum(2.0 * j.M * j.K for j in chunk))


def wgrad_job(A, T, toff, G, gs_n, gs_j, M, N, R, scale=1.0, a_rows=0, lda=None, A2=None):
    """A2: the reduction runs over fp16(A + A2) -- an adapter whose input is a sum of two tensors"""
    assert G.dtype == f32 and T.dtype == f32 and R <= 16
    return capi.LoraWgradJob(ptr(A, f16), lda if lda is not None else A.stride(0), ptr(T), T.stride(0), toff, ptr(G), gs_n, gs_j,
                             M, N, R, float(scale), a_rows, ptr(A2, f16) if A2 is not None else None,
                             A2.stride(0) if A2 is not None else 0)


def lora_wgrad_multi(jobs, device):
    """several adapter weight-gradient reductions in one launch (+ one fold launch); jobs are grouped by rank class"""
    groups = {}
    for j in jobs:
        groups.setdefault(4 if j.R <= 4 else (8 if j.R <= 8 else 16), []).append(j)
    wsb = capi.lib().cdll.clora_lora_wgrad_workspace_bytes
    for _, js in groups.items():
        for i in range(0, len(js), capi.WGRAD_JOBS_PER_LAUNCH):
            chunk = js[i:i + capi.WGRAD_JOBS_PER_LAUNCH]
            ws = workspace(sum(wsb(j.M, j.N, j.R) for j in chunk), device)
            arr = (capi.LoraWgradJob * len(chunk))(*chunk)
            _call("clora_lora_wgrad_multi_f16", arr, len(chunk), ptr(ws), ws.numel(), nbytes=sum(2.0 * j.M * j.N for j in chunk))


# ---- deferred adapter weight gradients.  dU / dD are leaves of the backward pass (only the optimizer reads them), so
# the autograd functions queue their reduction jobs instead of launching them one site at a time; the queue is flushed
# ONCE at the end of the backward pass, 32 jobs per launch (16 until round 6): ~110 small launch pairs per step become ~10 large ones.
# Tensors the jobs read are kept alive until the flush.  Jobs that target the same gradient buffer never share a launch.
_wgrad_queue = {"jobs": [], "refs": [], "task": None, "enabled": os.environ.get("CLORA_DEFER_WGRAD", "1") != "0", "unpack": []}
DEFER_UNPACK = os.environ.get("CLORA_DEFER_UNPACK", "1") != "0"        # "0": one unpack launch per hint-encoder convolution (A/B runs)


def _graph_task_id():
    f = getattr(torch._C, "_current_graph_task_id", None)
    return f() if f is not None else -1


def _defer_open():
    """inside a backward pass with deferral on: make sure THIS pass has its end-of-backward flush registered -> True"""
    if not _wgrad_queue["enabled"] or PROFILER is not None:
        return False
    task = _graph_task_id()
    if task < 0:                   # not inside a backward pass: nothing to wait for
        return False
    if _wgrad_queue["task"] != task:
        # first deferral of THIS backward pass.  Anything still queued belongs to an earlier pass that raised before
        # its end-of-backward callback ran: those jobs point at freed tensors -- drop them, never launch them.
        lora_wgrad_discard()
        _wgrad_queue["task"] = task
        torch.autograd.Variable._execution_engine.queue_callback(lora_wgrad_flush)   # runs when this pass ends
    return True


def lora_wgrad_defer(jobs, device, *keepalive):
    if not _defer_open():
        lora_wgrad_multi(jobs, device)
        return
    _wgrad_queue["jobs"].extend(jobs)
    _wgrad_queue["refs"].extend(keepalive)
    _wgrad_queue["device"] = device


def conv_unpack_defer(job, *keepalive) -> bool:
    """queue the staging -> OIHW unpack of one hint-encoder convolution for the end-of-backward flush (one multi-job launch for all
    of them; the gradients are leaves of the pass: only the optimizer reads them) -> False when there is no pass to wait for"""
    if not DEFER_UNPACK or not _defer_open():
        return False
    _wgrad_queue["unpack"].append(job)
    _wgrad_queue["refs"].extend(keepalive)
    return True


def lora_wgrad_discard():
    """drop queued jobs without running them (a backward pass that raised must not leak its jobs into the next step)"""
    _wgrad_queue["jobs"].clear()
    _wgrad_queue["unpack"].clear()
    _wgrad_queue["refs"].clear()
    _wgrad_queue["task"] = None


def lora_wgrad_flush():
    jobs, _wgrad_queue["jobs"] = _wgrad_queue["jobs"], []
    unpack, _wgrad_queue["unpack"] = _wgrad_queue["unpack"], []
    _wgrad_queue["task"] = None
    if unpack:
        conv_wgrad_unpack_multi(unpack)
    if jobs:
        batches = []               # greedy packing: a batch never holds two jobs with the same destination
        for j in jobs:
            for b in batches:
                if len(b[0]) < capi.WGRAD_JOBS_PER_LAUNCH and j.G not in b[1] and b[2] == (4 if j.R <= 4 else (8 if j.R <= 8 else 16)):
                    b[0].append(j); b[1].add(j.G)
                    break
            else:
                batches.append(([j], {j.G}, 4 if j.R <= 4 else (8 if j.R <= 8 else 16)))
        for b in batches:
            lora_wgrad_multi(b[0], _wgrad_queue["device"])
    _wgrad_queue["refs"].clear()


def lora_up(base, T, toff, U, M, N, scale, out=None, u_tr=False):
    """Y = fp16(base + scale * T[:, toff:toff+R] . U^T), one rounding (base None: fp16(scale * fp16(T . U^T))); U is [N, R], or with u_tr a down matrix [R, N]."""
    assert U.dtype == f32 and U.stride(1) == 1 and T.dtype == f32
    y = out if out is not None else torch.empty((M, N), dtype=f16, device=T.device)
    _call("clora_lora_up_f16", ptr(base, f16) if base is not None else None, base.stride(0) if base is not None else 0,
          ptr(T), T.stride(0), toff, ptr(U), U.stride(0), int(u_tr), ptr(y), y.stride(0), M, N,
          U.shape[0] if u_tr else U.shape[1], float(scale),
          nbytes=2.0 * M * N * (2 if base is not None else 1))
    return y


def lora_wgrad(A, T, toff, G, gs_n, gs_j, M, N, R, scale=1.0, a_rows=0, lda=None):
    """G[n*gs_n + j*gs_j] += scale * sum_m A[m,n] T[m,toff+j]"""
    assert G.dtype == f32 and T.dtype == f32
    ws = workspace(capi.lib().cdll.clora_lora_wgrad_workspace_bytes(M, N, R), A.device)
    _call("clora_lora_wgrad_f16", ptr(A, f16), lda if lda is not None else A.stride(0), ptr(T), T.stride(0), toff, ptr(G),
          gs_n, gs_j, M, N, R, float(scale), a_rows, ptr(ws), ws.numel(), nbytes=2.0 * M * N)
    return G


def wgrad_accumulate(dY: torch.Tensor, X: torch.Tensor, M: int, N: int, K: int, G: torch.Tensor):
    """G[N, K] (fp32, contiguous) += dY[M, N]^T . X[M, K] on the MFMA weight-gradient kernel (clora_conv_wgrad_f16, plain rows): the
    weight gradient of a WIDE-rank adapter (rank > 16: danbooru-sketch's 256), whose matrices are GEMM-sized"""
    assert G.dtype == f32 and G.is_contiguous() and G.shape == (N, K) and dY.dtype == f16 and X.dtype == f16
    assert dY.stride(1) == 1 and X.stride(1) == 1
    _call("clora_conv_wgrad_f16", ptr(dY, f16), dY.stride(0), ptr(X, f16), X.stride(0), ptr(G), None, M, N, K, None, 0,
          flops=2.0 * M * N * K)


def to_f16(x: torch.Tensor) -> torch.Tensor:
    """fp32 master weight -> the fp16 GEMM operand of this step"""
    assert x.dtype == f32 and x.is_contiguous()
    y = torch.empty(x.shape, dtype=f16, device=x.device)
    _call("clora_cast_f32_to_f16", ptr(x), ptr(y), x.numel())
    return y


# ------------------------------------------------------------------ elementwise / movement
def add(a, b):
    y = torch.empty_like(a)
    _call("clora_add_f16", ptr(a, f16), ptr(b, f16), ptr(y), a.numel())
    return y


def timestep_embedding(timestep, batch, freq):
    """[batch, 2 * len(freq)] fp16 = cat(cos(t f), sin(t f)); timestep: int64 or fp32, `batch` values or one"""
    t = timestep.reshape(-1)
    if t.dtype not in (torch.int64, f32):
        t = t.float() if t.is_floating_point() else t.long()
    t = t.contiguous()
    assert t.numel() in (1, batch) and freq.dtype == f32 and freq.is_contiguous()
    out = torch.empty((batch, 2 * freq.numel()), dtype=f16, device=freq.device)
    _call("clora_timestep_embedding_f16", ptr(t), int(t.dtype == torch.int64), t.numel(), ptr(freq, f32), ptr(out), batch, freq.numel())
    return out


def silu(x):
    y = torch.empty_like(x)
    _call("clora_silu_f16", ptr(x, f16), ptr(y), x.numel())
    return y


def quick_gelu(x):
    y = torch.empty_like(x)
    _call("clora_quick_gelu_f16", ptr(x, f16), ptr(y), x.numel())
    return y


def silu_bwd(x, dy):
    dx = torch.empty_like(x)
    _call("clora_silu_bwd_f16", ptr(x, f16), ptr(dy, f16), ptr(dx), x.numel())
    return dx


def copy2d(src, lds, dst, ldd, M, N):
    _call("clora_copy2d_f16", ptr(src, f16), lds, ptr(dst, f16), ldd, M, N)
    return dst


def concat_channels(a, b):
    """[.., Ca] ++ [.., Cb] along the channel (last) dim."""
    Ca, Cb = a.shape[-1], b.shape[-1]
    M = a.numel() // Ca
    y = torch.empty(a.shape[:-1] + (Ca + Cb,), dtype=f16, device=a.device)
    copy2d(a, Ca, y, Ca + Cb, M, Ca)
    copy2d(b, Cb, y[..., Ca:], Ca + Cb, M, Cb)
    return y


def split_channels(y, Ca):
    Ct = y.shape[-1]
    M = y.numel() // Ct
    a = torch.empty(y.shape[:-1] + (Ca,), dtype=f16, device=y.device)
    b = torch.empty(y.shape[:-1] + (Ct - Ca,), dtype=f16, device=y.device)
    copy2d(y, Ct, a, Ca, M, Ca)
    copy2d(y[..., Ca:], Ct, b, Ct - Ca, M, Ct - Ca)
    return a, b


def pool2x2_sum(dy, B, H, W, Cc):
    dx = torch.empty((B, H * W, Cc), dtype=f16, device=dy.device)
    _call("clora_pool2x2_sum_f16", ptr(dy, f16), ptr(dx), B, H, W, Cc)
    return dx


def colsum(A, M, N, out=None):
    o = out if out is not None else torch.zeros(N, dtype=f32, device=A.device)
    _call("clora_colsum_f16", ptr(A, f16), A.stride(0) if A.dim() == 2 else N, ptr(o), M, N)
    return o


def mse(pred, target, loss_sum, dpred, grad_scale, loss_scale=None):
    _call("clora_mse_f16", ptr(pred, f16), ptr(target, f16), ptr(loss_sum, f32), ptr(dpred, f16) if dpred is not None else None,
          pred.numel(), float(grad_scale), ptr(loss_scale, f32) if loss_scale is not None else None)


def grad_sumsq(g, state):
    _call("clora_grad_sumsq_f32", ptr(g, f32), g.numel(), ptr(state, f32))


def optim_prep(state, max_norm, beta1, beta2, dynamic, growth=2.0, backoff=0.5, interval=2000):
    _call("clora_optim_prep_f32", ptr(state, f32), float(max_norm), float(beta1), float(beta2), int(dynamic), float(growth),
          float(backoff), int(interval))


def adamw_flat(p, g, m, v, state, lr, beta1, beta2, eps, wd):
    _call("clora_adamw_flat_f32", ptr(p, f32), ptr(g, f32), ptr(m, f32), ptr(v, f32), p.numel(), ptr(state, f32), float(lr),
          float(beta1), float(beta2), float(eps), float(wd))
