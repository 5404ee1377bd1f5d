"""ctypes binding of the C ABI in ``include/clora.h`` (libclora.so, gfx950).

This is the ONLY way the Python host code reaches the GPU kernels.  There is no CPU or PyTorch
fallback: if the library was not built (``python -m controllora_amd.build``) importing the ops raises,
and every wrapper refuses tensors that are not on a HIP device.  PyTorch is used for device memory
and streams only -- wrappers pass ``tensor.data_ptr()`` and ``torch.cuda.current_stream().cuda_stream``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CLORA_LIB_PATH") or os.path.join(_HERE, "_build", "libclora.so")   # override: A/B of kernel builds

ABI_VERSION = 4                          # CLORA_ABI_VERSION of include/clora.h
OK, ERR_ARG, ERR_LAUNCH, ERR_WORKSPACE = 0, -1, -2, -3
_ERR = {ERR_ARG: "bad argument", ERR_LAUNCH: "kernel launch failed", ERR_WORKSPACE: "workspace too small"}


class CloraError(RuntimeError):
    pass


class ConvDesc(C.Structure):
    """mirror of clora_conv_t"""
    _fields_ = [(n, C.c_int) for n in (
        "enabled", "Hin", "Win", "Cin", "Hout", "Wout", "ksize", "mul", "kmul", "off", "lim_h", "lim_w", "shift",
        "need_even", "kchunk")]


class Epilogue(C.Structure):
    """mirror of clora_epilogue_t"""
    _fields_ = [("bias", C.c_void_p), ("rowadd", C.c_void_p), ("rows_per_batch", C.c_int), ("ld_rowadd", C.c_int),
                ("residual", C.c_void_p), ("ldr", C.c_int), ("lora_t", C.c_void_p), ("ldt", C.c_int),
                ("lora_u", C.c_void_p), ("ldu", C.c_int), ("lora_u_tr", C.c_int), ("lora_r", C.c_int), ("lora_seg", C.c_int), ("lora_scale", C.c_float),
                ("geglu", C.c_int), ("geglu_f", C.c_int), ("geglu_h", C.c_void_p), ("geglu_y", C.c_void_p),
                ("lora_dpack", C.c_void_p), ("lora_t_in", C.c_void_p), ("ldt_in", C.c_int), ("lora_t_in_rows", C.c_int),
                ("lora_t_in_mask", C.c_uint), ("defer", C.c_void_p),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_out", C.c_void_p), ("ln_eps", C.c_float),
                ("residual_lo", C.c_void_p), ("c_lo", C.c_void_p)]                                              # compensated trunk


class Deferred(C.Structure):
    """mirror of clora_deferred_t: a split-K GEMM whose finish pass is left to the consumer of its output"""
    _fields_ = [("partial", C.c_void_p), ("splits", C.c_int), ("M", C.c_int), ("N", C.c_int), ("C", C.c_void_p), ("ldc", C.c_int),
                ("epi", Epilogue)]


class LoraPackJob(C.Structure):
    """mirror of clora_lora_pack_job_t"""
    _fields_ = [("D", C.c_void_p), ("out", C.c_void_p), ("ldd", C.c_int), ("R", C.c_int), ("K", C.c_int), ("kmajor", C.c_int),
                ("scale", C.c_float), ("pad_", C.c_int)]


class LoraDownJob(C.Structure):
    """mirror of clora_lora_down_job_t"""
    _fields_ = [("X", C.c_void_p), ("ldx", C.c_int), ("D", C.c_void_p), ("ldd", C.c_int), ("T", C.c_void_p), ("ldt", C.c_int),
                ("toff", C.c_int), ("M", C.c_int), ("K", C.c_int), ("R", C.c_int), ("accumulate", C.c_int), ("x_rows", C.c_int),
                ("d_kmajor", C.c_int), ("d_scale", C.c_float), ("X2", C.c_void_p), ("ldx2", C.c_int), ("x2_rows", C.c_int),
                ("r2", C.c_int), ("T_in", C.c_void_p), ("ldt_in", C.c_int), ("t_in_rows", C.c_int), ("t_in_r", C.c_int)]


class RankSite(C.Structure):
    """mirror of clora_rank_site_t"""
    _fields_ = [("Dq", C.c_void_p), ("lddq", C.c_int), ("Uc", C.c_void_p), ("lduc", C.c_int), ("M", C.c_void_p),
                ("Tc", C.c_void_p), ("ldtc", C.c_int), ("toff", C.c_int), ("Tq", C.c_void_p), ("ldtq", C.c_int),
                ("dTc", C.c_void_p), ("lddtc", C.c_int), ("gDq", C.c_void_p), ("gUc", C.c_void_p),
                ("rows", C.c_int), ("C", C.c_int), ("rc", C.c_int), ("scale", C.c_float)]


class LoraUpJob(C.Structure):
    """mirror of clora_lora_up_job_t"""
    _fields_ = [("base", C.c_void_p), ("ldb", C.c_int), ("T", C.c_void_p), ("ldt", C.c_int), ("toff", C.c_int), ("U", C.c_void_p),
                ("ldu", C.c_int), ("u_transposed", C.c_int), ("Y", C.c_void_p), ("ldy", C.c_int), ("M", C.c_int), ("N", C.c_int),
                ("R", C.c_int), ("scale", C.c_float)]


class LoraWgradJob(C.Structure):
    """mirror of clora_lora_wgrad_job_t"""
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int), ("T", C.c_void_p), ("ldt", C.c_int), ("toff", C.c_int), ("G", C.c_void_p),
                ("gs_n", C.c_int), ("gs_j", C.c_int), ("M", C.c_int), ("N", C.c_int), ("R", C.c_int), ("scale", C.c_float),
                ("a_rows", C.c_int), ("A2", C.c_void_p), ("lda2", C.c_int)]


class ConvPackJob(C.Structure):
    """mirror of clora_conv_pack_job_t"""
    _fields_ = [("w", C.c_void_p), ("fwd", C.c_void_p), ("dgrad", C.c_void_p), ("Co", C.c_int), ("Ci", C.c_int), ("ksize", C.c_int),
                ("Cip", C.c_int), ("Cop", C.c_int), ("pad_", C.c_int)]


class ConvUnpackJob(C.Structure):
    """mirror of clora_conv_unpack_job_t"""
    _fields_ = [("stage", C.c_void_p), ("stage_b", C.c_void_p), ("grad_w", C.c_void_p), ("grad_b", C.c_void_p), ("Co", C.c_int),
                ("Ci", C.c_int), ("ksize", C.c_int), ("Cip", C.c_int)]


LORA_MAX_JOBS = 16
LORA_WGRAD_MAX_JOBS = 32       # CLORA_LORA_WGRAD_MAX_JOBS
WGRAD_JOBS_PER_LAUNCH = min(LORA_WGRAD_MAX_JOBS, max(1, int(os.environ.get("CLORA_WGRAD_JOBS", "32"))))   # A/B runs: 16 = the round-5 batching
CONV_MAX_JOBS = 32
_P, _I, _Z, _F = C.c_void_p, C.c_int, C.c_size_t, C.c_float
_PROTOS = {
    "clora_gemm_f16": [_P, _I, _P, _P, _I, _I, _I, _I, C.POINTER(ConvDesc), C.POINTER(Epilogue), _I, _P, _Z, _P],
    "clora_gemm_f16_ex": [_P, _I, _P, _P, _I, _I, _I, _I, C.POINTER(ConvDesc), C.POINTER(Epilogue), _I, _I, _P, _Z, _P],
    "clora_conv_patch_eligible": [_I, C.POINTER(ConvDesc), _I],
    "clora_conv_strip_eligible": [_I, _I, C.POINTER(ConvDesc)],
    "clora_set_option": [C.c_char_p, _I],
    "clora_conv_wgrad_f16": [_P, _I, _P, _I, _P, _P, _I, _I, _I, C.POINTER(ConvDesc), _I, _P],
    "clora_conv_weight_pack_f32": [_P, _I, _I, _I, _I, _I, _P, _P, _P],
    "clora_conv_wgrad_unpack_f32": [_P, _P, _P, _P, _I, _I, _I, _I, _P],
    "clora_conv_weight_pack_multi_f32": [C.POINTER(ConvPackJob), _I, _P],
    "clora_conv_wgrad_unpack_multi_f32": [C.POINTER(ConvUnpackJob), _I, _P],
    "clora_attn_fwd_f16": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F, _P],
    "clora_attn_fwd_causal_f16": [_P, _I, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _F, _P],
    "clora_attn_bwd_f16": [_P, _I, _P, _I, _P, _I, _P, _I, _P, _I, _P, _P, _P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _F, _P, _Z, _P],
    "clora_groupnorm_fwd_f16": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _Z, _P],
    "clora_groupnorm_bwd_f16": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P],
    "clora_groupnorm_fwd_f16_ex": [_P, _P, _I, C.POINTER(Deferred), _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _Z, _P],
    "clora_groupnorm_bwd_f16_ex": [_P, _P, C.POINTER(Deferred), _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P],
    "clora_groupnorm_fwd_f16_team": [_P, _P, _I, C.POINTER(Deferred), _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _P, _Z, _P, _Z, _P],
    "clora_groupnorm_bwd_f16_team": [_P, _P, C.POINTER(Deferred), _P, _P, _P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _Z, _P, _Z, _P],
    "clora_layernorm_bwd_f16_ex": [_P, _P, C.POINTER(Deferred), _P, _P, _P, _I, _I, _F, _P],
    "clora_finish_deferred": [C.POINTER(Deferred), _P],
    "clora_gemm_ln_fusable": [_I, _I, _I, _I, _I],
    "clora_layernorm_fwd_f16": [_P, _P, _P, _P, _I, _I, _F, _P],
    "clora_softmax_rows_f16": [_P, _P, _I, _I, _I, _F, _P],
    "clora_layernorm_bwd_f16": [_P, _P, _P, _P, _P, _I, _I, _F, _P],
    "clora_geglu_fwd_f16": [_P, _P, _I, _I, _P],
    "clora_geglu_bwd_f16": [_P, _P, _P, _I, _I, _P],
    "clora_lora_down_f16": [_P, _I, _P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _P],
    "clora_lora_down_multi_f16": [C.POINTER(LoraDownJob), _I, _P],
    "clora_lora_wgrad_multi_f16": [C.POINTER(LoraWgradJob), _I, _P, _Z, _P],
    "clora_lora_pack_f16": [_P, _I, _I, _P],
    "clora_rank_compose_f32": [C.POINTER(RankSite), _I, _P],
    "clora_rank_mix_f32": [C.POINTER(RankSite), _I, _I, _P, _Z, _P],
    "clora_rank_compose_bwd_f32": [C.POINTER(RankSite), _I, _P, _P],
    "clora_lora_up_f16": [_P, _I, _P, _I, _I, _P, _I, _I, _P, _I, _I, _I, _I, _F, _P],
    "clora_lora_up_multi_f16": [C.POINTER(LoraUpJob), _I, _P],
    "clora_lora_wgrad_f16": [_P, _I, _P, _I, _I, _P, _I, _I, _I, _I, _I, _F, _I, _P, _Z, _P],
    "clora_comm_unique_id": [_P],
    "clora_comm_init": [_P, _I, _I],
    "clora_comm_library": [C.c_char_p, _Z],
    "clora_comm_world": [],
    "clora_comm_rank": [],
    "clora_allreduce_flat_f32": [_P, _Z, _P],
    "clora_comm_destroy": [],
    "clora_add_f16": [_P, _P, _P, _Z, _P],
    "clora_silu_f16": [_P, _P, _Z, _P],
    "clora_timestep_embedding_f16": [_P, _I, _I, _P, _P, _I, _I, _P],
    "clora_silu_bwd_f16": [_P, _P, _P, _Z, _P],
    "clora_quick_gelu_f16": [_P, _P, _Z, _P],
    "clora_copy2d_f16": [_P, _I, _P, _I, _Z, _I, _P],
    "clora_pool2x2_sum_f16": [_P, _P, _I, _I, _I, _I, _P],
    "clora_colsum_f16": [_P, _I, _P, _I, _I, _P],
    "clora_mse_f16": [_P, _P, _P, _P, _Z, _F, _P, _P],
    "clora_cast_f32_to_f16": [_P, _P, _Z, _P],
    "clora_cast_f16_to_f32": [_P, _P, _Z, _P],
    "clora_grad_sumsq_f32": [_P, _Z, _P, _P],
    "clora_optim_prep_f32": [_P, _F, _F, _F, _I, _F, _F, _I, _P],
    "clora_adamw_flat_f32": [_P, _P, _P, _P, _Z, _P, _F, _F, _F, _F, _F, _P],
    "clora_abi_version": [],
    "clora_clock_probe": [_P, _I, _I, _P],
    "clora_groupnorm_workspace_bytes": [_I, _I, _I, _I, _I, _I],
    "clora_groupnorm_team_state_bytes": [],
    "clora_lora_wgrad_workspace_bytes": [_I, _I, _I],
    "clora_rank_gram_ws_bytes": [_I, _I],
}


class Lib:
    """A loaded libclora with typed prototypes."""

    def __init__(self, path: str = LIB_PATH, require_device: bool = True):
        if not os.path.exists(path):
            raise CloraError(
                f"{path} not found: the gfx950 kernel library is not built.  Run `python -m controllora_amd.build` "
                "(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for this path.")
        self.path = path
        self.require_device = require_device
        self.cdll = C.CDLL(path)
        # struct layouts are part of the ABI: a library of another version would read fields this binding does not send
        got = self.cdll.clora_abi_version() if hasattr(self.cdll, "clora_abi_version") else -1
        if got != ABI_VERSION and os.environ.get("CLORA_ABI_ANY", "") != "1":
            raise CloraError(f"{path} reports ABI {got}, this binding is written for ABI {ABI_VERSION} (include/clora.h); rebuild with "
                             "`python -m controllora_amd.build` (CLORA_ABI_ANY=1 overrides for A/B runs of older builds whose "
                             "struct layouts are known to match)")
        for name, argtypes in _PROTOS.items():
            if not hasattr(self.cdll, name) and os.environ.get("CLORA_ABI_ANY", "") == "1":
                continue                             # an older A/B build may lack the newest entry points
            fn = getattr(self.cdll, name)          # AttributeError (loud) if a symbol is missing
            fn.argtypes = argtypes
            fn.restype = C.c_int
        self.cdll.clora_build_info.restype = C.c_char_p
        self.cdll.clora_groupnorm_workspace_bytes.restype = C.c_size_t
        self.cdll.clora_groupnorm_team_state_bytes.restype = C.c_size_t
        self.cdll.clora_lora_wgrad_workspace_bytes.restype = C.c_size_t
        self.cdll.clora_rank_gram_ws_bytes.restype = C.c_size_t
        self._options_from_env()

    # The library reads no environment variable; A/B runs set its knobs (clora_set_option, the ABI's single piece of
    # process-global state) through these variables, forwarded here when the library is loaded.
    _ENV_OPTIONS = {"CLORA_TILE_ORDER": ("tile_order", {"m": 0, "n": 1, "auto": 2, "a": 2, "grid": 3, "g": 3}), "CLORA_LN_ROWS": ("ln_rows", None),
                    "CLORA_ATTN_FWD_WAVES": ("attn_fwd_waves", None), "CLORA_ATTN_BWD_WAVES": ("attn_bwd_waves", None),
                    "CLORA_GN_BLOCKS": ("gn_blocks", None), "CLORA_EPI_TWO_PHASE": ("epi_two_phase", None),
                    "CLORA_LORA_DOWN_MODE": ("lora_down_mode", None), "CLORA_GN_UNROLL": ("gn_unroll", None),
                    "CLORA_EPI_HOIST": ("epi_hoist", None), "CLORA_GN_RESIDENT": ("gn_resident", None),
                    "CLORA_DEFER_MAX_ROWS": ("defer_max_rows", None), "CLORA_WGRAD_PATCH": ("wgrad_patch", None), "CLORA_STRIP_BLOCKS": ("strip_blocks", None),
                    "CLORA_GN_TEAM": ("gn_team", None)}

    def _options_from_env(self):
        for var, (name, names) in self._ENV_OPTIONS.items():
            v = os.environ.get(var)
            if v is None or v == "":
                continue
            try:
                val = names[v] if (names and v in names) else int(v)
                self.call("clora_set_option", name.encode(), val)
            except (ValueError, CloraError) as e:
                allowed = f"one of {sorted(names)} or an integer" if names else "an integer in the range include/clora.h documents"
                raise CloraError(f"environment variable {var}={v!r} is not a valid value for the library option "
                                 f"{name!r} (expected {allowed})") from e

    def call(self, name: str, *args) -> None:
        rc = getattr(self.cdll, name)(*args)
        if rc != OK:
            raise CloraError(f"{name} failed: {_ERR.get(rc, rc)}")


_LIB: Optional[Lib] = None


def lib() -> Lib:
    global _LIB
    if _LIB is None:
        _LIB = Lib()
    return _LIB


def ptr(t: Optional[torch.Tensor], dtype=None) -> Optional[int]:
    if t is None:
        return None
    L = lib()
    if L.require_device and not t.is_cuda:
        raise CloraError("libclora kernels need tensors on a HIP device (no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise CloraError(f"expected {dtype}, got {t.dtype}")
    return t.data_ptr()


def stream() -> Optional[int]:
    if lib().require_device:
        return torch.cuda.current_stream().cuda_stream
    return None
