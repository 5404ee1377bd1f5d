"""CPU test: the gfx950 library built by __graft_entry__.build() loads and exports every symbol that
include/clora.h declares (no compute calls -- there is no GPU here), and the product binding refuses to
run without it / on CPU tensors (no silent fallback)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "clora.h")).read()
    return sorted(set(re.findall(r"\b(clora_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from controllora_amd import build
    lib = build.build(verbose=False)
    cdll = ctypes.CDLL(lib)
    syms = _declared_symbols()
    assert len(syms) >= 25
    missing = [s for s in syms if not hasattr(cdll, s)]
    assert not missing, missing
    assert cdll.clora_abi_version() == 4


def test_binding_covers_the_header():
    from controllora_amd import capi
    declared = set(_declared_symbols()) - {"clora_build_info", "clora_gemm_f16"}
    assert declared <= set(capi._PROTOS), declared - set(capi._PROTOS)


def test_no_cpu_fallback():
    from controllora_amd import capi, kernels as K
    with pytest.raises(capi.CloraError):
        capi.Lib(os.path.join(ROOT, "controllora_amd", "_build", "does_not_exist.so"))
    if not torch.cuda.is_available():
        x = torch.zeros(8, 8, dtype=torch.float16)
        with pytest.raises(capi.CloraError):          # product library + CPU tensors -> loud error, not a fallback
            K.silu(x)


def test_workspace_is_never_freed_or_moved_under_a_graph():
    """a captured hipGraph holds the scratch address: growing the workspace must keep the old buffer alive
    (regression: batch-32 inference after a captured train step grew it and the replayed step faulted)"""
    import torch
    from controllora_amd import kernels as K
    a = K.workspace(1000, "cpu")
    pa = a.data_ptr()
    b = K.workspace(a.numel() + 4096, "cpu")
    assert b.numel() >= a.numel() + 4096 and any(t.data_ptr() == pa for t in K._ws_retired)
    assert K.workspace(10, "cpu") is b                       # never shrinks


def test_options_are_the_only_global_state_and_no_getenv_in_the_library(monkeypatch):
    """clora_set_option validates names / ranges (host-only call: no kernel runs); the kernel sources read no environment
    variable -- CLORA_* variables are forwarded by the host binding when the library is loaded."""
    import glob
    from controllora_amd import build, capi
    for src in glob.glob(os.path.join(ROOT, "controllora_amd", "csrc", "*")):
        assert "getenv" not in open(src).read(), src
    monkeypatch.setenv("CLORA_TILE_ORDER", "auto")
    monkeypatch.setenv("CLORA_GN_BLOCKS", "256")
    L = capi.Lib(build.build(verbose=False), require_device=False)            # forwards the two variables
    so = L.cdll.clora_set_option
    assert so(b"tile_order", 0) == 0 and so(b"tile_order", 3) == 0 and so(b"tile_order", 4) == capi.ERR_ARG and so(b"tile_order", 2) == 0
    assert so(b"attn_fwd_waves", 6) == 0 and so(b"attn_fwd_waves", 16) == 0 and so(b"attn_fwd_waves", 5) == capi.ERR_ARG and so(b"attn_fwd_waves", 0) == 0
    assert so(b"attn_bwd_waves", 6) == capi.ERR_ARG and so(b"gn_blocks", 8) == capi.ERR_ARG and so(b"gn_blocks", 512) == 0
    assert so(b"no_such_knob", 1) == capi.ERR_ARG and so(None, 1) == capi.ERR_ARG
    monkeypatch.setenv("CLORA_TILE_ORDER", "sideways")
    with pytest.raises((ValueError, capi.CloraError)):
        capi.Lib(build.build(verbose=False), require_device=False)
