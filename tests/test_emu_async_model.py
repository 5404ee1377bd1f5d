"""The host emulator's LDS-DMA model has teeth: an asynchronous copy lands only when its wave executes a wait that covers it
(tests/hipemu/hipemu.h), so a kernel whose ``s_waitcnt vmcnt(N)`` lets one stage too many stay in flight must FAIL its parity
case here.  The emulator library is rebuilt once from a copy of the kernel sources with the counted waits of two different
kernels weakened, and the cases that pass on the real sources (tests/test_kernels_emu.py) must now break -- the guarantee
behind "the vmcnt protocol of every LDS-DMA kernel is checked on the CPU" in DESIGN.md section 2.  (The third counted wait, the
patch kernel's halo wait ``(NST - 2) * B_IN + PA_IN``, was mutated the same way when this test was written and is caught too;
it shares the kernel with ``patch_weights`` so it is not part of the single mutant build.)"""
import contextlib

import pytest

from controllora_amd import capi
from tests import kernel_cases as KC
from tests.hipemu import build_emu

EDITS = [
    # GEMM ring: the wait leaves NST-1 stages in flight instead of NST-2 -> stage kt is read before it landed
    ("CLORA_WAIT_VMCNT((NST - 2) * (A_IN + B_IN));", "CLORA_WAIT_VMCNT((NST - 1) * (A_IN + B_IN));"),
    # patch conv: the weight ring's steady-state wait one stage short
    ("else CLORA_WAIT_VMCNT((NST - 2) * B_IN);", "else CLORA_WAIT_VMCNT((NST - 1) * B_IN);"),
    # eight-phase GEMM: the K-tile's one counted wait leaves four half-tiles in flight instead of three -> A1 of the next K-tile is read early
    ("CLORA_WAIT_VMCNT(kInFlight8p);", "CLORA_WAIT_VMCNT(kInFlight8p + 2);"),
]
CASES = {
    "gemm_ring": lambda: KC.case_gemm_plain("cpu", 150, 72, 104, 1, tile_cfg=21),
    "patch_weights": lambda: KC.case_conv_patch("cpu", 2, 8, 8, 128, 64, 72),
    "gemm_8phase": lambda: KC.case_gemm_plain("cpu", 150, 72, 424, 1, tile_cfg=59),
}


@contextlib.contextmanager
def _use(lib_path):
    old = capi._LIB
    capi._LIB = capi.Lib(lib_path, require_device=False)
    try:
        yield
    finally:
        capi._LIB = old


@pytest.fixture(scope="module")
def mutant_lib(tmp_path_factory):
    build_emu.build()                                   # the regular objects the mutant build reuses
    return build_emu.build_mutant(str(tmp_path_factory.mktemp("mutant")), "clora_gemm.hip", EDITS)


@pytest.mark.parametrize("name", sorted(CASES))
def test_weakened_wait_is_caught(name, mutant_lib):
    with _use(mutant_lib):
        with pytest.raises(AssertionError):
            CASES[name]()
    with _use(build_emu.build()):
        CASES[name]()                                   # and the unmodified sources pass the very same case


# ---- experiment build of the GEMM epilogue (CLORA_EPI_SINGLE_PASS: the whole tile staged to LDS in one pass) gives the same results
@pytest.fixture(scope="module")
def single_pass_lib(tmp_path_factory):
    build_emu.build()
    return build_emu.build_mutant(str(tmp_path_factory.mktemp("epi1")), "clora_gemm.hip", [("#ifdef CLORA_EPI_SINGLE_PASS", "#if 1")])


@pytest.mark.parametrize("tile", [3, 23, 43, 53, 72, 76])
def test_single_pass_epilogue_variant(tile, single_pass_lib):
    with _use(single_pass_lib):
        if tile < 70:
            KC.case_gemm_plain("cpu", 150, 72, 104, 1, tile_cfg=tile)
            KC.case_gemm_epilogue("cpu", split_k=1, tile_cfg=tile)
            KC.case_conv("cpu", 1, 8, 8, 16, 24, tile_cfg=tile)
        else:
            KC.case_conv_patch("cpu", 2, 8, 8, 128, 64, tile)
