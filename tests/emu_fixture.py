"""TEST INFRASTRUCTURE: route controllora_amd.capi to the host-emulated kernel library
(tests/hipemu) so kernel logic and the Python host code can be exercised without a GPU.
Never imported by the product package."""
import contextlib

from controllora_amd import capi
from tests.hipemu import build_emu

_EMU = None


def emu_lib():
    global _EMU
    if _EMU is None:
        _EMU = capi.Lib(build_emu.build(), require_device=False)
    return _EMU


@contextlib.contextmanager
def use_emulator():
    old = capi._LIB
    capi._LIB = emu_lib()
    try:
        yield
    finally:
        capi._LIB = old
