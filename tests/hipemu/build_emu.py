"""TEST INFRASTRUCTURE: compile the SAME kernel sources for the host fiber emulator
(tests/hipemu/hipemu.h) -> tests/hipemu/_build/libclora_emu.so.  Used only by CPU tests."""
from __future__ import annotations

import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "controllora_amd", "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libclora_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _host_isa_flags():
    """F16C turns the emulated MFMA's fp16->fp32 conversions from library calls into one instruction (3.3x faster
    emulated step); only used when this host advertises it."""
    try:
        flags = open("/proc/cpuinfo").read()
    except OSError:
        return []
    return ["-mf16c", "-mavx2", "-mfma"] if all(f" {f}" in flags for f in ("f16c", "avx2", "fma")) else []


def build_mutant(out_dir: str, file_name: str, edits) -> str:
    """The emulator library with some source lines of ``csrc/<file_name>`` replaced (``edits`` = [(old, new)], each ``old`` must
    occur exactly once): used by the
    tests that prove the emulator's pessimistic LDS-DMA model catches a weakened ``s_waitcnt`` (tests/test_emu_async_model.py)."""
    import shutil
    src_dir = os.path.join(out_dir, "controllora_amd", "csrc")      # same depth as the real tree: the sources include
    shutil.copytree(CSRC, src_dir)                                  # "../../include/clora.h"
    shutil.copytree(os.path.join(ROOT, "include"), os.path.join(out_dir, "include"))
    path = os.path.join(src_dir, file_name)
    text = open(path).read()
    for old, new in edits:
        if text.count(old) != 1:
            raise ValueError(f"{file_name}: mutation anchor {old!r} occurs {text.count(old)} times")
        text = text.replace(old, new)
    open(path, "w").write(text)
    objs = []
    for src in sorted(glob.glob(os.path.join(src_dir, "*.hip"))):
        base = os.path.basename(src).rsplit(".", 1)[0] + ".emu.o"
        if os.path.basename(src) != file_name and os.path.exists(os.path.join(OUT_DIR, base)):
            objs.append(os.path.join(OUT_DIR, base))          # untouched translation units: reuse the regular build
            continue
        obj = os.path.join(out_dir, base)
        subprocess.check_call([CLANG, "-O2", "-g0", *_host_isa_flags(), "-std=c++17", "-fPIC", "-I", HERE,
                               "-Wno-unknown-pragmas", "-Wno-pass-failed", "-x", "c++", "-c", src, "-o", obj])
        objs.append(obj)
    objs.append(os.path.join(OUT_DIR, "hipemu.emu.o"))
    lib = os.path.join(out_dir, "libclora_emu_mutant.so")
    subprocess.check_call([CLANG, "-shared", "-fPIC", *objs, "-o", lib])
    return lib


def build(verbose: bool = False) -> str:
    """Serialised across processes by an exclusive file lock: the two gloo ranks of tests/test_distributed_cpu.py (or
    pytest-xdist workers) on a fresh clone would otherwise compile the same objects concurrently and one of them would
    link / dlopen a half-written file (VERDICT r02 "fresh-clone race")."""
    import fcntl
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> str:
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    deps = srcs + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "*.h")) + \
        glob.glob(os.path.join(HERE, "*.cpp")) + [os.path.join(ROOT, "include", "clora.h")]
    newest = max(os.path.getmtime(d) for d in deps)
    objs = []
    for src in srcs + [os.path.join(HERE, "hipemu.cpp")]:
        obj = os.path.join(OUT_DIR, os.path.basename(src).rsplit(".", 1)[0] + ".emu.o")
        if not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            cmd = [CLANG, "-O2", "-g0", *_host_isa_flags(), "-std=c++17", "-fPIC", "-I", HERE, "-Wno-unknown-pragmas", "-Wno-pass-failed",
                   "-x", "c++", "-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(obj)
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < max(os.path.getmtime(o) for o in objs):
        subprocess.check_call([CLANG, "-shared", "-fPIC", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True))
