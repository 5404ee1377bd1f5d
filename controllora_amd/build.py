"""Build libclora.so (gfx950) in-tree with hipcc.  Explicit commands, no cmake:
``python -m controllora_amd.build``.  The built library lives in ``controllora_amd/_build/`` (git-ignored,
but it travels to the GPU box with the gpurun snapshot)."""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT_DIR = os.path.join(HERE, "_build")
LIB = os.path.join(OUT_DIR, "libclora.so")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    """Serialised across processes (file lock): concurrent first builds must not link half-written objects."""
    import fcntl
    os.makedirs(OUT_DIR, exist_ok=True)
    with open(os.path.join(OUT_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build_locked(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(force: bool, verbose: bool) -> str:
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(os.path.dirname(HERE), "include", "clora.h")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, jobs = [], []
    for src in sources():
        obj = os.path.join(OUT_DIR, os.path.basename(src)[:-4] + ".o")
        if force or _stale(obj, deps):
            jobs.append([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", obj])
        objs.append(obj)
    if jobs:                                     # independent translation units: compile them side by side (clora_gemm.hip alone takes ~2 min)
        from concurrent.futures import ThreadPoolExecutor

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as ex:
            list(ex.map(run, jobs))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
