"""Diagnostic for the hoisted adapter epilogue: run the short-K projection shape through one tile variant of
the library named by CLORA_LIB_PATH and report WHERE the wrong elements sit in hardware terms -- lane of the wave, element of the
8-column chunk (= which hoisted register), row sweep `it`, wave of the block.

    CLORA_LIB_PATH=controllora_amd/_build_v_<name>/libclora.so python tools/hoist_diag.py <label> [tile ...]
"""
import collections
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controllora_amd import kernels as K

dev = "cuda"
f16, f32 = torch.float16, torch.float32
label = sys.argv[1] if len(sys.argv) > 1 else "default"
tiles = [int(a) for a in sys.argv[2:]] or [42, 22]
TILE_GEOM = {42: (128, 64), 22: (128, 64), 26: (128, 64), 21: (128, 128), 41: (128, 128), 23: (64, 64), 43: (64, 64), 55: (64, 320), 52: (64, 320)}
REPS = int(os.environ.get("HOIST_DIAG_REPS", "8"))


def locate(out, ref, BM, BN, NT=256):
    ref = ref.float().cpu()
    d = (out.float().cpu() - ref).abs()
    bad = d > 0.02 * ref.abs().max()
    n = int(bad.sum())
    if n == 0:
        return 0, ""
    r, c = bad.nonzero(as_tuple=True)
    CPR = BN // 8
    RPIT = NT // CPR
    lanes, es, its, waves = collections.Counter(), collections.Counter(), collections.Counter(), collections.Counter()
    for rr, cc in zip(r.tolist(), c.tolist()):
        mlp = (rr % BM) % 64                      # row inside the 64-row staging pass
        nc, e = (cc % BN) // 8, cc % 8
        t = (mlp % RPIT) * CPR + nc
        lanes[t % 64] += 1; es[e] += 1; its[mlp // RPIT] += 1; waves[t // 64] += 1
    q = collections.Counter({k // 16: 0 for k in range(64)})
    for k, v in lanes.items():
        q[k // 16] += v
    return n, (f"quarter-waves {dict(sorted(q.items()))} e {dict(sorted(es.items()))} it {dict(sorted(its.items()))} waves {dict(sorted(waves.items()))} "
               f"max {float(d.max()):.3f} tiles_m {sorted(set((r // BM).tolist()))[:8]}")


M, N, K_ = 16384, 320, 320
g = torch.Generator().manual_seed(41)
rnd = lambda shape, scale=1.0, dtype=f16: (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)
A, B = rnd((M, K_)), rnd((N, K_), 1 / math.sqrt(K_))
bias, res = rnd((N,), dtype=f32), rnd((M, N))
base = A.float() @ B.float().T
T, U, Ut = rnd((M, 4), dtype=f32), rnd((N, 4), dtype=f32), rnd((4, N), dtype=f32)
T2 = rnd((M, 8), dtype=f32)                                  # two adapter segments of 160 columns (the round-3 strict test's first case)
lora2 = torch.cat([T2[:, :4] @ U[:160].T, T2[:, 4:] @ U[160:].T], 1)
refs = {"U+bias+res": (base + bias + 0.7 * (T @ U.T)).half().float() + res.float(),
        "Ut+res": (base + T @ Ut).half().float() + res.float(),
        "Ut": (base + T @ Ut).half().float(),
        "U2seg": (base + lora2).half().float(),
        "bias+res": (base + bias).half().float() + res.float()}
for tile in tiles:
    BM, BN = TILE_GEOM[tile]
    kw = dict(split_k=1, tile_cfg=tile, _tuned=False)
    tot = collections.Counter()
    for rep in range(REPS):
        outs = {"U+bias+res": K.gemm(A, B, M, N, K_, bias=bias, residual=res, lora_t=T, lora_u=U, lora_seg=N, lora_scale=0.7, **kw),
                "Ut+res": K.gemm(A, B, M, N, K_, residual=res, lora_t=T, lora_u=Ut, lora_seg=N, lora_u_tr=True, lora_r=4, **kw),
                "Ut": K.gemm(A, B, M, N, K_, lora_t=T, lora_u=Ut, lora_seg=N, lora_u_tr=True, lora_r=4, **kw),
                "U2seg": K.gemm(A, B, M, N, K_, lora_t=T2, lora_u=U, lora_seg=160, lora_scale=1.0, **kw),
                "bias+res": K.gemm(A, B, M, N, K_, bias=bias, residual=res, **kw)}
        torch.cuda.synchronize()
        for name, o in outs.items():
            n, where = locate(o, refs[name], BM, BN)
            tot[name] += n
            if n:
                print(f"HOIST_DIAG {label} tile={tile} rep{rep} {name}: {n} bad; {where}", flush=True)
    print(f"HOIST_DIAG_TOTAL {label} tile={tile} reps={REPS}: " + " ".join(f"{k}={v}" for k, v in sorted(tot.items())) +
          (" CLEAN" if not sum(tot.values()) else " DIRTY"), flush=True)
