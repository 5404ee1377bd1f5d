"""Diagnostic for the projection-epilogue variants on the GPU: per sub-case rel error and where the wrong elements sit."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controllora_amd import kernels as K
dev = "cuda"
f16, f32 = torch.float16, torch.float32
def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))
def where(out, ref, BM=64, BN=64):
    ref = ref.float().cpu()
    d = (out.float().cpu() - ref).abs()
    bad = d > 0.02 * ref.abs().max()
    if not bool(bad.any()):
        return "none"
    r, c = bad.nonzero(as_tuple=True)
    return f"{int(bad.sum())} bad; rows%{BM} in {sorted(set((r % BM).tolist()))[:12]}.. cols%{BN} in {sorted(set((c % BN).tolist()))[:12]}.. tiles_m {sorted(set((r // BM).tolist()))[:6]} max {float(d.max()):.3f}"
for tp in (1, 0):
    K.set_option("epi_two_phase", tp)
    for tile in (23, 43, 55):
        for (M, N, K_) in ((16384, 320, 320), (1000, 1296, 1280)):
            g = torch.Generator().manual_seed(41)
            rnd = lambda shape, scale=1.0, dtype=f16: (torch.randn(shape, generator=g) * scale).to(dtype).to(dev)
            A, B = rnd((M, K_)), rnd((N, K_), 1 / math.sqrt(K_))
            bias, res = rnd((N,), dtype=f32), rnd((M, N))
            base = A.float() @ B.float().T
            T, U, Ut = rnd((M, 4), dtype=f32), rnd((N, 4), dtype=f32), rnd((4, N), dtype=f32)
            kw = dict(split_k=1, tile_cfg=tile, _tuned=False)
            for rep in range(2):
                o1 = K.gemm(A, B, M, N, K_, bias=bias, residual=res, lora_t=T, lora_u=U, lora_seg=N, lora_scale=0.7, **kw)
                r1 = (base + bias + 0.7 * (T @ U.T)).half().float() + res.float()
                o2 = K.gemm(A, B, M, N, K_, residual=res, lora_t=T, lora_u=Ut, lora_seg=N, lora_u_tr=True, lora_r=4, **kw)
                r2 = (base + T @ Ut).half().float() + res.float()
                o3 = K.gemm(A, B, M, N, K_, lora_t=T, lora_u=Ut, lora_seg=N, lora_u_tr=True, lora_r=4, **kw)
                r3 = (base + T @ Ut).half().float()
                o4 = K.gemm(A, B, M, N, K_, bias=bias, residual=res, **kw)
                r4 = (base + bias).half().float() + res.float()
                torch.cuda.synchronize()
                print(f"tp={tp} tile={tile} {M}x{N}x{K_} rep{rep}: U {rel(o1, r1):.2e} | Ut+res {rel(o2, r2):.2e} [{where(o2, r2)}] | Ut {rel(o3, r3):.2e} [{where(o3, r3)}] | bias+res {rel(o4, r4):.2e}", flush=True)
K.set_option("epi_two_phase", 1)
