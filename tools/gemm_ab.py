"""Fixed set of GEMM / conv shapes at forced configurations, timed inside a hipGraph: A/B harness for main-loop edits."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controllora_amd import kernels as K
dev = torch.device("cuda", 0)
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e3
cases = [("conv 64^2 320->320", 4, 64, 320, 320, 1, 2), ("conv 32^2 640->640", 4, 32, 640, 640, 1, 2), ("conv 16^2 1280->1280", 4, 16, 1280, 1280, 1, 4),
         ("conv 64^2 640->320", 4, 64, 640, 320, 1, 2), ("conv 64^2 320->320 t1s1", 4, 64, 320, 320, 1, 1), ("conv 128^2 320->320 t1s1", 4, 128, 320, 320, 1, 1)]
tot = 0
for name, B, H, Ci, Co, tile, sk in cases:
    x = torch.randn(B * H * H, Ci, device=dev).half()
    w = (torch.randn(Co, 9 * Ci, device=dev) / math.sqrt(9 * Ci)).half()
    cd, Ho, Wo = K.conv_fwd_desc(H, H, Ci)
    M = B * Ho * Wo
    out = torch.empty(M, Co, device=dev, dtype=torch.float16)
    res = torch.randn(M, Co, device=dev).half()
    us = timeit(lambda: K.gemm(x, w, M, Co, 9 * Ci, conv=cd, out=out, residual=res, split_k=sk, tile_cfg=tile, _tuned=False))
    tot += us
    print(f"{name:28s} tile{tile} sk{sk} {us:8.1f} us {2.0*M*Co*9*Ci/us/1e6:7.1f} TF", flush=True)
for name, M, N, Kd, tile, sk in [("ff1 16384x2560x320", 16384, 2560, 320, 1, 1), ("ff2 16384x320x1280", 16384, 320, 1280, 1, 1), ("big 8192^3", 8192, 8192, 8192, 1, 1),
                                 ("ff1.s1 4096x5120x640", 4096, 5120, 640, 1, 1), ("qkv 16384x960x320 t2", 16384, 960, 320, 2, 1), ("out 4096x640x640 t3", 4096, 640, 640, 3, 1),
                                 ("proj 1024x1280x1280 t3", 1024, 1280, 1280, 3, 1), ("proj 1024x1280x1280 t6", 1024, 1280, 1280, 6, 1),
                                 ("q 16384x320x320 t3", 16384, 320, 320, 3, 1), ("ff2.s2 1024x1280x5120 t3s4", 1024, 1280, 5120, 3, 4)]:
    A = torch.randn(M, Kd, device=dev).half()
    Bw = (torch.randn(N, Kd, device=dev) / math.sqrt(Kd)).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    us = timeit(lambda: K.gemm(A, Bw, M, N, Kd, out=out, split_k=sk, tile_cfg=tile, _tuned=False), iters=5 if M * N * Kd > 1e11 else 10)
    tot += us if M * N * Kd < 1e11 else 0
    print(f"{name:28s} tile{tile} sk{sk} {us:8.1f} us {2.0*M*N*Kd/us/1e6:7.1f} TF", flush=True)
print(f"sum (without 8192^3) {tot:.1f} us")
