"""SD-1.5 VAE conv shapes: the wide-patch conv variants (tile_cfg 77 / 78: one 128-pixel row or row segment per tile) against the
implicit-GEMM variants, timed inside a hipGraph with bias + residual (GPU box)."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controllora_amd import kernels as K
from controllora_amd.ops import conv_k_order
dev = torch.device("cuda", 0)
def timeit(fn, iters=6):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e3
for B, H, Ci, Co in [(4, 512, 128, 128), (4, 256, 256, 256), (4, 256, 128, 256), (4, 128, 512, 512), (4, 128, 256, 512), (4, 64, 512, 512)]:
    M = B * H * H
    x = torch.randn(M, Ci, device=dev).half()
    w = conv_k_order((torch.randn(Co, 9, Ci, device=dev) / math.sqrt(9 * Ci)).half(), 64)
    bias, res = torch.randn(Co, device=dev), torch.randn(M, Co, device=dev).half()
    out = torch.empty(M, Co, device=dev, dtype=torch.float16)
    cd, _, _ = K.conv_fwd_desc(H, H, Ci, 3, 1, 1, kchunk=64)
    row = {"shape": f"{B}x{H}^2 {Ci}->{Co}", "GF": round(2.0 * M * Co * 9 * Ci / 1e9, 1)}
    for name, tile in (("auto", 0), ("impl128x128bk64", 21), ("impl128x128bk32", 1), ("impl256x128", 7), ("patch77", 77), ("patch78", 78), ("patch72", 72), ("patch71", 71)):
        if tile >= 71 and not K.conv_patch_eligible(M, cd, tile):
            continue
        us = timeit(lambda: K.gemm(x, w, M, Co, 9 * Ci, conv=cd, bias=bias, residual=res, out=out, tile_cfg=tile, split_k=1 if tile else 0, _tuned=False))
        row[name] = f"{us:.0f}us/{2.0 * M * Co * 9 * Ci / us / 1e6:.0f}TF"
    print(json.dumps(row), flush=True)
