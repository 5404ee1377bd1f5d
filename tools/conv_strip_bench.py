"""conv3x3_strip_kernel (tile_cfg 61) against the implicit GEMM on the hint encoder's large-map 3x3 stride-1 convolutions and their dgrads (B = 4)."""
import os, sys
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
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best
B = 4
for H, Ci, Co, what in [(512, 8, 32, "conv_in fwd 3(8)->32"), (512, 32, 32, "fwd / dgrad 32->32"), (256, 32, 64, "fwd 32->64"), (256, 64, 32, "dgrad of 32->64")]:
    M = B * H * H
    x = torch.randn(M, Ci, device=dev).half()
    w = (torch.randn(Co, 9 * Ci, device=dev) / (9 * Ci) ** 0.5).half()
    bias = torch.randn(Co, device=dev)
    cd, _, _ = K.conv_fwd_desc(H, H, Ci, 3, 1, 1)
    out = torch.empty(M, Co, device=dev, dtype=torch.float16)
    K.CONV_STRIP = False
    t_old = timeit(lambda: K.gemm(x, w, M, Co, 9 * Ci, conv=cd, bias=bias, out=out))
    K.CONV_STRIP = True
    mb = (x.numel() + out.numel()) * 2 / 1e6
    line = f"H{H:4d} {what:22s} implicit GEMM (tuned table) {t_old:6.1f} us | strip at"
    for blocks in (256, 512, 1024, 2048, 4096):
        K.set_option("strip_blocks", blocks)
        t_new = timeit(lambda: K.gemm(x, w, M, Co, 9 * Ci, conv=cd, bias=bias, out=out))
        line += f" {blocks}: {t_new:5.1f}"
    K.set_option("strip_blocks", 512)
    print(line + f" us  ({mb:.0f} MB min)", flush=True)
