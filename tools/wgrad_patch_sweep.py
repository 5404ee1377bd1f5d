"""conv_wgrad_patch_kernel: block-count sweep on the hint encoder's large-map stride-1 layers (B = 4) against the gather kernel."""
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
for H, Ci, Co, st in [(512, 32, 32, 1), (256, 32, 64, 1), (512, 32, 32, 2), (256, 64, 64, 2), (128, 64, 64, 1)]:
    x = torch.randn(B * H * H, Ci, device=dev).half()
    cd, Ho, Wo = K.conv_fwd_desc(H, H, Ci, 3, st, 1 if st == 1 else 0, asym_pad=st == 2)
    M = B * Ho * Wo
    dy = torch.randn(M, Co, device=dev).half()
    line = f"H{H:4d} Ci{Ci:3d} Co{Co:4d} stride {st}"
    for v in (0, 1, 64, 128, 256, 512, 1024):
        K.set_option("wgrad_patch", v)
        us = timeit(lambda: K.conv_wgrad(dy, x, M, Co, 9 * Ci, cd, with_bias=True))
        line += f" | {'gather' if v == 0 else ('auto' if v == 1 else v)} {us:6.1f}"
    print(line, flush=True)
