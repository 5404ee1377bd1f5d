"""conv_wgrad kernel timings for the 18 hint-encoder weight gradients of one step (B=4, 512^2)."""
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
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3
B = 4
# (H_in, Cin_padded, Cout, ksize, stride)
layers = [(512, 8, 32, 3, 1), (512, 32, 32, 3, 1), (512, 32, 32, 3, 2), (256, 32, 64, 3, 1), (256, 64, 64, 3, 2), (128, 64, 128, 3, 1),
          (128, 128, 128, 3, 2), (64, 128, 256, 3, 1), (64, 256, 256, 3, 1), (64, 256, 256, 3, 2), (32, 256, 256, 3, 1), (32, 256, 256, 3, 2),
          (16, 256, 256, 3, 1), (16, 256, 256, 3, 2), (64, 256, 320, 1, 1), (32, 256, 640, 1, 1), (16, 256, 1280, 1, 1), (8, 256, 1280, 1, 1)]
tot = 0.0
for H, Ci, Co, k, st in layers:
    x = torch.randn(B * H * H, Ci, device=dev).half()
    if k == 3:
        cd, Ho, Wo = K.conv_fwd_desc(H, H, Ci, 3, st, 0 if st == 2 else 1, False, st == 2)
        Kd = 9 * Ci
    else:
        cd, Ho, Wo, Kd = None, H, H, Ci
    M = B * Ho * Wo
    dy = torch.randn(M, Co, device=dev).half()
    us = timeit(lambda: K.conv_wgrad(dy, x, M, Co, Kd, cd, with_bias=True))
    gW = torch.zeros((Co, Ci, k, k), dtype=torch.float32, device=dev)
    gb = torch.zeros((Co,), dtype=torch.float32, device=dev)
    stage = torch.zeros(Co * Kd + Co, dtype=torch.float32, device=dev)
    us2 = timeit((lambda: K.conv_wgrad_staged(dy, x, M, Co, Kd, cd, stage, gW, gb, Ci)) if k == 3 else
                 (lambda: K.conv_wgrad_into(dy, x, M, Co, Kd, cd, gW, gb)))
    tot += us
    tot2 = globals().get("tot2", 0.0) + us2
    globals()["tot2"] = tot2
    mb = (dy.numel() + x.numel()) * 2 / 1e6
    print(f"M{M:8d} N{Co:5d} K{Kd:5d} product-path {us2:8.1f} us | {us:8.1f} us  {2.0*M*Co*Kd/us/1e6:7.1f} TF  min-bytes {mb:6.1f} MB -> {mb/us:5.2f} TB/s", flush=True)
print(f"total {tot/1e3:.3f} ms   (product path: staged + unpack / direct for 1x1: {tot2/1e3:.3f} ms)")
