"""GroupNorm kernel timings for every (HW, C) of the SD-1.5 + hint-encoder step (B=4)."""
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
shapes = [(4096, 320), (4096, 640), (4096, 960), (1024, 320), (1024, 640), (1024, 960), (1024, 1280), (1024, 1920),
          (256, 640), (256, 1280), (256, 1920), (256, 2560), (64, 1280), (64, 2560),
          (262144, 32), (65536, 32), (65536, 64), (16384, 64), (16384, 128), (4096, 128), (4096, 256), (1024, 256)]
for HW, C in shapes:
    x = torch.randn(B, HW, C, device=dev).half()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mb = x.numel() * 2 / 1e6
    f = timeit(lambda: K.groupnorm_fwd(x, g, b, 32, 1e-5, True))
    y, st = K.groupnorm_fwd(x, g, b, 32, 1e-5, True)
    bw = timeit(lambda: K.groupnorm_bwd(x, y, g, b, st, 32, True))
    print(f"HW{HW:7d} C{C:5d} {mb:7.1f} MB  fwd {f:7.1f} us ({3*mb/f:5.2f} TB/s)  bwd {bw:7.1f} us ({5*mb/bw:5.2f} TB/s)", flush=True)
