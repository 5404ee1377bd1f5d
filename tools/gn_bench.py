"""GroupNorm kernel timings for every (HW, C) of the SD-1.5 + hint-encoder step (B=4), per setting of the "gn_team" knob:
0 = two launches / one block per slab, 2 = team kernels at HW >= 1024, 3 = at HW >= 256, 4 = at HW >= 64."""
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
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
shapes = [(4096, 320), (4096, 640), (4096, 960), (1024, 320), (1024, 640), (1024, 960), (1024, 1280), (1024, 1920),
          (256, 640), (256, 1280), (256, 1920), (256, 2560), (64, 1280), (64, 2560), (16384, 128), (4096, 256)]
K.gn_team_state(dev)
for HW, C in shapes:
    x = torch.randn(B, HW, C, device=dev).half()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mb = x.numel() * 2 / 1e6
    line = f"HW{HW:7d} C{C:5d} {mb:7.1f} MB |"
    for mode in (0, 2, 3, 4):
        K.set_option("gn_team", mode)
        f = timeit(lambda: K.groupnorm_fwd(x, g, b, 32, 1e-5, True))
        y, st = K.groupnorm_fwd(x, g, b, 32, 1e-5, True)
        bw = timeit(lambda: K.groupnorm_bwd(x, y, g, b, st, 32, True))
        line += f" team={mode} fwd {f:6.1f} bwd {bw:6.1f} |"
    K.set_option("gn_team", 2)
    print(line, flush=True)
print("gn_team_errors", K.gn_team_errors(dev))
