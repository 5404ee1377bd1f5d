"""Patch-conv tile variants against each other on the ResnetBlock2D 3x3 conv shapes (forward form, residual epilogue), interleaved rounds
inside hipGraphs, median; every candidate's output is compared with the first one's.   python tools/patch_ab.py [rounds] [cfg,cfg,...]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controllora_amd import kernels as K
dev = torch.device("cuda", 0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
CF = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [76, 72, 71, 79]

def graph_of(fn, iters):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters): fn()
    g.replay(); torch.cuda.synchronize()
    return g

def t_graph(g, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

SHAPES = [("L0 64^2 320->320", 4, 64, 320, 320, 1), ("L0 64^2 640->320", 4, 64, 640, 320, 1), ("L0 64^2 960->320", 4, 64, 960, 320, 1),
          ("L1 32^2 640->640", 4, 32, 640, 640, 2), ("L1 32^2 1280->640", 4, 32, 1280, 640, 2), ("L1 32^2 1920->640", 4, 32, 1920, 640, 3),
          ("L2 16^2 1280->1280", 4, 16, 1280, 1280, 4), ("L2 16^2 2560->1280", 4, 16, 2560, 1280, 6), ("L3 8^2 1280->1280", 4, 8, 1280, 1280, 6),
          ("b8 64^2 320->320", 8, 64, 320, 320, 1), ("b32 64^2 320->320", 32, 64, 320, 320, 1), ("b32 32^2 640->640", 32, 32, 640, 640, 1),
          ("b32 16^2 1280->1280", 32, 16, 1280, 1280, 1)]
for name, B, H, Ci, Co, sk in SHAPES:
    g_ = torch.Generator(device=dev).manual_seed(1)
    x = (torch.rand(B * H * H, Ci, device=dev, generator=g_) - 0.5).half()
    w = ((torch.rand(Co, 9 * Ci, device=dev, generator=g_) - 0.5) * (2 / math.sqrt(9 * Ci))).half()
    cd, Ho, Wo = K.conv_fwd_desc(H, H, Ci, kchunk=64)
    M = B * Ho * Wo
    res = (torch.rand(M, Co, device=dev, generator=g_) - 0.5).half()
    out = torch.empty(M, Co, device=dev, dtype=torch.float16)
    ws = K.workspace(8 * M * Co * 4, dev) if hasattr(K, "workspace") else None
    make = lambda c: (lambda: K.gemm(x, w, M, Co, 9 * Ci, conv=cd, out=out, residual=res, split_k=sk, tile_cfg=c, _tuned=False))
    iters = 4 if B >= 32 else 10
    outs, graphs = {}, {}
    for c in CF:
        if not K.conv_patch_eligible(M, cd, c):
            continue
        outs[c] = make(c)().clone()
        graphs[c] = graph_of(make(c), iters)
    ts = {c: [] for c in graphs}
    for _ in range(rounds):
        for c, g in graphs.items():
            ts[c].append(t_graph(g, iters))
    med = {c: sorted(v)[len(v) // 2] for c, v in ts.items()}
    base = outs[next(iter(outs))]
    fl = 2.0 * M * Co * 9 * Ci
    line = "  ".join(f"cfg{c} {med[c]:7.1f}us {fl / med[c] / 1e6:5.0f}TF{'' if torch.equal(outs[c], base) else ' DIFF %.1e' % float((outs[c].float() - base.float()).norm() / base.float().norm())}" for c in med)
    print(f"{name:22s} sk{sk}  {line}", flush=True)
