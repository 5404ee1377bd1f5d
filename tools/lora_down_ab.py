"""Adapter down-projection launch modes (option "lora_down_mode") at the step's sizes, timed inside a hipGraph (GPU box)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controllora_amd import kernels as K
dev = torch.device("cuda", 0)
def timeit(fns, iters=24):
    fns[0](); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters): fns[i % len(fns)]()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e3
for M, Kd, R, two in [(16384, 320, 12, True), (16384, 320, 4, False), (16384, 320, 4, True), (16384, 256, 4, False), (4096, 640, 12, True), (1024, 1280, 12, True), (1024, 1280, 4, False)]:
    sets = []
    for _ in range(6):
        X, X2 = torch.randn(M, Kd, device=dev).half(), torch.randn(M, Kd, device=dev).half()
        D = torch.randn(R, Kd, device=dev) * 0.1
        T = torch.empty(M, R, device=dev)
        sets.append((X, X2, D, T))
    row = {"shape": f"M={M} K={Kd} R={R} {'two inputs' if two else 'one input'}"}
    for mode in (0, 1, 2):
        K.set_option("lora_down_mode", mode)
        row[f"mode{mode}_us"] = round(timeit([(lambda s=s: K.lora_down_multi([K.down_job(s[0], s[2], s[3], 0, M, Kd, X2=s[1] if two else None, r2=4 if two else 0)])) for s in sets]), 2)
    K.set_option("lora_down_mode", 0)
    print(json.dumps(row), flush=True)
