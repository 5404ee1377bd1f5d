"""Where does a short-K projection GEMM spend its time?  (GPU box, ~20 s.)

For the dominant short-K shapes of the SD-1.5 step (profiles/r02_step_trace_by_grid_final.txt) at their tuned tile: time, inside
a hipGraph, the launch with (a) no epilogue operands, (b) bias, (c) bias + residual, (d) bias + rank-4 adapter, (e) bias + adapter
+ residual (the real to_out), and with K cut to one ring stage (64) -- fixed cost vs per-k-step cost vs epilogue cost.  Operands
rotate through enough copies that weights come from HBM / MALL like in the step, not from L2."""
import json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controllora_amd import kernels as K

dev = torch.device("cuda", 0)
f16, f32 = torch.float16, torch.float32


def timeit(fns, iters=24):
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for i in range(iters):
            fns[i % len(fns)]()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e3


def case(M, N, Kd, tile, sk, rot=8):
    sets = []
    for _ in range(rot):
        A = torch.randn(M, Kd, device=dev).half()
        W = (torch.randn(N, Kd, device=dev) / math.sqrt(Kd)).half()
        sets.append(dict(A=A, W=W, out=torch.empty(M, N, device=dev, dtype=f16), res=torch.randn(M, N, device=dev).half(),
                         T=torch.randn(M, 4, device=dev), U=torch.randn(N, 4, device=dev) * 0.02, bias=torch.randn(N, device=dev)))
    row = {"shape": f"{M}x{N}x{Kd}", "tile": tile, "split_k": sk}
    variants = {"plain": {}, "bias": {"bias": 1}, "bias_res": {"bias": 1, "res": 1}, "bias_lora": {"bias": 1, "lora": 1},
                "bias_lora_res": {"bias": 1, "lora": 1, "res": 1}}
    for name, v in variants.items():
        def mk(s, kd=Kd):
            kw = dict(out=s["out"], split_k=sk, tile_cfg=tile, _tuned=False)
            if v.get("bias"): kw["bias"] = s["bias"]
            if v.get("res"): kw["residual"] = s["res"]
            if v.get("lora"): kw.update(lora_t=s["T"], lora_u=s["U"], lora_seg=N)
            return lambda: K.gemm(s["A"], s["W"], M, N, kd, **kw)
        row[name] = round(timeit([mk(s) for s in sets]), 2)
    # one ring stage of K: fixed cost (launch + prologue + first-byte latency + epilogue) of the plain and the full variant
    sets1 = [dict(s, A=s["A"][:, :64].contiguous(), W=s["W"][:, :64].contiguous()) for s in sets]
    row["k64_plain"] = round(timeit([(lambda s=s: K.gemm(s["A"], s["W"], M, N, 64, out=s["out"], split_k=1, tile_cfg=tile, _tuned=False)) for s in sets1]), 2)
    row["k64_full"] = round(timeit([(lambda s=s: K.gemm(s["A"], s["W"], M, N, 64, out=s["out"], bias=s["bias"], residual=s["res"], lora_t=s["T"],
                                                        lora_u=s["U"], lora_seg=N, split_k=1, tile_cfg=tile, _tuned=False)) for s in sets1]), 2)
    row["TF_full"] = round(2.0 * M * N * Kd / row["bias_lora_res"] / 1e6, 1)
    print(json.dumps(row), flush=True)
    return row


if __name__ == "__main__":
    tab = json.load(open(os.path.join(os.path.dirname(os.path.abspath(K.__file__)), "gemm_tuning_gfx950.json")))["table"]
    shapes = [(16384, 320, 320), (16384, 320, 1280), (16384, 960, 320), (16384, 320, 960), (4096, 640, 640), (4096, 640, 2560), (4096, 1920, 640),
              (1024, 1280, 1280), (1024, 1280, 5120), (1024, 3840, 1280), (256, 1280, 1280)]
    rows = []
    for M, N, Kd in shapes:
        hit = tab.get(f"{M}x{N}x{Kd}")
        tile, sk = hit if hit else (0, 0)
        rows.append(case(M, N, Kd, tile, sk, rot=8 if N * Kd < 3e6 else 24))
    # an empty kernel's launch floor inside a graph, for reference
    x = torch.zeros(1024, device=dev, dtype=f16)
    y = torch.zeros(1024, device=dev, dtype=f16)
    print(json.dumps({"launch_floor_us(K.add 1024 elems)": round(timeit([lambda: K.add(x, y)], iters=50), 2)}))
    if len(sys.argv) > 1:
        json.dump(rows, open(sys.argv[1], "w"), indent=1)
