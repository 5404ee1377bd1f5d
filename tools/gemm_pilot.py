"""Pilot sweep of the GEMM main-loop variants (tile_cfg of clora_gemm_f16_ex) on representative shapes of the train step
(B=4, 512^2) and of the batch-32 inference forward.  Each (shape, cfg, split-K) is timed as 10 launches captured into one
hipGraph (the regime of the captured step).  Run on the GPU box:
    python tools/gemm_pilot.py gpurun_out/gemm_pilot.json            (timing sweep)
    rocprofv3 --pmc ... -- python tools/gemm_pilot.py --pmc          (a few launches per variant, for counter collection)"""
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from controllora_amd import kernels as K

dev = torch.device("cuda", 0)
CFGS = [1, 2, 3, 4, 5, 6, 7, 8, 21, 22, 23, 26, 31, 32, 33, 41, 42, 43, 51, 52, 53, 54, 55, 56, 57, 58, 71, 72, 73, 74, 75, 76, 79]
# (M, N, K, conv side H (0 = plain GEMM), Cin)
SHAPES = [
    (16384, 320, 320, 0, 0), (4096, 640, 640, 0, 0), (1024, 1280, 1280, 0, 0),
    (16384, 320, 2880, 64, 320), (4096, 640, 5760, 32, 640), (1024, 1280, 11520, 16, 1280), (256, 1280, 11520, 8, 1280),
    (16384, 2560, 320, 0, 0), (16384, 320, 1280, 0, 0), (4096, 5120, 640, 0, 0), (1024, 10240, 1280, 0, 0),
    (16384, 640, 5760, 64, 640), (16384, 960, 320, 0, 0), (1024, 1280, 5120, 0, 0), (8192, 8192, 8192, 0, 0),
    (131072, 320, 2880, 64, 320), (32768, 640, 5760, 32, 640), (131072, 2560, 320, 0, 0), (8192, 1280, 11520, 16, 1280),
    (131072, 320, 320, 0, 0),
]


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(iters):
            fn()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    g.replay(); g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * iters) * 1e3


def operands(M, N, Kd, H, Cin):
    if H:
        cd, Ho, Wo = K.conv_fwd_desc(H, H, Cin)
        Bn = M // (H * H)
        A = torch.randn(Bn * H * H, Cin, device=dev).half()
    else:
        cd = None
        A = torch.randn(M, Kd, device=dev).half()
    Bw = (torch.randn(N, Kd, device=dev) / math.sqrt(Kd)).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    res = torch.randn(M, N, device=dev).half()
    return A, Bw, out, res, cd


def main():
    pmc = "--pmc" in sys.argv
    outp = next((a for a in sys.argv[1:] if not a.startswith("--")), None)
    rows = []
    shapes = SHAPES if not pmc else [SHAPES[0], SHAPES[3], SHAPES[14], SHAPES[2]]
    cfgs = CFGS if not pmc else [1, 21, 41, 7, 8, 3, 23, 51, 53, 71, 76]
    for (M, N, Kd, H, Cin) in shapes:
        A, Bw, out, res, cd = operands(M, N, Kd, H, Cin)
        ksteps = Kd // 32
        best = None
        for cfg in cfgs:
            sks = [1] if pmc else [1, 2, 3, 4, 6, 8, 12]
            for sk in sks:
                if sk > 1 and (ksteps // sk < 4 or sk * M * N * 4 > K.GEMM_WS_BYTES or M * N >= 16384 * 1280):
                    continue
                fn = lambda: K.gemm(A, Bw, M, N, Kd, conv=cd, out=out, residual=res, split_k=sk, tile_cfg=cfg, _tuned=False)
                try:
                    if pmc:
                        for _ in range(3):
                            fn()
                        torch.cuda.synchronize()
                        continue
                    us = timeit(fn)
                except Exception as ex:          # noqa: BLE001
                    print("ERR", M, N, Kd, cfg, sk, repr(ex)[:80], flush=True)
                    continue
                r = dict(M=M, N=N, K=Kd, conv=bool(H), cfg=cfg, sk=sk, us=round(us, 2), TF=round(2.0 * M * N * Kd / us / 1e6, 1))
                rows.append(r)
                if best is None or us < best["us"]:
                    best = r
        if not pmc:
            base = min((r for r in rows if (r["M"], r["N"], r["K"]) == (M, N, Kd) and r["cfg"] in (1, 2, 3, 4, 5, 6)), key=lambda r: r["us"])
            per_cfg = {}
            for r in rows:
                if (r["M"], r["N"], r["K"]) == (M, N, Kd):
                    if r["cfg"] not in per_cfg or r["us"] < per_cfg[r["cfg"]]["us"]:
                        per_cfg[r["cfg"]] = r
            line = " ".join(f"{c}:{per_cfg[c]['us']:.0f}/{per_cfg[c]['sk']}" for c in sorted(per_cfg))
            print(f"{M}x{N}x{Kd}{'c' if H else ' '} r01-best cfg{base['cfg']} sk{base['sk']} {base['us']:.1f}us {base['TF']}TF | "
                  f"best cfg{best['cfg']} sk{best['sk']} {best['us']:.1f}us {best['TF']}TF ({base['us'] / best['us']:.2f}x) | {line}", flush=True)
        del A, Bw, out, res
    if outp and not pmc:
        json.dump(rows, open(outp, "w"))


if __name__ == "__main__":
    main()
