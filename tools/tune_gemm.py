"""Autotune the GEMM launch configuration (tile shape x split-K) for every distinct GEMM of one train step on the
current GPU and write controllora_amd/gemm_tuning_gfx950.json.  kernels.gemm() consults the table (exact key
match) and falls back to the library's latency model for unknown shapes.   Run on the GPU box:
    python tools/tune_gemm.py [--batch 4 --res 512]"""
import argparse, ctypes as C, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from controllora_amd import kernels as K
from controllora_amd.capi import ConvDesc
from controllora_amd.schedulers import DDPMScheduler
from controllora_amd.train import ControlLoRATrainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--infer-batch", type=int, default=32)
args = ap.parse_args()
dev = torch.device("cuda", 0)
unet, clora = bench.build_models(dev)
trainer = ControlLoRATrainer(unet, clora)
batch = bench.synthetic_batch(args.batch, args.res, dev, 42)
noisy = DDPMScheduler().add_noise(batch["latents"], batch["noise"], batch["timesteps"]).half()

# 1) record every distinct GEMM signature of a train step (+ the inference forward at the DDIM batch)
seen = {}
orig = K.gemm
def rec(A, Bw, M, N, Kd, **kw):
    conv = kw.get("conv")
    ck = tuple(getattr(conv, f) for f, _ in ConvDesc._fields_) if conv is not None else None
    seen.setdefault((M, N, Kd, ck), 0)
    seen[(M, N, Kd, ck)] += 1
    return orig(A, Bw, M, N, Kd, **kw)
K.gemm = rec
import controllora_amd.ops as ops
trainer.step(noisy, batch["timesteps"], batch["ehs"], batch["guide"], batch["noise"])
with torch.no_grad():
    nb = args.infer_batch
    clora(batch["guide"][:1])
    unet(torch.randn(nb, 4, args.res // 8, args.res // 8, device=dev).half(), 10, torch.randn(nb, 77, 768, device=dev).half())
K.gemm = orig
torch.cuda.synchronize()
print(f"{len(seen)} distinct GEMM signatures", flush=True)

def timeit(fn, iters=10, warm=2):
    """GPU-side time per call: the calls are captured into one hipGraph (no Python / launch overhead between
    kernels -- the same regime as the captured train step) and the replay is timed with events."""
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

table = {}
for (M, N, Kd, ck), cnt in sorted(seen.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2]):
    if ck is None:
        A = torch.randn(M, Kd, device=dev).half(); conv = None
    else:
        conv = ConvDesc(*ck)
        A = torch.randn((M // (conv.Hout * conv.Wout)) * conv.Hin * conv.Win, conv.Cin, device=dev).half()
    Bw = (torch.randn(N, Kd, device=dev) / math.sqrt(Kd)).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    res_ = torch.randn(M, N, device=dev).half()
    best, err = None, None
    auto = timeit(lambda: K.gemm(A, Bw, M, N, Kd, conv=conv, out=out, residual=res_, split_k=0, tile_cfg=0, _tuned=False))
    for tile in (1, 2, 3, 4, 5, 6):
        for sk in (1, 2, 3, 4, 6, 8, 12, 16):
            if sk > 1 and (Kd // 32) // sk < 4:
                continue
            if sk > 1 and sk * M * N * 4 > K.GEMM_WS_BYTES:
                continue
            try:
                us = timeit(lambda: K.gemm(A, Bw, M, N, Kd, conv=conv, out=out, residual=res_, split_k=sk, tile_cfg=tile, _tuned=False))
            except Exception as ex:
                err = ex
                continue
            if best is None or us < best[0]:
                best = (us, tile, sk)
    key = K.tuning_key(M, N, Kd, conv)
    if best is None:
        print(f"{key}: no configuration ran ({err!r})", flush=True)
        continue
    table[key] = [best[1], best[2]]
    print(f"{key:48s} x{cnt:3d} auto {auto:8.1f}us best tile={best[1]} sk={best[2]:2d} {best[0]:8.1f}us  ({auto / best[0]:.2f}x)", flush=True)
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "controllora_amd", "gemm_tuning_gfx950.json")
json.dump(dict(device=torch.cuda.get_device_name(0), note="tile: 1=128x128 2=128x64 3=64x64, 4-6 = same tiles with the deep LDS ring; value = [tile, split_k]", table=table),
          open(path, "w"), indent=0)
print("wrote", path)
