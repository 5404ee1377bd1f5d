"""Pick the tile of every projection launch that carries its adapter's down-projection (clora_epilogue_t.lora_dpack): the launch
table's entries were timed WITHOUT the extra operand rows.  Records the fused launches of one train step (+ one inference forward),
times each signature on every tile that can take it and writes `"<M>x<N>x<K>:x": [tile, 1]` entries into
controllora_amd/gemm_tuning_gfx950.json (ops._fuse_plan reads them).   python tools/tune_fused.py [--batch 4] [--infer-batch 32]"""
import argparse, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from controllora_amd import kernels as K, ops
from controllora_amd.schedulers import DDPMScheduler
from controllora_amd.train import ControlLoRATrainer

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--res", type=int, default=512)
ap.add_argument("--infer-batch", type=int, default=32)
ap.add_argument("--config", default="fill50k.json")
args = ap.parse_args()
dev = torch.device("cuda", 0)
unet, clora = bench.build_models(dev, config=args.config)
trainer = ControlLoRATrainer(unet, clora)
batch = bench.synthetic_batch(args.batch, args.res, dev, 42)
noisy = DDPMScheduler().add_noise(batch["latents"], batch["noise"], batch["timesteps"]).half()
seen = {}
orig = K.gemm
def rec(A, Bw, M, N, Kd, **kw):
    if kw.get("lora_dpack") is not None:
        sig = (M, N, Kd, int(kw.get("lora_seg") or N), bool(kw.get("lora_u_tr")), kw.get("bias") is not None, kw.get("residual") is not None,
               kw.get("lora_t_in") is not None)
        seen[sig] = seen.get(sig, 0) + 1
    return orig(A, Bw, M, N, Kd, **kw)
K.gemm = rec
ops.K.gemm = rec
trainer.step(noisy, batch["timesteps"], batch["ehs"], batch["guide"], batch["noise"])
if args.infer_batch > 0:
    with torch.no_grad():
        nb = args.infer_batch
        clora(batch["guide"][:1])
        unet(torch.randn(nb, 4, args.res // 8, args.res // 8, device=dev).half(), 10, torch.randn(nb, 77, 768, device=dev).half())
K.gemm = orig
ops.K.gemm = orig
torch.cuda.synchronize()
print(f"{len(seen)} fused signatures", flush=True)

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

TILES = {51: 320, 52: 320, 54: 320, 55: 320, 21: 128, 41: 128, 22: 64, 42: 64, 26: 64, 23: 64, 43: 64}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "controllora_amd", "gemm_tuning_gfx950.json")
doc = json.load(open(path))
table = doc["table"]
tot_cur = tot_best = 0.0
for (M, N, Kd, seg, u_tr, has_b, has_r, has_tin), cnt in sorted(seen.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2]):
    A = torch.randn(M, Kd, device=dev).half()
    Bw = (torch.randn(N, Kd, device=dev) / math.sqrt(Kd)).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    nseg = N // seg
    T = torch.empty(M, 4 * nseg, device=dev)
    U = torch.randn(4, N, device=dev) if u_tr else torch.randn(N, 4, device=dev)
    pack = torch.randn(8 * nseg, Kd, device=dev).half()
    kw = dict(lora_t=T, lora_u=U, lora_seg=seg, lora_u_tr=u_tr, lora_r=4, lora_dpack=pack, out=out, _tuned=False, split_k=1)
    if has_b:
        kw["bias"] = torch.randn(N, device=dev)
    if has_r:
        kw["residual"] = torch.randn(M, N, device=dev).half()
    if has_tin:
        kw.update(lora_t_in=torch.randn(M, 4, device=dev), lora_t_in_mask=1)
    cur = ops._fuse_plan(M, N, Kd, seg)
    best = None
    res = {}
    for tile, bn in TILES.items():
        if seg % bn:
            continue
        if bn == 320 and M * (N // 320) < 64 * 64:          # a handful of blocks: not worth timing
            continue
        us = timeit(lambda: K.gemm(A, Bw, M, N, Kd, tile_cfg=tile, **kw), iters=10 if M * N * Kd < 4e10 else 4)
        res[tile] = us
        if best is None or us < best[0]:
            best = (us, tile)
    cur_tile = cur if cur else (54 if M >= 32768 else 55)
    cur_us = res.get(cur_tile, best[0])
    tot_cur += cnt * cur_us; tot_best += cnt * best[0]
    table[f"{M}x{N}x{Kd}:x"] = [best[1], 1]
    print(f"{M}x{N}x{Kd} seg {seg} u_tr {int(u_tr)} x{cnt:3d}: current tile {cur_tile} {cur_us:7.1f} us, best tile {best[1]} {best[0]:7.1f} us   " +
          " ".join(f"{t}:{u:.1f}" for t, u in sorted(res.items())), flush=True)
    del A, Bw, out, T, U, pack
print(f"sum over the recorded fused launches: current {tot_cur / 1e3:.3f} ms, best {tot_best / 1e3:.3f} ms")
json.dump(doc, open(path, "w"), indent=0)
print("wrote", path)
