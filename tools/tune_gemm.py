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
ap.add_argument("--infer-batch", type=int, default=32, help="0 = skip the inference forward")
ap.add_argument("--config", default="fill50k.json")
ap.add_argument("--merge", action="store_true", help="keep the entries already in the table and add / refresh the measured ones")
ap.add_argument("--cfgs", default="", help="comma list of tile_cfg values to sweep instead of the full set (with --merge the current "
                "table entry of a signature is timed too and only replaced by a faster candidate)")
ap.add_argument("--plain-only", action="store_true", help="only the plain (non-conv) GEMM signatures")
ap.add_argument("--vae", type=int, default=0, metavar="B", help="tune the GEMM / conv signatures of the SD-1.5 VAE instead (encode + decode "
                "of B images at --res: reference train...:753-754, apps/gradio_canny2image.py:88-92); implies --merge")
ap.add_argument("--geglu", action="store_true", help="only the FeedForward GEMMs that carry the fused GEGLU activation (forward / backward), "
                "timed WITH it (the activation is most of such a launch: the default pass times the bare GEMM); implies --merge")
ap.add_argument("--patch-only", action="store_true", help="only the signatures the patch-staged 3x3 conv kernel can take (tile_cfg 71..75)")
args = ap.parse_args()
dev = torch.device("cuda", 0)
unet, clora = bench.build_models(dev, config=args.config)
trainer = ControlLoRATrainer(unet, clora)
batch = bench.synthetic_batch(args.batch, args.res, dev, 42)
noisy = DDPMScheduler().add_noise(batch["latents"], batch["noise"], batch["timesteps"]).half()

# 1) record every distinct GEMM signature of a train step (+ the inference forward at the DDIM batch)
seen = {}
epis = {}            # signature -> {epilogue shape: count}: each signature is timed with its most frequent epilogue
orig = K.gemm
wide_only = set()           # signatures used with the GEGLU-forward epilogue: tiles must be >= 128 columns wide
geglu_mode = {}             # signature -> (1 forward | 2 backward, keep_h): timed WITH the fused activation (it is most of the launch)
def rec(A, Bw, M, N, Kd, **kw):
    conv = kw.get("conv")
    ck = tuple(getattr(conv, f) for f, _ in ConvDesc._fields_) if conv is not None else None
    seen.setdefault((M, N, Kd, ck), 0)
    seen[(M, N, Kd, ck)] += 1
    lt, lu = kw.get("lora_t"), kw.get("lora_u")
    epi = (lt is not None, bool(kw.get("lora_u_tr")), int(kw.get("lora_seg") or 0),
           (kw.get("lora_r") or ((lu.shape[0] if kw.get("lora_u_tr") else lu.shape[1]) if lu is not None else 0)),
           (lt.shape[1] if lt is not None else 0), kw.get("bias") is not None, kw.get("rowadd") is not None,
           int(kw.get("rows_per_batch") or 0), kw.get("residual") is not None)
    epis.setdefault((M, N, Kd, ck), {}).setdefault(epi, 0)
    epis[(M, N, Kd, ck)][epi] += 1
    if kw.get("geglu") == 1:
        wide_only.add((M, N, Kd, ck))
        geglu_mode[(M, N, Kd, ck)] = (1, bool(kw.get("geglu_keep_h", True)))
    elif kw.get("geglu") == 2:
        geglu_mode[(M, N, Kd, ck)] = (2, True)
    return orig(A, Bw, M, N, Kd, **kw)
K.gemm = rec
import controllora_amd.ops as ops
if args.vae:
    from controllora_amd import vae as V
    args.merge, args.infer_batch = True, 0
    vm = V.AutoencoderKL(**V.SD15_VAE); V.init_random_(vm, 1); vm.to(dev)
    with torch.no_grad():
        xv = (torch.rand(args.vae, 3, args.res, args.res, device=dev) * 2 - 1).half()
        zv = vm.encode(xv).latent_dist.sample()
        vm.decode(zv.half())
else:
    trainer.step(noisy, batch["timesteps"], batch["ehs"], batch["guide"], batch["noise"])
if args.infer_batch > 0:
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

CFGS = [1, 2, 3, 4, 5, 6, 7, 8, 21, 22, 23, 26, 31, 32, 33, 41, 42, 43, 51, 52, 53, 54, 55, 56, 57, 58, 71, 72, 73, 74, 75, 76, 79]
if args.cfgs:
    CFGS = [int(c) for c in args.cfgs.split(",")]
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "controllora_amd", "gemm_tuning_gfx950.json")
if args.geglu:
    args.merge = True
table = json.load(open(path))["table"] if (args.merge and os.path.exists(path)) else {}
tot_auto = tot_best = tot_r01 = 0.0
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
    has_lora, u_tr, lseg, lr, tcols, has_bias, has_rowadd, rpb, has_res = max(epis[(M, N, Kd, ck)].items(), key=lambda kv: kv[1])[0]
    ekw = dict(residual=res_ if has_res else None)
    if has_bias:
        ekw["bias"] = torch.randn(N, device=dev)
    if has_rowadd and rpb > 0:
        ekw.update(rowadd=torch.randn((M + rpb - 1) // rpb, N, device=dev).half(), rows_per_batch=rpb)
    if has_lora:                                               # the adapter epilogue of the attention projections
        ekw.update(lora_t=torch.randn(M, max(tcols, lr), device=dev), lora_seg=lseg, lora_u_tr=u_tr, lora_r=lr,
                   lora_u=(torch.randn(max(1, tcols), Kd if False else N, device=dev) if u_tr else torch.randn(N, lr, device=dev)))
    run = lambda sk, tile: K.gemm(A, Bw, M, N, Kd, conv=conv, out=out, split_k=sk, tile_cfg=tile, _tuned=False, **ekw)
    gm = geglu_mode.get((M, N, Kd, ck))
    if gm is not None and args.geglu:
        if gm[0] == 1:
            gy = torch.empty(M, N // 2, device=dev, dtype=torch.float16)
            gb = torch.randn(N, device=dev) if has_bias else None
            run = lambda sk, tile: K.gemm(A, Bw, M, N, Kd, bias=gb, geglu=1, geglu_y=gy, geglu_keep_h=gm[1], tile_cfg=tile, _tuned=False)
        else:
            gh = torch.randn(M, 2 * N, device=dev).half()
            go = torch.empty(M, 2 * N, device=dev, dtype=torch.float16)
            run = lambda sk, tile: K.gemm(A, Bw, M, N, Kd, geglu=2, geglu_h=gh, out=go, tile_cfg=tile, _tuned=False)
    elif args.geglu:
        del A, Bw, out, res_
        continue
    if (args.patch_only and not (conv is not None and any(K.conv_patch_eligible(M, conv, c) for c in K.PATCH_TILE_CFGS))) or \
            (args.plain_only and conv is not None):
        del A, Bw, out, res_
        continue
    auto = timeit(lambda: run(0, 0))
    iters = 10 if auto < 300 else 4
    cur = table.get(K.tuning_key(M, N, Kd, conv))
    if args.cfgs and cur is not None:                      # the incumbent defends its entry
        best = (timeit(lambda: run(cur[1], cur[0]), iters=iters), cur[0], cur[1])
        auto = best[0]
    r01 = None
    geglu_sig = (M, N, Kd, ck) in wide_only
    for tile in CFGS:
        if geglu_sig and tile not in K.WIDE_TILE_CFGS:
            continue
        if tile in K.PATCH_TILE_CFGS and not (conv is not None and K.conv_patch_eligible(M, conv, tile)):
            continue                                       # would only re-time the fallback
        prev = None
        for sk in (1, 2, 3, 4, 6, 8, 12, 16):
            if sk > 1 and (geglu_sig or (gm is not None and args.geglu) or (Kd // 32) // sk < 4 or sk * M * N * 4 > K.GEMM_WS_BYTES or M * N > 16384 * 1280):
                break
            if tile in K.PATCH_TILE_CFGS and sk > conv.Cin // 64:
                break
            try:
                us = timeit(lambda: run(sk, tile), iters=iters)
            except Exception as ex:
                err = ex
                break
            # with --cfgs the incumbent defends its entry: a challenger must win by more than the timing noise (2 %)
            if best is None or us < best[0] * (0.98 if (args.cfgs and cur is not None and best[1] == cur[0] and best[2] == cur[1]) else 1.0):
                best = (us, tile, sk)
            if tile <= 6 and (r01 is None or us < r01):
                r01 = us
            if prev is not None and us > prev * 1.03:      # split-K stopped paying for this tile
                break
            prev = us
    key = K.tuning_key(M, N, Kd, conv)
    if best is None:
        print(f"{key}: no configuration ran ({err!r})", flush=True)
        continue
    table[key] = [best[1], best[2]]
    tot_auto += cnt * auto; tot_best += cnt * best[0]; tot_r01 += cnt * (r01 or best[0])
    print(f"{key:48s} x{cnt:3d} auto {auto:8.1f}us r01-tiles {r01 or 0:8.1f}us best tile={best[1]:2d} sk={best[2]:2d} {best[0]:8.1f}us  "
          f"({(r01 or best[0]) / best[0]:.2f}x) {2.0 * M * N * Kd / best[0] / 1e6:7.1f}TF", flush=True)
    del A, Bw, out, res_
print(f"sum over the recorded launches: latency model {tot_auto / 1e3:.2f} ms, round-1 tile set {tot_r01 / 1e3:.2f} ms, full set {tot_best / 1e3:.2f} ms")
json.dump(dict(device=torch.cuda.get_device_name(0),
               note="value = [tile_cfg, split_k]; tile_cfg as documented at clora_gemm_f16_ex (1-3 BK32 ring, 4-6 deep ring, 7/8 256x128, "
                    "21/22/23/26 BK64 whole-line rows, 3x/4x fragment reads before the ring refill)", table=table),
          open(path, "w"), indent=0)
print("wrote", path, len(table), "entries")
