"""Same-process A/B of the tile -> XCD assignment (clora_set_option "tile_order": "m" = contiguous m-major tile ranges per XCD,
"n" = n-major, "auto" = the fabric-bytes model of clora_gemm.hip pick_tile_order) on the weight-heavy GEMM / conv shapes of
the SD-1.5 step (batch 4, 512^2) and a few activation-heavy controls.  Every shape runs with the tile / split-K the tuning table
gives it; each order is captured into its own hipGraph (the order is fixed at capture) and replayed.
    python tools/tile_order_ab.py [--json out.json]"""
import argparse, json, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from controllora_amd import kernels as K

ap = argparse.ArgumentParser()
ap.add_argument("--json", default="")
args = ap.parse_args()
dev = torch.device("cuda", 0)

PLAIN = [(1024, 10240, 1280), (1024, 5120, 1280), (1024, 1280, 5120), (256, 10240, 1280), (256, 1280, 5120), (1024, 1280, 1280),
         (4096, 5120, 640), (4096, 640, 2560), (4096, 640, 640), (16384, 2560, 320), (16384, 320, 1280)]
CONVS = [(4, 16, 16, 1280, 1280), (4, 16, 16, 2560, 1280), (4, 8, 8, 1280, 1280), (4, 8, 8, 2560, 1280), (4, 32, 32, 640, 640),
         (4, 32, 32, 1280, 640), (4, 64, 64, 320, 320), (4, 64, 64, 640, 320)]


def timeit(fn, iters=20):
    for _ in range(2):
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


rows = []
def ab(name, fn, flops):
    r = {"shape": name}
    for order in ("m", "n", "auto", "m"):
        K.set_tile_order(order)
        us = timeit(fn)
        r[order] = min(us, r.get(order, 1e9))
    K.set_tile_order(K.DEFAULT_TILE_ORDER)
    r["tflops_m"], r["tflops_auto"] = flops / r["m"] / 1e6, flops / r["auto"] / 1e6
    rows.append(r)
    print(f"{name:34s} m {r['m']:7.1f} us   n {r['n']:7.1f} us   auto {r['auto']:7.1f} us   ({r['m'] / r['auto']:.2f}x, {r['tflops_auto']:.0f} TF)", flush=True)


for M, N, Kd in PLAIN:
    A = torch.randn(M, Kd, device=dev).half()
    Bw = (torch.randn(N, Kd, device=dev) / math.sqrt(Kd)).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    ab(f"gemm {M}x{N}x{Kd}", lambda: K.gemm(A, Bw, M, N, Kd, out=out), 2.0 * M * N * Kd)
for Bn, H, W, Ci, Co in CONVS:
    M = Bn * H * W
    cd, _, _ = K.conv_fwd_desc(H, W, Ci, 3, 1, 1, kchunk=64)
    x = torch.randn(M, Ci, device=dev).half()
    w = (torch.randn(Co, 9 * Ci, device=dev) / math.sqrt(9 * Ci)).half()
    out = torch.empty(M, Co, device=dev, dtype=torch.float16)
    ab(f"conv3x3 {H}x{W} {Ci}->{Co}", lambda: K.gemm(x, w, M, Co, 9 * Ci, conv=cd, out=out), 2.0 * M * Co * 9 * Ci)
# attention: whole heads per XCD ("auto" / "n") against launch order ("m")
for B, H, Nq, Nk, D in [(4, 8, 4096, 4096, 40), (4, 8, 1024, 1024, 80), (4, 8, 256, 256, 160), (32, 8, 4096, 4096, 40)]:
    q, k, v = (torch.randn(B * n, H * D, device=dev).half() for n in (Nq, Nk, Nk))
    dO = torch.randn(B * Nq, H * D, device=dev).half()
    o, lse = K.attn_fwd(q, k, v, B, H, Nq, Nk, D, D ** -0.5)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    ab(f"attn_fwd B{B} N{Nq} D{D}", lambda: K.attn_fwd(q, k, v, B, H, Nq, Nk, D, D ** -0.5), 4.0 * B * H * Nq * Nk * D)
    ab(f"attn_bwd B{B} N{Nq} D{D}", lambda: K.attn_bwd(q, k, v, o, dO, lse, B, H, Nq, Nk, D, D ** -0.5, dq, dk, dv), 10.0 * B * H * Nq * Nk * D)
if args.json:
    json.dump({"what": "tools/tile_order_ab.py: us per launch under each tile -> XCD order, table-chosen tile / split-K", "rows": rows},
              open(args.json, "w"), indent=1)
