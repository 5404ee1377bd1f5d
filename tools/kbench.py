"""Micro-benchmark of the individual HIP kernels at the SD-1.5 site shapes (B=4, 512^2).
Prints achieved TFLOP/s (MFMA kernels) or GB/s (HBM-bound kernels).  Run on the GPU box."""
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch

from controllora_amd import kernels as K

dev = "cuda"
f16 = torch.float16


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


rows = []


def report(name, secs, flops=None, bytes_=None):
    r = dict(kernel=name, us=round(secs * 1e6, 1))
    if flops:
        r["TFLOPs"] = round(flops / secs / 1e12, 1)
    if bytes_:
        r["GBs"] = round(bytes_ / secs / 1e9, 1)
    rows.append(r)
    print(json.dumps(r), flush=True)


SWEEP = False


def _sweep_or_auto(name, fn, flops, nbytes):
    s = timeit(lambda: fn(0, 0))
    report(f"{name} auto", s, flops, nbytes)
    if SWEEP:
        for tile in (1, 2, 3, 11, 12, 13):
            for split in (1, 2, 4, 8):
                try:
                    s = timeit(lambda: fn(split, tile), iters=8, warm=2)
                except Exception as e:      # workspace too small etc.
                    continue
                report(f"{name} tile={tile} sk={split}", s, flops, nbytes)


def bench_gemm(name, M, N, Kd):
    A = torch.randn(M, Kd, device=dev).half()
    B = (torch.randn(N, Kd, device=dev) / math.sqrt(Kd)).half()
    out = torch.empty(M, N, device=dev, dtype=f16)
    _sweep_or_auto(f"gemm {name} {M}x{N}x{Kd}", lambda sk, tc: K.gemm(A, B, M, N, Kd, out=out, split_k=sk, tile_cfg=tc),
                   2.0 * M * N * Kd, 2.0 * (M * Kd + N * Kd + M * N))


def bench_conv(name, Bn, H, Ci, Co):
    x = torch.randn(Bn, H * H, Ci, device=dev).half()
    w = (torch.randn(Co, 9 * Ci, device=dev) / math.sqrt(9 * Ci)).half()
    cd, Ho, Wo = K.conv_fwd_desc(H, H, Ci)
    M = Bn * Ho * Wo
    out = torch.empty(M, Co, device=dev, dtype=f16)
    _sweep_or_auto(f"conv3x3 {name} B{Bn} {H}^2 {Ci}->{Co}",
                   lambda sk, tc: K.gemm(x, w, M, Co, 9 * Ci, conv=cd, out=out, split_k=sk, tile_cfg=tc),
                   2.0 * M * Co * 9 * Ci, 2.0 * (M * Ci + Co * 9 * Ci + M * Co))


def bench_attn(B, H, N, Nk, D):
    q = torch.randn(B * N, H * D, device=dev).half()
    k = torch.randn(B * Nk, H * D, device=dev).half()
    v = torch.randn(B * Nk, H * D, device=dev).half()
    sc = D ** -0.5
    o, lse = K.attn_fwd(q, k, v, B, H, N, Nk, D, sc)
    s = timeit(lambda: K.attn_fwd(q, k, v, B, H, N, Nk, D, sc, out=o))
    fl = 4.0 * B * H * N * Nk * D
    report(f"attn_fwd B{B} H{H} N{N} Nk{Nk} D{D}", s, fl)
    dO = torch.randn_like(q)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    s = timeit(lambda: K.attn_bwd(q, k, v, o, dO, lse, B, H, N, Nk, D, sc, dq, dk, dv))
    report(f"attn_bwd B{B} H{H} N{N} Nk{Nk} D{D}", s, 2.5 * fl)


def bench_norm(B, HW, C):
    x = torch.randn(B, HW, C, device=dev).half()
    g, b = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    s = timeit(lambda: K.groupnorm_fwd(x, g, b, 32, 1e-5, True))
    report(f"groupnorm_silu_fwd B{B} HW{HW} C{C}", s, bytes_=3.0 * x.numel() * 2)
    y, st = K.groupnorm_fwd(x, g, b, 32, 1e-5, True)
    s = timeit(lambda: K.groupnorm_bwd(x, y, g, b, st, 32, True))
    report(f"groupnorm_silu_bwd B{B} HW{HW} C{C}", s, bytes_=5.0 * x.numel() * 2)
    x2 = x.reshape(B * HW, C)
    s = timeit(lambda: K.layernorm_fwd(x2, g, b, 1e-5))
    report(f"layernorm_fwd M{B*HW} C{C}", s, bytes_=2.0 * x.numel() * 2)


def bench_lora(M, Kd):
    X = torch.randn(M, Kd, device=dev).half()
    D = torch.randn(8, Kd, device=dev)
    T = torch.zeros(M, 8, device=dev)
    s = timeit(lambda: K.lora_down(X, D, T, 0, M, Kd))
    report(f"lora_down M{M} K{Kd} R8", s, bytes_=X.numel() * 2.0)
    G = torch.zeros(Kd, 8, device=dev)
    s = timeit(lambda: K.lora_wgrad(X, T, 0, G, 8, 1, M, Kd, 8))
    report(f"lora_wgrad M{M} N{Kd} R8", s, bytes_=X.numel() * 2.0)


if __name__ == "__main__":
    torch.manual_seed(0)
    SWEEP = "--sweep" in sys.argv
    sys.argv = [a for a in sys.argv if a != "--sweep"]
    Bn = 4
    bench_gemm("qkv.s0", Bn * 4096, 960, 320)
    bench_gemm("out.s0", Bn * 4096, 320, 320)
    bench_gemm("ff1.s0", Bn * 4096, 2560, 320)
    bench_gemm("ff2.s0", Bn * 4096, 320, 1280)
    bench_gemm("qkv.s1", Bn * 1024, 1920, 640)
    bench_gemm("ff1.s1", Bn * 1024, 5120, 640)
    bench_gemm("ff1.s2", Bn * 256, 10240, 1280)
    bench_gemm("ff2.s2", Bn * 256, 1280, 5120)
    bench_gemm("big", 8192, 8192, 8192)
    bench_gemm("sq.s2", Bn * 256, 1280, 1280)
    bench_gemm("sq.s1", Bn * 1024, 640, 640)
    bench_gemm("qkvT.s2", Bn * 256, 1280, 3840)
    bench_gemm("sq.s3", Bn * 64, 1280, 1280)
    bench_gemm("kv.s2", Bn * 77, 2560, 768)
    bench_conv("res.s0", Bn, 64, 320, 320)
    bench_conv("res.s1", Bn, 32, 640, 640)
    bench_conv("res.s2", Bn, 16, 1280, 1280)
    bench_conv("res.s3", Bn, 8, 1280, 1280)
    bench_conv("res.s3", Bn, 8, 2560, 1280)
    bench_conv("up.s0", Bn, 64, 960, 320)
    bench_attn(Bn, 8, 4096, 4096, 40)
    bench_attn(Bn, 8, 4096, 77, 40)
    bench_attn(Bn, 8, 1024, 1024, 80)
    bench_attn(Bn, 8, 256, 256, 160)
    bench_norm(Bn, 4096, 320)
    bench_norm(Bn, 1024, 640)
    bench_lora(Bn * 4096, 320)
    bench_norm(Bn, 262144, 32)
    bench_attn(Bn, 8, 1024, 77, 80)
    if len(sys.argv) > 1:
        with open(sys.argv[1], "w") as f:
            json.dump(rows, f, indent=1)
