"""DIAGNOSTIC (never imported by the package): the vendor libraries beside the clora kernels on the heaviest signatures of the train
step, same box, same process -- so that the next optimisation target is chosen from MEASURED gaps.

    python tools/vendor_yardstick.py > profiles/r06_vendor_yardstick.txt

Per signature: the clora kernel as the step launches it (tuned tile / split-K, plain epilogue) against
  plain GEMM  : torch.nn.functional.linear, fp16                     (hipBLASLt / rocBLAS behind torch)
  3x3 conv    : torch.nn.functional.conv2d, fp16, channels_last      (MIOpen)
  attention   : torch.nn.functional.scaled_dot_product_attention     (torch's flash / mem-efficient kernels on ROCm), fwd and fwd+bwd
Medians of interleaved repetitions, HIP events.  The vendor ops are a YARDSTICK: they compute the bare contraction (no fused
bias / adapter / residual / GEGLU epilogue, NCHW<->NHWC conversions not counted), so a ratio near 1.0 means "the main loop is at the
vendor's level", not "replace the kernel"."""
from __future__ import annotations

import os
import statistics
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from controllora_amd import kernels as K  # noqa: E402

# (M, N, K) of the plain GEMMs and (B, H, W, Cin, Cout) of the 3x3 stride-1 convs that carry most of the step's flops
# (tools/count_launches.py --detail, configs/fill50k.json 512x512 bs 4), calls per step in the comment
GEMMS = [(16384, 320, 320), (4096, 640, 640), (1024, 1280, 1280),             # q/out/proj projections: 49 / 50 / 50
         (16384, 960, 320), (4096, 1920, 640), (1024, 3840, 1280),            # q|k|v: 5 + dgrads
         (16384, 2560, 320), (4096, 5120, 640), (1024, 10240, 1280),          # GEGLU proj: 5 each
         (16384, 320, 1280), (4096, 640, 2560), (1024, 1280, 5120),           # FF out: 5 each
         (16384, 320, 2560), (4096, 640, 5120), (1024, 1280, 10240),          # dgrad of the GEGLU proj
         (256, 1280, 1280), (308, 15360, 768)]
CONVS = [(4, 64, 64, 320, 320), (4, 32, 32, 640, 640), (4, 16, 16, 1280, 1280), (4, 8, 8, 1280, 1280),   # 13 / 13 / 15 / 23
         (4, 64, 64, 640, 320), (4, 64, 64, 960, 320), (4, 32, 32, 1280, 640), (4, 32, 32, 1920, 640),
         (4, 16, 16, 2560, 1280), (4, 16, 16, 1920, 1280), (4, 32, 32, 320, 640), (4, 16, 16, 640, 1280)]
ATTN = [(4, 8, 4096, 4096, 40), (4, 8, 1024, 1024, 80), (4, 8, 256, 256, 160), (4, 8, 4096, 77, 40), (4, 8, 1024, 77, 80)]


def timed(fns, reps=7, inner=3):
    """interleaved medians (us) of several callables"""
    for f in fns:
        f()
    torch.cuda.synchronize()
    ts = [[] for _ in fns]
    for _ in range(reps):
        for i, f in enumerate(fns):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(inner):
                f()
            e1.record()
            torch.cuda.synchronize()
            ts[i].append(e0.elapsed_time(e1) * 1e3 / inner)
    return [statistics.median(t) for t in ts]


def main():
    dev = "cuda"
    g = torch.Generator(device=dev).manual_seed(0)
    rnd = lambda *s: (torch.rand(*s, device=dev, generator=g) - 0.5).half()
    print(f"# vendor yardstick on {torch.cuda.get_device_name(0)}, torch {torch.__version__}; times in us, TFLOP/s = 2MNK / t")
    print(f"{'signature':38s} {'clora us':>9s} {'TF/s':>7s} {'vendor us':>10s} {'TF/s':>7s} {'clora/vendor':>12s}")
    tot_c = tot_v = 0.0
    for M, N, Kd in GEMMS:
        A, W = rnd(M, Kd), rnd(N, Kd)
        tc, tv = timed([lambda: K.gemm(A, W, M, N, Kd), lambda: F.linear(A, W)])
        fl = 2.0 * M * N * Kd
        print(f"gemm {M}x{N}x{Kd:<24d} {tc:9.1f} {fl / tc / 1e6:7.0f} {tv:10.1f} {fl / tv / 1e6:7.0f} {tc / tv:12.2f}")
        tot_c += tc; tot_v += tv
    from controllora_amd import ops
    for B, H, W_, Ci, Co in CONVS:
        w = (torch.randn(Co, Ci, 3, 3, device=dev, generator=g) / (3 * Ci ** 0.5)).half()
        pack = ops.ConvPack(w, None)
        x = rnd(B * H * W_, Ci)
        cd, Ho, Wo = K.conv_fwd_desc(H, W_, pack.Cip, 3, 1, 1, False, False, pack.kchunk)
        xn = x.reshape(B, H, W_, Ci).permute(0, 3, 1, 2)                      # NCHW view of NHWC memory = channels_last
        wc = w.contiguous(memory_format=torch.channels_last)
        tc, tv = timed([lambda: K.gemm(x, pack.w, B * H * W_, pack.Cop, 9 * pack.Cip, conv=cd), lambda: F.conv2d(xn, wc, padding=1)])
        fl = 2.0 * B * H * W_ * Co * 9 * Ci
        print(f"conv3x3 {B}x{H}x{W_} {Ci}->{Co:<17d} {tc:9.1f} {fl / tc / 1e6:7.0f} {tv:10.1f} {fl / tv / 1e6:7.0f} {tc / tv:12.2f}")
        tot_c += tc; tot_v += tv
    for B, Hh, Nq, Nk, D in ATTN:
        C_ = Hh * D
        q, k, v = rnd(B * Nq, C_), rnd(B * Nk, C_), rnd(B * Nk, C_)
        dO = rnd(B * Nq, C_)
        sc = D ** -0.5
        q4 = q.reshape(B, Nq, Hh, D).transpose(1, 2).detach().requires_grad_(True)
        k4 = k.reshape(B, Nk, Hh, D).transpose(1, 2).detach().requires_grad_(True)
        v4 = v.reshape(B, Nk, Hh, D).transpose(1, 2).detach().requires_grad_(True)
        dO4 = dO.reshape(B, Nq, Hh, D).transpose(1, 2)

        def c_fwd():
            return K.attn_fwd(q, k, v, B, Hh, Nq, Nk, D, sc)

        o, lse = c_fwd()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)

        def c_bwd():
            K.attn_bwd(q, k, v, o, dO, lse, B, Hh, Nq, Nk, D, sc, dq, dk, dv)

        def v_fwd():
            with torch.no_grad():
                return F.scaled_dot_product_attention(q4, k4, v4)

        def v_fwd_bwd():
            y = F.scaled_dot_product_attention(q4, k4, v4)
            y.backward(dO4)
            q4.grad = k4.grad = v4.grad = None

        try:
            tcf, tcb, tvf, tvfb = timed([c_fwd, c_bwd, v_fwd, v_fwd_bwd], reps=5, inner=2)
        except Exception as e:                                   # noqa: BLE001 -- a vendor path that is missing on this build
            print(f"attention B{B} H{Hh} {Nq}x{Nk} d{D}: vendor path failed: {e!r}"[:160])
            continue
        ff = 4.0 * B * Hh * Nq * Nk * D
        print(f"attn fwd B{B} H{Hh} {Nq}x{Nk} d{D:<14d} {tcf:9.1f} {ff / tcf / 1e6:7.0f} {tvf:10.1f} {ff / tvf / 1e6:7.0f} {tcf / tvf:12.2f}")
        tvb = max(tvfb - tvf, 1e-3)
        print(f"attn bwd B{B} H{Hh} {Nq}x{Nk} d{D:<14d} {tcb:9.1f} {2.5 * ff / tcb / 1e6:7.0f} {tvb:10.1f} {2.5 * ff / tvb / 1e6:7.0f} {tcb / tvb:12.2f}"
              "   (vendor bwd = fwd+bwd - fwd)")
    print(f"# sum over the GEMM / conv signatures above (one call each): clora {tot_c:.0f} us, vendor {tot_v:.0f} us, ratio {tot_c / tot_v:.2f}")


if __name__ == "__main__":
    main()
