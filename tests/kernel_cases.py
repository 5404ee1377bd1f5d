"""Per-kernel parity cases shared by the CPU (host-emulated kernels, small shapes) and GPU (-m gpu, real
gfx950 library, real site shapes) test modules.  Every case compares a C-ABI entry point with the same
operation written in plain torch fp32 on the SAME fp16-rounded inputs; tolerances are stated per case
(fp16 output rounding is 2^-11 = 4.9e-4 relative)."""
import math

import os

import torch
import torch.nn.functional as F

from controllora_amd import kernels as K

f16, f32 = torch.float16, torch.float32


def rel(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return float((a - b).norm() / (b.norm() + 1e-20))


def no_outliers(out, ref, what=""):
    """elementwise guard beside the rel-L2 checks: a handful of wrong elements in a large tensor (a race that corrupts a few
    rows of a few tiles -- seen on hardware with an epilogue variant in round 3) moves rel-L2 by 1e-4..1e-3 only and would
    pass a norm-wise limit; no element may be further from the reference than a few fp16 ulps of the largest value."""
    out, ref = out.detach().float().cpu(), ref.detach().float().cpu()
    lim = 6e-3 * max(1.0, float(ref.abs().max()))
    worst = float((out - ref).abs().max())
    assert worst < lim, f"{what}: {int(((out - ref).abs() >= lim).sum())} elements off by up to {worst:.3f} (limit {lim:.3f})"


def rnd(shape, dev, gen, scale=1.0, dtype=f16):
    return (torch.randn(shape, generator=gen) * scale).to(dtype).to(dev)


def case_gemm_plain(dev, M, N, K_, split_k=1, seed=0, tile_cfg=0):
    g = torch.Generator().manual_seed(seed)
    A, B = rnd((M, K_), dev, g), rnd((N, K_), dev, g, 1 / math.sqrt(K_))
    out = K.gemm(A, B, M, N, K_, split_k=split_k, tile_cfg=tile_cfg)
    assert rel(out, A.float() @ B.float().T) < 6e-4


def case_gemm_epilogue(dev, M=200, N=96, K_=64, split_k=1, tile_cfg=0):
    g = torch.Generator().manual_seed(1)
    A, B = rnd((M, K_), dev, g), rnd((N, K_), dev, g, 1 / math.sqrt(K_))
    bias, rowadd, res = rnd((N,), dev, g, dtype=f32), rnd((4, N), dev, g), rnd((M, N), dev, g)
    T, U = rnd((M, 8), dev, g, dtype=f32), rnd((N, 4), dev, g, dtype=f32)
    out = K.gemm(A, B, M, N, K_, bias=bias, rowadd=rowadd, rows_per_batch=M // 4, residual=res, lora_t=T, lora_u=U,
                 lora_seg=N // 2, lora_scale=0.7, split_k=split_k, tile_cfg=tile_cfg)
    seg = torch.arange(N, device=dev) // (N // 2)
    lora = torch.stack([T[:, int(s) * 4:(int(s) + 1) * 4] @ U[n] for n, s in enumerate(seg.tolist())], 1)
    ref = (A.float() @ B.float().T + bias + rowadd.float().repeat_interleave(M // 4, 0) + 0.7 * lora).half().float() + res.float()
    assert rel(out, ref) < 6e-4


def case_gemm_epilogue_no_rowadd(dev, M=300, N=320, K_=128, tile_cfg=0, split_k=1):
    """The epilogue shapes of the attention / transformer projections (no time-embedding row add): the 8-wave tiles take
    their two-phase chunk loop here (all T rows / residual chunks requested before the first is used) -- (a) bias + rank-4
    adapter + residual, N split in two adapter segments; (b) the transposed-U form of the dgrad launches; (c) bias +
    residual without an adapter; (d) no operands at all.  Ragged M (rows past the matrix re-read the tile's first row)."""
    g = torch.Generator().manual_seed(41)
    A, B = rnd((M, K_), dev, g), rnd((N, K_), dev, g, 1 / math.sqrt(K_))
    bias, res = rnd((N,), dev, g, dtype=f32), rnd((M, N), dev, g)
    base = A.float() @ B.float().T
    nseg = 2 if N % 32 == 0 else 1                     # lora_seg must be a multiple of 16
    seg_w = N // nseg
    T, U = rnd((M, 4 * nseg), dev, g, dtype=f32), rnd((N, 4), dev, g, dtype=f32)
    seg = (torch.arange(N) // seg_w).tolist()
    lora = torch.stack([T[:, s_ * 4:(s_ + 1) * 4] @ U[n] for n, s_ in enumerate(seg)], 1)
    kw = dict(split_k=split_k, tile_cfg=tile_cfg)
    out = K.gemm(A, B, M, N, K_, bias=bias, residual=res, lora_t=T, lora_u=U, lora_seg=seg_w, lora_scale=0.7, **kw)
    assert rel(out, (base + bias + 0.7 * lora).half().float() + res.float()) < 6e-4
    no_outliers(out, (base + bias + 0.7 * lora).half().float() + res.float(), "bias + adapter + residual")
    again = K.gemm(A, B, M, N, K_, bias=bias, residual=res, lora_t=T, lora_u=U, lora_seg=seg_w, lora_scale=0.7, **kw)
    assert torch.equal(out, again), "the same launch twice must give the same bits"
    out = K.gemm(A, B, M, N, K_, lora_t=T, lora_u=U, lora_seg=seg_w, lora_scale=1.0, **kw)            # no bias, no residual
    assert rel(out, (base + lora).half().float()) < 6e-4
    no_outliers(out, (base + lora).half().float(), "adapter only")
    Ut = rnd((4, N), dev, g, dtype=f32)                                                                  # dgrad form: u(n, j) = Ut[j, n]
    out = K.gemm(A, B, M, N, K_, residual=res, lora_t=T[:, :4].contiguous(), lora_u=Ut, lora_seg=N, lora_u_tr=True, lora_r=4, **kw)
    assert rel(out, (base + T[:, :4] @ Ut).half().float() + res.float()) < 6e-4
    no_outliers(out, (base + T[:, :4] @ Ut).half().float() + res.float(), "transposed-U adapter + residual")
    again = K.gemm(A, B, M, N, K_, residual=res, lora_t=T[:, :4].contiguous(), lora_u=Ut, lora_seg=N, lora_u_tr=True, lora_r=4, **kw)
    assert torch.equal(out, again), "the same launch twice must give the same bits"
    out = K.gemm(A, B, M, N, K_, bias=bias, residual=res, **kw)
    assert rel(out, (base + bias).half().float() + res.float()) < 6e-4
    no_outliers(out, (base + bias).half().float() + res.float(), "bias + residual")
    out = K.gemm(A, B, M, N, K_, residual=res, **kw)
    assert rel(out, base.half().float() + res.float()) < 6e-4


def case_gemm_fused_down(dev, M=150, N=320, K_=128, nseg=1, tile_cfg=0, u_tr=False, t_in_rows=None, bias=True, residual=True):
    """clora_epilogue_t.lora_dpack: the rank-4 adapter down-projection T = A . D^T evaluated inside the projection GEMM
    (reference models.py:232-282: `attn.to_x(h) + scale * to_x_lora(h)`).  Checks (a) the T the launch WRITES against fp64
    A . D^T (+ the precomputed part) and against the stand-alone clora_lora_down kernel, (b) the GEMM output against the
    unfused formula, elementwise, (c) bit-identical repeat launches.  t_in_rows: None = no precomputed part, 0 = one row per GEMM
    row, > 0 = broadcast rows (control batch 1).  nseg column segments of N / nseg columns (a multiple of 320)."""
    from controllora_amd import ops
    g = torch.Generator().manual_seed(43)
    A, B = rnd((M, K_), dev, g), rnd((N, K_), dev, g, 1 / math.sqrt(K_))
    bias_t = rnd((N,), dev, g, dtype=f32) if bias else None
    res = rnd((M, N), dev, g) if residual else None
    seg_w = N // nseg
    Ds = [rnd((4, K_), dev, g, 1 / math.sqrt(K_), dtype=f32) for _ in range(nseg)]
    if u_tr:
        assert nseg == 1
        U = rnd((4, N), dev, g, dtype=f32)
    else:
        U = rnd((N, 4), dev, g, dtype=f32)
    t_in, mask = None, 0
    if t_in_rows is not None:
        rows = t_in_rows or M
        t_in = rnd((rows, 4 * nseg), dev, g, dtype=f32)
        mask = 1                                                  # segment 0 only (the q adapter)
    T_ref = torch.cat([A.double() @ D.double().T for D in Ds], 1)
    if t_in is not None:
        T_ref[:, :4] += (t_in.double().repeat(M // rows, 1) if rows != M else t_in.double())[:, :4]
    pack = ops.ADAPTER_PACKS.get(Ds)
    kw = dict(bias=bias_t, residual=res, lora_u=U, lora_seg=seg_w, lora_scale=0.7, lora_u_tr=u_tr, lora_r=4,
              split_k=1, tile_cfg=tile_cfg, _tuned=False)
    T = torch.full((M, 4 * nseg), float("nan"), dtype=f32, device=dev)
    out = K.gemm(A, B, M, N, K_, lora_t=T, lora_dpack=pack, lora_t_in=t_in, lora_t_in_mask=mask,
                 lora_t_in_rows=(t_in_rows or 0) if t_in is not None else 0, **kw)
    eT = float((T.double().cpu() - T_ref.cpu()).norm() / T_ref.norm())
    assert eT < 2e-6, f"T written by the fused launch: rel {eT:.2e}"
    T2 = torch.empty_like(T)
    for s_, D in enumerate(Ds):                                   # the stand-alone kernel computes the same quantity
        K.lora_down(A, D, T2, 4 * s_, M, K_)
    if t_in is None:
        assert rel(T, T2) < 2e-6
    seg = (torch.arange(N) // seg_w).tolist()
    Tr = T_ref.float().to(dev)
    if u_tr:
        lora = Tr @ U
    else:
        lora = torch.stack([Tr[:, s_ * 4:(s_ + 1) * 4] @ U[n] for n, s_ in enumerate(seg)], 1)
    ref = A.float() @ B.float().T + 0.7 * lora
    if bias:
        ref = ref + bias_t
    ref = ref.half().float()
    if residual:
        ref = ref + res.float()
    assert rel(out, ref) < 6e-4
    no_outliers(out, ref, "fused down-projection")
    T3 = torch.empty_like(T)
    again = K.gemm(A, B, M, N, K_, lora_t=T3, lora_dpack=pack, lora_t_in=t_in, lora_t_in_mask=mask,
                   lora_t_in_rows=(t_in_rows or 0) if t_in is not None else 0, **kw)
    assert torch.equal(out, again) and torch.equal(T, T3), "the same launch twice must give the same bits"
    # the unfused launch fed with the T the fused one wrote gives the same output bits (same epilogue arithmetic)
    bn = {51: 320, 52: 320, 54: 320, 55: 320, 21: 128, 41: 128, 22: 64, 42: 64, 26: 64, 23: 64, 43: 64}.get(tile_cfg, 0)
    if bn == 0 or seg_w % bn:                                    # the library's replacement rule (include/clora.h, lora_dpack)
        tile_cfg = (54 if M >= 32768 else 55) if seg_w % 320 == 0 else 43
    kw["tile_cfg"] = tile_cfg                                    # the tile the fused launch ran on
    unf = K.gemm(A, B, M, N, K_, lora_t=T, **kw)
    assert rel(unf, out) < 2e-5 or torch.equal(unf, out)      # same T, same epilogue formula; the non-hoisted variants order the four products differently
    return eT


def case_rank_control(dev, Mc=300, Cc=64, C_=128, rc=4, n=3, scale=0.8, strided_grad=True):
    """ops.control_terms_rank (clora_rank_compose / _mix / _compose_bwd + the deferred weight-gradient jobs) against the formula it
    replaces, evaluated by torch autograd in fp32: c_l = s U_c,l (D_c,l ctrl), Tq_l = c_l D_q,l^T (reference models.py:214-218,
    237-238) -- outputs, d ctrl and the gradients of all three matrices of every site."""
    from controllora_amd import ops
    g = torch.Generator().manual_seed(17)
    ctrl = rnd((Mc, Cc), dev, g).requires_grad_(True)
    P = lambda *sh, sc=1.0: torch.nn.Parameter((torch.randn(sh, generator=g) * sc).to(dev))
    layers = [(P(rc, Cc, sc=Cc ** -0.5), P(C_, rc, sc=0.5), P(4, C_, sc=C_ ** -0.5)) for _ in range(n)]
    outs = ops.control_terms_rank(ctrl, layers, scale)
    # the incoming gradients: column blocks of one [Mc, 12] buffer (what lora_proj hands back for a fused q | k | v projection)
    gbuf = [torch.randn((Mc, 12), generator=g).to(dev) for _ in range(n)]
    grads = [b[:, :4] if strided_grad else b[:, :4].contiguous() for b in gbuf]
    torch.autograd.backward(outs, grads)
    K.lora_wgrad_flush()
    c32 = ctrl.detach().float().clone().requires_grad_(True)
    refp = [tuple(w.detach().clone().requires_grad_(True) for w in tri) for tri in layers]
    ref = [(scale * ((c32 @ Dc.T) @ Uc.T)) @ Dq.T for Dc, Uc, Dq in refp]
    torch.autograd.backward(ref, [b[:, :4].float() for b in gbuf])
    errs = {"Tq": max(rel(o, r_) for o, r_ in zip(outs, ref)), "dctrl": rel(ctrl.grad, c32.grad)}
    for k, idx in (("dDc", 0), ("dUc", 1), ("dDq", 2)):
        errs[k] = max(rel(layers[l][idx].grad, refp[l][idx].grad) for l in range(n))
    assert errs["Tq"] < 1e-5 and errs["dctrl"] < 1e-3 and errs["dUc"] < 1e-5 and errs["dDq"] < 1e-5 and errs["dDc"] < 5e-4, errs
    return errs


def case_conv(dev, Bn, H, W, Ci, Co, stride=1, pad=1, ups=False, asym=False, seed=2, tile_cfg=0, kchunk=0):
    """forward, dgrad and wgrad of one 3x3 conv configuration against F.conv2d autograd.
    kchunk > 0: forward and dgrad additionally run with the channel-chunk-major K order (clora_conv_t.kchunk)."""
    g = torch.Generator().manual_seed(seed)
    x = rnd((Bn, Ci, H, W), dev, g)
    w = rnd((Co, Ci, 3, 3), dev, g, 1 / math.sqrt(Ci * 9))
    xin = x.float()
    if ups:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    xin = xin.clone().requires_grad_(True)
    w32 = w.float().clone().requires_grad_(True)
    y = F.conv2d(F.pad(xin, (0, 1, 0, 1)), w32, stride=2) if asym else F.conv2d(xin, w32, stride=stride, padding=pad)
    Ho, Wo = y.shape[2:]
    dy = rnd((Bn, Co, Ho, Wo), dev, g)
    y.backward(dy.float())
    xn = x.permute(0, 2, 3, 1).contiguous()
    wp = w.permute(0, 2, 3, 1).contiguous().reshape(Co, 9 * Ci)
    cd, Ho2, Wo2 = K.conv_fwd_desc(H, W, Ci, 3, stride, pad, upsample=ups, asym_pad=asym)
    assert (Ho2, Wo2) == (Ho, Wo)
    M = Bn * Ho * Wo
    out = K.gemm(xn, wp, M, Co, 9 * Ci, conv=cd, tile_cfg=tile_cfg, split_k=1 if tile_cfg else 0)
    assert rel(out, y.detach().permute(0, 2, 3, 1).reshape(M, Co)) < 6e-4
    if kchunk:
        from controllora_amd.ops import conv_k_order
        cdk, _, _ = K.conv_fwd_desc(H, W, Ci, 3, stride, pad, upsample=ups, asym_pad=asym, kchunk=kchunk)
        for sk in (1, 2, 3):
            outk = K.gemm(xn, conv_k_order(wp.reshape(Co, 9, Ci), kchunk), M, Co, 9 * Ci, conv=cdk, tile_cfg=tile_cfg, split_k=sk)
            assert rel(outk, y.detach().permute(0, 2, 3, 1).reshape(M, Co)) < 6e-4, (sk, rel(outk, out))
    Hi, Wi = xin.shape[2:]
    wd = w.permute(1, 2, 3, 0).contiguous().reshape(Ci, 9 * Co)
    dyn = dy.permute(0, 2, 3, 1).contiguous()
    cdd = K.conv_dgrad_desc(Ho, Wo, Co, Hi, Wi, 3, 2 if asym else stride, pad, asym_pad=asym)
    dx = K.gemm(dyn, wd, Bn * Hi * Wi, Ci, 9 * Co, conv=cdd, tile_cfg=tile_cfg, split_k=2 if tile_cfg else 0)
    assert rel(dx, xin.grad.permute(0, 2, 3, 1).reshape(-1, Ci)) < 6e-4
    if kchunk and Co % kchunk == 0:
        from controllora_amd.ops import conv_k_order
        cddk = K.conv_dgrad_desc(Ho, Wo, Co, Hi, Wi, 3, 2 if asym else stride, pad, asym_pad=asym, kchunk=kchunk)
        dxk = K.gemm(dyn, conv_k_order(wd.reshape(Ci, 9, Co), kchunk), Bn * Hi * Wi, Ci, 9 * Co, conv=cddk, tile_cfg=tile_cfg,
                     split_k=2 if tile_cfg else 0)
        assert rel(dxk, xin.grad.permute(0, 2, 3, 1).reshape(-1, Ci)) < 6e-4
    if ups:
        pooled = K.pool2x2_sum(dx.reshape(Bn, Hi * Wi, Ci), Bn, H, W, Ci)
        ref = F.avg_pool2d(xin.grad, 2) * 4
        assert rel(pooled, ref.permute(0, 2, 3, 1).reshape(Bn, H * W, Ci)) < 1e-3
    dW, db = K.conv_wgrad(dyn, xn, M, Co, 9 * Ci, cd, with_bias=True)
    assert rel(dW, w32.grad.permute(0, 2, 3, 1).reshape(Co, 9 * Ci)) < 1e-4
    assert rel(db, dy.float().sum((0, 2, 3))) < 1e-4
    assert rel(K.colsum(dyn.reshape(M, Co), M, Co), dy.float().sum((0, 2, 3))) < 1e-4
    if Ci % 8 == 0 and not ups:
        # trainable-conv path: one pack launch for both fp16 operands, gradients accumulated into OIHW .grad buffers
        wf, wdg = K.conv_weight_pack(w.float().contiguous(), Ci, True)
        assert torch.equal(wf, wp) and torch.equal(wdg, wd)
        gW = torch.ones((Co, Ci, 3, 3), dtype=f32, device=dev)
        gb = torch.ones((Co,), dtype=f32, device=dev)
        K.conv_wgrad_into(dyn, xn, M, Co, 9 * Ci, cd, gW, gb)
        assert rel(gW - 1, w32.grad) < 1e-4 and rel(gb - 1, dy.float().sum((0, 2, 3))) < 1e-4
        stage = torch.zeros(Co * 9 * Ci + Co, dtype=f32, device=dev)          # persistent staging + unpack (the product path)
        gW.fill_(1.0), gb.fill_(1.0)
        for rep in (1, 2):
            K.conv_wgrad_staged(dyn, xn, M, Co, 9 * Ci, cd, stage, gW, gb, Ci)
            assert rel(gW - 1, rep * w32.grad) < 1e-4 and rel(gb - 1, rep * dy.float().sum((0, 2, 3))) < 1e-4
            assert float(stage.abs().max()) == 0.0


def case_conv_wgrad_patch(dev, Bn, H, W, Ci, Co, seed=21, stride=1):
    """Round 6 (conv_wgrad_patch_kernel): the weight / bias gradient of a large-map 3x3 stride-1 pad-1 convolution with 32 or 64 input
    channels (the hint encoder's 512^2 / 256^2 / 128^2 stages, reference models.py:470-543 ConvBlock2D / SimpleDownEncoderBlock2D under
    loss.backward(), train...:786) on the patch-staged kernel vs fp32 torch autograd, and vs the gather kernel it replaces there
    ("wgrad_patch" 0) -- same products, another summation order (fp32 atomics in both)."""
    g = torch.Generator().manual_seed(seed)
    x = rnd((Bn, Ci, H, W), dev, g)
    w = rnd((Co, Ci, 3, 3), dev, g, 1 / math.sqrt(9 * Ci))
    Ho, Wo = (H, W) if stride == 1 else (H // 2, W // 2)              # stride 2: the downsamplers' F.pad(0, 1, 0, 1) + stride 2 (H, W even)
    dy = rnd((Bn, Co, Ho, Wo), dev, g)
    w32 = w.float().clone().requires_grad_(True)
    (F.conv2d(x.float(), w32, padding=1) if stride == 1 else F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w32, stride=2)).backward(dy.float())
    xn = x.permute(0, 2, 3, 1).contiguous().reshape(-1, Ci)
    dyn = dy.permute(0, 2, 3, 1).contiguous().reshape(-1, Co)
    cd, Ho2, Wo2 = K.conv_fwd_desc(H, W, Ci, 3, stride, 1 if stride == 1 else 0, asym_pad=stride == 2)
    assert (Ho2, Wo2) == (Ho, Wo)
    M = Bn * Ho * Wo
    want_w, want_b = w32.grad.permute(0, 2, 3, 1).reshape(Co, 9 * Ci), dy.float().sum((0, 2, 3))
    dW, db = K.conv_wgrad(dyn, xn, M, Co, 9 * Ci, cd, with_bias=True)
    assert rel(dW, want_w) < 1e-4 and rel(db, want_b) < 1e-4, (rel(dW, want_w), rel(db, want_b))
    no_outliers(dW, want_w, "conv_wgrad_patch dW")
    K.set_option("wgrad_patch", 0)
    try:
        dW0, db0 = K.conv_wgrad(dyn, xn, M, Co, 9 * Ci, cd, with_bias=True)
    finally:
        K.set_option("wgrad_patch", int(os.environ.get("CLORA_WGRAD_PATCH", "1")))
    assert rel(dW, dW0) < 2e-5 and rel(db, db0) < 2e-5
    dW2 = K.conv_wgrad(dyn, xn, M, Co, 9 * Ci, cd)                      # without the bias gradient
    assert rel(dW2, want_w) < 1e-4


def case_conv_strip(dev, Bn, H, W, Ci, Co, seed=22):
    """Round 6 (conv3x3_strip_kernel, tile_cfg 61): forward (+ bias) and dgrad of a large-map 3x3 stride-1 pad-1 convolution with 32 / 64
    channels (the hint encoder's 512^2 / 256^2 stages, reference models.py:470-543) through the default launch path -- kernels.gemm picks
    the strip kernel where clora_conv_strip_eligible says so -- vs fp32 torch, and vs the implicit GEMM it replaces there."""
    import ctypes
    from controllora_amd import capi
    g = torch.Generator().manual_seed(seed)
    x = rnd((Bn, Ci, H, W), dev, g)
    w = rnd((Co, Ci, 3, 3), dev, g, 1 / math.sqrt(9 * Ci))
    bias = rnd((Co,), dev, g, dtype=f32)
    xin = x.float().clone().requires_grad_(True)
    y = F.conv2d(xin, w.float(), bias.float(), padding=1)
    dy = rnd((Bn, Co, H, W), dev, g)
    y.backward(dy.float())
    xn = x.permute(0, 2, 3, 1).contiguous().reshape(-1, Ci)
    wp = w.permute(0, 2, 3, 1).contiguous().reshape(Co, 9 * Ci)
    cd, _, _ = K.conv_fwd_desc(H, W, Ci, 3, 1, 1)
    M = Bn * H * W
    lib = capi.lib().cdll
    assert lib.clora_conv_strip_eligible(M, Co, ctypes.byref(cd)) == 1
    out = K.gemm(xn, wp, M, Co, 9 * Ci, conv=cd, bias=bias)
    want = y.detach().permute(0, 2, 3, 1).reshape(M, Co)
    assert rel(out, want) < 6e-4, rel(out, want)
    no_outliers(out, want, "conv_strip fwd")
    ref = K.gemm(xn, wp, M, Co, 9 * Ci, conv=cd, bias=bias, tile_cfg=2, split_k=1)         # the implicit GEMM on the same operands
    assert rel(out, ref) < 3e-4
    assert torch.equal(out, K.gemm(xn, wp, M, Co, 9 * Ci, conv=cd, bias=bias))             # bit-stable
    wd = w.permute(1, 2, 3, 0).contiguous().reshape(Ci, 9 * Co)
    dyn = dy.permute(0, 2, 3, 1).contiguous().reshape(-1, Co)
    cdd = K.conv_dgrad_desc(H, W, Co, H, W, 3, 1, 1)
    wantd = xin.grad.permute(0, 2, 3, 1).reshape(-1, Ci)
    if lib.clora_conv_strip_eligible(M, Ci, ctypes.byref(cdd)) == 1:
        dx = K.gemm(dyn, wd, M, Ci, 9 * Co, conv=cdd)
        assert rel(dx, wantd) < 6e-4, rel(dx, wantd)
        no_outliers(dx, wantd, "conv_strip dgrad")
    # a launch the strip kernel cannot take (residual in the epilogue) falls back to the library's own choice under tile_cfg 61
    res = rnd((M, Co), dev, g)
    fb = K.gemm(xn, wp, M, Co, 9 * Ci, conv=cd, bias=bias, residual=res, tile_cfg=61, split_k=1)
    assert rel(fb, want.float().cpu() + res.float().cpu()) < 6e-4


def case_conv_patch_upsampled(dev, Bn, H, W, Ci, Co, tile_cfg, seed=13):
    """conv3x3_patch_kernel on conv(nearest-2x(x)) (Upsample2D): the patch lives at the output resolution"""
    from controllora_amd.ops import conv_k_order
    g = torch.Generator().manual_seed(seed)
    x = rnd((Bn, Ci, H, W), dev, g)
    w = rnd((Co, Ci, 3, 3), dev, g, 1 / math.sqrt(Ci * 9))
    y = F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), padding=1)
    M = Bn * 4 * H * W
    xn = x.permute(0, 2, 3, 1).contiguous().reshape(Bn * H * W, Ci)
    wp = conv_k_order(w.permute(0, 2, 3, 1).contiguous().reshape(Co, 9, Ci), 64)
    cd, Ho, Wo = K.conv_fwd_desc(H, W, Ci, 3, 1, 1, upsample=True, kchunk=64)
    assert (Ho, Wo) == (2 * H, 2 * W) and K.conv_patch_eligible(M, cd, tile_cfg)
    yref = y.permute(0, 2, 3, 1).reshape(M, Co)
    for sk in (1, 2):
        out = K.gemm(xn, wp, M, Co, 9 * Ci, conv=cd, tile_cfg=tile_cfg, split_k=sk)
        assert rel(out, yref) < 6e-4, (sk, rel(out, yref))


def case_conv_patch(dev, Bn, H, W, Ci, Co, tile_cfg, seed=12, fwd_only=False):
    """conv3x3_patch_kernel (tile_cfg 71..75): forward and dgrad of a stride-1 pad-1 conv with the slab-major K order,
    split-K 1..3, bias + residual epilogue, against F.conv2d autograd -- and the shape must really take the patch path."""
    from controllora_amd.ops import conv_k_order
    g = torch.Generator().manual_seed(seed)
    x = rnd((Bn, Ci, H, W), dev, g)
    w = rnd((Co, Ci, 3, 3), dev, g, 1 / math.sqrt(Ci * 9))
    xin = x.float().clone().requires_grad_(True)
    w32 = w.float().clone().requires_grad_(True)
    y = F.conv2d(xin, w32, padding=1)
    dy = rnd((Bn, Co, H, W), dev, g)
    y.backward(dy.float())
    M = Bn * H * W
    xn = x.permute(0, 2, 3, 1).contiguous().reshape(M, Ci)
    wp = conv_k_order(w.permute(0, 2, 3, 1).contiguous().reshape(Co, 9, Ci), 64)
    cd, _, _ = K.conv_fwd_desc(H, W, Ci, 3, 1, 1, kchunk=64)
    assert K.conv_patch_eligible(M, cd, tile_cfg), "shape does not take the patch kernel"
    yref = y.detach().permute(0, 2, 3, 1).reshape(M, Co)
    bias = rnd((Co,), dev, g, dtype=f32)
    res = rnd((M, Co), dev, g)
    for sk in (1, 2, 3):
        out = K.gemm(xn, wp, M, Co, 9 * Ci, conv=cd, tile_cfg=tile_cfg, split_k=sk)
        assert rel(out, yref) < 6e-4, (sk, rel(out, yref))
    out = K.gemm(xn, wp, M, Co, 9 * Ci, conv=cd, tile_cfg=tile_cfg, split_k=1, bias=bias, residual=res)
    assert rel(out, yref.float().cpu() + bias.float().cpu() + res.float().cpu()) < 6e-4
    out = K.gemm(xn, wp, M, Co, 9 * Ci, conv=cd, _tuned=False)                 # the library's own choice for an untuned shape
    assert rel(out, yref) < 6e-4
    if Co % 64 == 0 and not fwd_only:
        wd = conv_k_order(w.permute(1, 2, 3, 0).contiguous().reshape(Ci, 9, Co), 64)
        dyn = dy.permute(0, 2, 3, 1).contiguous().reshape(M, Co)
        cdd = K.conv_dgrad_desc(H, W, Co, H, W, 3, 1, 1, kchunk=64)
        assert K.conv_patch_eligible(M, cdd, tile_cfg)
        for sk in (1, 2):
            dx = K.gemm(dyn, wd, M, Ci, 9 * Co, conv=cdd, tile_cfg=tile_cfg, split_k=sk)
            assert rel(dx, xin.grad.permute(0, 2, 3, 1).reshape(M, Ci)) < 6e-4, (sk,)


def case_tile_order(dev, tile_cfg, order, seed=21):
    """the tile -> XCD assignment permutes which workgroup computes which tile and nothing else: plain GEMM with split-K and the
    adapter epilogue, and a 3x3 conv (patch kernel for tile_cfg 71..76), give the same bits under every order"""
    from controllora_amd.ops import conv_k_order
    g = torch.Generator().manual_seed(seed)
    M, N, K_ = 600, 336, 192                       # several tiles each way, ragged edges (N % 16 == 0 for the adapter epilogue)
    if order == "grid":                            # whole divisors each way for every tile shape up to 256 x 256: the rectangle assignment applies
        M, N = 1024, 512
    A, B = rnd((M, K_), dev, g), rnd((N, K_), dev, g, 1 / math.sqrt(K_))
    bias, res = rnd((N,), dev, g, dtype=f32), rnd((M, N), dev, g)
    T, U = rnd((M, 4), dev, g, dtype=f32), rnd((N, 4), dev, g, dtype=f32)
    Bn, H, W, Ci, Co = (8, 8, 8, 128, 256) if order == "grid" else (4, 8, 8, 128, 136)
    x = rnd((Bn * H * W, Ci), dev, g)
    w = conv_k_order(rnd((Co, 9, Ci), dev, g, 1 / math.sqrt(Ci * 9)), 64)
    cd, _, _ = K.conv_fwd_desc(H, W, Ci, 3, 1, 1, kchunk=64)
    patch = tile_cfg >= 71

    def run():
        outs = []
        if not patch:
            for sk in ((1, 2, 3) if order == "grid" else (1, 3)):
                outs.append(K.gemm(A, B, M, N, K_, split_k=sk, tile_cfg=tile_cfg))
            outs.append(K.gemm(A, B, M, N, K_, bias=bias, residual=res, lora_t=T, lora_u=U, lora_scale=0.5, split_k=1, tile_cfg=tile_cfg))
        for sk in (1, 2):
            outs.append(K.gemm(x, w, Bn * H * W, Co, 9 * Ci, conv=cd, tile_cfg=tile_cfg, split_k=sk))
        return outs

    try:
        K.set_tile_order("m")
        base = run()
        K.set_tile_order(order)
        other = run()
    finally:
        K.set_tile_order(K.DEFAULT_TILE_ORDER)
    assert rel(base[0], A.float() @ B.float().T) < 6e-4 if not patch else True
    for a, b in zip(base, other):
        assert torch.equal(a, b)


def case_conv_padded_channels(dev, seed=9):
    """3 -> padded 8 input channels (the hint encoder's conv_in): packing pads with zeros, the OIHW gradient drops them"""
    g = torch.Generator().manual_seed(seed)
    Bn, H, W, Ci, Cip, Co = 2, 8, 8, 3, 8, 16
    x = rnd((Bn, Ci, H, W), dev, g)
    w = rnd((Co, Ci, 3, 3), dev, g, dtype=f32)
    xin = x.float().clone()
    w32 = w.half().float().clone().requires_grad_(True)
    y = F.conv2d(xin, w32, padding=1)
    dy = rnd((Bn, Co, H, W), dev, g)
    y.backward(dy.float())
    xn = torch.zeros((Bn, H, W, Cip), dtype=f16, device=dev)
    xn[..., :Ci] = x.permute(0, 2, 3, 1)
    wf, _ = K.conv_weight_pack(w, Cip, False)
    cd, Ho, Wo = K.conv_fwd_desc(H, W, Cip, 3, 1, 1)
    M = Bn * Ho * Wo
    out = K.gemm(xn.reshape(M, Cip), wf, M, Co, 9 * Cip, conv=cd)
    assert rel(out, y.detach().permute(0, 2, 3, 1).reshape(M, Co)) < 1e-3
    gW = torch.zeros((Co, Ci, 3, 3), dtype=f32, device=dev)
    dyn = dy.permute(0, 2, 3, 1).contiguous().reshape(M, Co)
    K.conv_wgrad_into(dyn, xn.reshape(M, Cip), M, Co, 9 * Cip, cd, gW, None)
    assert rel(gW, w32.grad) < 1e-4
    stage = torch.zeros(Co * 9 * Cip + Co, dtype=f32, device=dev)
    K.conv_wgrad_staged(dyn, xn.reshape(M, Cip), M, Co, 9 * Cip, cd, stage, gW, None, Cip)
    assert rel(gW, 2 * w32.grad) < 1e-4 and float(stage.abs().max()) == 0.0


def _attn_ref(q, k, v, H, scale):
    B, Nq, HD = q.shape
    D = HD // H
    qh = q.float().reshape(B, Nq, H, D).permute(0, 2, 1, 3)
    kh = k.float().reshape(B, -1, H, D).permute(0, 2, 1, 3)
    vh = v.float().reshape(B, -1, H, D).permute(0, 2, 1, 3)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * scale, dim=-1)
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, Nq, HD)


def case_attention(dev, B, H, Nq, Nk, D, seed=3, fused_qkv=False, tol=2e-3, ramp=0.0):
    """ramp > 0: keys grow along the sequence (k_j scaled by 1 + ramp*j/Nk) and q is 3x larger, so the row maxima keep
    rising from KV tile to KV tile by more than the forward kernel's lazy-rescale threshold: the rebase path runs on
    later tiles too, for some queries of a wave and not for others."""
    g = torch.Generator().manual_seed(seed)
    scale = D ** -0.5
    if fused_qkv:   # q,k,v are column slices of one [B*N, 3*H*D] buffer (self-attention layout)
        qkv = rnd((B * Nq, 3 * H * D), dev, g)
        q2, k2, v2 = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]
    else:
        q2, k2, v2 = rnd((B * Nq, H * D), dev, g), rnd((B * Nk, H * D), dev, g), rnd((B * Nk, H * D), dev, g)
    if ramp > 0:
        assert not fused_qkv
        w = 1.0 + ramp * torch.arange(Nk, dtype=f32).repeat(B)[:, None] / Nk
        k2 = (k2.float().cpu() * w).to(f16).to(dev)
        q2 = (q2.float() * 3.0).to(f16)
    q3 = q2.reshape(B, Nq, H * D).float().clone().requires_grad_(True)
    k3 = k2.reshape(B, Nk, H * D).float().clone().requires_grad_(True)
    v3 = v2.reshape(B, Nk, H * D).float().clone().requires_grad_(True)
    ref = _attn_ref(q3, k3, v3, H, scale)
    o, lse = K.attn_fwd(q2, k2, v2, B, H, Nq, Nk, D, scale)
    assert rel(o, ref.detach().reshape(B * Nq, H * D)) < tol, rel(o, ref.detach().reshape(B * Nq, H * D))
    dO = rnd((B * Nq, H * D), dev, g)
    ref.backward(dO.reshape(B, Nq, H * D).float())
    dq, dk, dv = torch.empty_like(q2.contiguous()), torch.empty_like(k2.contiguous()), torch.empty_like(v2.contiguous())
    K.attn_bwd(q2, k2, v2, o, dO, lse, B, H, Nq, Nk, D, scale, dq, dk, dv)
    for name, a, b_ in (("dq", dq, q3.grad), ("dk", dk, k3.grad), ("dv", dv, v3.grad)):
        e = rel(a, b_.reshape(a.shape))
        assert e < 2 * tol, (name, e)


def case_attention_negative_logits(dev, B, H, Nq, Nk, D, seed=31):
    """every logit of the FIRST KV tile far below -88 (q = +8, k = -8 on every channel: logit
    -64 * sqrt(D)): the lazy exponent reference must not rescale the still-empty accumulators by exp2(+huge) = inf
    (0 * inf = NaN).  Later tiles carry ordinary keys, so the result is an ordinary softmax over those."""
    g = torch.Generator().manual_seed(seed)
    scale = D ** -0.5
    q2 = torch.full((B * Nq, H * D), 8.0, dtype=f16, device=dev)
    k2 = rnd((B * Nk, H * D), dev, g, scale=0.05)
    v2 = rnd((B * Nk, H * D), dev, g)
    k3 = k2.reshape(B, Nk, H * D).clone()
    k3[:, :min(64, Nk)] = -8.0                               # the whole first tile (or every key when Nk <= 64)
    k2 = k3.reshape(B * Nk, H * D).contiguous()
    ref = _attn_ref(q2.reshape(B, Nq, H * D), k2.reshape(B, Nk, H * D), v2.reshape(B, Nk, H * D), H, scale)
    o, lse = K.attn_fwd(q2, k2, v2, B, H, Nq, Nk, D, scale)
    assert bool(torch.isfinite(o.float()).all()) and bool(torch.isfinite(lse).all())
    e = rel(o, ref.reshape(B * Nq, H * D))
    assert e < 2e-3, e
    dO = rnd((B * Nq, H * D), dev, g)
    dq, dk, dv = torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(v2)
    K.attn_bwd(q2, k2, v2, o, dO, lse, B, H, Nq, Nk, D, scale, dq, dk, dv)
    for t_ in (dq, dk, dv):
        assert bool(torch.isfinite(t_.float()).all())


def case_attention_block_order(dev, B, H, Nq, Nk, D, seed=23):
    """whole heads per XCD (tile_order != "m") only permutes which workgroup handles which (head, block): the forward
    output, LSE and the three gradients are bit-identical; the grid (3 x 6 blocks at Nq = 300, B*H = 6) is not a multiple of 8"""
    g = torch.Generator().manual_seed(seed)
    scale = D ** -0.5
    q2, k2, v2 = rnd((B * Nq, H * D), dev, g), rnd((B * Nk, H * D), dev, g), rnd((B * Nk, H * D), dev, g)
    dO = rnd((B * Nq, H * D), dev, g)

    def run():
        o, lse = K.attn_fwd(q2, k2, v2, B, H, Nq, Nk, D, scale)
        dq, dk, dv = torch.empty_like(q2), torch.empty_like(k2), torch.empty_like(v2)
        K.attn_bwd(q2, k2, v2, o, dO, lse, B, H, Nq, Nk, D, scale, dq, dk, dv)
        return o, lse, dq, dk, dv

    try:
        K.set_tile_order("m")
        base = run()
        K.set_tile_order("auto")
        other = run()
    finally:
        K.set_tile_order(K.DEFAULT_TILE_ORDER)
    for a, b_ in zip(base, other):
        assert torch.equal(a, b_)


def case_groupnorm(dev, B, HW, C, G, silu, eps=1e-5, seed=4, train_params=False):
    g = torch.Generator().manual_seed(seed)
    x = (rnd((B, HW, C), dev, g).float() * 1.5 + 0.3).half()
    gamma, beta = (1 + 0.2 * rnd((C,), dev, g, dtype=f32)), 0.2 * rnd((C,), dev, g, dtype=f32)
    x32 = x.float().clone().requires_grad_(True)
    g32, b32 = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y = F.group_norm(x32.permute(0, 2, 1), G, g32, b32, eps).permute(0, 2, 1)
    if silu:
        y = F.silu(y)
    out, stats = K.groupnorm_fwd(x, gamma, beta, G, eps, silu)
    assert rel(out, y.detach()) < 8e-4
    dy = rnd((B, HW, C), dev, g)
    y.backward(dy.float())
    dx, dg, db = K.groupnorm_bwd(x, dy, gamma, beta, stats, G, silu, want_param_grads=train_params)
    assert rel(dx, x32.grad) < 2e-3
    if train_params:
        assert rel(dg, g32.grad) < 1e-3 and rel(db, b32.grad) < 1e-3
    # fused second gradient (residual branch) and in-place accumulation of the affine gradients
    dres = rnd((B, HW, C), dev, g)
    acc_g, acc_b = torch.ones(C, dtype=f32, device=dev), torch.ones(C, dtype=f32, device=dev)
    dx2, _, _ = K.groupnorm_bwd(x, dy, gamma, beta, stats, G, silu, grads_into=(acc_g, acc_b) if train_params else None, dres=dres)
    assert rel(dx2, x32.grad + dres.float()) < 2e-3
    if train_params:
        assert rel(acc_g - 1, g32.grad) < 1e-3 and rel(acc_b - 1, b32.grad) < 1e-3
    else:
        # the one-launch (register-resident) passes against the two-launch scheme on the same inputs: same arithmetic, a different
        # summation order -- a few fp16 ulps on a handful of elements at most; shapes the one-launch plan does not take run the
        # same kernels twice
        K.set_option("gn_resident", 0)
        try:
            out0, stats0 = K.groupnorm_fwd(x, gamma, beta, G, eps, silu)
            dx0, _, _ = K.groupnorm_bwd(x, dy, gamma, beta, stats0, G, silu, dres=dres)
        finally:
            K.set_option("gn_resident", int(os.environ.get("CLORA_GN_RESIDENT", "1")))
        assert rel(out, out0) < 2e-4 and rel(stats, stats0) < 1e-5 and rel(dx2, dx0) < 3e-4
        out1, stats1 = K.groupnorm_fwd(x, gamma, beta, G, eps, silu)             # bit-stable
        assert torch.equal(out1, out) and torch.equal(stats1, stats)
        # the team kernels of the large maps (one launch, in-launch exchange of the partial sums) against the scheme they replace;
        # repeated calls walk the exchange epochs of the persistent state: bit-stable, and no exchange ever gave up
        K.set_option("gn_team", 0)
        try:
            out_t0, stats_t0 = K.groupnorm_fwd(x, gamma, beta, G, eps, silu)
            dx_t0, _, _ = K.groupnorm_bwd(x, dy, gamma, beta, stats_t0, G, silu, dres=dres)
        finally:
            K.set_option("gn_team", int(os.environ.get("CLORA_GN_TEAM", "2")))
        assert rel(out, out_t0) < 2e-4 and rel(stats, stats_t0) < 1e-5 and rel(dx2, dx_t0) < 3e-4
        for _ in range(2):
            dx3, _, _ = K.groupnorm_bwd(x, dy, gamma, beta, stats, G, silu, dres=dres)
            assert torch.equal(dx3, dx2)
        assert K.gn_team_errors(x.device) == 0


def case_groupnorm_concat(dev, B, HW, Ca, Cb, G, silu, eps=1e-5, seed=11):
    """Round 6 (include/clora.h clora_groupnorm_*_ex, x2 / dx2): GroupNorm over the channel concatenation of two tensors read in
    place == GroupNorm over the materialised concatenation, BIT for bit (same loads, same arithmetic, same order): output, statistics,
    the concatenated copy the kernel writes, and the backward's two gradient tensors."""
    g = torch.Generator().manual_seed(seed)
    a = (rnd((B, HW, Ca), dev, g).float() * 1.5 + 0.3).half()
    b = (rnd((B, HW, Cb), dev, g).float() * 0.7 - 0.2).half()
    C_ = Ca + Cb
    gamma, beta = (1 + 0.2 * rnd((C_,), dev, g, dtype=f32)), 0.2 * rnd((C_,), dev, g, dtype=f32)
    xc = torch.cat([a, b], -1).contiguous()
    y0, st0 = K.groupnorm_fwd(xc, gamma, beta, G, eps, silu)
    y1, st1, xcat = K.groupnorm_fwd(a, gamma, beta, G, eps, silu, x2=b)
    assert torch.equal(xcat, xc) and torch.equal(y0, y1) and torch.equal(st0, st1)
    dy, dres = rnd((B, HW, C_), dev, g), rnd((B, HW, C_), dev, g)
    for dr in (None, dres):
        dx0, _, _ = K.groupnorm_bwd(xc, dy, gamma, beta, st0, G, silu, dres=dr)
        (da, db), _, _ = K.groupnorm_bwd(xc, dy, gamma, beta, st0, G, silu, dres=dr, split_at=Ca)
        assert da.shape == (B, HW, Ca) and db.shape == (B, HW, Cb) and da.is_contiguous() and db.is_contiguous()
        assert torch.equal(da, dx0[..., :Ca]) and torch.equal(db, dx0[..., Ca:])


def case_deferred_finish(dev, B, HW, C, Kd, G, split, silu=True, seed=12, lora=False):
    """both settings of the "defer_max_rows" knob: the default (fold only in the few-rows-per-thread GroupNorm plans, finish first
    elsewhere) and 16 (fold wherever a folding kernel exists, incl. the LayerNorm backward) -- bit-identical either way"""
    for rows in (16, int(os.environ.get("CLORA_DEFER_MAX_ROWS", "4"))):
        K.set_option("defer_max_rows", rows)
        try:
            _case_deferred_finish(dev, B, HW, C, Kd, G, split, silu, seed, lora)
        finally:
            K.set_option("defer_max_rows", int(os.environ.get("CLORA_DEFER_MAX_ROWS", "4")))


def _case_deferred_finish(dev, B, HW, C, Kd, G, split, silu=True, seed=12, lora=False):
    """Round 6 (clora_deferred_t): a split-K GEMM whose finish pass is left to the GroupNorm / LayerNorm launch that reads its output
    == the same GEMM with its own finish pass followed by the plain norm launch, BIT for bit -- forward GroupNorm (the finished
    tensor must also land in the GEMM's C), GroupNorm backward and LayerNorm backward with the GEMM producing dy; epilogue = bias +
    row add + residual (+ a rank-4 adapter term for the LayerNorm case, what the attention projections' dgrad carries)."""
    g = torch.Generator().manual_seed(seed)
    M = B * HW
    A = rnd((M, Kd), dev, g)
    Wt = (rnd((C, Kd), dev, g).float() * 0.1).half()
    bias = rnd((C,), dev, g, dtype=f32)
    res = rnd((M, C), dev, g)
    rowadd = rnd((B, C), dev, g)
    gamma, beta = (1 + 0.2 * rnd((C,), dev, g, dtype=f32)), 0.2 * rnd((C,), dev, g, dtype=f32)
    kw = dict(bias=bias, residual=res, rowadd=rowadd, rows_per_batch=HW, split_k=split)
    if lora:
        kw.update(lora_t=rnd((M, 8), dev, g, dtype=f32), lora_u=rnd((8, C), dev, g, dtype=f32), lora_seg=C, lora_u_tr=True, lora_r=8)
    assert not K._PENDING
    ref = K.gemm(A, Wt, M, C, Kd, **kw)
    y0, st0 = K.groupnorm_fwd(ref.reshape(B, HW, C), gamma, beta, G, 1e-5, silu)
    out = K.gemm(A, Wt, M, C, Kd, defer=True, **kw)
    deferred = bool(K._PENDING)
    assert deferred == (split > 1 and K.DEFER_FINISH)
    y1, st1 = K.groupnorm_fwd(out.reshape(B, HW, C), gamma, beta, G, 1e-5, silu)
    assert not K._PENDING and torch.equal(out, ref) and torch.equal(y0, y1) and torch.equal(st0, st1)
    x = rnd((B, HW, C), dev, g)
    dres = rnd((B, HW, C), dev, g)
    _, stx = K.groupnorm_fwd(x, gamma, beta, G, 1e-5, silu)
    dx0, _, _ = K.groupnorm_bwd(x, ref.reshape(B, HW, C), gamma, beta, stx, G, silu, dres=dres)
    dyd = K.gemm(A, Wt, M, C, Kd, defer=True, **kw)
    dx1, _, _ = K.groupnorm_bwd(x, dyd.reshape(B, HW, C), gamma, beta, stx, G, silu, dres=dres)
    assert not K._PENDING and torch.equal(dx0, dx1)
    if C <= 1536:
        x2 = x.reshape(M, C)
        l0 = K.layernorm_bwd(x2, ref, gamma, 1e-5, dres=dres.reshape(M, C))
        dyd = K.gemm(A, Wt, M, C, Kd, defer=True, **kw)
        l1 = K.layernorm_bwd(x2, dyd, gamma, 1e-5, dres=dres.reshape(M, C))
        assert not K._PENDING and torch.equal(l0, l1)
    # the safety net: any OTHER kernel call first finishes what is pending
    dyd = K.gemm(A, Wt, M, C, Kd, defer=True, **kw)
    z = K.add(dyd, dyd)
    assert not K._PENDING and torch.equal(dyd, ref) and torch.equal(z, K.add(ref, ref))


def case_gemm_fused_layernorm(dev, M, Kd, tile_cfg, seed=13, lora=False, residual=True):
    """Round 6 (clora_epilogue_t.ln_out): the LayerNorm that follows a projection, written by the projection's own launch where one
    320-column tile spans the row, == clora_layernorm_fwd_f16 on the stored output (same partition / formulas / summation order: the
    compiler may contract the final affine differently in the two kernels, so equality is asserted up to one fp16 ulp on a handful of
    elements); C itself must be bit-identical to the launch without the fused norm."""
    g = torch.Generator().manual_seed(seed)
    N = 320
    A = rnd((M, Kd), dev, g)
    Wt = (rnd((N, Kd), dev, g).float() * 0.1).half()
    bias = rnd((N,), dev, g, dtype=f32)
    res = rnd((M, N), dev, g) if residual else None
    gamma, beta = (1 + 0.2 * rnd((N,), dev, g, dtype=f32)), 0.2 * rnd((N,), dev, g, dtype=f32)
    kw = dict(bias=bias, residual=res, tile_cfg=tile_cfg, split_k=1)
    if lora:
        kw.update(lora_t=rnd((M, 4), dev, g, dtype=f32), lora_u=rnd((N, 4), dev, g, dtype=f32) * 0.1, lora_seg=N)
    ref = K.gemm(A, Wt, M, N, Kd, **kw)
    ln_ref = K.layernorm_fwd(ref, gamma, beta, 1e-5)
    slot = K.LayerNormSlot(gamma, beta, 1e-5)
    out = K.gemm(A, Wt, M, N, Kd, ln=slot, **kw)
    assert slot.out is not None, "the launch did not take the fused LayerNorm"
    assert torch.equal(out, ref)
    diff = (slot.out.float() - ln_ref.float()).abs()
    ulp = ln_ref.float().abs().clamp_min(2.0 ** -10) * 2.0 ** -10
    assert bool((diff <= ulp).all()), float((diff / ulp).max())
    assert float((diff > 0).float().mean()) < 0.02
    # a shape / tile that cannot fuse leaves the slot empty (the caller launches the norm itself)
    slot2 = K.LayerNormSlot(gamma, beta, 1e-5)
    K.gemm(A, Wt, M, N, Kd, ln=slot2, bias=bias, tile_cfg=43, split_k=1)
    assert slot2.out is None


def case_softmax_rows(dev, rows, cols, scale=0.37, seed=6):
    g = torch.Generator().manual_seed(seed)
    x = (rnd((rows, cols), dev, g).float() * 4).half()
    y = K.softmax_rows(x, scale)
    ref = torch.softmax(x.float() * scale, -1)
    assert float((y.float() - ref).abs().max()) < 1e-3 and float((y.float().sum(-1) - 1).abs().max()) < 4e-3
    K.softmax_rows(x, scale, out=x)                      # in place
    assert torch.equal(x, y)


def case_layernorm(dev, M, C, seed=5):
    g = torch.Generator().manual_seed(seed)
    x = rnd((M, C), dev, g)
    gamma, beta = (1 + 0.2 * rnd((C,), dev, g, dtype=f32)), 0.2 * rnd((C,), dev, g, dtype=f32)
    x32 = x.float().clone().requires_grad_(True)
    y = F.layer_norm(x32, (C,), gamma, beta, 1e-5)
    assert rel(K.layernorm_fwd(x, gamma, beta, 1e-5), y.detach()) < 6e-4
    dy = rnd((M, C), dev, g)
    y.backward(dy.float())
    assert rel(K.layernorm_bwd(x, dy, gamma, 1e-5), x32.grad) < 1e-3
    dres = rnd((M, C), dev, g)                      # fused residual-branch gradient
    assert rel(K.layernorm_bwd(x, dy, gamma, 1e-5, dres=dres), x32.grad + dres.float()) < 1e-3


def case_layernorm_rows(dev, M, C, seed=25):
    """several rows in flight per wave (clora_set_option "ln_rows"): same per-row arithmetic as the one-row kernel, ragged last
    block (M not a multiple of the rows per block), forward / backward / backward with the residual-branch gradient"""
    g = torch.Generator().manual_seed(seed)
    x, dy, dres = rnd((M, C), dev, g), rnd((M, C), dev, g), rnd((M, C), dev, g)
    gamma, beta = (1 + 0.2 * rnd((C,), dev, g, dtype=f32)), 0.2 * rnd((C,), dev, g, dtype=f32)

    def run():
        return (K.layernorm_fwd(x, gamma, beta, 1e-5), K.layernorm_bwd(x, dy, gamma, 1e-5), K.layernorm_bwd(x, dy, gamma, 1e-5, dres=dres))

    try:
        K.set_option("ln_rows", 0)
        base = run()
        K.set_option("ln_rows", 1)
        rows = run()
    finally:
        K.set_option("ln_rows", 1)                                # the library default
    x32 = x.float().clone().requires_grad_(True)
    y = F.layer_norm(x32, (C,), gamma, beta, 1e-5)
    y.backward(dy.float())
    assert rel(rows[0], y.detach()) < 6e-4 and rel(rows[1], x32.grad) < 1e-3
    for a, b_ in zip(base, rows):
        assert rel(a, b_) < 1e-5, rel(a, b_)


def case_geglu(dev, M, Fdim, seed=6):
    g = torch.Generator().manual_seed(seed)
    h = rnd((M, 2 * Fdim), dev, g)
    h32 = h.float().clone().requires_grad_(True)
    a, gg = h32.chunk(2, -1)
    y = a * F.gelu(gg)
    assert rel(K.geglu_fwd(h), y.detach()) < 6e-4
    dy = rnd((M, Fdim), dev, g)
    y.backward(dy.float())
    assert rel(K.geglu_bwd(h, dy), h32.grad) < 8e-4


def case_lora(dev, M, Kd, N, R, x_rows=0, seed=7):
    g = torch.Generator().manual_seed(seed)
    rows = x_rows if x_rows else M
    X = rnd((rows, Kd), dev, g)
    D, U = rnd((R, Kd), dev, g, 0.3, f32), rnd((N, R), dev, g, 0.3, f32)
    T = torch.zeros((M, R + 4), dtype=f32, device=dev)
    K.lora_down(X, D, T, 4, M, Kd, x_rows=x_rows)
    Xf = X.float().repeat(M // rows, 1) if x_rows else X.float()
    Tref = Xf @ D.T
    assert rel(T[:, 4:], Tref) < 1e-5
    K.lora_down(X, D, T, 4, M, Kd, accumulate=True, x_rows=x_rows)
    assert rel(T[:, 4:], 2 * Tref) < 1e-5
    if R <= 16:                                   # second input by linearity: T = (X + X2) . D^T in one job
        X2 = rnd((M, Kd), dev, g)
        T2 = torch.zeros((M, R + 4), dtype=f32, device=dev)
        K.lora_down_multi([K.down_job(X, D, T2, 4, M, Kd, x_rows=x_rows, X2=X2)])
        assert rel(T2[:, 4:], Tref + X2.float() @ D.T) < 1e-5
    T[:, 4:] = Tref
    base = rnd((M, N), dev, g)
    y = K.lora_up(base, T, 4, U, M, N, 0.6)
    ref = base.float() + (0.6 * (Tref @ U.T).half().float()).half().float()
    assert rel(y, ref) < 6e-4
    y0 = K.lora_up(None, T, 4, U, M, N, 1.0)
    assert rel(y0, Tref @ U.T) < 6e-4
    A = rnd((rows, N), dev, g)
    G1 = torch.zeros((N, R), dtype=f32, device=dev)
    K.lora_wgrad(A, T, 4, G1, R, 1, M, N, R, scale=0.5, a_rows=x_rows)
    Af = A.float().repeat(M // rows, 1) if x_rows else A.float()
    assert rel(G1, 0.5 * Af.T @ Tref) < 1e-4
    G2 = torch.zeros((R, N), dtype=f32, device=dev)
    K.lora_wgrad(A, T, 4, G2, 1, N, M, N, R, scale=1.0, a_rows=x_rows)
    assert rel(G2, (Af.T @ Tref).T) < 1e-4
    if R <= 16:                                   # reduction over fp16(A + A2)
        A2 = rnd((M, N), dev, g)
        G3 = torch.zeros((N, R), dtype=f32, device=dev)
        K.lora_wgrad_multi([K.wgrad_job(A, T, 4, G3, R, 1, M, N, R, a_rows=x_rows, A2=A2)], dev)
        assert rel(G3, (Af + A2.float()).half().float().T @ Tref) < 1e-4


def case_elementwise(dev, seed=8):
    g = torch.Generator().manual_seed(seed)
    a, b = rnd((6, 40, 24), dev, g), rnd((6, 40, 24), dev, g)
    assert rel(K.add(a, b), a.float() + b.float()) < 5e-4
    assert rel(K.silu(a), F.silu(a.float())) < 5e-4
    a32 = a.float().clone().requires_grad_(True)
    F.silu(a32).backward(b.float())
    assert rel(K.silu_bwd(a, b), a32.grad) < 6e-4
    c = rnd((6, 40, 16), dev, g)
    cat = K.concat_channels(a, c)
    assert torch.equal(cat.cpu(), torch.cat([a, c], -1).cpu())
    sa, sc = K.split_channels(cat, 24)
    assert torch.equal(sa.cpu(), a.cpu()) and torch.equal(sc.cpu(), c.cpu())
    # sinusoidal timestep embedding in one launch (round 6) vs upstream's op sequence (get_timestep_embedding, flip_sin_to_cos, shift 0):
    # int64 per-sample timesteps (training), one fp32 value for the whole batch (samplers)
    half = 160
    freq = torch.exp(-math.log(10000) * torch.arange(half, dtype=f32, device=dev) / half).contiguous()
    for t, batch in ((torch.tensor([0, 999, 1, 500, 37], dtype=torch.long, device=dev), 5), (torch.tensor([981.0], dtype=f32, device=dev), 3)):
        arg = t.reshape(-1).expand(batch)[:, None].float() * freq[None, :]
        ref = torch.cat([torch.cos(arg), torch.sin(arg)], -1)
        out = K.timestep_embedding(t, batch, freq)
        assert out.shape == (batch, 2 * half) and out.dtype == f16
        assert float((out.float() - ref).abs().max()) <= 1e-3            # one fp16 rounding of a value in [-1, 1]
        frac_equal = float((out == ref.to(f16)).float().mean())
        assert frac_equal > 0.99, frac_equal


def case_loss_and_optimizer(dev, n=5000, seed=9):
    g = torch.Generator().manual_seed(seed)
    pred, tgt = rnd((n * 8,), dev, g), rnd((n * 8,), dev, g)
    loss = torch.zeros(1, dtype=f32, device=dev)
    dpred = torch.empty_like(pred)
    state = torch.zeros(16, dtype=f32, device=dev)
    state[3] = 1024.0
    K.mse(pred, tgt, loss, dpred, 2.0 / pred.numel(), state[3:4])
    d = pred.float() - tgt.float()
    assert abs(float(loss) / pred.numel() - float((d * d).mean())) < 1e-4
    assert rel(dpred, 1024.0 * 2.0 / pred.numel() * d) < 6e-4
    # optimizer vs torch.optim.AdamW + clip_grad_norm_ for 3 steps (one of them with an inf -> skipped)
    p0 = rnd((n,), dev, g, dtype=f32)
    p, m, v = p0.clone(), torch.zeros_like(p0), torch.zeros_like(p0)
    pt = torch.nn.Parameter(p0.clone())
    opt = torch.optim.AdamW([pt], lr=1e-2, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    for step in range(3):
        gr = rnd((n,), dev, g, 0.1, f32)
        scaled = gr * state[3]
        if step == 1:
            scaled = scaled.clone()
            scaled[7] = float("inf")
        K.grad_sumsq(scaled, state)
        K.optim_prep(state, 1.0, 0.9, 0.999, True, 2.0, 0.5, 2000)
        K.adamw_flat(p, scaled, m, v, state, 1e-2, 0.9, 0.999, 1e-8, 1e-2)
        if step != 1:
            pt.grad = gr.clone()
            torch.nn.utils.clip_grad_norm_([pt], 1.0)
            opt.step()
            assert float(state[3]) == (1024.0 if step == 0 else 512.0)
        else:
            assert float(state[6]) == 1.0 and float(state[3]) == 512.0
        assert rel(p, pt.detach()) < 1e-6, (step, rel(p, pt.detach()))
    assert float(state[2]) == 2.0


def case_feed_forward_fused(dev, M=200, C=64, seed=12, tile_cfg=0):
    """FeedForward(GEGLU) with the activation fused into the GEMM epilogues (ops.feed_forward: GegluPack + geglu = 1 / 2)
    against (i) the unfused kernels -- bit-identical, same rounding points -- and (ii) an fp32 torch reference
    (upstream FeedForward: Linear(C, 8C) -> a * gelu_erf(g) -> Linear(4C, C), + residual), forward and d(input)."""
    from controllora_amd import ops
    g = torch.Generator().manual_seed(seed)
    Fd = 4 * C
    x = rnd((M, C), dev, g)
    w1, b1 = rnd((2 * Fd, C), dev, g, C ** -0.5), rnd((2 * Fd,), dev, g, 0.1)
    w2, b2 = rnd((C, Fd), dev, g, Fd ** -0.5), rnd((C,), dev, g, 0.1)
    res = rnd((M, C), dev, g)
    dout = rnd((M, C), dev, g)
    p1, p2 = ops.GegluPack(w1, b1), ops.LinearPack(w2, b2)
    xf = x.clone().requires_grad_(True)
    old_tuned = K._TUNING
    if tile_cfg:
        K._TUNING = {K.tuning_key(M, 2 * Fd, C, None): [tile_cfg, 1], K.tuning_key(M, Fd, C, None): [tile_cfg, 1]}
    try:
        out = ops.feed_forward(xf, p1, p2, res)
        out.backward(dout)
    finally:
        K._TUNING = old_tuned
    # (i) unfused kernels
    lp1 = ops.LinearPack(w1, b1)
    xu = x.clone().requires_grad_(True)
    out_u = ops.frozen_linear(ops.geglu(ops.frozen_linear(xu, lp1)), p2, res)
    out_u.backward(dout)
    # round 6: the fused forward forms a * gelu(g) from the fp32 accumulators and adds the residual in fp32 (ONE rounding each, the
    # unfused kernels round a, g and the branch first): the two differ by fp16 rounding, and the fused result must not be farther
    # from the fp32 reference than the unfused one (checked below); the backward reads the same stored fp16 h in both
    assert rel(out, out_u) < 6e-4, rel(out, out_u)
    assert rel(xf.grad, xu.grad) < 2e-4, rel(xf.grad, xu.grad)
    # (ii) fp32 reference
    x32 = x.float().cpu().requires_grad_(True)
    h = F.linear(x32, w1.float().cpu(), b1.float().cpu())
    a, gg = h.chunk(2, -1)
    ref = F.linear(a * F.gelu(gg), w2.float().cpu(), b2.float().cpu()) + res.float().cpu()
    ref.backward(dout.float().cpu())
    assert rel(out, ref.detach()) < 2e-3, rel(out, ref.detach())
    assert rel(out, ref.detach()) <= 1.05 * rel(out_u, ref.detach()), (rel(out, ref.detach()), rel(out_u, ref.detach()))
    assert rel(xf.grad, x32.grad) < 3e-3, rel(xf.grad, x32.grad)
    # inference: no h is written
    with torch.no_grad():
        y_inf, h_inf = K.gemm(x, p1.w, M, p1.N, p1.K, bias=p1.bias, geglu=1, geglu_keep_h=False)
        assert h_inf is None
        out_inf = ops.feed_forward(x, p1, p2, res)
    assert rel(out_inf, out.detach()) < 1e-6


def case_gemm_trunk_lo(dev, M, N, K_, split_k=1, tile_cfg=0, conv_like=False, lora=False, seed=41):
    """Round 6, compensated residual trunk (include/clora.h clora_epilogue_t.residual_lo / c_lo): inside a kernels.TrunkLo window a chain
    of three residual launches  x1 = A1 W^T + x0,  x2 = A2 W^T + x1,  x3 = A3 W^T + x2  (x0 from a launch flagged `trunk`) carries the
    rounding remainder of every sum to the next add.  Checked: hi stays ONE fp16 rounding (2.0e-4 rel-L2) away from the un-rounded fp32
    chain at every link, hi + lo reproduces the fp32 chain to fp32-accumulation accuracy (the plain chain is 3-4e-4
    away from it), the window leaves nothing registered, and outside a window the same calls give the plain results bit for bit."""
    g = torch.Generator().manual_seed(seed)
    W = rnd((N, K_), dev, g, 1 / math.sqrt(K_))
    As = [rnd((M, K_), dev, g) for _ in range(4)]
    bias = rnd((N,), dev, g, dtype=f32)
    kw = dict(split_k=split_k, tile_cfg=tile_cfg)
    if lora:
        kw.update(lora_t=rnd((M, 4), dev, g, dtype=f32), lora_u=rnd((N, 4), dev, g, 0.1, dtype=f32), lora_seg=N)
    plain = [K.gemm(As[0], W, M, N, K_, bias=bias, **kw)]
    for i in range(1, 4):
        plain.append(K.gemm(As[i], W, M, N, K_, bias=bias, residual=plain[-1], **kw))
    with K.TrunkLo(True):
        hi = [K.gemm(As[0], W, M, N, K_, bias=bias, trunk=True, **kw)]
        los = [K._TRUNK_LO[hi[0].data_ptr()][1]]
        for i in range(1, 4):
            hi.append(K.gemm(As[i], W, M, N, K_, bias=bias, residual=hi[-1].reshape(M, N), **kw))
            los.append(K._TRUNK_LO[hi[-1].data_ptr()][1])
            assert hi[-2].data_ptr() not in K._TRUNK_LO              # consumed by the add that read it
    assert not K._TRUNK_LO and not K._TRUNK_LO_ON[0]
    Wf = W.float().cpu()
    upd = None
    if lora:
        upd = (kw["lora_t"].float().cpu() @ kw["lora_u"].float().cpu().T)
    ref = torch.zeros(M, N)
    for i in range(4):
        ref = ref + As[i].float().cpu() @ Wf.T + bias.float().cpu() + (upd if upd is not None else 0)
        comp = hi[i].float().cpu() + los[i].float().cpu()
        assert rel(comp, ref) < 3e-6 * (i + 1) + 2e-6, (i, rel(comp, ref))
        assert rel(hi[i], ref) < 2.3e-4                           # hi = ONE fp16 rounding away from the un-rounded chain at every link
    assert torch.equal(hi[0], plain[0])                              # the first launch has no residual: same C, plus its remainder
    assert rel(plain[3], ref) > 1.2 * rel(hi[3], ref) or rel(plain[3], ref) < 3e-4
    again = K.gemm(As[1], W, M, N, K_, bias=bias, residual=plain[0], **kw)
    assert torch.equal(again, plain[1])
