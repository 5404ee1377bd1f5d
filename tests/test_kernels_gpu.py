"""GPU parity tests (-m gpu): the gfx950 library through the C ABI vs plain torch fp32, at the real
attention-site / resnet shapes of SD-1.5 plus ragged edge cases."""
import os

import pytest
import torch

from controllora_amd import kernels as K
from tests import kernel_cases as KC

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(autouse=True, scope="module")
def _native_lib_loaded():
    from controllora_amd import capi
    L = capi.lib()
    assert L.require_device and L.path.endswith("libclora.so")
    yield


@pytest.mark.parametrize("M,N,K,split", [(300, 72, 96, 1), (4096, 960, 320, 1), (1024, 1280, 1280, 1), (256, 1280, 1280, 4),
                                         (308, 640, 768, 1), (4096, 2560, 320, 1), (1024, 320, 1280, 2), (64, 8, 40, 1)])
def test_gemm_plain(M, N, K, split):
    KC.case_gemm_plain(DEV, M, N, K, split)


@pytest.mark.parametrize("split", [1, 2])
def test_gemm_epilogue(split):
    KC.case_gemm_epilogue(DEV, M=2000, N=320, K_=320, split_k=split)


@pytest.mark.parametrize("kw", [dict(stride=1, pad=1), dict(stride=2, pad=1), dict(asym=True, stride=2, pad=0), dict(ups=True)])
def test_conv_fwd_dgrad_wgrad(kw):
    KC.case_conv(DEV, 2, 32, 32, 64, 128, **kw)


def test_conv_unet_shape():
    KC.case_conv(DEV, 1, 32, 32, 320, 640)


@pytest.mark.parametrize("Bn,H,W,Ci,Co", [(4, 512, 512, 32, 32), (4, 256, 256, 32, 64), (4, 128, 128, 64, 64), (2, 37, 192, 64, 64), (1, 5, 128, 32, 16)])
def test_conv_wgrad_patch_kernel(Bn, H, W, Ci, Co):
    KC.case_conv_wgrad_patch(DEV, Bn, H, W, Ci, Co)


@pytest.mark.parametrize("Bn,H,W,Ci,Co", [(4, 512, 512, 32, 32), (4, 256, 256, 64, 64), (2, 38, 384, 32, 64), (1, 6, 128, 64, 24)])
def test_conv_wgrad_patch_kernel_stride2(Bn, H, W, Ci, Co):
    """the downsamplers (reference models.py SimpleDownEncoderBlock2D: F.pad(0, 1, 0, 1) + 3x3 stride 2); H, W = INPUT size"""
    KC.case_conv_wgrad_patch(DEV, Bn, H, W, Ci, Co, stride=2)


@pytest.mark.parametrize("Bn,H,W,Ci,Co", [(4, 512, 512, 32, 32), (4, 256, 256, 32, 64), (2, 37, 384, 64, 32), (1, 5, 128, 32, 24), (4, 512, 512, 8, 32), (1, 6, 256, 8, 64)])
def test_conv_strip_kernel(Bn, H, W, Ci, Co):
    KC.case_conv_strip(DEV, Bn, H, W, Ci, Co)


def test_conv_small_channels():
    KC.case_conv(DEV, 1, 64, 64, 8, 32)


ALL_TILE_CFGS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 21, 22, 23, 26, 31, 32, 33, 41, 42, 43, 51, 52, 53, 54, 55, 56, 57, 58, 59]


@pytest.mark.parametrize("tile", ALL_TILE_CFGS)
def test_gemm_tile_configs(tile):
    """every main-loop variant (tile shape x ring depth), with ragged M/N/K, split-K and the fused epilogue"""
    KC.case_gemm_plain(DEV, 1000, 328, 1256, 1, tile_cfg=tile)
    KC.case_gemm_plain(DEV, 333, 640, 2568, 3, tile_cfg=tile)
    KC.case_gemm_epilogue(DEV, M=2000, N=320, K_=320, split_k=1, tile_cfg=tile)
    KC.case_conv(DEV, 2, 32, 32, 64, 128, tile_cfg=tile)
    KC.case_conv(DEV, 1, 32, 32, 320, 320, tile_cfg=tile)          # Cin % 64 == 0: the BK = 64 variants take the fast tap walk


@pytest.mark.parametrize("M,N,K,split", [(4096, 2560, 320, 1), (1000, 520, 1096, 2), (16384, 1280, 320, 1), (513, 264, 8200, 1), (2048, 2048, 2048, 1)])
def test_gemm_eight_phase_tile(M, N, K, split):
    """tile_cfg 59 (gemm_8p_kernel: 256x256 tile, two wave rows one barrier apart, half-tile LDS-DMA seven ahead): the level-0
    FeedForward shapes, ragged M / N / K tail, split-K, a long K; elementwise outlier guard and repeat-launch bit equality inside
    the case (a rare early read of a staged buffer would show as a few wrong tiles)"""
    import math
    import torch
    from controllora_amd import kernels as K_
    g = torch.Generator().manual_seed(M + K)
    A, B = KC.rnd((M, K), DEV, g), KC.rnd((N, K), DEV, g, 1 / math.sqrt(K))
    ref = A.float() @ B.float().T
    first = None
    for _ in range(4):
        out = K_.gemm(A, B, M, N, K, split_k=split, tile_cfg=59, _tuned=False)
        assert KC.rel(out, ref) < 6e-4
        KC.no_outliers(out, ref)
        if first is None:
            first = out.clone()
        assert torch.equal(out, first)


@pytest.mark.parametrize("tile", [71, 72, 73, 74, 75, 76, 79])  # 79 = 256x160 (64x80 wave tiles, all 160 KB of LDS): first hardware run in round 5
@pytest.mark.parametrize("Bn,H,W,Ci,Co", [(4, 64, 64, 320, 320), (4, 32, 32, 640, 640), (2, 16, 16, 1280, 640), (4, 8, 8, 1280, 1280),
                                          (1, 32, 32, 320, 320), (3, 8, 8, 128, 72)])
def test_conv_patch_kernel(tile, Bn, H, W, Ci, Co):
    """patch-staged 3x3 conv at the ResnetBlock2D shapes of every UNet level (forward, dgrad, split-K, fused epilogue)"""
    KC.case_conv_patch(DEV, Bn, H, W, Ci, Co, tile)


@pytest.mark.parametrize("tile", [71, 72, 76])
@pytest.mark.parametrize("Bn,H,Ci", [(2, 32, 640), (2, 16, 1280), (1, 8, 1280)])
def test_conv_patch_kernel_upsampled(tile, Bn, H, Ci):
    KC.case_conv_patch_upsampled(DEV, Bn, H, H, Ci, Ci, tile)


@pytest.mark.parametrize("tile", [77, 78])
@pytest.mark.parametrize("Bn,H,W,Ci,Co", [(1, 512, 512, 128, 128), (2, 256, 256, 256, 256), (2, 128, 128, 512, 512), (1, 256, 256, 128, 256)])
def test_conv_patch_kernel_row_segments_vae_shapes(tile, Bn, H, W, Ci, Co):
    """tile_cfg 77 / 78 (392-pixel patch, one 128-pixel row or row segment per tile) at the SD-1.5 VAE's level shapes"""
    KC.case_conv_patch(DEV, Bn, H, W, Ci, Co, tile, fwd_only=True)


@pytest.mark.parametrize("kw", [dict(M=16384, N=320, K_=320, tile_cfg=55), dict(M=16384, N=320, K_=320, tile_cfg=52, t_in_rows=0),
                                dict(M=16384, N=320, K_=320, tile_cfg=54, t_in_rows=4096), dict(M=16384, N=320, K_=320, tile_cfg=51, u_tr=True, bias=False),
                                dict(M=16384, N=960, K_=320, nseg=3, tile_cfg=0, t_in_rows=0), dict(M=32768, N=960, K_=320, nseg=3, tile_cfg=0, t_in_rows=4096),
                                dict(M=4096, N=640, K_=640, tile_cfg=0, t_in_rows=0), dict(M=1024, N=1280, K_=1280, tile_cfg=0, u_tr=True, residual=False),
                                dict(M=4099, N=1920, K_=640, nseg=3, tile_cfg=55, t_in_rows=0), dict(M=131072, N=320, K_=320, tile_cfg=0, t_in_rows=4096),
                                dict(M=4096, N=640, K_=640, tile_cfg=43, t_in_rows=0), dict(M=1024, N=1280, K_=1280, tile_cfg=43, u_tr=True, residual=False),
                                dict(M=4096, N=640, K_=640, tile_cfg=42), dict(M=256, N=1280, K_=1280, tile_cfg=23, t_in_rows=64),
                                dict(M=308, N=1280, K_=768, nseg=2, tile_cfg=43), dict(M=2048, N=1280, K_=1280, tile_cfg=22, u_tr=True, bias=False),
                                dict(M=1024, N=1280, K_=1280, tile_cfg=21, t_in_rows=0), dict(M=8192, N=640, K_=640, tile_cfg=41),
                                dict(M=4096, N=640, K_=640, tile_cfg=26)])
def test_gemm_with_adapter_down_projection_in_the_launch(kw):
    """clora_epilogue_t.lora_dpack at the projection shapes of the step (level 0 q | k | v, out, cross-attention q; the deeper
    levels; the batch-8 and batch-32 row counts; ragged M): T written == A . D^T, output == the unfused formula elementwise,
    repeat launches bit-identical (reference models.py:232-282)"""
    e = KC.case_gemm_fused_down(DEV, **kw)
    print("FUSED_DOWN", kw, f"T rel {e:.2e}")


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8, 21, 22, 23, 26, 31, 32, 33, 41, 42, 43, 51, 52, 53, 54, 55, 56, 57, 58])
def test_gemm_epilogue_without_rowadd(tile):
    """projection epilogues at real shapes (M = 16384 x N = 320 / 960, ragged variants): two-phase chunk loop of the 8-wave tiles"""
    KC.case_gemm_epilogue_no_rowadd(DEV, M=16384, N=320, K_=320, tile_cfg=tile)
    KC.case_gemm_epilogue_no_rowadd(DEV, M=4100, N=960, K_=320, tile_cfg=tile)
    KC.case_gemm_epilogue_no_rowadd(DEV, M=1000, N=1296, K_=1280, tile_cfg=tile, split_k=2 if tile else 0)


@pytest.mark.parametrize("order", ["n", "auto", "grid"])
@pytest.mark.parametrize("tile", [21, 43, 53, 58, 72, 76])
def test_tile_order_does_not_change_results(tile, order):
    """clora_set_option("tile_order"): which XCD computes which tile is a permutation -- bit-identical GEMM / conv outputs"""
    KC.case_tile_order(DEV, tile, order)


@pytest.mark.parametrize("B,H,Nq,Nk,D", [(2, 8, 4096, 4096, 40), (2, 8, 1024, 77, 80), (3, 5, 300, 300, 160)])
def test_attention_block_order_does_not_change_results(B, H, Nq, Nk, D):
    KC.case_attention_block_order(DEV, B, H, Nq, Nk, D)


@pytest.mark.parametrize("B,H,Nq,Nk,D,fused", [
    (1, 8, 4096, 4096, 40, True), (2, 8, 4096, 77, 40, False), (2, 8, 1024, 1024, 80, True), (2, 8, 1024, 77, 80, False),
    (2, 8, 256, 256, 160, True), (2, 8, 64, 77, 160, False), (1, 2, 70, 70, 40, True), (1, 1, 150, 77, 64, False),
    (1, 2, 20, 20, 8, False)])
def test_attention(B, H, Nq, Nk, D, fused):
    KC.case_attention(DEV, B, H, Nq, Nk, D, fused_qkv=fused)


@pytest.mark.parametrize("Nq,Nk,D", [(4096, 4096, 40), (300, 64, 40), (300, 77, 64), (1024, 1024, 80), (256, 256, 160)])
def test_attention_first_tile_far_below_zero(Nq, Nk, D):
    """first KV tile with every logit below -88: finite outputs, ordinary softmax over the later tiles"""
    KC.case_attention_negative_logits(DEV, 2, 8, Nq, Nk, D)


@pytest.mark.parametrize("Nq,Nk,D", [(1024, 4096, 40), (300, 1000, 80), (70, 300, 160)])
def test_attention_rising_maxima(Nq, Nk, D):
    """forward: lazy exponent reference, rebased on later KV tiles for a subset of the queries (D = 40: rowsum from the ones column)"""
    KC.case_attention(DEV, 2, 8, Nq, Nk, D, ramp=4.0, tol=3e-3)


@pytest.mark.parametrize("B,HW,C,G,silu,train", [(2, 4096, 320, 32, True, False), (2, 1024, 960, 32, True, False),
                                                 (1, 64, 2560, 32, True, False), (2, 4096, 32, 32, True, True),
                                                 (1, 1024, 640, 32, False, False), (1, 37, 64, 8, False, True),
                                                 (4, 256, 1280, 32, True, True), (4, 1024, 1920, 32, True, False),
                                                 (4, 256, 1280, 32, True, False), (4, 64, 2560, 32, False, False), (4, 256, 640, 32, True, False),
                                                 # the UNet's one-launch shapes (bs 4 / 8 / inference batch 32) + ragged row counts
                                                 (4, 64, 1280, 32, True, False), (4, 256, 2560, 32, True, False), (4, 256, 1920, 32, True, False),
                                                 (8, 1024, 640, 32, True, False), (4, 1024, 1280, 32, True, False), (32, 256, 1280, 32, True, False),
                                                 (2, 1000, 640, 32, False, False), (3, 130, 1920, 32, True, False),
                                                 # team plan: the train step's 64x64 / 32x32 maps (bs 4, bs 8), batch 1 / 2, ragged rows
                                                 (4, 4096, 320, 32, True, False), (4, 4096, 640, 32, True, False), (4, 4096, 960, 32, True, False),
                                                 (4, 1024, 320, 32, True, False), (4, 1024, 960, 32, True, False), (8, 4096, 320, 32, True, False),
                                                 (1, 4096, 320, 32, False, False), (2, 1500, 640, 32, True, False)])
def test_groupnorm(B, HW, C, G, silu, train):
    KC.case_groupnorm(DEV, B, HW, C, G, silu, train_params=train)


@pytest.mark.parametrize("B,HW,Ca,Cb,G,silu", [(4, 4096, 320, 320, 32, True), (4, 4096, 640, 320, 32, True), (4, 1024, 640, 640, 32, True),
                                               (4, 1024, 1280, 640, 32, True), (4, 256, 1280, 1280, 32, True), (4, 64, 1280, 1280, 32, True),
                                               (1, 300, 64, 32, 8, False)])
def test_groupnorm_concat_in_place(B, HW, Ca, Cb, G, silu):
    """the up-path resnets' norm1 at the train shapes (512^2 bs 4): concatenation read in place == materialised, bit for bit"""
    KC.case_groupnorm_concat(DEV, B, HW, Ca, Cb, G, silu)


@pytest.mark.parametrize("B,HW,C,Kd,G,split,lora", [(4, 256, 1280, 2560, 32, 4, False), (4, 1024, 640, 640, 32, 2, True), (4, 64, 1280, 1280, 32, 6, True),
                                                    (4, 4096, 320, 320, 32, 2, False),   # 64x64 maps: two-launch plan, finished first
                                                    (2, 16, 128, 256, 8, 2, False), (4, 256, 1280, 1280, 32, 3, True), (2, 64, 64, 128, 8, 1, False)])
def test_deferred_split_k_finish_in_the_norm_kernels(B, HW, C, Kd, G, split, lora):
    KC.case_deferred_finish(DEV, B, HW, C, Kd, G, split, lora=lora)


@pytest.mark.parametrize("M,Kd,tile,lora,res", [(16384, 320, 55, True, True), (16384, 320, 55, False, False), (16384, 1280, 55, False, True),
                                                (32768, 320, 54, True, True), (16384, 320, 52, True, True), (1000, 320, 51, False, True)])
def test_gemm_fused_layernorm(M, Kd, tile, lora, res):
    """the level-0 projections (16384 x 320) with the LayerNorm that follows them written by the same launch"""
    KC.case_gemm_fused_layernorm(DEV, M, Kd, tile, lora=lora, residual=res)


@pytest.mark.parametrize("M,C", [(4096, 320), (1024, 640), (259, 1280)])
def test_layernorm(M, C):
    KC.case_layernorm(DEV, M, C)


@pytest.mark.parametrize("M,C", [(16384, 320), (4099, 640), (2049, 1280), (1024, 1280), (308, 768)])
def test_layernorm_rows_in_flight(M, C):
    KC.case_layernorm_rows(DEV, M, C)


def test_geglu():
    KC.case_geglu(DEV, 4096, 1280)


@pytest.mark.parametrize("M,K,N,R,xr", [(8192, 320, 320, 4, 0), (8192, 320, 320, 8, 4096), (308, 768, 640, 8, 0), (512, 576, 320, 32, 0)])
def test_lora(M, K, N, R, xr):
    KC.case_lora(DEV, M, K, N, R, x_rows=xr)


@pytest.mark.parametrize("kw", [dict(Mc=16384, Cc=320, C_=320, rc=4, n=10), dict(Mc=4096, Cc=640, C_=640, rc=4, n=10),
                                dict(Mc=256, Cc=1280, C_=1280, rc=8, n=2, strided_grad=False)])
def test_control_terms_in_rank_space(kw):
    """reference models.py:214-218, 237-238 at the level shapes of the step: the rank-space kernels vs torch autograd of the formula"""
    print("RANK_CONTROL", kw, KC.case_rank_control(DEV, **kw))


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_lora_down_launch_modes(mode):
    """option "lora_down_mode": how an adapter down-projection is spread over waves (K-split everywhere / eight k-steps in flight):
    same T within fp32 summation-order noise at the level-0 size, stacked q|k|v rows with a second input on the first adapter"""
    for M, Kd in ((16384, 320), (1024, 1280), (200, 1288), (256, 2560)):
        _lora_down_mode_case(mode, M, Kd)


def _lora_down_mode_case(mode, M, Kd):
    g = torch.Generator().manual_seed(61)
    X, X2 = KC.rnd((M, Kd), DEV, g), KC.rnd((M, Kd), DEV, g)
    D = KC.rnd((12, Kd), DEV, g, 0.25, dtype=torch.float32)
    ref = X.float() @ D.T
    ref[:, :4] += X2.float() @ D[:4].T
    try:
        K.set_option("lora_down_mode", mode)
        T = torch.empty((M, 12), dtype=torch.float32, device=DEV)
        K.lora_down_multi([K.down_job(X, D, T, 0, M, Kd, X2=X2, r2=4)])
        again = torch.empty_like(T)
        K.lora_down_multi([K.down_job(X, D, again, 0, M, Kd, X2=X2, r2=4)])
    finally:
        K.set_option("lora_down_mode", 1)                   # the library default
    assert KC.rel(T, ref) < 1e-5 and torch.equal(T, again), (mode, M, Kd, KC.rel(T, ref))
    KC.no_outliers(T, ref, f"lora_down mode {mode} M={M} K={Kd}")


def test_elementwise():
    KC.case_elementwise(DEV)


def test_loss_and_optimizer():
    KC.case_loss_and_optimizer(DEV, n=100000)


@pytest.mark.parametrize("rows,cols", [(4096, 4096), (1000, 72), (7, 8192)])
def test_softmax_rows(rows, cols):
    KC.case_softmax_rows(DEV, rows, cols)


def test_conv_padded_channels_pack_and_oihw_grad():
    KC.case_conv_padded_channels(DEV)


@pytest.mark.parametrize("kw", [dict(stride=2, pad=1), dict(asym=True, stride=2, pad=0), dict(ups=True)])
def test_conv_fast_path_variants(kw):
    """strided / asymmetric-pad / upsampled 3x3 gathers and their dgrads at Cin % 32 == 0 (generic gather path; the
    wave-uniform fast path is reserved for stride-1: a tabulated-offset generalisation measured no faster than generic)"""
    KC.case_conv(DEV, 2, 16, 16, 64, 96, **kw)
    KC.case_conv(DEV, 1, 32, 32, 320, 320, **kw)


@pytest.mark.parametrize("tile", [0, 1, 7, 8, 21, 31, 41, 53, 56])
def test_feed_forward_fused_geglu(tile):
    KC.case_feed_forward_fused(DEV, M=1000, C=320, tile_cfg=tile)
    KC.case_feed_forward_fused(DEV, M=300, C=1280, tile_cfg=tile)


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 7, 8, 21, 22, 23, 26, 41, 43])
def test_conv_channel_chunk_major_k_order(tile):
    """clora_conv_t.kchunk = 64 (what ops.ConvPack uses for every UNet conv) at real widths, forward and dgrad, plus the
    strided / upsampled gathers"""
    KC.case_conv(DEV, 2, 32, 32, 320, 640, tile_cfg=tile, kchunk=64)
    KC.case_conv(DEV, 1, 16, 16, 1280, 1280, tile_cfg=tile, kchunk=64)
    for kw in (dict(stride=2, pad=1), dict(asym=True, stride=2, pad=0), dict(ups=True)):
        KC.case_conv(DEV, 1, 16, 16, 128, 64, tile_cfg=tile, kchunk=64, **kw)


@pytest.mark.parametrize("family", ["attention", "groupnorm", "layernorm", "lora_up", "lora_wgrad", "conv_patch", "conv_patch_splitk"])
def test_kernels_are_bit_stable_run_to_run(family):
    """Two identical launches must give identical bits at the level-0 shapes of the step (every kernel here is atomics-free by
    design): the guard that caught a sporadically wrong GEMM epilogue variant in round 3, extended to the other kernel families."""
    import math
    g = torch.Generator().manual_seed(71)
    f32 = torch.float32
    big = DEV == "cuda"                       # (the CPU harness in tests/test_kernels_emu.py drives the same body at small sizes)
    if family == "attention":
        B, H, N, D = (2, 8, 4096, 40) if big else (1, 2, 200, 40)
        qkv, dO = KC.rnd((B * N, 3 * H * D), DEV, g), KC.rnd((B * N, H * D), DEV, g)
        q, k, v = qkv[:, :H * D], qkv[:, H * D:2 * H * D], qkv[:, 2 * H * D:]

        def run():
            o, lse = K.attn_fwd(q, k, v, B, H, N, N, D, D ** -0.5)
            d = torch.empty_like(qkv)
            K.attn_bwd(q, k, v, o, dO, lse, B, H, N, N, D, D ** -0.5, d[:, :H * D], d[:, H * D:2 * H * D], d[:, 2 * H * D:])
            return o, lse, d
    elif family == "groupnorm":
        x, dy, dres = (KC.rnd((4, 4096, 320) if big else (2, 60, 320), DEV, g) for _ in range(3))
        gamma, beta = KC.rnd((320,), DEV, g, dtype=f32), KC.rnd((320,), DEV, g, dtype=f32)

        def run():
            y, st = K.groupnorm_fwd(x, gamma, beta, 32, 1e-5, True)
            dx, _, _ = K.groupnorm_bwd(x, dy, gamma, beta, st, 32, True, dres=dres)
            return y, st, dx
    elif family == "layernorm":
        x, dy, dres = (KC.rnd((16384 if big else 70, 320), DEV, g) for _ in range(3))
        gamma, beta = KC.rnd((320,), DEV, g, dtype=f32), KC.rnd((320,), DEV, g, dtype=f32)

        def run():
            return K.layernorm_fwd(x, gamma, beta, 1e-5), K.layernorm_bwd(x, dy, gamma, 1e-5, dres=dres)
    elif family == "lora_up":
        M, N = (16384 if big else 200), 320
        base, T, U = KC.rnd((M, N), DEV, g), KC.rnd((M, 4), DEV, g, dtype=f32), KC.rnd((N, 4), DEV, g, dtype=f32)

        def run():
            return (K.lora_up(base, T, 0, U, M, N, 0.7),)
    elif family == "lora_wgrad":
        M, N = (16384 if big else 300), 320
        A, T = KC.rnd((M, N), DEV, g), KC.rnd((M, 4), DEV, g, dtype=f32)

        def run():
            G = torch.zeros((N, 4), dtype=f32, device=DEV)
            K.lora_wgrad(A, T, 0, G, 4, 1, M, N, 4, scale=0.5)
            return (G,)
    else:
        from controllora_amd.ops import conv_k_order
        Bn, Hh, Ci, Co = (4, 64, 320, 320) if big else (1, 16, 128, 160)
        M = Bn * Hh * Hh
        x = KC.rnd((M, Ci), DEV, g)
        w = conv_k_order(KC.rnd((Co, 9, Ci), DEV, g, 1 / math.sqrt(9 * Ci)), 64)
        res = KC.rnd((M, Co), DEV, g)
        cd, _, _ = K.conv_fwd_desc(Hh, Hh, Ci, 3, 1, 1, kchunk=64)
        sk = 2 if family == "conv_patch_splitk" else 1

        def run():
            return (K.gemm(x, w, M, Co, 9 * Ci, conv=cd, residual=res, tile_cfg=76, split_k=sk, _tuned=False),)
    a, b = run(), run()
    torch.cuda.synchronize() if DEV == "cuda" else None
    for u, v_ in zip(a, b):
        assert torch.equal(u, v_), f"{family}: two identical launches differ in {int((u != v_).sum())} elements"
        assert bool(torch.isfinite(u.float()).all())


def test_every_plain_entry_of_the_tuned_table_is_exact_and_bit_stable():
    """The autotuned launch table (controllora_amd/gemm_tuning_gfx950.json) entry by entry, for the plain (non-conv) signatures of the
    train step and the batch-32 inference forward: each (shape, tile_cfg, split_k) with a bias + rank-4 adapter + residual epilogue
    against an fp32 matmul -- rel-L2, no outlier element, and two identical launches bit-identical (the guard that caught tile_cfg 42)."""
    import json
    import math
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(K.__file__)), "gemm_tuning_gfx950.json")
    table = json.load(open(path))["table"]
    g = torch.Generator().manual_seed(81)
    done = 0
    for key, (tile, sk) in sorted(table.items()):
        if ":" in key:
            continue
        M, N, Kd = (int(v) for v in key.split("x"))
        if M * N * Kd > 6e10 or M * N > 3.4e7 or N % 16 or M < 16:        # keep the whole sweep to seconds; the big ones share their kernels
            continue
        A, B = KC.rnd((M, Kd), DEV, g), KC.rnd((N, Kd), DEV, g, 1 / math.sqrt(Kd))
        bias, res = KC.rnd((N,), DEV, g, dtype=torch.float32), KC.rnd((M, N), DEV, g)
        T, U = KC.rnd((M, 4), DEV, g, dtype=torch.float32), KC.rnd((N, 4), DEV, g, 0.3, dtype=torch.float32)
        ref = ((A.float() @ B.float().T) + bias + T @ U.T).half().float() + res.float()
        kw = dict(bias=bias, residual=res, lora_t=T, lora_u=U, lora_seg=N, tile_cfg=tile, split_k=sk, _tuned=False)
        out = K.gemm(A, B, M, N, Kd, **kw)
        again = K.gemm(A, B, M, N, Kd, **kw)
        e = float((out.float() - ref).norm() / ref.norm())
        assert e < 8e-4, (key, tile, sk, e)
        assert torch.equal(out, again), (key, tile, sk, "two identical launches differ")
        lim = 6e-3 * max(1.0, float(ref.abs().max()))
        worst = float((out.float() - ref).abs().max())
        assert worst < lim, (key, tile, sk, worst, lim)
        done += 1
    assert done >= 60, done


@pytest.mark.parametrize("waves", [4, 8, 16])
@pytest.mark.parametrize("B,H,Nq,Nk,D", [(2, 8, 4096, 4096, 40), (1, 2, 600, 300, 40), (2, 1, 513, 64, 40), (1, 8, 1024, 1100, 64)])
def test_attention_forward_block_widths(waves, B, H, Nq, Nk, D):
    """clora_set_option("attn_fwd_waves"): 4 / 8 / 16 waves = 128 / 256 / 512 queries per forward block (the 16-wave block halves the
    K/V tile stream per flop once more: the batch-32 sampler's level-0 self-attention); same results, ragged query counts included"""
    from controllora_amd import kernels as K_
    K_.set_option("attn_fwd_waves", waves)
    try:
        KC.case_attention(DEV, B, H, Nq, Nk, D)
    finally:
        K_.set_option("attn_fwd_waves", 0)


def test_groupnorm_team_exchange_under_graph_replay_and_changing_geometry():
    """The team kernels' in-launch exchange (include/clora.h clora_groupnorm_*_team) with everything that could expose a stale word:
    launches of four geometries (8 / 16 / 32 units, 32 / 16 / 8 / 64 members) interleaved so that granule slots are re-used by units
    whose epochs differ, 40 launch pairs captured in ONE hipGraph and replayed (frozen arguments: the epoch must come from the state),
    consumers that have just read the same lines (L1-warm), a GEMM between the launches so that workgroups start unevenly.  Every
    replay must give the bits of the first eager pass, which in turn agree with the two-launch scheme; no exchange may give up."""
    g = torch.Generator().manual_seed(5)
    f32 = torch.float32
    shapes = [(4, 4096, 320), (4, 1024, 640), (8, 1024, 320), (2, 4096, 320), (4, 1024, 1280)]
    data = []
    for B, HW, C in shapes:
        x, dy = KC.rnd((B, HW, C), DEV, g), KC.rnd((B, HW, C), DEV, g)
        gamma, beta = 1 + 0.2 * KC.rnd((C,), DEV, g, dtype=f32), 0.2 * KC.rnd((C,), DEV, g, dtype=f32)
        data.append((x, dy, gamma, beta))
    A, W = KC.rnd((4096, 640), DEV, g), KC.rnd((640, 640), DEV, g)

    def one_pass():
        outs = []
        for rep in range(4):
            for x, dy, gamma, beta in data:
                y, st = K.groupnorm_fwd(x, gamma, beta, 32, 1e-5, True)
                K.gemm(A, W, 4096, 640, 640)
                dx, _, _ = K.groupnorm_bwd(x, dy, gamma, beta, st, 32, True)
                if rep == 3:
                    outs += [y, st, dx]
        return outs

    K.set_option("gn_team", 0)
    try:
        ref = one_pass()
    finally:
        K.set_option("gn_team", int(os.environ.get("CLORA_GN_TEAM", "2")))
    first = one_pass()
    for a, b in zip(first, ref):
        assert KC.rel(a, b) < 3e-4
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        captured = one_pass()
    for _ in range(5):
        graph.replay()
        torch.cuda.synchronize()
        for a, b in zip(captured, first):
            assert torch.equal(a, b)
    assert K.gn_team_errors(DEV) == 0


@pytest.mark.parametrize("M,N,K_,split,tile,lora", [(16384, 320, 320, 1, 0, False), (16384, 320, 320, 1, 55, True), (1024, 1280, 1280, 3, 0, False),
                                                   (4096, 640, 2560, 1, 0, False), (300, 96, 64, 1, 1, False)])
def test_gemm_compensated_trunk(M, N, K_, split, tile, lora):
    KC.case_gemm_trunk_lo(DEV, M, N, K_, split_k=split, tile_cfg=tile, lora=lora)
