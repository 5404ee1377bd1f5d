"""CPU tests of the kernel LOGIC: the same .hip sources compiled for the host fiber emulator
(tests/hipemu) against plain torch fp32.  Small shapes; the real-size GPU runs are in
tests/test_kernels_gpu.py (-m gpu)."""
import pytest

from tests import kernel_cases as KC
from tests.emu_fixture import use_emulator


@pytest.fixture(autouse=True)
def _emu():
    with use_emulator():
        yield


@pytest.mark.parametrize("M,N,K,split", [(300, 72, 96, 1), (130, 320, 64, 1), (520, 136, 160, 2), (256, 640, 320, 3),
                                         (64, 8, 40, 1)])
def test_gemm_plain(M, N, K, split):
    KC.case_gemm_plain("cpu", M, N, K, split)


@pytest.mark.parametrize("split", [1, 2])
def test_gemm_epilogue(split):
    KC.case_gemm_epilogue("cpu", split_k=split)


@pytest.mark.parametrize("kw", [dict(stride=1, pad=1), dict(stride=2, pad=1), dict(asym=True, stride=2, pad=0), dict(ups=True)])
def test_conv_fwd_dgrad_wgrad(kw):
    KC.case_conv("cpu", 2, 12, 10, 16, 24, **kw)


@pytest.mark.parametrize("Bn,H,W,Ci,Co", [(1, 3, 128, 32, 32), (2, 5, 64, 32, 64), (1, 2, 64, 64, 64), (1, 3, 64, 64, 48), (1, 2, 128, 32, 64)])
def test_conv_wgrad_patch_kernel(Bn, H, W, Ci, Co):
    KC.case_conv_wgrad_patch("cpu", Bn, H, W, Ci, Co)


@pytest.mark.parametrize("Bn,H,W,Ci,Co", [(1, 6, 128, 32, 32), (2, 4, 128, 32, 64), (1, 4, 128, 64, 64), (1, 2, 256, 32, 16)])
def test_conv_wgrad_patch_kernel_stride2(Bn, H, W, Ci, Co):
    """the downsamplers (reference models.py SimpleDownEncoderBlock2D: F.pad(0, 1, 0, 1) + 3x3 stride 2); H, W = INPUT size"""
    KC.case_conv_wgrad_patch("cpu", Bn, H, W, Ci, Co, stride=2)


@pytest.mark.parametrize("Bn,H,W,Ci,Co", [(1, 3, 128, 32, 32), (2, 2, 128, 32, 64), (1, 2, 128, 64, 32), (1, 5, 256, 32, 24), (2, 3, 128, 8, 32), (1, 2, 128, 8, 48)])
def test_conv_strip_kernel(Bn, H, W, Ci, Co):
    KC.case_conv_strip("cpu", Bn, H, W, Ci, Co)


def test_conv_small_channels():
    KC.case_conv("cpu", 1, 16, 16, 8, 32)       # hint-encoder conv_in shape class (3 -> padded 8 channels)


ALL_TILE_CFGS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 21, 22, 23, 26, 31, 32, 33, 41, 42, 43, 51, 52, 53, 54, 55, 56, 57, 58, 59]


@pytest.mark.parametrize("tile", ALL_TILE_CFGS)
def test_gemm_tile_configs(tile):
    """every main-loop variant (tile shape x ring depth), with ragged M/N/K, split-K and the fused epilogue"""
    KC.case_gemm_plain("cpu", 150, 72, 104, 1, tile_cfg=tile)
    KC.case_gemm_plain("cpu", 70, 136, 424, 3, tile_cfg=tile)
    KC.case_gemm_epilogue("cpu", split_k=1, tile_cfg=tile)
    KC.case_conv("cpu", 1, 8, 8, 16, 24, tile_cfg=tile)


@pytest.mark.parametrize("M,N,K,split", [(300, 520, 1096, 1), (513, 256, 640, 2), (256, 264, 64, 1), (40, 1288, 72, 1)])
def test_gemm_eight_phase_tile(M, N, K, split, zero_latency_dma_param):
    """tile_cfg 59 (gemm_8p_kernel, 256x256, two wave rows one barrier apart): several tiles, many K-tiles, a ragged K tail,
    split-K, under both ends of the emulator's DMA latency range"""
    KC.case_gemm_plain("cpu", M, N, K, split, tile_cfg=59)


@pytest.mark.parametrize("tile", [1, 3, 4, 5, 6, 7, 21, 23, 43, 51, 52, 53, 54, 55, 56, 57, 58, 59])
def test_gemm_epilogue_without_rowadd(tile):
    """projection epilogues (adapter / bias / residual, no row add): the two-phase chunk loop of the 8-wave tiles, ragged M and N"""
    KC.case_gemm_epilogue_no_rowadd("cpu", M=300, N=320, K_=128, tile_cfg=tile)
    KC.case_gemm_epilogue_no_rowadd("cpu", M=70, N=144, K_=64, tile_cfg=tile)


@pytest.mark.parametrize("kw", [dict(M=150, N=320, K_=128, tile_cfg=55), dict(M=150, N=320, K_=128, tile_cfg=52, t_in_rows=0),
                                dict(M=140, N=640, K_=64, nseg=2, tile_cfg=55, t_in_rows=70), dict(M=200, N=960, K_=192, nseg=3, tile_cfg=54, t_in_rows=0),
                                dict(M=130, N=320, K_=320, tile_cfg=51, u_tr=True, bias=False), dict(M=70, N=640, K_=64, nseg=1, tile_cfg=0, residual=False),
                                dict(M=100, N=320, K_=128, tile_cfg=21), dict(M=150, N=128, K_=128, nseg=2, tile_cfg=43, t_in_rows=0),
                                dict(M=200, N=192, K_=64, nseg=1, tile_cfg=23, u_tr=True, bias=False), dict(M=140, N=256, K_=192, nseg=2, tile_cfg=42, t_in_rows=70),
                                dict(M=130, N=128, K_=64, nseg=2, tile_cfg=22), dict(M=260, N=256, K_=128, nseg=2, tile_cfg=21, t_in_rows=0),
                                dict(M=129, N=128, K_=64, nseg=1, tile_cfg=41, residual=False), dict(M=100, N=64, K_=64, tile_cfg=26),
                                dict(M=100, N=192, K_=64, nseg=3, tile_cfg=0)])
def test_gemm_with_adapter_down_projection_in_the_launch(kw):
    """clora_epilogue_t.lora_dpack on the 8-wave 320-column tiles (an incapable tile_cfg is replaced by the library's choice)"""
    KC.case_gemm_fused_down("cpu", **kw)


@pytest.mark.parametrize("kw", [dict(), dict(Mc=70, Cc=40, C_=64, rc=8, n=2, strided_grad=False), dict(Mc=513, Cc=32, C_=320, rc=4, n=10)])
def test_control_terms_in_rank_space(kw):
    print(KC.case_rank_control("cpu", **kw))


def test_tile_order_grid_random_shapes():
    """seeded sweep of tile shape x grid size x split-K (whole and ragged edges): the rectangle assignment of clora_set_option
    "tile_order" = 3 covers every (split, tile) exactly once -- outputs bit-identical to the m-major order and equal to fp32 matmul"""
    import math
    import random
    import torch
    from controllora_amd import kernels as K
    rng = random.Random(3)
    dims = {43: (64, 64), 21: (128, 128), 22: (128, 64), 53: (128, 256), 58: (256, 256), 54: (128, 320)}
    n = 0
    try:
        while n < 24:
            tile = rng.choice(sorted(dims))
            tm, tn, sk = rng.choice([2, 4, 8, 16]), rng.choice([1, 2, 4, 8]), rng.choice([1, 2, 4])
            M, N = tm * dims[tile][0] - rng.choice([0, 0, 8]), tn * dims[tile][1] - rng.choice([0, 0, 16])
            Kd = 64 * sk * rng.choice([1, 2])
            if M * N > 700000:
                continue
            n += 1
            g = torch.Generator().manual_seed(n)
            A = torch.randn(M, Kd, generator=g).half()
            B = (torch.randn(N, Kd, generator=g) / math.sqrt(Kd)).half()
            K.set_tile_order("m")
            base = K.gemm(A, B, M, N, Kd, split_k=sk, tile_cfg=tile)
            K.set_tile_order("grid")
            other = K.gemm(A, B, M, N, Kd, split_k=sk, tile_cfg=tile)
            assert torch.equal(base, other), (tile, M, N, Kd, sk)
            assert float((base.float() - A.float() @ B.float().T).abs().max()) < 0.05
    finally:
        K.set_tile_order(K.DEFAULT_TILE_ORDER)


@pytest.mark.parametrize("order", ["n", "auto", "grid"])
@pytest.mark.parametrize("tile", [21, 43, 53, 72, 76])
def test_tile_order_does_not_change_results(tile, order):
    """the tile -> XCD assignment (clora_set_option "tile_order") only permutes which workgroup computes which tile: bit-identical outputs"""
    KC.case_tile_order("cpu", tile, order)


@pytest.mark.parametrize("tile", [71, 72, 73, 74, 75, 76, 79])
@pytest.mark.parametrize("Bn,H,W,Ci,Co", [(2, 8, 8, 128, 64), (1, 16, 16, 64, 72), (3, 4, 4, 64, 64)])
def test_conv_patch_kernel(tile, Bn, H, W, Ci, Co):
    """patch-staged 3x3 conv: whole images per tile (8x8, 4x4: tiles straddle the batch, rows past M), whole rows (16x16)"""
    if tile in (71, 74, 79) and W == 4:
        pytest.skip("16 images of 4x4 exceed the 256-pixel tile's patch budget (falls back by design)")
    KC.case_conv_patch("cpu", Bn, H, W, Ci, Co, tile)


@pytest.mark.parametrize("tile", [71, 72, 76, 79])
def test_conv_patch_kernel_upsampled(tile):
    KC.case_conv_patch_upsampled("cpu", 2, 8, 8, 128, 72, tile)
    KC.case_conv_patch_upsampled("cpu", 1, 4, 4, 64, 64, tile)


@pytest.mark.parametrize("tile", [77, 78])
@pytest.mark.parametrize("Bn,H,W,Ci,Co", [(1, 3, 128, 64, 64), (1, 2, 256, 64, 128), (2, 1, 128, 128, 64)])
def test_conv_patch_kernel_row_segments(tile, Bn, H, W, Ci, Co):
    """tile_cfg 77 / 78 (392-pixel patch): one 128-pixel image row per tile (W = 128) or 128-pixel SEGMENTS of a wider row (W = 256):
    the VAE's levels; forward, dgrad, split-K, epilogue"""
    KC.case_conv_patch("cpu", Bn, H, W, Ci, Co, tile)
    if tile == 77:
        KC.case_conv_patch_upsampled("cpu", 1, 2, 64, 64, 64, tile)        # output rows of 128 pixels from a 64-wide source


def test_conv_patch_kernel_wide_rows():
    KC.case_conv_patch("cpu", 1, 32, 32, 64, 64, 71)
    KC.case_conv_patch("cpu", 1, 64, 64, 64, 64, 73)


@pytest.mark.parametrize("B,H,Nq,Nk,D,fused", [(1, 2, 70, 70, 40, True), (2, 2, 64, 77, 40, False), (1, 2, 33, 130, 80, False),
                                               (1, 1, 40, 40, 160, True), (1, 2, 20, 20, 8, False), (1, 1, 150, 77, 64, False),
                                               (1, 2, 640, 77, 40, False)])
def test_attention(B, H, Nq, Nk, D, fused):
    KC.case_attention("cpu", B, H, Nq, Nk, D, fused_qkv=fused)


@pytest.fixture(params=[0, 1], ids=["late_dma", "eager_dma"])
def zero_latency_dma_param(request):
    from tests.emu_fixture import emu_lib
    emu_lib().cdll.hipemu_set_dma_eager(request.param)
    yield
    emu_lib().cdll.hipemu_set_dma_eager(0)


@pytest.fixture
def zero_latency_dma():
    """LDS-DMA copies land at issue instead of at the covering wait (tests/hipemu/hipemu.cpp): the other end of the latency
    range.  A copy issued into a ring stage / double-buffer half that a slower wave still reads corrupts that wave's operands."""
    from tests.emu_fixture import emu_lib
    emu_lib().cdll.hipemu_set_dma_eager(1)
    yield
    emu_lib().cdll.hipemu_set_dma_eager(0)


@pytest.mark.parametrize("tile", [21, 33, 43, 53, 57, 59])
def test_gemm_rings_zero_latency_dma(tile, zero_latency_dma):
    KC.case_gemm_plain("cpu", 70, 136, 424, 3, tile_cfg=tile)
    KC.case_gemm_epilogue("cpu", split_k=1, tile_cfg=tile)


@pytest.mark.parametrize("tile", [71, 72, 75, 76])
def test_conv_patch_zero_latency_dma(tile, zero_latency_dma):
    KC.case_conv_patch("cpu", 2, 8, 8, 128, 64, tile)
    KC.case_conv_patch_upsampled("cpu", 1, 4, 4, 64, 64, tile) if tile in (71, 72, 76) else None


def test_conv_wgrad_patch_zero_latency_dma(zero_latency_dma):
    """the double-buffered row loop of conv_wgrad_patch_kernel when every copy lands at issue (a row staged into a buffer that a slower
    wave still reads would corrupt that wave's operands)"""
    KC.case_conv_wgrad_patch("cpu", 1, 4, 64, 32, 64)


@pytest.mark.parametrize("B,H,Nq,Nk,D", [(1, 2, 70, 300, 40), (1, 2, 640, 77, 40), (1, 1, 150, 200, 64)])
def test_attention_zero_latency_dma(B, H, Nq, Nk, D, zero_latency_dma):
    KC.case_attention("cpu", B, H, Nq, Nk, D)


@pytest.mark.parametrize("B,H,Nq,Nk,D", [(2, 3, 300, 300, 40), (1, 5, 200, 77, 80), (3, 1, 130, 130, 160)])
def test_attention_block_order_does_not_change_results(B, H, Nq, Nk, D):
    KC.case_attention_block_order("cpu", B, H, Nq, Nk, D)


@pytest.mark.parametrize("Nk,D", [(40, 40), (64, 64), (200, 40), (130, 64), (100, 80)])
def test_attention_first_tile_far_below_zero(Nk, D):
    """first KV tile with every logit below -88 (single-tile and multi-tile, ones-column and summed row sums)"""
    KC.case_attention_negative_logits("cpu", 1, 2, 40, Nk, D)


@pytest.mark.parametrize("D", [40, 80])
def test_attention_rising_maxima(D):
    """forward: lazy exponent reference, rebased on later KV tiles for a subset of the queries (D = 40: rowsum from the ones column)"""
    KC.case_attention("cpu", 1, 2, 70, 300, D, ramp=4.0, tol=3e-3)


@pytest.mark.parametrize("B,HW,C,G,silu,train", [(2, 50, 320, 32, True, False), (1, 37, 64, 8, False, False),
                                                 (2, 64, 32, 32, True, True), (1, 9, 2560, 32, True, False),
                                                 (2, 300, 64, 8, True, False), (2, 256, 320, 32, False, False),
                                                 (1, 40, 2056, 8, True, True),      # one 2056-channel slab: two column chunks per thread
                                                 (3, 70, 1280, 32, True, False),    # row chunks that end inside a load batch
                                                 # one-launch (register-resident) plan: 5 / 10 / 15 chunk slabs, 256 and 512 threads, row counts
                                                 # below / not a multiple of the row lanes, a slab of four 10-channel groups, 4-channel groups
                                                 (2, 64, 1280, 32, True, False), (1, 256, 2560, 32, True, False), (2, 130, 1920, 32, False, False),
                                                 (1, 600, 640, 32, True, False), (2, 20, 320, 32, True, False), (1, 97, 128, 32, True, False),
                                                 # team plan (HW >= 1024, B * slabs dividing 256): 2 / 4 / 8 slabs, ragged member row ranges
                                                 (1, 1024, 320, 32, True, False), (2, 1030, 64, 8, False, False), (1, 1100, 640, 32, True, False)])
def test_groupnorm(B, HW, C, G, silu, train):
    KC.case_groupnorm("cpu", B, HW, C, G, silu, train_params=train)


def test_groupnorm_random_shapes():
    """seeded sweep over group sizes 1..80 channels, 1..513 rows, 1..32 groups: whichever plan a shape gets (one launch, two launches,
    channel slabs, two column chunks per thread) agrees with torch, with the other plan and with itself (case_groupnorm)"""
    import random
    rng = random.Random(7)
    n = 0
    while n < 48:
        G = rng.choice([1, 2, 4, 8, 16, 32, 32, 32])
        C = G * rng.choice([1, 2, 3, 4, 5, 6, 8, 10, 12, 16, 20, 40, 60, 80])
        HW, B = rng.choice([1, 2, 7, 16, 37, 64, 100, 130, 256, 300, 513]), rng.choice([1, 2, 3])
        if C % 8 or C > 2560 or B * HW * C > 600000:
            continue
        n += 1
        KC.case_groupnorm("cpu", B, HW, C, G, rng.random() < 0.5, seed=n)


@pytest.mark.parametrize("B,HW,C,G", [(1, 64, 1280, 4), (2, 16, 2048, 1), (1, 37, 512, 4), (1, 16, 1024, 8), (2, 9, 2560, 8), (1, 64, 640, 5)])
def test_groupnorm_wide_groups(B, HW, C, G):
    """groups wider than a block has threads (cpg 128 .. 2048; ADVICE r04: the one-launch plan reduced only the first NT / seg
    channels of a slab and read the rest of chs[] uninitialised -- C = 1280, G = 4, HW = 64 gave rel err 0.12): such shapes must
    not take the one-launch plan, and whatever plan they take agrees with torch and with the other plan (case_groupnorm)"""
    KC.case_groupnorm("cpu", B, HW, C, G, True, seed=B + HW)


@pytest.mark.parametrize("B,HW,Ca,Cb,G,silu", [(2, 64, 320, 320, 32, True), (1, 256, 640, 320, 32, True), (2, 16, 1280, 1280, 32, True),
                                               (1, 300, 64, 32, 8, False), (2, 9, 1280, 640, 32, True), (1, 1024, 320, 640, 32, True),
                                               (3, 37, 8, 56, 8, True)])
def test_groupnorm_concat_in_place(B, HW, Ca, Cb, G, silu):
    KC.case_groupnorm_concat("cpu", B, HW, Ca, Cb, G, silu)


@pytest.mark.parametrize("B,HW,C,Kd,G,split,lora", [(2, 16, 128, 256, 8, 2, False), (2, 64, 320, 320, 32, 3, True), (1, 256, 640, 128, 32, 2, False),
                                                    (2, 300, 64, 96, 8, 2, True),       # two-launch GroupNorm plan: finished first
                                                    (1, 16, 1280, 256, 32, 4, True), (2, 64, 64, 128, 8, 1, False)])
def test_deferred_split_k_finish_in_the_norm_kernels(B, HW, C, Kd, G, split, lora):
    KC.case_deferred_finish("cpu", B, HW, C, Kd, G, split, lora=lora)


@pytest.mark.parametrize("M,Kd,tile,lora,res", [(150, 320, 55, False, True), (200, 128, 52, True, True), (130, 320, 54, True, False),
                                                (300, 64, 51, False, True), (64, 1280, 55, True, True)])
def test_gemm_fused_layernorm(M, Kd, tile, lora, res):
    KC.case_gemm_fused_layernorm("cpu", M, Kd, tile, lora=lora, residual=res)


@pytest.mark.parametrize("M,C", [(37, 320), (9, 1280), (5, 64)])
def test_layernorm(M, C):
    KC.case_layernorm("cpu", M, C)


@pytest.mark.parametrize("M,C", [(2051, 320), (2050, 640), (2049, 1280), (309, 320), (77, 1280), (130, 768)])
def test_layernorm_rows_in_flight(M, C):
    KC.case_layernorm_rows("cpu", M, C)


def test_geglu():
    KC.case_geglu("cpu", 33, 256)


@pytest.mark.parametrize("M,K,N,R,xr", [(300, 320, 64, 4, 0), (128, 72, 40, 8, 32), (70, 96, 32, 20, 0)])
def test_lora(M, K, N, R, xr):
    KC.case_lora("cpu", M, K, N, R, x_rows=xr)


def test_elementwise():
    KC.case_elementwise("cpu")


def test_loss_and_optimizer():
    KC.case_loss_and_optimizer("cpu")


@pytest.mark.parametrize("rows,cols", [(5, 64), (9, 1032)])
def test_softmax_rows(rows, cols):
    KC.case_softmax_rows("cpu", rows, cols)


def test_conv_padded_channels_pack_and_oihw_grad():
    KC.case_conv_padded_channels("cpu")


@pytest.mark.parametrize("tile", [1, 3, 5, 7, 9, 21, 23, 26, 42, 33, 51, 52, 56])
def test_conv_fast_path_uniform_taps(tile):
    """3x3 stride-1 convs with Cin % 32 == 0 take the wave-uniform tap walk (CONV == 2) in forward and dgrad"""
    KC.case_conv("cpu", 1, 6, 5, 32, 64, tile_cfg=tile)
    KC.case_conv("cpu", 2, 4, 4, 64, 32, tile_cfg=tile)
    for kw in (dict(stride=2, pad=1), dict(asym=True, stride=2, pad=0), dict(ups=True)):     # generic gather at Cin % 32 == 0
        KC.case_conv("cpu", 1, 6, 6, 32, 32, tile_cfg=tile, **kw)


@pytest.mark.parametrize("tile", [0, 1, 7, 21, 41, 53, 56])
def test_feed_forward_fused_geglu(tile):
    KC.case_feed_forward_fused("cpu", M=150, C=32, tile_cfg=tile)
    KC.case_feed_forward_fused("cpu", M=70, C=64, tile_cfg=tile)


@pytest.mark.parametrize("tile", [0, 1, 3, 7, 21, 23, 42, 12])
def test_conv_channel_chunk_major_k_order(tile):
    """clora_conv_t.kchunk: slabs of kchunk channels with their 9 taps back to back -- fast tap walk (kchunk % BK == 0),
    generic gather (strided / upsampled / kchunk < BK), split-K starting inside a slab, forward and dgrad"""
    KC.case_conv("cpu", 1, 6, 5, 64, 64, tile_cfg=tile, kchunk=64)
    KC.case_conv("cpu", 2, 4, 4, 128, 64, tile_cfg=tile, kchunk=64)
    KC.case_conv("cpu", 1, 6, 6, 64, 32, tile_cfg=tile, kchunk=32)
    KC.case_conv("cpu", 1, 5, 5, 32, 32, tile_cfg=tile, kchunk=16)
    for kw in (dict(stride=2, pad=1), dict(asym=True, stride=2, pad=0), dict(ups=True)):
        KC.case_conv("cpu", 1, 6, 6, 64, 64, tile_cfg=tile, kchunk=64, **kw)


@pytest.mark.parametrize("family", ["attention", "groupnorm", "layernorm", "lora_up", "lora_wgrad", "conv_patch", "conv_patch_splitk"])
def test_kernels_are_bit_stable_run_to_run(family):
    """the body of the GPU test of the same name at small sizes (host logic of the check; the emulator is deterministic by construction)"""
    import tests.test_kernels_gpu as TG
    old = TG.DEV
    TG.DEV = "cpu"
    try:
        TG.test_kernels_are_bit_stable_run_to_run(family)
    finally:
        TG.DEV = old


@pytest.mark.parametrize("waves", [4, 8, 16])
@pytest.mark.parametrize("B,H,Nq,Nk,D", [(1, 2, 600, 300, 40), (2, 1, 513, 64, 40), (1, 1, 100, 1100, 64)])
def test_attention_forward_block_widths(waves, B, H, Nq, Nk, D):
    """clora_set_option("attn_fwd_waves"): 4 / 8 / 16 waves = 128 / 256 / 512 queries per forward block (the 16-wave block halves the
    K/V tile stream per flop once more: the batch-32 sampler's level-0 self-attention); same results, ragged query counts included"""
    from controllora_amd import kernels as K_
    K_.set_option("attn_fwd_waves", waves)
    try:
        KC.case_attention("cpu", B, H, Nq, Nk, D)
    finally:
        K_.set_option("attn_fwd_waves", 0)


def test_gemm_eight_phase_random_shapes(zero_latency_dma_param):
    """seeded sweep of tile_cfg 59 over ragged M / N / K (K tails that are not whole 64-deep K-tiles, fewer K-tiles than the seven
    half-tiles the prologue runs ahead, several tiles in both directions, split-K): every shape against an fp32 matmul, under the
    late- and the eager-landing DMA model"""
    import random
    rng = random.Random(59)
    for _ in range(10):
        M = rng.choice([1, 17, 64, 255, 256, 257, 300, 520])
        N = rng.choice([8, 24, 72, 256, 264, 520])
        Kd = rng.choice([8, 40, 64, 72, 128, 136, 320, 456, 1032])
        split = rng.choice([1, 1, 2, 3]) if Kd >= 256 else 1
        KC.case_gemm_plain("cpu", M, N, Kd, split, seed=M + N + Kd, tile_cfg=59)


@pytest.mark.parametrize("M,N,K_,split,tile,lora", [(150, 320, 128, 1, 0, False), (150, 320, 128, 1, 55, True), (200, 96, 64, 2, 0, False),
                                                   (130, 64, 320, 1, 43, False), (100, 128, 64, 1, 1, False)])
def test_gemm_compensated_trunk(M, N, K_, split, tile, lora):
    KC.case_gemm_trunk_lo("cpu", M, N, K_, split_k=split, tile_cfg=tile, lora=lora)
