// clora_gemm.hip -- fp16 MFMA GEMM / implicit-GEMM convolution for gfx950.
//
//   C[M,N] = epilogue( A[M,K] . B[N,K]^T )
//
// One kernel family serves every dense contraction of the path: Linear
// and 1x1 conv (plain A), 3x3 conv forward / strided / nearest-upsampled and the dgrad of each
// (A rows gathered on the fly from the NHWC activation, include/clora.h clora_conv_t), with the
// frozen weight always presented K-contiguous as B[N,K] (packed once at load time; the transposed /
// tap-flipped dgrad copy is a second packed tensor -- HBM is 288 GB, the UNet is 1.7 GB).
//
// Tiling (gemm_dma_kernel, the product path): 256 threads = 4 waves, BMxBN output tile (256x128 ... 64x64), ring stages
// of BK = 32 or 64 (v_mfma_f32_16x16x32_f16 k-steps) filled by LDS-DMA (`global_load_lds_dwordx4`, no staging VGPRs)
// into an NST-deep ring with counted `s_waitcnt vmcnt` and ONE raw barrier per stage, source-side XOR swizzle
// (conflict-free ds_read_b128 fragments from unpadded rows), XCD-aware tile order, accumulators staged through LDS in
// the epilogue so C / residual traffic is 16-byte coalesced.  split-K (grid.y) writes fp32 slabs that a second kernel
// reduces + finishes.  gemm_kernel below is the round-1 register-staged loop, kept for A/B runs (tile_cfg 11..13).
#include <string.h>
#include "clora_common.h"
#include "../../include/clora.h"
#include "clora_epilogue.h"

namespace {

struct GemmArgs {
    const half_t* A;
    const half_t* B;
    half_t* C;
    float* partial;
    int lda, ldc, M, N, K;
    int k_per_split;
    int tiles_n;
    int two_phase;             // option "epi_two_phase" at launch time (A/B switch of the two-phase chunk loop)
    int hoist_on;              // option "epi_hoist" at launch time (0: the one-chunk-at-a-time epilogue on the 8-wave tiles too)
    int tiles_m, n_major;      // n_major: logical tile t = n * tiles_m + m (else m * tiles_n + n); see pick_tile_order
    int xg_s, xg_m, xg_n;      // > 0: XCD x owns the (split, m, n) RECTANGLE x -> (x / (xg_m * xg_n), (x / xg_n) % xg_m, x % xg_n) of an xg_s x xg_m x xg_n grid
    clora_conv_t conv;
    clora_epilogue_t epi;
};

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    constexpr int BK = 32, LD = BK + 8;
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
    constexpr int A_CH = BM / 64, B_CH = BN / 64;
    constexpr int STAGE = (BM + BN) * LD;
    constexpr int C_LD = BN + 8;
    constexpr int SMEM = (2 * STAGE > BM * C_LD) ? 2 * STAGE : BM * C_LD;
    __shared__ __attribute__((aligned(16))) half_t smem[SMEM];

    const int t = threadIdx.x;
    const int tile_m = blockIdx.x / p.tiles_n, tile_n = blockIdx.x % p.tiles_n;
    const int split = blockIdx.y;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * p.k_per_split;
    const int kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
    const int nk = (kend - kbeg + BK - 1) / BK;

    // ---- loader state: thread owns 16-byte chunk column kc of rows r0 + 64*i
    const int kc = t & 3, r0 = t >> 2;
    const bool conv = p.conv.enabled != 0;
    bool a_ok[A_CH];
    size_t a_base[A_CH];
    int a_ty[A_CH], a_tx[A_CH];
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
        const int m = m0 + r0 + 64 * i;
        a_ok[i] = m < p.M;
        a_base[i] = 0; a_ty[i] = 0; a_tx[i] = 0;
        if (a_ok[i]) {
            if (!conv) {
                a_base[i] = (size_t)m * p.lda;
            } else {
                const int hw = p.conv.Hout * p.conv.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int yo = rem / p.conv.Wout, xo = rem - yo * p.conv.Wout;
                a_base[i] = (size_t)b * p.conv.Hin * p.conv.Win;
                a_ty[i] = yo * p.conv.mul + p.conv.off;
                a_tx[i] = xo * p.conv.mul + p.conv.off;
            }
        }
    }
    bool b_ok[B_CH];
    size_t b_base[B_CH];
#pragma unroll
    for (int i = 0; i < B_CH; ++i) {
        const int n = n0 + r0 + 64 * i;
        b_ok[i] = n < p.N;
        b_base[i] = (size_t)n * p.K;
    }
    int k = kbeg + kc * 8;  // this thread's k for the tile being loaded
    // K order (clora_conv_t.kchunk): slabs of kcs channels, the taps of a slab back to back (kcs = Cin: plain tap-major)
    const int kcs = p.conv.kchunk > 0 ? p.conv.kchunk : p.conv.Cin, ntap = p.conv.ksize * p.conv.ksize;
    int tap = 0, ci = k, cb = 0;       // tap, channel inside the slab, first channel of the slab
    if (conv) { cb = (k / (ntap * kcs)) * kcs; const int r = k % (ntap * kcs); tap = r / kcs; ci = r - tap * kcs; }

    half8 ra[A_CH], rb[B_CH];
    auto load_tile = [&]() {
        const bool kok = k < kend;
        int ky = 0, kx = 0;
        if (conv) { ky = tap / p.conv.ksize; kx = tap - ky * p.conv.ksize; }
#pragma unroll
        for (int i = 0; i < A_CH; ++i) {
            half8 v = zero8();
            if (a_ok[i] && kok) {
                if (!conv) {
                    v = ld8(p.A + a_base[i] + k);
                } else {
                    const int ty = a_ty[i] + ky * p.conv.kmul, tx = a_tx[i] + kx * p.conv.kmul;
                    bool ok = ty >= 0 && ty < p.conv.lim_h && tx >= 0 && tx < p.conv.lim_w;
                    if (p.conv.need_even) ok = ok && (((ty | tx) & 1) == 0);
                    if (ok)
                        v = ld8(p.A + (a_base[i] + (size_t)(ty >> p.conv.shift) * p.conv.Win + (tx >> p.conv.shift)) * p.conv.Cin + cb + ci);
                }
            }
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < B_CH; ++i) {
            half8 v = zero8();
            if (b_ok[i] && kok) v = ld8(p.B + b_base[i] + k);
            rb[i] = v;
        }
        k += BK;
        if (conv) {
            ci += BK;
            while (ci >= kcs) { ci -= kcs; if (++tap == ntap) { tap = 0; cb += kcs; } }
        }
    };
    auto store_tile = [&](int buf) {
        half_t* As = smem + buf * STAGE;
        half_t* Bs = As + BM * LD;
#pragma unroll
        for (int i = 0; i < A_CH; ++i) st8(As + (r0 + 64 * i) * LD + kc * 8, ra[i]);
#pragma unroll
        for (int i = 0; i < B_CH; ++i) st8(Bs + (r0 + 64 * i) * LD + kc * 8, rb[i]);
    };

    const int w = t >> 6, l = t & 63, g = l >> 4, li = l & 15;
    const int wm = w / WN, wn = w % WN;
    floatx4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = zero4f();

    if (nk > 0) {
        load_tile();
        store_tile(0);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile();
        const half_t* As = smem + buf * STAGE;
        const half_t* Bs = As + BM * LD;
        half8 af[FM], bf[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) af[i] = ld8(As + (wm * FM * 16 + i * 16 + li) * LD + g * 8);
#pragma unroll
        for (int j = 0; j < FN; ++j) bf[j] = ld8(Bs + (wn * FN * 16 + j * 16 + li) * LD + g * 8);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- split-K: raw fp32 slab, finished by splitk_finish_kernel
    if (p.partial) {
        float* slab = p.partial + (size_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * FM * 16 + i * 16 + 4 * g + r;
                    const int n = n0 + wn * FN * 16 + j * 16 + li;
                    if (m < p.M && n < p.N) slab[(size_t)m * p.N + n] = acc[i][j][r];
                }
        return;
    }

    // ---- fused epilogue: acc -> (+bias, +rowadd, +LoRA) -> fp16 -> LDS -> (+residual) -> 16-B stores
    half_t* Cs = smem;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ml = wm * FM * 16 + i * 16 + 4 * g + r;
                const int nl = wn * FN * 16 + j * 16 + li;
                const int m = m0 + ml, n = n0 + nl;
                float v = acc[i][j][r];
                if (m < p.M && n < p.N) v = epi_pre(v, m, n, p.epi);
                Cs[ml * C_LD + nl] = (half_t)v;
            }
    __syncthreads();
    constexpr int CPR = BN / 8;  // chunks per row
    for (int c = t; c < BM * CPR; c += 256) {
        const int ml = c / CPR, nc = c - ml * CPR;
        const int m = m0 + ml, n = n0 + nc * 8;
        if (m < p.M && n < p.N) {
            half8 v = ld8(Cs + ml * C_LD + nc * 8);
            if (p.epi.residual_lo || p.epi.c_lo) {            // compensated trunk (this round-1 kernel rounds the branch first)
                half8 rr = zero8(), rl = zero8(), lo;
                if (p.epi.residual) rr = ld8((const half_t*)p.epi.residual + (size_t)m * p.epi.ldr + n);
                if (p.epi.residual_lo) rl = ld8((const half_t*)p.epi.residual_lo + (size_t)m * p.epi.ldr + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float sum = (float)v[e] + (float)rr[e] + (float)rl[e];
                    v[e] = (half_t)sum;
                    lo[e] = (half_t)(sum - (float)v[e]);
                }
                if (p.epi.c_lo) st8((half_t*)p.epi.c_lo + (size_t)m * p.ldc + n, lo);
            } else if (p.epi.residual) {
                const half8 rr = ld8((const half_t*)p.epi.residual + (size_t)m * p.epi.ldr + n);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rr[e]);
            }
            st8(p.C + (size_t)m * p.ldc + n, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// v2 main loop: operands go global -> LDS by LDS-DMA (no staging VGPRs, no ds_write pass) into a 3-stage ring;
// the copy of tile kt+2 is issued right after the barrier of iteration kt and only has to land two
// iterations later (counted s_waitcnt vmcnt, raw s_barrier: one barrier per k-step, loads stay in flight
// across it).  LDS rows are 64 B (one BK=32 row) unpadded -- the DMA writes lane-linearly -- and bank
// conflicts are removed by an XOR swizzle applied on the SOURCE side: the lane that fills 16-byte slot `pos`
// of row r fetches logical k-chunk  pos ^ ((r >> 2) & 3); fragment reads apply the same involution.
// Out-of-range / padding chunks are fetched from a 16-byte zero page.
__device__ __attribute__((aligned(16))) const unsigned g_clora_zero16[4] = {0u, 0u, 0u, 0u};

// Epilogue shared by the LDS-DMA main loops (gemm_dma_kernel, conv3x3_patch_kernel): split-K slab, or fp32 accumulators ->
// LDS (64 rows per pass) -> per row-chunk: bias / time-embedding / rank-r adapter update (float4 operand loads) -> fp16 ->
// + residual (or the fused GEGLU forms) -> 16-byte coalesced stores.  NT threads = WM x WN waves, wave tile FM x FN MFMA tiles.
// The rank-4 update of one 8-column chunk from registers: v[e] += sum_j t4[j] * u(e, j), with the up-matrix values of the thread's
// column in `ureg` (TR: ureg[2j + (e >> 2)][e & 3], else ureg[e][j]).
// Every multiply-add is ONE explicit v_fma_f32 (CLORA_FMA_F32).  Written as plain C++ the block compiles to runs of dependent
// v_pk_fma_f32 with op_sel broadcasts of t4[j], and THAT form is what produced the sporadic wrong elements of round 3 on MI355X
// (profiles/r04_hoist_diag*.txt): only with two or three workgroups resident per CU, only in lanes 48..63,
// only in the LOW half of a packed pair (chunk elements 0 / 2 / 4), with every s_waitcnt of the ISA in place; the same kernel is
// clean with -mllvm -amdgpu-waitcnt-forcezero, with one workgroup per CU (larger LDS request, same instructions), with
// -fno-slp-vectorize, and with nothing changed but this block forced to scalar FMAs.  -DCLORA_HOIST_PACKED_FMA restores the
// packed form for diagnosis.
template <bool TR>
__device__ __forceinline__ void hoisted_rank4(float (&v)[8], const floatx4& t4, const floatx4 (&ureg)[8]) {
#ifndef CLORA_HOIST_PACKED_FMA
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float u = TR ? ureg[2 * j + (e >> 2)][e & 3] : ureg[e][j], tj = t4[j];
            CLORA_FMA_F32(v[e], tj, u);
        }
#else
    if (TR) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += t4[j] * ureg[2 * j + (e >> 2)][e & 3];
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += t4[0] * ureg[e][0] + t4[1] * ureg[e][1] + t4[2] * ureg[e][2] + t4[3] * ureg[e][3];
    }
#endif
}

// EXT > 0 (gemm_dma_kernel<..., EXT = 8>: the adapter down-projection rides in the main loop, clora_epilogue_t.lora_dpack): tacc
// holds this wave's share of the tile's extra columns (4 "hi" + 4 "lo" partial sums per row); they are staged in LDS
// next to the accumulators, every chunk takes its T row from there (+ the optional lora_t_in part), and the first tile of a
// column segment writes T to lora_t for the backward.
template <int BM, int BN, int WM, int WN, int NT, int SMEM, bool HOIST = false, int EXT = 0>
__device__ __forceinline__ void dma_epilogue(const GemmArgs& p, floatx4 (&acc)[BM / WM / 16][BN / WN / 16], int m0, int n0, int split,
                                             half_t* smem, int t, const floatx4* tacc = nullptr) {
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
    const int w = t >> 6, l = t & 63, g = l >> 4, li = l & 15;
    const int wm = w / WN, wn = w % WN;
    if (p.partial) {
        float* slab = p.partial + (size_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * FM * 16 + i * 16 + 4 * g + r;
                    const int n = n0 + wn * FN * 16 + j * 16 + li;
                    if (m < p.M && n < p.N) slab[(size_t)m * p.N + n] = acc[i][j][r];
                }
        return;
    }
    // ---- fp32 accumulators -> LDS (64 rows per pass) -> per row-chunk: bias / time-embedding / rank-r
    // adapter update (float4 operand loads) -> fp16 -> + residual -> 16-byte coalesced stores.  Doing the fused
    // math AFTER the LDS hop keeps it out of the main loop's register budget.
    constexpr int F_LD = BN + 4;
#ifdef CLORA_EPI_SINGLE_PASS
    // experiment build (tools/build_variant_lib.sh -DCLORA_EPI_SINGLE_PASS, A/B through CLORA_LIB_PATH): the whole tile is staged at once
    // where the ring allocation has room, so every accumulator is dead before the first chunk is processed.  Compiled figures in
    // DESIGN.md section 7 (peak VGPRs 255 -> 186 on 128x256, the 128x64 BK32 spills disappear); untimed, hence not the default.
    constexpr int PR = ((BM * F_LD + (EXT > 0 ? BM * 8 : 0)) * 2 <= SMEM) ? BM : 64;
#else
    constexpr int PR = 64;
#endif
    constexpr int NPASS = BM / PR;
    static_assert(PR * F_LD * 2 <= SMEM, "fp32 staging must fit in the LDS allocation");
    float* Cf = reinterpret_cast<float*>(smem);
    static_assert(EXT == 0 || (PR * F_LD + BM * 8) * 2 <= SMEM, "T staging must fit behind the accumulator staging");
    float* const Ts = Cf + PR * F_LD;                          // EXT: [BM][8] raw hi | lo sums, valid from pass 0 to the end
    constexpr int TM = EXT > 0 ? (BM / WM / 16 + WN - 1) / WN : 0;
    constexpr int CPR = BN / 8;
    constexpr int RPIT = NT / CPR;                             // rows one sweep of the block covers (threads past RPIT*CPR idle: BN = 160 / 320)
    static_assert(RPIT >= 1, "tile wider than the block");
    const int nc = t % CPR, n = n0 + nc * 8;
    // rank-4 adapter in the epilogue (every attention projection): the 32 up-matrix values and the bias of this thread's 8 columns
    // live in registers for the whole tile -- per output chunk only the 16-byte T row is fetched.  (Fetching U per chunk made the
    // epilogue's L1 traffic as long as a 5-step main loop: 8 x 16-byte loads per chunk.)
    // (Staging U in LDS for the other kernels instead was measured too: 24.24 -> 24.30 ms/step, not kept.)
    // HOIST: only the 8-wave tiles (one block per CU: registers to spare, and see the note at the call site about the variants with
    // more than one block per CU), and a thread must own >= 4 output chunks for the hoist to amortise (2 on the 64x64 tile)
    constexpr bool HOIST_PAYS = HOIST && (BM * CPR / NT) >= 4 && (BM / WM / 16) * (BN / WN / 16) * 4 < 128;
    // the two-phase chunk loop (below) only where the register file has the room: the 8-wave tiles up to 128 rows (64x320, 128x320,
    // 128x256: one block per CU, 256 VGPRs per wave); the 256-row tiles and the 2-blocks-per-CU kernels would spill
    // SMALL2: the 64x64 BK = 64 tile (4 waves; ~170 launches per step: out / proj / FF2 projections at the 32x32 .. 8x8 levels).  A thread
    // owns TWO chunks of one column there; with its accumulators dead after the single staging pass, U, bias and both rows' T /
    // residual chunks are one batch of loads instead of ~12 dependent ones per chunk.  Round 3 switched it off after it produced
    // sporadic wrong elements on MI355X; round 4 traced those to the packed-fp32 form of the rank-4 update (hoisted_rank4 above),
    // which is now explicit scalar FMAs -- the strict epilogue tests (no outlier element, bit-identical repeats) run over this tile.
#ifdef CLORA_SMALL2_OFF
    constexpr bool SMALL2 = false;                              // the round-3 state (A/B)
#else
    constexpr bool SMALL2 = BM == 64 && BN == 64 && NT == 256 && NPASS == 1 && SMEM >= 24576;
#endif
    constexpr bool LN_TILE = BN == 320 && NT == 512 && (BM == 64 || BM == 128);      // tiles that may carry clora_epilogue_t.ln_out
    constexpr bool FIXED_COL = HOIST_PAYS || SMALL2;           // thread -> one fixed chunk column, rows t / CPR + it * RPIT
    static_assert(EXT == 0 || FIXED_COL, "EXT tiles take the fixed-column hoisted epilogue");
    constexpr bool TWO_PHASE = (HOIST_PAYS && NT == 512 && BM <= 128) || SMALL2;
    const bool hoist = FIXED_COL && (EXT > 0 || p.hoist_on) && p.epi.lora_t != nullptr && p.epi.lora_r == 4 && p.epi.geglu == 0 && n < p.N &&
                       ((p.epi.ldt | ((n / p.epi.lora_seg) * 4)) & 3) == 0 && (p.epi.lora_u_tr ? (p.epi.ldu & 3) == 0 : p.epi.ldu == 4);
    // `two`: this thread takes the two-phase chunk loop -- adapter launches it can hoist, and launches without an adapter
    // (proj_in / proj_out / FF2 and the dgrads: bias and / or residual only)
    const bool two = TWO_PHASE && p.two_phase && p.epi.geglu == 0 && !p.epi.rowadd && n < p.N && (hoist || p.epi.lora_t == nullptr) &&
                     !p.epi.residual_lo && !p.epi.c_lo;         // (the compensated-trunk operands live in the one-chunk form only)
    floatx4 ureg[8];
    float bias8[8];
    int utoff = 0;
    if (two && !hoist) {
        floatx4 b0 = zero4f(), b1 = zero4f();
        if (p.epi.bias) { b0 = *reinterpret_cast<const floatx4*>(p.epi.bias + n); b1 = *reinterpret_cast<const floatx4*>(p.epi.bias + n + 4); }
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
    }
    if (hoist) {
        utoff = (n / p.epi.lora_seg) * 4;
#pragma unroll
        for (int q = 0; q < 8; ++q)
            ureg[q] = p.epi.lora_u_tr ? *reinterpret_cast<const floatx4*>(p.epi.lora_u + (size_t)(q >> 1) * p.epi.ldu + n + (q & 1) * 4)
                                      : *reinterpret_cast<const floatx4*>(p.epi.lora_u + (size_t)(n + q) * 4);
        floatx4 b0 = zero4f(), b1 = zero4f();
        if (p.epi.bias) { b0 = *reinterpret_cast<const floatx4*>(p.epi.bias + n); b1 = *reinterpret_cast<const floatx4*>(p.epi.bias + n + 4); }
#pragma unroll
        for (int e = 0; e < 4; ++e) { bias8[e] = b0[e]; bias8[4 + e] = b1[e]; }
    }
#pragma unroll
    for (int ph = 0; ph < NPASS; ++ph) {
        __syncthreads();                                       // ring (or previous pass) fully consumed
        const int wrow0 = wm * FM * 16;                        // first tile row of this wave
#pragma unroll
        for (int i = 0; i < FM; ++i) {
            const int frow = wrow0 + i * 16;                   // this fragment's 16 rows lie inside one 64-row pass
            if (frow >= ph * PR && frow < (ph + 1) * PR) {
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Cf[(frow - ph * PR + 4 * g + r) * F_LD + wn * FN * 16 + j * 16 + li] = acc[i][j][r];
            }
        }
        if constexpr (EXT > 0) {
            if (ph == 0) {
#pragma unroll
                for (int q = 0; q < TM; ++q) {
                    const int i = q * WN + wn;                 // the M fragment whose T columns this wave accumulated
                    if (i < FM && li < 8) {                    // columns 8..15 of the MFMA tile repeat 0..7 (the operand has 8 rows)
#pragma unroll
                        for (int r = 0; r < 4; ++r) Ts[(wrow0 + i * 16 + 4 * g + r) * 8 + li] = tacc[q][r];
                    }
                }
            }
        }
        __syncthreads();
        if (p.epi.geglu == 1) {
            // forward GEGLU: tile columns come in groups of [64 a | 64 g]; one thread takes an a-chunk and its g-chunk
            if constexpr (BN % 128 == 0) {
                constexpr int HPR = CPR / 2;                   // a-chunks per tile row
                const int F = p.epi.geglu_f;
                for (int c = t; c < PR * HPR; c += NT) {
                    const int ml = c / HPR, hc = c - ml * HPR;
                    const int col = (hc >> 3) * 128 + (hc & 7) * 8;        // tile column of the a-chunk; g sits 64 further
                    const int m = m0 + ph * PR + ml, n = n0 + col;
                    if (m < p.M && n < p.N) {
                        float va[8], vg[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) { va[e] = Cf[ml * F_LD + col + e]; vg[e] = Cf[ml * F_LD + col + 64 + e]; }
                        epi_chunk8(va, m, n, p.epi);
                        epi_chunk8(vg, m, n + 64, p.epi);
                        const int j = (n >> 7) * 64 + (hc & 7) * 8;       // column in the standard [a | g] layout
                        half8 a8, g8, y8;
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            a8[e] = (half_t)va[e];
                            g8[e] = (half_t)vg[e];
#ifdef CLORA_RES_ADD_TWICE
                            y8[e] = (half_t)((float)a8[e] * gelu_f((float)g8[e]));
#else
                            y8[e] = (half_t)(va[e] * gelu_f(vg[e]));             // from the fp32 values: one rounding (see CLORA_RES_ADD)
#endif
                        }
                        st8((half_t*)p.epi.geglu_y + (size_t)m * F + j, y8);
                        if (p.C) {
                            st8(p.C + (size_t)m * p.ldc + j, a8);
                            st8(p.C + (size_t)m * p.ldc + F + j, g8);
                        }
                    }
                }
            }
            continue;
        }
        // EXT: T row of tile row mt = hi + lo sums from LDS (+ the precomputed part), written to lora_t by the thread of chunk
        // column 0 when this tile is the first of its column segment
        const bool ext_in = EXT > 0 && p.epi.lora_t_in != nullptr && ((p.epi.lora_t_in_mask >> (n0 / p.epi.lora_seg)) & 1u);
        const bool ext_store = EXT > 0 && (n0 % p.epi.lora_seg) == 0 && nc == 0;
        auto ext_t4 = [&](int mt, floatx4 tin) -> floatx4 {
            const floatx4 hi = *reinterpret_cast<const floatx4*>(Ts + mt * 8);
            const floatx4 lo = *reinterpret_cast<const floatx4*>(Ts + mt * 8 + 4);
            return hi + lo + tin;
        };
        // one 8-column chunk of one staged row: bias / time embedding / adapter update -> fp16 -> + residual (or GEGLU') -> store
        auto chunk = [&](int ml, int nc, int n) {
            const int m = m0 + ph * PR + ml;
            if (m < p.M && n < p.N) {
                const floatx4 f0 = *reinterpret_cast<const floatx4*>(Cf + ml * F_LD + nc * 8);
                const floatx4 f1 = *reinterpret_cast<const floatx4*>(Cf + ml * F_LD + nc * 8 + 4);
                float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
#ifdef CLORA_DIAG_EPI_NOP
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // diagnostic build: drain + idle slots after the LDS reads
#endif
                if (hoist) {
                    if (p.epi.bias) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += bias8[e];
                    }
                    if (p.epi.rowadd) {
                        const half8 ra = ld8((const half_t*)p.epi.rowadd + (size_t)(m / p.epi.rows_per_batch) * p.epi.ld_rowadd + n);
#pragma unroll
                        for (int e = 0; e < 8; ++e) v[e] += (float)ra[e];
                    }
                    floatx4 t4;
                    if constexpr (EXT > 0) {
                        floatx4 tin = zero4f();
                        if (ext_in) tin = *reinterpret_cast<const floatx4*>(p.epi.lora_t_in + (size_t)(p.epi.lora_t_in_rows > 0 ? m % p.epi.lora_t_in_rows : m) * p.epi.ldt_in + utoff);
                        t4 = ext_t4(ph * PR + ml, tin);
                        if (ext_store) *reinterpret_cast<floatx4*>(const_cast<float*>(p.epi.lora_t) + (size_t)m * p.epi.ldt + utoff) = t4;
                    } else {
                        t4 = *reinterpret_cast<const floatx4*>(p.epi.lora_t + (size_t)m * p.epi.ldt + utoff);
                    }
                    t4 *= p.epi.lora_scale;
                    if (p.epi.lora_u_tr) hoisted_rank4<true>(v, t4, ureg);
                    else hoisted_rank4<false>(v, t4, ureg);
                } else {
                    epi_chunk8(v, m, n, p.epi);
                }
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                if (p.epi.geglu == 2) {                        // backward GEGLU: o = dy (fp16-rounded like the unfused path)
                    const int F = p.epi.geglu_f;
                    const half_t* hrow = (const half_t*)p.epi.geglu_h + (size_t)m * 2 * F + n;
                    const half8 a = ld8(hrow), gg = ld8(hrow + F);
                    half8 da, dg;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gf = (float)gg[e], df = (float)o[e];
                        float cdf, pdf;
                        gelu_parts(gf, cdf, pdf);
                        da[e] = (half_t)(df * gf * cdf);
                        dg[e] = (half_t)(df * (float)a[e] * (cdf + gf * pdf));
                    }
                    st8(p.C + (size_t)m * p.ldc + n, da);
                    st8(p.C + (size_t)m * p.ldc + F + n, dg);
                    return;
                }
                if (p.epi.residual_lo || p.epi.c_lo) {         // compensated trunk: the sum continues from (residual + residual_lo) and its
                    half8 rr = zero8(), rl = zero8(), lo;      // own rounding remainder goes to c_lo (include/clora.h)
                    if (p.epi.residual) rr = ld8((const half_t*)p.epi.residual + (size_t)m * p.epi.ldr + n);
                    if (p.epi.residual_lo) rl = ld8((const half_t*)p.epi.residual_lo + (size_t)m * p.epi.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float sum = v[e] + (float)rr[e] + (float)rl[e];
                        o[e] = (half_t)sum;
                        lo[e] = (half_t)(sum - (float)o[e]);
                    }
                    if (p.epi.c_lo) st8((half_t*)p.epi.c_lo + (size_t)m * p.ldc + n, lo);
                } else if (p.epi.residual) {                   // the sum is formed in fp32 and rounded ONCE (CLORA_RES_ADD, clora_epilogue.h)
                    const half8 rr = ld8((const half_t*)p.epi.residual + (size_t)m * p.epi.ldr + n);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = CLORA_RES_ADD(v[e], o[e], rr[e]);
                }
                st8(p.C + (size_t)m * p.ldc + n, o);
                if constexpr (LN_TILE) {                       // the stored fp16 values back to the staging rows: the LayerNorm pass below
                    if (p.epi.ln_out) {
                        floatx4 w0, w1;
#pragma unroll
                        for (int e = 0; e < 4; ++e) { w0[e] = (float)o[e]; w1[e] = (float)o[4 + e]; }
                        *reinterpret_cast<floatx4*>(Cf + ml * F_LD + nc * 8) = w0;
                        *reinterpret_cast<floatx4*>(Cf + ml * F_LD + nc * 8 + 4) = w1;
                    }
                }
            }
        };
        if constexpr (FIXED_COL) {
            // thread -> one fixed chunk column, rows ml = t / CPR + it * RPIT: what depends on the column only is already in registers
            if (two) {
                // Two-phase form: the accumulators are dead (staged in LDS), so the T row and the residual chunk of EVERY row this
                // thread owns are requested before the first one is used.  In the loop below each iteration is load -> wait -> use ->
                // store, and the compiler may not move a load above the previous iteration's store (C / residual / T may alias):
                // five to six dependent L2 round trips per thread on the short-K projections, whose main loop is only 5 k-steps.
                // Rows past the tile / matrix re-read the tile's first row (always valid) and are masked at the store.
                constexpr int NIT = (PR + RPIT - 1) / RPIT;
                floatx4 t4s[NIT];
                half8 rrs[NIT];
                const bool col_ok = t < RPIT * CPR;
                const bool has_res = p.epi.residual != nullptr;
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int ml = t / CPR + it * RPIT, m = m0 + ph * PR + ml;
                    const int mc = (col_ok && ml < PR && m < p.M) ? m : m0;
                    // unconditional loads (an absent operand reads the 16-byte zero page): no branch between them, one batch
                    const float* tp;
                    if constexpr (EXT > 0)
                        tp = ext_in ? p.epi.lora_t_in + (size_t)(p.epi.lora_t_in_rows > 0 ? mc % p.epi.lora_t_in_rows : mc) * p.epi.ldt_in + utoff
                                    : reinterpret_cast<const float*>(g_clora_zero16);
                    else
                        tp = hoist ? p.epi.lora_t + (size_t)mc * p.epi.ldt + utoff : reinterpret_cast<const float*>(g_clora_zero16);
                    const half_t* rp = has_res ? (const half_t*)p.epi.residual + (size_t)mc * p.epi.ldr + n
                                               : reinterpret_cast<const half_t*>(g_clora_zero16);
                    t4s[it] = *reinterpret_cast<const floatx4*>(tp);
                    rrs[it] = ld8(rp);
                }
                // pin the loaded registers only AFTER every load has been issued (an asm that consumes a register waits for its
                // load): stops the compiler from sinking each load into the conditional block that uses it
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    CLORA_KEEP(t4s[it]);
                    CLORA_KEEP(rrs[it]);
                }
#pragma unroll
                for (int it = 0; it < NIT; ++it) {
                    const int ml = t / CPR + it * RPIT, m = m0 + ph * PR + ml;
                    if (col_ok && ml < PR && m < p.M) {
                        const floatx4 f0 = *reinterpret_cast<const floatx4*>(Cf + ml * F_LD + nc * 8);
                        const floatx4 f1 = *reinterpret_cast<const floatx4*>(Cf + ml * F_LD + nc * 8 + 4);
                        float v[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
#ifdef CLORA_DIAG_EPI_NOP
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#endif
                        if (p.epi.bias) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) v[e] += bias8[e];
                        }
                        if (hoist) {
                            floatx4 t4 = t4s[it];
                            if constexpr (EXT > 0) {
                                t4 = ext_t4(ph * PR + ml, t4);
                                if (ext_store) *reinterpret_cast<floatx4*>(const_cast<float*>(p.epi.lora_t) + (size_t)m * p.epi.ldt + utoff) = t4;
                            }
                            t4 *= p.epi.lora_scale;
                            if (p.epi.lora_u_tr) hoisted_rank4<true>(v, t4, ureg);
                            else hoisted_rank4<false>(v, t4, ureg);
                        }
                        half8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
                        if (has_res) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) o[e] = CLORA_RES_ADD(v[e], o[e], rrs[it][e]);
                        }
                        st8(p.C + (size_t)m * p.ldc + n, o);
                        if constexpr (LN_TILE) {
                            if (p.epi.ln_out) {
                                floatx4 w0, w1;
#pragma unroll
                                for (int e = 0; e < 4; ++e) { w0[e] = (float)o[e]; w1[e] = (float)o[4 + e]; }
                                *reinterpret_cast<floatx4*>(Cf + ml * F_LD + nc * 8) = w0;
                                *reinterpret_cast<floatx4*>(Cf + ml * F_LD + nc * 8 + 4) = w1;
                            }
                        }
                    }
                }
            } else {
                for (int ml = t / CPR; ml < PR && t < RPIT * CPR; ml += RPIT) chunk(ml, nc, n);
            }
        } else {
            for (int c = t; c < PR * CPR; c += NT) {
                const int ml = c / CPR, ncc = c - ml * CPR;
                chunk(ml, ncc, n0 + ncc * 8);
            }
        }
        // ---- fused LayerNorm of the rows just stored (clora_epilogue_t.ln_out; the tile spans the row: N == BN == 320).  One wave per
        // row, lane l < 40 owns chunk l -- the partition, the formulas and the summation order of layernorm_rows_kernel
        // (clora_norm.hip), on the fp16 values that went to C (read back from the staging rows).
        if constexpr (LN_TILE) {
            if (p.epi.ln_out) {
                __syncthreads();
                const int lane = t & 63, wv = t >> 6;
                const bool lok = lane < CPR;
                const int cl = lok ? lane : 0;
                float gm[8], bt[8];
                {
                    const floatx4 g0 = *reinterpret_cast<const floatx4*>(p.epi.ln_gamma + cl * 8), g1 = *reinterpret_cast<const floatx4*>(p.epi.ln_gamma + cl * 8 + 4);
                    const floatx4 b0 = *reinterpret_cast<const floatx4*>(p.epi.ln_beta + cl * 8), b1 = *reinterpret_cast<const floatx4*>(p.epi.ln_beta + cl * 8 + 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e) { gm[e] = g0[e]; gm[4 + e] = g1[e]; bt[e] = b0[e]; bt[4 + e] = b1[e]; }
                }
                for (int r = wv; r < PR; r += NT / 64) {
                    const int m = m0 + ph * PR + r;
                    const floatx4 x0 = *reinterpret_cast<const floatx4*>(Cf + r * F_LD + cl * 8);
                    const floatx4 x1 = *reinterpret_cast<const floatx4*>(Cf + r * F_LD + cl * 8 + 4);
                    float x[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    float sum = 0.f;
                    if (lok) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) sum += x[e];
                    }
                    const float mean = wave_sum(sum) / (float)BN;
                    float q = 0.f;
                    if (lok) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float d = x[e] - mean; q += d * d; }
                    }
                    const float rstd = rsqrtf(wave_sum(q) / (float)BN + p.epi.ln_eps);
                    if (lok && m < p.M) {
                        half8 y;
#pragma unroll
                        for (int e = 0; e < 8; ++e) y[e] = (half_t)((x[e] - mean) * rstd * gm[e] + bt[e]);
                        st8((half_t*)p.epi.ln_out + (size_t)m * BN + cl * 8, y);
                    }
                }
            }
        }
    }
}

// blocks per CU the LDS ring allows, capped by what the accumulators leave room for in the register file: the register
// allocator is told to leave room for them.  (NST = 3, BK = 32: 48 / 36 / 24 KB -> 3 / 4 / 6 blocks; the deep rings for
// grids that cannot fill a CU with blocks anyway -- 5 x 16 KB, 6 x 12 KB, 8 x 8 KB -- leave 2 blocks per CU but 2-3x the
// bytes in flight per block; the 256x128 tile holds 128 accumulator registers per lane -> 2 blocks.)
template <int BM, int BN, int NST, int BK, int NW = 4> struct DmaOcc {
    static constexpr int lds = NST * (BM + BN) * BK * 2;
    static constexpr int fit = (160 * 1024) / lds;
    static constexpr int cap = NW > 4 ? 1 : ((BM * BN >= 256 * 128) ? 2 : ((BM * BN >= 128 * 128) ? 3 : ((BM * BN >= 128 * 64) ? 4 : 5)));   // 64x64 at 6 would spill (80 VGPRs)
    static constexpr int v = fit < 1 ? 1 : (fit < cap ? fit : cap);
};

// CONV: 0 = plain GEMM rows; 1 = generic gather (strided / asymmetric-pad / upsampled convs and their dgrads);
// 2 = the common case -- 3x3, stride 1 or 2 without upsampling / parity holes, Cin % BK == 0 (every ResnetBlock conv and
// its dgrad, the down-samplers, the hint-encoder stages): a BK step then lies
// inside ONE filter tap for the whole wave, so the tap walk and its address delta are scalar (SALU) work and each
// DMA instruction costs a bit test, an add and a select.  The generic path spends ~35 VALU/branch instructions per DMA
// instruction; with four of them per 16 MFMAs that made instruction issue, not the matrix pipe, the limiter of the conv
// GEMMs (plain GEMMs ran at ~600 TFLOP/s, the same shapes as convs at 340-540).
//
// BK: k-depth of one ring stage = one barrier.  BK = 32: 64-byte LDS rows, one DMA wave-instruction copies 16 rows x 64 B.
// BK = 64: 128-byte LDS rows, one DMA wave-instruction copies 8 rows x 128 B -- whole cache lines per row (half the
// address-unit work per byte, cdna_hip_programming.md "x through LDS in full lines") and half the barriers per MFMA.
// Swizzle (applied on the SOURCE side, the DMA image is lane-linear): the lane that fills 16-byte slot `pos` of row r
// fetches logical k-chunk pos ^ key(r), key(r) = (r >> 2) & 3 for 64-byte rows and r & 7 for 128-byte rows; fragment
// reads apply the same involution (both conflict-free for ds_read_b128's lane groups: tools/lds_bank_check.py).
// FLAGS bit 0 (ORD): 0 = refill the ring first, then read the fragments; 1 = fragment reads first (their latency overlaps
// the DMA issue).  FLAGS bit 1 (64-byte rows only, A/B runs): the round-1 key (r >> 2) & 3, which is 2-way conflicted for
// ds_read_b128's REAL lane groups ({0-3,12-15,20-27}, ...: MI355X_MICROARCH.md section LDS) -- PMC on MI355X:
// SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE with it, 0 with the default key (-(r >> 2)) & 3
// (profiles/r02_pmc_gemm_variants.md, tools/lds_bank_check.py).
// EXT = 8 (BK = 64 plain GEMMs: the 8-wave 320-column tiles and the 4-wave 64x64 / 128x64 / 128x128 tiles): 8 extra B rows per
// stage = the packed down matrix of this tile's column segment (clora_epilogue_t.lora_dpack: 4 rows fp16(D), 4 rows
// fp16(D - fp16(D))), filled by ONE extra DMA instruction of wave 0; T = A . D^T costs one extra MFMA per k-substep and M fragment
// (the MFMA's columns 8..15 re-read rows 0..7 and are dropped) on the waves with wn < FM, and never leaves the CU before the
// epilogue uses it.  Every n-tile of a column segment recomputes T (+25 % MFMAs on the 64-column tiles, which sit at 13 % MFMA
// busy); the first one writes it out.
template <int BM, int BN, int WM, int WN, int NST, int CONV, int BK = 32, int FLAGS = 0, int EXT = 0>
__global__ __launch_bounds__(WM * WN * 64, (DmaOcc<BM, BN, NST, BK, WM * WN>::v)) void gemm_dma_kernel(GemmArgs p) {
    constexpr int NW = WM * WN, NT = NW * 64;                // 4 waves, or 8 for the wide tiles (128x320, 64x320, 128x256: one block per CU)
    constexpr int ORD = FLAGS & 1;
    constexpr bool ALTKEY = (FLAGS & 2) == 0;
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
    constexpr int CH = BK / 8;                              // 16-byte chunks per LDS row
    constexpr int RPI = 64 / CH;                            // rows one DMA wave-instruction fills (16 or 8)
    constexpr int KS = BK / 32;                             // MFMA k-substeps per stage
    constexpr int A_IN = BM / RPI / NW, B_IN = BN / RPI / NW; // DMA wave-instructions per stage per wave
    static_assert(BM % (RPI * NW) == 0 && BN % (RPI * NW) == 0, "tile rows must split evenly over the waves' DMA instructions");
    static_assert(EXT == 0 || (EXT == 8 && CONV == 0 && BK == 64), "extra operand rows: BK = 64 plain GEMMs");
    constexpr int STAGE = (BM + BN + EXT) * BK;             // halves
    constexpr int SMEM_RING = NST * STAGE;
    constexpr int SMEM_EPI = (64 * (BN + 4) + (EXT ? BM * 8 : 0)) * 2;    // fp32 staging of 64 output rows (+ the tile's T rows), in halves
#ifdef CLORA_DIAG_MIN_SMEM
    // diagnostic build: a larger LDS request than the ring needs, so that fewer blocks fit on a CU (same instruction stream)
    constexpr int SMEM0 = SMEM_RING > SMEM_EPI ? SMEM_RING : SMEM_EPI;
    constexpr int SMEM = SMEM0 > (CLORA_DIAG_MIN_SMEM) ? SMEM0 : (CLORA_DIAG_MIN_SMEM);
#else
    constexpr int SMEM = SMEM_RING > SMEM_EPI ? SMEM_RING : SMEM_EPI;
#endif
    __shared__ __attribute__((aligned(16))) half_t smem[SMEM];

    const int t = threadIdx.x;
    // XCD-aware block order (cdna_hip_programming.md T1): workgroup L runs on XCD L % 8, each with a private L2.
    // Remap so that every XCD walks a CONTIGUOUS range of logical tiles (n fastest): the N-tiles that share one
    // block of A rows -- and the 9 taps of a conv that re-read the same input pixels -- hit the same L2.
    // (PMC evidence in profiles/r01_pmc_traffic.json: fabric reads were several x the algorithmic bytes.)
    const int tiles = gridDim.x, nwg = gridDim.x * gridDim.y;
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    int split = logical / tiles, tile = logical - split * tiles;
    int tile_m = p.n_major ? tile % p.tiles_m : tile / p.tiles_n, tile_n = p.n_major ? tile / p.tiles_m : tile % p.tiles_n;
    if (p.xg_m) {                                              // rectangles (pick_tile_order, tile_order = 3): nwg % 8 == 0 by construction
        const int rn = p.tiles_n / p.xg_n, rm = p.tiles_m / p.xg_m, rs = (int)gridDim.y / p.xg_s, idx = lin >> 3;
        const int sl = idx / (rm * rn), r = idx - sl * (rm * rn);
        split = (xcd / (p.xg_m * p.xg_n)) * rs + sl;
        tile_m = ((xcd / p.xg_n) % p.xg_m) * rm + r / rn;
        tile_n = (xcd % p.xg_n) * rn + r % rn;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * p.k_per_split;
    const int kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
    const int nk = (kend - kbeg + BK - 1) / BK;

    const int w = t >> 6, l = t & 63, g = l >> 4, li = l & 15;
    constexpr bool conv = CONV == 1;
    const half_t* zero_page = reinterpret_cast<const half_t*>(g_clora_zero16);
    // EXT: wave id as a scalar (the extra DMA instruction and its counted wait are wave-uniform branches)
    const int wu = EXT > 0 ? __builtin_amdgcn_readfirstlane(t >> 6) : 0;
    constexpr int X_W = EXT > 0 ? EXT / (64 / (BK / 8)) : 0;   // waves that issue one extra DMA instruction per stage
    const half_t* const bx_base = EXT > 0 ? reinterpret_cast<const half_t*>(p.epi.lora_dpack) + (size_t)(n0 / p.epi.lora_seg) * EXT * p.K : nullptr;

    // ---- loader: wave w, instruction i fills rows (w*IN + i)*RPI .. +RPI of the tile; lane -> (row l/CH, slot l%CH)
    const int lrow = l / CH, pos = l % CH;
    const int kc = BK == 32 ? (pos ^ ((ALTKEY ? -(lrow >> 2) : (lrow >> 2)) & 3)) : (pos ^ (lrow & 7));   // logical k-chunk this lane fetches (same for all its rows)
    bool a_ok[A_IN];
    size_t a_base[A_IN];
    int a_ty[A_IN], a_tx[A_IN];
    unsigned a_mask[A_IN];                                  // CONV == 2: bit t = filter tap t of this row is in bounds
#pragma unroll
    for (int i = 0; i < A_IN; ++i) {
        const int m = m0 + (w * A_IN + i) * RPI + lrow;
        a_ok[i] = m < p.M;
        a_base[i] = 0; a_ty[i] = 0; a_tx[i] = 0; a_mask[i] = 0;
        if (a_ok[i]) {
            if (CONV == 0) {
                a_base[i] = (size_t)m * p.lda;
            } else if (CONV == 2) {
                const int hw = p.conv.Hout * p.conv.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int yo = rem / p.conv.Wout, xo = rem - yo * p.conv.Wout;
                const int yc = yo * p.conv.mul, xc = xo * p.conv.mul;                      // stride: still linear in the tap
                a_base[i] = (((size_t)b * p.conv.Hin + yc) * p.conv.Win + xc) * p.conv.Cin + kc * 8;   // centre pixel + lane chunk
#pragma unroll
                for (int tp = 0; tp < 9; ++tp) {
                    const int ty = yc + p.conv.off + (tp / 3) * p.conv.kmul, tx = xc + p.conv.off + (tp % 3) * p.conv.kmul;
                    if (ty >= 0 && ty < p.conv.lim_h && tx >= 0 && tx < p.conv.lim_w) a_mask[i] |= 1u << tp;
                }
            } else {
                const int hw = p.conv.Hout * p.conv.Wout;
                const int b = m / hw, rem = m - b * hw;
                const int yo = rem / p.conv.Wout, xo = rem - yo * p.conv.Wout;
                a_base[i] = (size_t)b * p.conv.Hin * p.conv.Win;
                a_ty[i] = yo * p.conv.mul + p.conv.off;
                a_tx[i] = xo * p.conv.mul + p.conv.off;
            }
        }
    }
    bool b_ok[B_IN];
    size_t b_base[B_IN];
#pragma unroll
    for (int i = 0; i < B_IN; ++i) {
        const int n = n0 + (w * B_IN + i) * RPI + lrow;
        b_ok[i] = n < p.N;
        b_base[i] = (size_t)n * p.K;
    }
    int k = kbeg + kc * 8;
    // K order (clora_conv_t.kchunk): slabs of kcs channels, the 9 taps of a slab back to back (kcs = Cin: plain tap-major)
    const int kcs = p.conv.kchunk > 0 ? p.conv.kchunk : p.conv.Cin, ntap = p.conv.ksize * p.conv.ksize;
    int tap = 0, ci = k, cb = 0;       // generic gather: tap, channel inside the slab, first channel of the slab (per lane)
    if (conv) { cb = (k / (ntap * kcs)) * kcs; const int r = k % (ntap * kcs); tap = r / kcs; ci = r - tap * kcs; }
    // CONV == 2: wave-uniform walk (kq = first k of the stage, its tap, slab and offset inside the slab), kept in SGPRs
    int kq = kbeg, qtap = 0, qci = 0, qcb = 0;
    if (CONV == 2) { qcb = (kbeg / (9 * kcs)) * kcs; const int r = kbeg % (9 * kcs); qtap = r / kcs; qci = r - qtap * kcs; }

    auto issue_stage = [&](int buf) {
        half_t* As = smem + buf * STAGE;
        half_t* Bs = As + BM * BK;
        if (CONV == 2) {
            const bool kokq = kq < kend;                    // K = 9*Cin and the split size are multiples of BK: whole stages only
            const int qky = qtap / 3, qkx = qtap - qky * 3;
            const long delta = ((long)(p.conv.off + qky * p.conv.kmul) * p.conv.Win + (p.conv.off + qkx * p.conv.kmul)) * p.conv.Cin + qcb + qci;
#pragma unroll
            for (int i = 0; i < A_IN; ++i) {
                const bool ok = kokq && ((a_mask[i] >> qtap) & 1u);
                const half_t* src = ok ? p.A + (long)a_base[i] + delta : zero_page;
#ifdef CLORA_DMA_PROBE
                if (FLAGS & 4) src = zero_page;             // timing probe: same instruction stream, no operand traffic
#endif
                CLORA_GLDS16(src, As + (w * A_IN + i) * RPI * BK);
            }
#pragma unroll
            for (int i = 0; i < B_IN; ++i) {
                const half_t* src = (b_ok[i] && kokq) ? p.B + b_base[i] + kq + kc * 8 : zero_page;
#ifdef CLORA_DMA_PROBE
                if (FLAGS & 8) src = zero_page;
#endif
                CLORA_GLDS16(src, Bs + (w * B_IN + i) * RPI * BK);
            }
            kq += BK;
            qci += BK;
            if (qci >= kcs) { qci = 0; if (++qtap == 9) { qtap = 0; qcb += kcs; } }
            return;
        }
        const bool kok = k < kend;
        int ky = 0, kx = 0;
        if (conv) { ky = tap / p.conv.ksize; kx = tap - ky * p.conv.ksize; }
#pragma unroll
        for (int i = 0; i < A_IN; ++i) {
            const half_t* src = zero_page;
            if (a_ok[i] && kok) {
                if (!conv) {
                    src = p.A + a_base[i] + k;
                } else {
                    const int ty = a_ty[i] + ky * p.conv.kmul, tx = a_tx[i] + kx * p.conv.kmul;
                    bool ok = ty >= 0 && ty < p.conv.lim_h && tx >= 0 && tx < p.conv.lim_w;
                    if (p.conv.need_even) ok = ok && (((ty | tx) & 1) == 0);
                    if (ok)
                        src = p.A + (a_base[i] + (size_t)(ty >> p.conv.shift) * p.conv.Win + (tx >> p.conv.shift)) * p.conv.Cin + cb + ci;
                }
            }
#ifdef CLORA_DMA_PROBE
            if (FLAGS & 4) src = zero_page;
#endif
            CLORA_GLDS16(src, As + (w * A_IN + i) * RPI * BK);
        }
#pragma unroll
        for (int i = 0; i < B_IN; ++i) {
            const half_t* src = (b_ok[i] && kok) ? p.B + b_base[i] + k : zero_page;
#ifdef CLORA_DMA_PROBE
            if (FLAGS & 8) src = zero_page;
#endif
            CLORA_GLDS16(src, Bs + (w * B_IN + i) * RPI * BK);
        }
        if constexpr (EXT > 0) {
            if (wu < X_W) {
                const half_t* src = kok ? bx_base + (size_t)(wu * RPI + lrow) * p.K + k : zero_page;
                CLORA_GLDS16(src, Bs + (BN + wu * RPI) * BK);
            }
        }
        k += BK;
        if (conv) {
            ci += BK;
            while (ci >= kcs) { ci -= kcs; if (++tap == ntap) { tap = 0; cb += kcs; } }
        }
    };

    const int wm = w / WN, wn = w % WN;
    // swizzled 16-byte slot of logical chunk (ks*4 + g) for fragment rows == li (mod 16)
    int fsw[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) fsw[ks] = (BK == 32 ? ((g ^ (ALTKEY ? -(li >> 2) : (li >> 2))) & 3) : ((ks * 4 + g) ^ (li & 7))) * 8;
    floatx4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = zero4f();
    constexpr int TM = EXT > 0 ? (FM + WN - 1) / WN : 1;     // M fragments of the T columns per wave: i = q * WN + wn
    floatx4 tacc[TM];
#pragma unroll
    for (int q = 0; q < TM; ++q) tacc[q] = zero4f();
    const int wnu = EXT > 0 ? wu % WN : 0;

    // NST-1 stages are always in flight: stages past the end of K fetch the zero page (cheap L2 hits), which keeps
    // the counted wait a compile-time constant
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) issue_stage(st);
    int rd = 0, wr = NST - 1;
    for (int kt = 0; kt < nk; ++kt) {
        if (EXT > 0 && wu < X_W) CLORA_WAIT_VMCNT((NST - 2) * (A_IN + B_IN + 1));
        else CLORA_WAIT_VMCNT((NST - 2) * (A_IN + B_IN));      // stage kt has landed (loads retire in order)
        CLORA_RAW_BARRIER();                                   // ... for every wave, and stage kt-1 is fully consumed
        if (ORD == 0) { issue_stage(wr); wr = (wr + 1 == NST) ? 0 : wr + 1; }
        const half_t* As = smem + rd * STAGE;
        const half_t* Bs = As + BM * BK;
        rd = (rd + 1 == NST) ? 0 : rd + 1;
        half8 af[KS][FM], bf[KS][FN];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
            for (int i = 0; i < FM; ++i) af[ks][i] = ld8(As + (wm * FM * 16 + i * 16 + li) * BK + fsw[ks]);
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[ks][j] = ld8(Bs + (wn * FN * 16 + j * 16 + li) * BK + fsw[ks]);
        }
        if (ORD == 1) { issue_stage(wr); wr = (wr + 1 == NST) ? 0 : wr + 1; }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(af[ks][i], bf[ks][j], acc[i][j]);
        if constexpr (EXT > 0) {
            const half_t* Es = Bs + BN * BK;
#pragma unroll
            for (int q = 0; q < TM; ++q) {
                if (q * WN + wnu < FM) {                       // wave-uniform
#pragma unroll
                    for (int ks = 0; ks < KS; ++ks) {
                        half8 a = af[ks][q * WN];
#pragma unroll
                        for (int j = 1; j < WN; ++j)
                            if (q * WN + j < FM && wnu == j) a = af[ks][q * WN + j < FM ? q * WN + j : 0];
                        tacc[q] = mfma16(a, ld8(Es + (li & (EXT - 1)) * BK + fsw[ks]), tacc[q]);
                    }
                }
            }
        }
    }
    CLORA_WAIT_VMCNT(0);                                       // trailing zero-page stages: LDS is reused below
    if constexpr (EXT > 0) {
        dma_epilogue<BM, BN, WM, WN, NT, SMEM, true, EXT>(p, acc, m0, n0, split, smem, t, tacc);
        return;
    }
    // HOIST (U / bias of a thread's column in registers for the whole tile) only on the 8-wave tiles = ONE block per CU.  Round 2 had it
    // on every kernel built for <= 2 blocks per CU; the strict epilogue test of round 3 (no element off by more than a few ulps, two
    // launches bit-identical) caught those variants -- 128x64 BK64 at 2 blocks per CU, and the 64x64 experiment at 3 -- producing a handful
    // of wrong elements per launch at M = 16384, N = K = 320 on hardware, sporadically; the one-block-per-CU kernels and the
    // non-hoisted epilogue are clean and bit-stable (profiles/r03_epi_diag*.txt, r03_gputest_11.log).  Cause not established.
#ifdef CLORA_HOIST_8WAVE_ONLY
    dma_epilogue<BM, BN, WM, WN, NT, SMEM, (NW > 4)>(p, acc, m0, n0, split, smem, t);                   // the round-3 restriction (A/B)
#else
    // the hoisted epilogue on every kernel built for <= 2 blocks per CU (the round-2 scope), restored in round 4 with the scalar
    // rank-4 update of hoisted_rank4
    dma_epilogue<BM, BN, WM, WN, NT, SMEM, (NW > 4 || DmaOcc<BM, BN, NST, BK, NW>::v <= 2)>(p, acc, m0, n0, split, smem, t);
#endif
}

// ------------------------------------------------------------------------------------------------
// gemm_8p_kernel (tile_cfg 59): the plain GEMM on a 256 x 256 tile with the eight-phase ping-pong schedule of the CDNA4 guide
// (cdna_hip_programming.md section 5, "The 256^2 8-phase template"), for the launches with M >= 16384 and wide N (FeedForward
// projections, q|k|v at batch 32, the 8192^3 calibration GEMM).  What differs from gemm_dma_kernel's one-barrier-per-stage loop:
//   * 8 waves = 2 (M) x 4 (N), wave tile 128 x 64 (FM = 8, FN = 4: 12 fragment reads per 64 MFMAs of a BK = 64 step instead of
//     the 32x128 / 64x64 wave tiles' 20 / 16: the LDS fragment traffic was what kept the other tiles at 23 % MFMA busy);
//   * a K-tile is FOUR phases of 16 MFMAs, one quadrant (64 rows x 32 columns) of the wave tile each: (A0,B0) (A0,B1) (A1,B1) (A1,B0);
//     phase 0 reads the A0 and B0 fragments, phase 1 B1, phase 2 A1, phase 3 nothing (B0 stays in registers): every fragment is read
//     from LDS exactly once per K-tile;
//   * the two wave rows run ONE BARRIER APART (the wr == 1 waves pass an extra barrier before the loop, the wr == 0 waves one after
//     it): between two barriers one wave of every SIMD issues its fragment reads and LDS-DMA while the other one owns the matrix
//     pipe (s_setprio 1 around the MFMA cluster) -- the roles swap at every barrier;
//   * operands arrive as 16-KB HALF-TILES by LDS-DMA (128 rows x 128 B; A-half h = the rows with (row % 128) / 64 == h of both wave
//     rows, B-half h = the columns with (col % 64) / 32 == h of all four wave columns: exactly what the phases consume), one half-tile
//     = 2 DMA instructions per thread per phase, SEVEN half-tiles ahead, in an 8-slot ring (2 K-tiles x {A0, B0, B1, A1} = 128 KB).
//     Counted wait ONCE per K-tile: s_waitcnt vmcnt(6) in phase 3 (three half-tiles stay in flight across the barriers) retires the
//     whole next K-tile; it sits before the phase's first barrier, so every wave -- of either wave row -- reads the buffer only after
//     a barrier that all issuing waves passed behind their wait.
//   * write-after-read: half-tile G + 7 lands in the slot of half-tile G - 1, whose last fragment read was issued one phase (A0) or
//     two phases earlier and RETIRED (s_waitcnt lgkmcnt(0)) before the first barrier of its phase by both wave rows.
// Rows past M / N are clamped to the last valid row (they feed accumulator rows / columns the epilogue never stores), k past the end
// of the split fetches the zero page (the tail K-tile of K % 64 != 0, and the trailing prefetches that keep the wait a constant).
// LDS rows are 128 B, key r & 7 on the source side as in gemm_dma_kernel (BK = 64): conflict-free ds_read_b128.
template <int CONV>
__global__ __launch_bounds__(512) void gemm_8p_kernel(GemmArgs p) {
    static_assert(CONV == 0, "plain GEMM rows only");
    constexpr int BM = 256, BN = 256, BK = 64, WM = 2, WN = 4, NT = 512;
    constexpr int FM = 8, FN = 4;
    constexpr int HT = 128 * BK;                             // halves per half-tile (16 KB)
    constexpr int kInFlight8p = 6;                           // DMA instructions of the three half-tiles that stay in flight across the K-tile's wait
    constexpr int SMEM = 8 * HT;                             // 128 KB
    static_assert(64 * (BN + 4) * 2 <= SMEM, "epilogue staging");
    __shared__ __attribute__((aligned(16))) half_t smem[SMEM];

    const int t = threadIdx.x;
    const int tiles = gridDim.x, nwg = gridDim.x * gridDim.y;  // XCD-aware block order, as in gemm_dma_kernel
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    int split = logical / tiles, tile = logical - split * tiles;
    int tile_m = p.n_major ? tile % p.tiles_m : tile / p.tiles_n, tile_n = p.n_major ? tile / p.tiles_m : tile % p.tiles_n;
    if (p.xg_m) {
        const int rn = p.tiles_n / p.xg_n, rm = p.tiles_m / p.xg_m, rs = (int)gridDim.y / p.xg_s, idx = lin >> 3;
        const int sl = idx / (rm * rn), r = idx - sl * (rm * rn);
        split = (xcd / (p.xg_m * p.xg_n)) * rs + sl;
        tile_m = ((xcd / p.xg_n) % p.xg_m) * rm + r / rn;
        tile_n = (xcd % p.xg_n) * rn + r % rn;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int kbeg = split * p.k_per_split;
    const int kend = (kbeg + p.k_per_split < p.K) ? kbeg + p.k_per_split : p.K;
    const int nk = (kend - kbeg + BK - 1) / BK;

    const int w = t >> 6, l = t & 63, g = l >> 4, li = l & 15;
    const int wr = w >> 2, wc = w & 3;                       // wave row / column: rows wr*128.., columns wc*64..
    const half_t* zero_page = reinterpret_cast<const half_t*>(g_clora_zero16);

    // ---- loader: per half-tile a thread issues two DMA instructions; instruction i of wave w fills local rows (w*2 + i)*8 .. +8,
    // lane -> (row l >> 3, slot l & 7) and fetches logical k-chunk kc = slot ^ (row & 7)
    const int lrow8 = l >> 3, kc = (l & 7) ^ (lrow8 & 7);
    const half_t* srcA[2][2];                                // [half][instruction]: row pointer + lane chunk, K-tile 0 of the split
    const half_t* srcB[2][2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = (w * 2 + i) * 8 + lrow8;          // local row of the half-tile, 0..127
            int m = m0 + (lr >> 6) * 128 + h * 64 + (lr & 63);
            int n = n0 + (lr >> 5) * 64 + h * 32 + (lr & 31);
            m = m < p.M ? m : p.M - 1;
            n = n < p.N ? n : p.N - 1;
            srcA[h][i] = p.A + (size_t)m * p.lda + kbeg + kc * 8;
            srcB[h][i] = p.B + (size_t)n * p.K + kbeg + kc * 8;
        }
    const int klane = kbeg + kc * 8;
    // stage half-tile `slot` (0: A0, 1: B0, 2: B1, 3: A1) of K-tile ts of this split into ring buffer ts & 1
    auto stage = [&](int ts, int slot) {
        half_t* dst = smem + ((ts & 1) * 4 + slot) * HT + w * 2 * 512;
        const bool kok = klane + ts * BK < kend;
        const size_t ko = (size_t)ts * BK;
        const half_t* s0 = slot == 0 ? srcA[0][0] : slot == 1 ? srcB[0][0] : slot == 2 ? srcB[1][0] : srcA[1][0];
        const half_t* s1 = slot == 0 ? srcA[0][1] : slot == 1 ? srcB[0][1] : slot == 2 ? srcB[1][1] : srcA[1][1];
        CLORA_GLDS16(kok ? s0 + ko : zero_page, dst);
        CLORA_GLDS16(kok ? s1 + ko : zero_page, dst + 512);
    };

    // ---- fragment addressing: A fragment i (rows wr*128 + i*16 + li) lives in A-half i / 4 at local row wr*64 + (i % 4)*16 + li;
    // B fragment j (columns wc*64 + j*16 + li) in B-half j / 2 at local row wc*32 + (j % 2)*16 + li; 16-byte slot (ks*4 + g) ^ (li & 7)
    int fsw[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) fsw[ks] = ((ks * 4 + g) ^ (li & 7)) * 8;
    const int a_row = (wr * 64 + li) * BK, b_row = (wc * 32 + li) * BK;

    floatx4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = zero4f();

    // prologue: half-tiles 0..6 (K-tile 0 complete, K-tile 1 without its A1), then K-tile 0 landed and published
    stage(0, 0); stage(0, 1); stage(0, 2); stage(0, 3);
    stage(1, 0); stage(1, 1); stage(1, 2);
    CLORA_WAIT_VMCNT(6);
    CLORA_RAW_BARRIER();
    if (wr == 1) CLORA_RAW_BARRIER();                        // the second wave row runs one barrier behind the first

    half8 a0[2][4], a1[2][4], b0[2][2], b1[2][2];
    for (int kt = 0; kt < nk; ++kt) {
        const half_t* buf = smem + (kt & 1) * 4 * HT;
        // ---- phase 0: A0, B0 fragments; quadrant (rows 0..63, columns 0..31) of the wave tile; stages A1 of K-tile kt + 1
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) b0[ks][j] = ld8(buf + 1 * HT + b_row + j * 16 * BK + fsw[ks]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) a0[ks][i] = ld8(buf + 0 * HT + a_row + i * 16 * BK + fsw[ks]);
        stage(kt + 1, 3);
        CLORA_WAIT_LGKMCNT(0);
        CLORA_RAW_BARRIER();
        CLORA_SCHED_BARRIER();
        CLORA_SETPRIO(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(a0[ks][i], b0[ks][j], acc[i][j]);
        CLORA_SETPRIO(0);
        CLORA_SCHED_BARRIER();
        CLORA_RAW_BARRIER();
        // ---- phase 1: B1 fragments; quadrant (rows 0..63, columns 32..63); stages A0 of K-tile kt + 2 (A0 of kt was read a phase ago)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) b1[ks][j] = ld8(buf + 2 * HT + b_row + j * 16 * BK + fsw[ks]);
        stage(kt + 2, 0);
        CLORA_WAIT_LGKMCNT(0);
        CLORA_RAW_BARRIER();
        CLORA_SCHED_BARRIER();
        CLORA_SETPRIO(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][2 + j] = mfma16(a0[ks][i], b1[ks][j], acc[i][2 + j]);
        CLORA_SETPRIO(0);
        CLORA_SCHED_BARRIER();
        CLORA_RAW_BARRIER();
        // ---- phase 2: A1 fragments; quadrant (rows 64..127, columns 32..63); stages B0 of K-tile kt + 2
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i) a1[ks][i] = ld8(buf + 3 * HT + a_row + i * 16 * BK + fsw[ks]);
        stage(kt + 2, 1);
        CLORA_WAIT_LGKMCNT(0);
        CLORA_RAW_BARRIER();
        CLORA_SCHED_BARRIER();
        CLORA_SETPRIO(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][2 + j] = mfma16(a1[ks][i], b1[ks][j], acc[4 + i][2 + j]);
        CLORA_SETPRIO(0);
        CLORA_SCHED_BARRIER();
        CLORA_RAW_BARRIER();
        // ---- phase 3: no fragment reads (B0 is still in registers); quadrant (rows 64..127, columns 0..31); stages B1 of K-tile kt + 2;
        // the K-tile's one counted wait: everything but the three newest half-tiles has landed = all of K-tile kt + 1
        stage(kt + 2, 2);
        CLORA_WAIT_VMCNT(kInFlight8p);
        CLORA_RAW_BARRIER();
        CLORA_SCHED_BARRIER();
        CLORA_SETPRIO(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[4 + i][j] = mfma16(a1[ks][i], b0[ks][j], acc[4 + i][j]);
        CLORA_SETPRIO(0);
        CLORA_SCHED_BARRIER();
        CLORA_RAW_BARRIER();
    }
    if (wr == 0) CLORA_RAW_BARRIER();                        // re-join the two wave rows
    CLORA_WAIT_VMCNT(0);                                     // trailing zero-page prefetches: LDS is reused below
    dma_epilogue<BM, BN, WM, WN, NT, SMEM, true>(p, acc, m0, n0, split, smem, t);
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride-1 / pad-1 convolution and its dgrad with the input staged ONCE per channel slab as a spatial patch in LDS.
//
// The implicit GEMM above fetches every input pixel nine times from L2 (once per filter tap); with its operand DMAs
// redirected to a zero page the same kernel runs 32-40 % faster on the ResnetBlock convs (tools/dma_probe.py,
// profiles/r02_dma_probe.txt): the gather traffic, not the matrix pipe, bounds it.  Here a workgroup owns BM output pixels =
// whole image rows (or whole images at 8x8 and below) and per 64-channel slab loads the (rows+2) x (W+2) halo patch by
// LDS-DMA once -- 1.4-2.1 x the output pixels instead of 9 x -- into a double-buffered patch; the nine taps of the slab are
// nine k-steps whose A fragments are read from the patch at a wave-uniform pixel offset.  The weight operand streams
// through an NST-stage ring exactly as in gemm_dma_kernel (its K order is already slab-major: clora_conv_t.kchunk = 64).
//   patch pixel pp = (image * (rows+2) + patch_row) * (W+2) + patch_col, 128 bytes (64 channels) each; the 16-byte chunk c of
//   pixel pp lives in slot c ^ (pp & 7): conflict-free ds_read_b128 for 16 consecutive pixels (source-side swizzle, the DMA
//   image is lane-linear: the lane that fills slot `pos` of pixel pp fetches chunk pos ^ (pp & 7)).
//   Out-of-image halo pixels and rows past M come from the zero page: padding costs nothing.
// One workgroup = WM x WN waves (512 threads: the patch pair + ring take most of the CU's LDS, so the eight waves of ONE
// block provide the two waves per SIMD).  vmcnt waits are compile-time constants: the tap loop is unrolled and every
// wave issues the same number of DMA instructions per k-step (B_IN, plus PA_IN at tap 0; surplus ones fetch the zero page
// into a dump slot).
template <int BM, int BN> struct PatchCfg {
    static constexpr int MAXPP = (BM == 256) ? 400 : 288;    // patch pixels (multiple of 8): 4x66 / 6x34 / 10x18 / nx10x10 ... see eligibility
};
// WIDE (tile_cfg 77, 78): a 392-pixel patch = 3 x (128 + 2): ONE image row of 128 pixels, or a 128-pixel SEGMENT of a wider row
// (W = 256, 512: the VAE's 128..256-channel levels; the tile's pixels are still consecutive rows of the NHWC operand).  The
// halo patch is then 3.05x the output pixels instead of the 9x the implicit GEMM gathers.
constexpr int kPatchWide = 392;

template <int BM, int BN, int WM, int WN, int NST, int MAXPP_ = 0>
__global__ __launch_bounds__(WM * WN * 64, 1) void conv3x3_patch_kernel(GemmArgs p) {
    constexpr int NT = WM * WN * 64, NW = WM * WN, BK = 64;
    constexpr int FM = BM / WM / 16, FN = BN / WN / 16;
    constexpr int MAXPP = MAXPP_ > 0 ? MAXPP_ : PatchCfg<BM, BN>::MAXPP;
    constexpr int PA_IN = (MAXPP / 8 + NW - 1) / NW;           // patch DMA wave-instructions per wave per slab
    constexpr int B_NI = BN / 8;                               // ring DMA wave-instructions per stage (8 rows x 128 B each)
    constexpr int B_IN = (B_NI + NW - 1) / NW;                 // ... per wave (instruction ib = w + NW*i; surplus ones fetch the zero page into a dump slot)
    // 256x160 (tile_cfg 79; 64x80 wave tiles: 18 fragment reads per 40 MFMAs, the only patch tile on the matrix-pipe side of the LDS
    // budget at N = 320) needs exactly the CU's 160 KB: no dump slots -- a surplus DMA instruction repeats its wave's previous one
    // (same source, same destination: a duplicate write of identical bytes), the instruction counts the vmcnt waits rely on are unchanged
    constexpr bool NODUMP = BM == 256 && BN == 160;
    constexpr int PATCH = MAXPP * 64 + (NODUMP ? 0 : 512);     // halves per patch buffer (+ one 1-KB dump slot)
    constexpr int BST = BN * BK + (((B_NI % NW) && !NODUMP) ? 512 : 0);   // halves per ring stage (+ dump slot when the split is uneven: BN = 160)
    constexpr int SMEM = 2 * PATCH + NST * BST;
    static_assert(SMEM * 2 <= 160 * 1024, "LDS per workgroup");
    static_assert(!NODUMP || (PA_IN >= 2 && (NW * PA_IN - NW) * 8 <= MAXPP && B_IN >= 2 && NW * B_IN - NW <= B_NI), "a surplus instruction needs a valid predecessor in its wave");
    static_assert(64 * (BN + 4) * 2 <= SMEM, "epilogue staging");
    __shared__ __attribute__((aligned(16))) half_t smem[SMEM];
    half_t* const Pb = smem;
    half_t* const Bring = smem + 2 * PATCH;

    const int t = threadIdx.x;
    const int tiles = gridDim.x, nwg = gridDim.x * gridDim.y;  // XCD-aware block order, as in gemm_dma_kernel
    const int lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    int split = logical / tiles, tile = logical - split * tiles;
    int tile_m = p.n_major ? tile % p.tiles_m : tile / p.tiles_n, tile_n = p.n_major ? tile / p.tiles_m : tile % p.tiles_n;
    if (p.xg_m) {                                              // rectangles (pick_tile_order, tile_order = 3): nwg % 8 == 0 by construction
        const int rn = p.tiles_n / p.xg_n, rm = p.tiles_m / p.xg_m, rs = (int)gridDim.y / p.xg_s, idx = lin >> 3;
        const int sl = idx / (rm * rn), r = idx - sl * (rm * rn);
        split = (xcd / (p.xg_m * p.xg_n)) * rs + sl;
        tile_m = ((xcd / p.xg_n) % p.xg_m) * rm + r / rn;
        tile_n = (xcd % p.xg_n) * rn + r % rn;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int w = t >> 6, l = t & 63, g = l >> 4, li = l & 15;
    const int wm = w / WN, wn = w % WN;
    const half_t* zero_page = reinterpret_cast<const half_t*>(g_clora_zero16);

    // ---- geometry of this tile
    // (nearest-2x upsampled convs: the tile / patch live at the OUTPUT resolution and the loader reads source pixel (y>>1, x>>1))
    const int H = p.conv.Hout, W = p.conv.Wout, C = p.conv.Cin, HW = H * W, sh = p.conv.shift;
    // images per tile, image rows per image, tile width: whole rows (TW = W), or for W > BM a BM-pixel segment of ONE row
    const int nimg = HW >= BM ? 1 : BM / HW, rpi = HW >= BM ? (W >= BM ? 1 : BM / W) : H, TW = W > BM ? BM : W;
    const int PW = TW + 2, PRI = rpi + 2;
    const int npp = nimg * PRI * PW;
    const int b0 = m0 / HW, y0 = (m0 - b0 * HW) / W, x0 = W > BM ? (m0 - b0 * HW - y0 * W) : 0;
    const int nb = p.M / HW;
    const int slabs = C / 64;
    const int cs_beg = split * p.k_per_split, cs_end_ = cs_beg + p.k_per_split;   // k_per_split counts SLABS here
    const int cs_end = cs_end_ < slabs ? cs_end_ : slabs;

    // ---- patch loader: instruction i = w + NW*j fills slots i*64 .. +63; slot s -> pixel s >> 3, position s & 7
    int poff[PA_IN];                                           // element offset of this lane's 16-byte chunk at slab 0; -1: zero page
#pragma unroll
    for (int j = 0; j < PA_IN; ++j) {
        const int ip = (NODUMP && (w + NW * j) * 8 >= MAXPP) ? w + NW * (j - 1) : w + NW * j;
        const int sl = ip * 64 + l, pp = sl >> 3, pos = sl & 7;
        poff[j] = -1;
        if (pp < npp) {
            const int img = pp / (PRI * PW), rem = pp - img * (PRI * PW);
            const int pr = rem / PW, pc = rem - pr * PW;
            const int y = y0 - 1 + pr, x = x0 + pc - 1, b = b0 + img;
            if (b < nb && y >= 0 && y < H && x >= 0 && x < W)
                poff[j] = ((b * p.conv.Hin + (y >> sh)) * p.conv.Win + (x >> sh)) * C + ((pos ^ (pp & 7)) * 8);
        }
    }
    auto issue_patch = [&](int cs, int buf) {
        half_t* dst = Pb + buf * PATCH;
        const bool live = cs < cs_end;
#pragma unroll
        for (int j = 0; j < PA_IN; ++j) {
            const int i = w + NW * j;
            const half_t* src = (live && poff[j] >= 0) ? p.A + poff[j] + cs * 64 : zero_page;
            CLORA_GLDS16(src, (i * 8 < MAXPP) ? dst + i * 512 : dst + (NODUMP ? (i - NW) * 512 : MAXPP * 64));   // surplus: dump slot (or a repeat)
        }
    };
    // ---- ring loader (rows of the [N, K] weight operand, 128-byte rows, key r & 7): instruction ib = w + NW*i fills rows ib*8..+8
    const int lrow = l >> 3, bpos = l & 7, kc = bpos ^ (lrow & 7);
    bool b_ok[B_IN];
    size_t b_base[B_IN];
#pragma unroll
    for (int i = 0; i < B_IN; ++i) {
        const int ib = w + NW * i, ibe = (NODUMP && ib >= B_NI) ? ib - NW : ib, n = n0 + ibe * 8 + lrow;
        b_ok[i] = ibe < B_NI && n < p.N;
        b_base[i] = (size_t)n * p.K + kc * 8;
    }
    const int kbeg = cs_beg * 576, kend = cs_end * 576;
    int kq = kbeg;
    auto issue_b = [&](int buf) {
        half_t* Bs = Bring + buf * BST;
        const bool kok = kq < kend;
#pragma unroll
        for (int i = 0; i < B_IN; ++i) {
            const int ib = w + NW * i;
            const half_t* src = (b_ok[i] && kok) ? p.B + b_base[i] + kq : zero_page;
            CLORA_GLDS16(src, Bs + (ib < B_NI ? ib * 8 * BK : (NODUMP ? (ib - NW) * 8 * BK : BN * BK)));
        }
        kq += BK;
    };

    // ---- fragment addressing
    int pp0[FM];                                               // patch pixel of (output pixel, tap offset 0,0) per A fragment
#pragma unroll
    for (int i = 0; i < FM; ++i) {
        const int ml = wm * FM * 16 + i * 16 + li;
        const int img = ml / (rpi * TW), rem = ml - img * (rpi * TW);
        const int yl = rem / TW, x = rem - yl * TW;
        pp0[i] = (img * PRI + yl) * PW + x;
    }
    int bsw[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) bsw[ks] = ((ks * 4 + g) ^ (li & 7)) * 8;
    const int toff0 = 1 + p.conv.off, tmul = p.conv.kmul;      // patch row/col of tap (ky,kx): local + toff0 + k*tmul

    floatx4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = zero4f();

    issue_patch(cs_beg, 0);
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) issue_b(st);
    int rd = 0, wr = NST - 1;
    for (int cs = cs_beg; cs < cs_end; ++cs) {
        const half_t* P = Pb + ((cs - cs_beg) & 1) * PATCH;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // loads younger than ring stage kt: stages kt+1 .. kt+NST-2, plus the patch issued at tap 0 of this slab (an
            // iteration ago for tap 1, ..., NST-1 iterations ago for tap NST-1: it was issued AFTER stage kt+NST-1-tap)
            if (tap >= 1 && tap <= NST - 1) CLORA_WAIT_VMCNT((NST - 2) * B_IN + PA_IN);
            else CLORA_WAIT_VMCNT((NST - 2) * B_IN);
            CLORA_RAW_BARRIER();
            issue_b(wr);
            wr = (wr + 1 == NST) ? 0 : wr + 1;
            if (tap == 0) issue_patch(cs + 1, ((cs - cs_beg) & 1) ^ 1);      // consumed through tap 8 of the previous slab: free
            const half_t* Bs = Bring + rd * BST;
            rd = (rd + 1 == NST) ? 0 : rd + 1;
            const int ky = tap / 3, kx = tap - ky * 3;
            const int tapoff = (toff0 + ky * tmul) * PW + (toff0 + kx * tmul);
            half8 af[2][FM], bf[2][FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                const int pp = pp0[i] + tapoff;
                const half_t* q = P + pp * 64;
                const int k7 = pp & 7;
                af[0][i] = ld8(q + ((g ^ k7) * 8));
                af[1][i] = ld8(q + (((4 + g) ^ k7) * 8));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < FN; ++j) bf[ks][j] = ld8(Bs + (wn * FN * 16 + j * 16 + li) * BK + bsw[ks]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j) acc[i][j] = mfma16(af[ks][i], bf[ks][j], acc[i][j]);
        }
    }
    CLORA_WAIT_VMCNT(0);
    dma_epilogue<BM, BN, WM, WN, NT, SMEM>(p, acc, m0, n0, split, smem, t);
}

__global__ __launch_bounds__(256) void splitk_finish_kernel(GemmArgs p, int splits) {
    const size_t chunks = (size_t)p.M * (p.N / 8);
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < chunks; c += (size_t)gridDim.x * 256) {
        const int m = (int)(c / (p.N / 8));
        const int n = (int)(c - (size_t)m * (p.N / 8)) * 8;
        if (p.epi.c_lo) {                                // compensated trunk: the rounding remainder beside C
            half8 lo;
            st8(p.C + (size_t)m * p.ldc + n, finish_chunk8(p.partial, splits, p.M, p.N, p.epi, m, n, &lo));
            st8((half_t*)p.epi.c_lo + (size_t)m * p.ldc + n, lo);
        } else {
            st8(p.C + (size_t)m * p.ldc + n, finish_chunk8(p.partial, splits, p.M, p.N, p.epi, m, n));     // clora_epilogue.h
        }
    }
}

// ------------------------------------------------------------------------------------------------
// weight gradient of a trainable convolution: dW[N, K] += dY[M,N]^T . gather(X)[M,K]
// Reduction runs over the GEMM rows m, so both operands are staged TRANSPOSED in LDS
// (At[n][m], Bt[kcol][m]) and the M axis plays the MFMA k role.  One block = 64(n) x 64(kcol)
// output tile over one M chunk; fp32 atomics combine the M chunks.
struct WgradArgs {
    const half_t* dY;
    const half_t* X;
    float* dW;
    float* db;   // optional bias gradient: handled as one extra all-ones input column k == K
    int ldy, ldx, M, N, K;
    int oihw_ci;   // > 0: dW is the parameter itself, [N][oihw_ci][ks][ks] (OIHW); the padded input channels are dropped
    int m_per_block;
    int tiles_n, tiles_k;
    clora_conv_t conv;
};

// One block = 64(n) x 64(kcol) tile of dW over one chunk of GEMM rows m; the reduction index m is the MFMA k index.
// Both operands are k-strided in memory (dY [m][n], gather(X) [m][kcol]), so they are staged ROW-MAJOR by LDS-DMA
// (no staging registers, no transposing ds_write pass) as 16-column sub-tiles [64 m][16] and fetched with the gfx950
// transpose read ds_read_b64_tr_b16: a 16-lane group reads a [4 m][16 col] block (4 rows x 32 B = 128 contiguous
// bytes, conflict free) and every lane receives its column's 4 consecutive m -- two of them make one MFMA operand.
// 3-stage ring of 64-row stages (16 KB each), counted vmcnt waits, one barrier per stage, as in gemm_dma_kernel.
// The bias gradient rides along as an all-ones input column at kcol == K (fetched from a 16-byte "one page").
__device__ __attribute__((aligned(16))) const unsigned short g_clora_one16[8] = {0x3C00u, 0, 0, 0, 0, 0, 0, 0};

__global__ __launch_bounds__(256, 3) void conv_wgrad_kernel(WgradArgs p) {
    constexpr int BMR = 64, NST = 3;
    constexpr int SUB = BMR * 16;                 // halves per [64 m][16 col] sub-tile
    constexpr int STAGE = 8 * SUB;                // 4 dY sub-tiles + 4 X sub-tiles
    __shared__ __attribute__((aligned(16))) half_t smem[NST * STAGE];
    const int t = threadIdx.x;
    const int tile = blockIdx.x;
    const int tn = tile % p.tiles_n, tk = tile / p.tiles_n;
    const int n0 = tn * 64, k0 = tk * 64;
    const int mbeg = blockIdx.y * p.m_per_block;
    const int mend = (mbeg + p.m_per_block < p.M) ? mbeg + p.m_per_block : p.M;
    const int nst = (mend - mbeg + BMR - 1) / BMR;
    const bool conv = p.conv.enabled != 0;
    const half_t* zero_page = reinterpret_cast<const half_t*>(g_clora_zero16);
    const half_t* one_page = reinterpret_cast<const half_t*>(g_clora_one16);

    const int w = t >> 6, l = t & 63, g = l >> 4, li = l & 15;
    // ---- loader: wave w fills, per stage, rows (w&1)*32 .. +32 of sub-tiles {w>>1, 2+(w>>1)} of dY and of X;
    // lane -> (row l>>1, 8-column chunk l&1) so that the lane-linear DMA image is exactly [32 m][16 col]
    const int lm = (w & 1) * 32 + (l >> 1);
    const int c8 = (l & 1) * 8;
    int ncol[2], kcol[2], ky[2], kx[2], ci[2];
    bool n_ok[2], k_ok[2], k_one[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int sub = q * 2 + (w >> 1);
        ncol[q] = n0 + sub * 16 + c8;
        kcol[q] = k0 + sub * 16 + c8;
        n_ok[q] = ncol[q] < p.N;
        k_ok[q] = kcol[q] < p.K;
        k_one[q] = p.db != nullptr && kcol[q] == p.K;          // K % 8 == 0: the ones column starts a chunk
        ky[q] = 0; kx[q] = 0; ci[q] = kcol[q];
        if (conv && k_ok[q]) {
            const int tap = kcol[q] / p.conv.Cin;
            ci[q] = kcol[q] - tap * p.conv.Cin;
            ky[q] = tap / p.conv.ksize;
            kx[q] = tap - ky[q] * p.conv.ksize;
        }
    }
    int mrow = mbeg + lm;
    auto issue_stage = [&](int buf) {
        half_t* base = smem + buf * STAGE + (w & 1) * 512;     // second half of a sub-tile = rows 32..63 = +512 halves
        const bool mok = mrow < mend;
        size_t xrow = 0;
        int ty0 = 0, tx0 = 0;
        if (mok) {
            if (!conv) {
                xrow = (size_t)mrow * p.ldx;
            } else {
                const int hw = p.conv.Hout * p.conv.Wout;
                const int b = mrow / hw, rem = mrow - b * hw;
                const int yo = rem / p.conv.Wout, xo = rem - yo * p.conv.Wout;
                xrow = (size_t)b * p.conv.Hin * p.conv.Win;
                ty0 = yo * p.conv.mul + p.conv.off;
                tx0 = xo * p.conv.mul + p.conv.off;
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int sub = q * 2 + (w >> 1);
            const half_t* src = (mok && n_ok[q]) ? p.dY + (size_t)mrow * p.ldy + ncol[q] : zero_page;
            CLORA_GLDS16(src, base + sub * SUB);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int sub = q * 2 + (w >> 1);
            const half_t* src = zero_page;
            if (mok && k_one[q]) src = one_page;
            else if (mok && k_ok[q]) {
                if (!conv) {
                    src = p.X + xrow + kcol[q];
                } else {
                    const int ty = ty0 + ky[q] * p.conv.kmul, tx = tx0 + kx[q] * p.conv.kmul;
                    bool ok = ty >= 0 && ty < p.conv.lim_h && tx >= 0 && tx < p.conv.lim_w;
                    if (p.conv.need_even) ok = ok && (((ty | tx) & 1) == 0);
                    if (ok) src = p.X + (xrow + (size_t)(ty >> p.conv.shift) * p.conv.Win + (tx >> p.conv.shift)) * p.conv.Cin + ci[q];
                }
            }
            CLORA_GLDS16(src, base + (4 + sub) * SUB);
        }
        mrow += BMR;
    };

    const int wm = w >> 1, wn = w & 1;               // 2x2 waves, each 32(n) x 32(kcol)
    // transpose-read address of this lane inside a sub-tile: row g*8 + (li>>2) (+4 for the second half), cols (l&3)*4
    const int troff = (g * 8 + (li >> 2)) * 16 + (l & 3) * 4;
    floatx4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = zero4f();

#pragma unroll
    for (int st = 0; st < NST - 1; ++st) issue_stage(st);
    int rd = 0, wr = NST - 1;
    for (int it = 0; it < nst; ++it) {
        CLORA_WAIT_VMCNT((NST - 2) * 4);
        CLORA_RAW_BARRIER();
        issue_stage(wr);
        wr = (wr + 1 == NST) ? 0 : wr + 1;
        const half_t* Ys = smem + rd * STAGE;
        const half_t* Xs = Ys + 4 * SUB;
        rd = (rd + 1 == NST) ? 0 : rd + 1;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                        // two MFMA k-steps of 32 rows
            half8 af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const half_t* q = Ys + (wm * 2 + i) * SUB + ks * 512 + troff;
                const half4v lo = CLORA_DS_READ_TR16(q), hi = CLORA_DS_READ_TR16(q + 64);
                af[i] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const half_t* q = Xs + (wn * 2 + j) * SUB + ks * 512 + troff;
                const half4v lo = CLORA_DS_READ_TR16(q), hi = CLORA_DS_READ_TR16(q + 64);
                bf[j] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mfma16(af[i], bf[j], acc[i][j]);
        }
    }
    CLORA_WAIT_VMCNT(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wm * 32 + i * 16 + 4 * g + r;
                const int kk = k0 + wn * 32 + j * 16 + li;
                if (n < p.N && kk < p.K) {
                    if (p.oihw_ci > 0) {                       // (ky,kx,ci) gather order -> OIHW parameter layout
                        const int cin = p.conv.enabled ? p.conv.Cin : p.K;
                        const int tap = kk / cin, ci = kk - tap * cin;
                        const int taps = p.conv.enabled ? p.conv.ksize * p.conv.ksize : 1;
                        if (ci < p.oihw_ci) atomicAdd(p.dW + ((size_t)n * p.oihw_ci + ci) * taps + tap, acc[i][j][r]);
                    } else {
                        atomicAdd(p.dW + (size_t)n * p.K + kk, acc[i][j][r]);
                    }
                } else if (n < p.N && kk == p.K && p.db) atomicAdd(p.db + n, acc[i][j][r]);
            }
}

// ------------------------------------------------------------------------------------------------
// Weight gradient of the hint encoder's LARGE-map 3x3 convolutions (512^2 / 256^2 / 128^2, 32 or 64 input channels; round 6).
// conv_wgrad_kernel above gathers X once per 64-column k tile and re-reads dY for every one of them: at 512^2 x 32 channels that is
// 1.3 GB of L2 -> LDS traffic for two 67 MB tensors (96 us; the launch is L2-bandwidth bound at 8 % MFMA busy).  Here a block owns a
// column strip of TW output pixels and walks down its rows; per output row it stages, ONCE, the three input rows of the strip as a
// halo patch (3 x (TW + 2) pixels, all CIN channels) and the row's dY (TW x N) by LDS-DMA, double buffered, and forms ALL 9 x CIN
// columns of dW from them: dY crosses L2 -> LDS once, X three times (the rows above / below are re-staged for the next row).
//   * LDS planes of 16 channels: [pixel][16] halves (32 bytes per pixel, contiguous): the reduction index (pixels) is the MFMA k index,
//     both operands are fetched with ds_read_b64_tr_b16 exactly as in conv_wgrad_kernel; for tap (ky, kx) the B operand of k-step s is
//     the plane at pixel ky * PW + kx + 32 s -- consecutive output pixels are consecutive patch pixels, so a tap is an address offset;
//   * waves = (CIN / 16 input-channel blocks) x 2 output-channel halves, each wave: 9 taps x NFW fragments of accumulators;
//   * the bias gradient is one more MFMA per k-step against an all-ones operand (registers); fp32 atomics into the gather-ordered
//     staging buffer like conv_wgrad_kernel (same destination layout).
struct WgradPatchArgs {
    const half_t* dY;
    const half_t* X;
    float* dW;
    float* db;
    int N, K, H, W, nimg, nseg, rows_per_block, row_chunks;      // H, W: OUTPUT rows / columns per image
    int Hin, Win;
};

//   * STRIDE 2 (the downsamplers: 3x3, stride 2, zero row / column at the bottom / right only -- F.pad(0, 1, 0, 1)): output row yo reads
//     input rows 2 yo .. 2 yo + 2 and input columns 2 x0 .. 2 x0 + 2 TW; a patch row holds the EVEN columns (TW + 1 of them) followed by
//     the ODD ones (the DMA's per-lane source address de-interleaves for free), so that tap kx is again an address offset:
//     kx = 0 -> even[m], kx = 1 -> odd[m], kx = 2 -> even[m + 1].
template <int CIN, int NFW, int TW, int STRIDE = 1>
__global__ __launch_bounds__(CIN / 16 * 128) void conv_wgrad_patch_kernel(WgradPatchArgs p) {
    constexpr int WC = CIN / 16, NW = WC * 2, NP = 2 * NFW;
    constexpr int PW = STRIDE == 1 ? TW + 2 : 2 * TW + 1, PPIX = 3 * PW;
    constexpr int P_NI = (PPIX * 2 + 63) / 64;                   // DMA wave-instructions per patch plane (64 chunks of 16 bytes each)
    constexpr int PLANE = P_NI * 512;                            // halves
    constexpr int Y_NI = TW * 2 / 64, YPLANE = TW * 16;
    constexpr int BUF = WC * PLANE + NP * YPLANE;
    constexpr int NI = WC * P_NI + NP * Y_NI, NIW = (NI + NW - 1) / NW;
    constexpr int KS = TW / 32;
    static_assert((2 * BUF + 512) * 2 <= 160 * 1024, "LDS per workgroup");
    __shared__ __attribute__((aligned(16))) half_t smem[2 * BUF + 512];      // two buffers + one dump slot for surplus DMA instructions
    const int t = threadIdx.x, w = t >> 6, l = t & 63, g = l >> 4, li = l & 15;
    const int cb = w >> 1, wn = w & 1;                           // this wave's input-channel block and output-channel half
    const half_t* zero_page = reinterpret_cast<const half_t*>(g_clora_zero16);

    // ---- which strip: block -> (image, column segment, row chunk)
    int bid = blockIdx.x;
    const int rc = bid % p.row_chunks; bid /= p.row_chunks;
    const int seg = bid % p.nseg, img = bid / p.nseg;
    const int x0 = seg * TW, y_beg = rc * p.rows_per_block;
    const int y_end = (y_beg + p.rows_per_block < p.H) ? y_beg + p.rows_per_block : p.H;

    // ---- loader: DMA instruction i = w + NW * j of an output row; per lane: kind, source offset relative to the row, destination
    int kind[NIW], soff[NIW], doff[NIW];                         // kind: -1 zero page always, 0..2 patch row, 3 dY
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
        const int i = w + NW * j;
        kind[j] = -1; soff[j] = 0; doff[j] = 2 * BUF;            // surplus instruction: zero page into the dump slot
        if (i < WC * P_NI) {
            const int pl = i / P_NI, pi = i - pl * P_NI;
            const int c = pi * 64 + l, pp = c >> 1, h = c & 1;
            doff[j] = pl * PLANE + pi * 512;
            if (pp < PPIX) {
                const int pr = pp / PW, pc = pp - pr * PW;
                if constexpr (STRIDE == 1) {
                    const int x = x0 - 1 + pc;
                    if (x >= 0 && x < p.W) { kind[j] = pr; soff[j] = ((pr - 1) * p.W + (pc - 1)) * CIN + pl * 16 + h * 8; }
                } else {
                    const int dx = pc <= TW ? 2 * pc : 2 * (pc - TW - 1) + 1;       // input column relative to 2 x0
                    if (2 * x0 + dx < p.Win) { kind[j] = pr; soff[j] = (pr * p.Win + dx) * CIN + pl * 16 + h * 8; }
                }
            }
        } else if (i < NI) {
            const int q = i - WC * P_NI, nb = q / Y_NI, yi = q - nb * Y_NI;
            const int c = yi * 64 + l, m = c >> 1, h = c & 1;
            doff[j] = WC * PLANE + nb * YPLANE + yi * 512;
            if (nb * 16 + h * 8 < p.N) { kind[j] = 3; soff[j] = m * p.N + nb * 16 + h * 8; }
        }
    }
    auto issue_row = [&](int y, int buf) {
        half_t* dst = smem + buf * BUF;
        const bool live = y < y_end;
        const size_t row = ((size_t)img * p.H + (live ? y : y_beg)) * p.W + x0;       // first output pixel of the row segment
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const half_t* src = zero_page;
            if (live && kind[j] >= 0) {
                if (kind[j] == 3) src = p.dY + row * p.N + soff[j];
                else if constexpr (STRIDE == 1) {
                    const int yy = y - 1 + kind[j];
                    if (yy >= 0 && yy < p.H) src = p.X + (ptrdiff_t)row * CIN + soff[j];
                } else {
                    if (2 * y + kind[j] < p.Hin) src = p.X + (((size_t)img * p.Hin + 2 * y) * p.Win + 2 * x0) * CIN + soff[j];
                }
            }
            half_t* d = (doff[j] == 2 * BUF) ? smem + 2 * BUF : dst + doff[j];
            CLORA_GLDS16(src, d);
        }
    };

    // ---- fragments: transpose reads of [4 pixels][16 channels] blocks (32 bytes per pixel), as in conv_wgrad_kernel
    const int troff = (g * 8 + (li >> 2)) * 16 + (l & 3) * 4;
    floatx4 acc[9][NFW], accb[NFW];
#pragma unroll
    for (int f = 0; f < NFW; ++f) {
        accb[f] = zero4f();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) acc[tap][f] = zero4f();
    }
    const half8 ones = {(half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f, (half_t)1.f};

    issue_row(y_beg, 0);
    for (int y = y_beg; y < y_end; ++y) {
        const int buf = (y - y_beg) & 1;
        issue_row(y + 1, buf ^ 1);                               // (past the strip: zero-page instructions, the count stays the same)
        CLORA_WAIT_VMCNT(NIW);                                   // everything but the row just issued has landed: row y
        CLORA_RAW_BARRIER();
        const half_t* P = smem + buf * BUF + cb * PLANE;
        const half_t* Y = smem + buf * BUF + WC * PLANE + (wn * NFW) * YPLANE;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            half8 af[NFW];
#pragma unroll
            for (int f = 0; f < NFW; ++f) {
                const half_t* q = Y + f * YPLANE + s * 512 + troff;
                const half4v lo = CLORA_DS_READ_TR16(q), hi = CLORA_DS_READ_TR16(q + 64);
                af[f] = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
            if (cb == 0) {                                       // wave-uniform: the bias gradient rides with channel block 0
#pragma unroll
                for (int f = 0; f < NFW; ++f) accb[f] = mfma16(af[f], ones, accb[f]);
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int ky = tap / 3, kx = tap - ky * 3;
                const int koff = STRIDE == 1 ? kx : (kx == 1 ? TW + 1 : kx / 2);
                const half_t* q = P + (ky * PW + koff + s * 32) * 16 + troff;
                const half4v lo = CLORA_DS_READ_TR16(q), hi = CLORA_DS_READ_TR16(q + 64);
                const half8 bf = half8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
                for (int f = 0; f < NFW; ++f) acc[tap][f] = mfma16(af[f], bf, acc[tap][f]);
            }
        }
        CLORA_RAW_BARRIER();                                     // row y consumed by every wave: its buffer is the target of the next issue
    }
    CLORA_WAIT_VMCNT(0);
#pragma unroll
    for (int f = 0; f < NFW; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = (wn * NFW + f) * 16 + 4 * g + r;
            if (n < p.N) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) atomicAdd(p.dW + (size_t)n * p.K + tap * CIN + cb * 16 + li, acc[tap][f][r]);
                if (cb == 0 && li == 0 && p.db) atomicAdd(p.db + n, accb[f][r]);
            }
        }
}

template <int CIN, int NFW, int TW, int STRIDE = 1>
int launch_wgrad_patch(WgradPatchArgs& a, hipStream_t s) {
    a.nseg = a.W / TW;
    const long strips = (long)a.nimg * a.nseg;
    // Every block adds its whole N x (K + 1) result with fp32 atomics, ~6 ns each on MI355X (block-count sweep, profiles/r06_wgrad_patch_sweep.txt:
    // 512^2 32->32 takes 76 / 74 / 103 / 167 us at 256 / 512 / 1024 / 2048 blocks): as many blocks as keep the launch under ~4.7 M atomics,
    // between 64 and 512 (A/B: "wgrad_patch" >= 64 forces the count)
    long target = 4700000L / ((long)a.N * (a.K + 1));
    target = target < 64 ? 64 : (target > 512 ? 512 : target);
    if (clora_option(CLORA_OPT_WGRAD_PATCH) >= 64) target = clora_option(CLORA_OPT_WGRAD_PATCH);
    long rpb = ((long)a.H * strips + target - 1) / target;
    if (rpb < 4) rpb = 4;
    if (rpb > a.H) rpb = a.H;
    a.rows_per_block = (int)rpb;
    a.row_chunks = clora_cdiv(a.H, rpb);
    hipLaunchKernelGGL((conv_wgrad_patch_kernel<CIN, NFW, TW, STRIDE>), dim3((unsigned)(strips * a.row_chunks)), dim3(CIN / 16 * 128), 0, s, a);
    return clora_check_launch();
}

// ------------------------------------------------------------------------------------------------
// Forward / dgrad of the same large-map convolutions (tile_cfg 61, round 6): 3x3, stride 1, pad 1, CIN = 32 or 64 input channels, at
// most 64 output channels, K order (tap, channel) (kchunk 0) -- the hint encoder's 512^2 / 256^2 convolutions and their dgrads.  The
// implicit GEMM gathers every input pixel nine times through L2 -> LDS (600 MB for two 67 MB tensors at 512^2 x 32: 95 us); the
// 64-channel-slab patch kernel cannot take 32 channels.  Same strip walk as conv_wgrad_patch_kernel: a block owns a TW-pixel column
// strip, stages per output row the three input rows as a halo patch of 16-channel planes (double buffered LDS-DMA) and multiplies it
// with the WHOLE weight operand, which lives in registers (9 taps x CIN/32 k-steps x N/16 fragments: 72-144 VGPRs, loaded once).
// D = W . patch^T: rows = output channels, columns = pixels, so a lane holds four consecutive channels of one pixel (8-byte stores);
// the B operand of pixel group mf / tap (ky, kx) / k-step ks is a ds_read_b128 at plane 2 ks + (g >> 1), pixel (ky' PW + kx' + 16 mf + li),
// half g & 1 -- conflict free for the hardware's ds_read_b128 lane groups at this 32-byte pixel pitch.  Epilogue: bias only.
struct StripArgs {
    const half_t* X;
    const half_t* Wt;        // [N][9 * CIN], column = tap * CIN + channel
    half_t* C;
    const float* bias;
    int N, ldc, H, W, nimg, nseg, rows_per_block, row_chunks;
    int off, kmul;           // input offset of tap k: off + k * kmul  (forward -1 / +1, dgrad +1 / -1)
};

template <int CIN, int NF, int TW>
__global__ __launch_bounds__(256) void conv3x3_strip_kernel(StripArgs p) {
    constexpr int WC = CIN / 16, NW = 4, KSTEPS = CIN / 32;
    constexpr int PW = TW + 2, PPIX = 3 * PW;
    constexpr int P_NI = (PPIX * 2 + 63) / 64, PLANE = P_NI * 512;
    constexpr int BUF = WC * PLANE;
    constexpr int NI = WC * P_NI, NIW = (NI + NW - 1) / NW;
    constexpr int MF = TW / 16 / NW;                             // 16-pixel groups per wave
    static_assert(MF >= 1 && (2 * BUF + 512) * 2 <= 160 * 1024, "strip geometry");
    __shared__ __attribute__((aligned(16))) half_t smem[2 * BUF + 512];
    const int t = threadIdx.x, w = t >> 6, l = t & 63, g = l >> 4, li = l & 15;
    const half_t* zero_page = reinterpret_cast<const half_t*>(g_clora_zero16);

    int bid = blockIdx.x;
    const int rc = bid % p.row_chunks; bid /= p.row_chunks;
    const int seg = bid % p.nseg, img = bid / p.nseg;
    const int x0 = seg * TW, y_beg = rc * p.rows_per_block;
    const int y_end = (y_beg + p.rows_per_block < p.H) ? y_beg + p.rows_per_block : p.H;

    // ---- the weight operand: A fragments, lane (row n = li, k-group g) holds 8 consecutive channels of one tap
    half8 wf[9][KSTEPS][NF];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int n = f * 16 + li;
                wf[tap][ks][f] = (n < p.N) ? ld8(p.Wt + (size_t)n * (9 * CIN) + tap * CIN + ks * 32 + g * 8) : zero8();
            }
    float bias[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = f * 16 + 4 * g + r;
            bias[f][r] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
        }

    // ---- patch loader (conv_wgrad_patch_kernel's, stride 1)
    int kind[NIW], soff[NIW], doff[NIW];
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
        const int i = w + NW * j;
        kind[j] = -1; soff[j] = 0; doff[j] = 2 * BUF;
        if (i < NI) {
            const int pl = i / P_NI, pi = i - pl * P_NI;
            const int c = pi * 64 + l, pp = c >> 1, h = c & 1;
            doff[j] = pl * PLANE + pi * 512;
            if (pp < PPIX) {
                const int pr = pp / PW, pc = pp - pr * PW;
                const int x = x0 - 1 + pc;
                if (x >= 0 && x < p.W) { kind[j] = pr; soff[j] = ((pr - 1) * p.W + (pc - 1)) * CIN + pl * 16 + h * 8; }
            }
        }
    }
    auto issue_row = [&](int y, int buf) {
        half_t* dst = smem + buf * BUF;
        const bool live = y < y_end;
        const size_t row = ((size_t)img * p.H + (live ? y : y_beg)) * p.W + x0;
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const half_t* src = zero_page;
            if (live && kind[j] >= 0) {
                const int yy = y - 1 + kind[j];
                if (yy >= 0 && yy < p.H) src = p.X + (ptrdiff_t)row * CIN + soff[j];
            }
            half_t* d = (doff[j] == 2 * BUF) ? smem + 2 * BUF : dst + doff[j];
            CLORA_GLDS16(src, d);
        }
    };

    const int rd = (g >> 1) * PLANE + (g & 1) * 8;               // this lane's plane (of a 32-channel k-step) and half
    issue_row(y_beg, 0);
    for (int y = y_beg; y < y_end; ++y) {
        const int buf = (y - y_beg) & 1;
        issue_row(y + 1, buf ^ 1);
        CLORA_WAIT_VMCNT(NIW);
        CLORA_RAW_BARRIER();
        const half_t* P = smem + buf * BUF + rd;
        floatx4 acc[NF][MF];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int m = 0; m < MF; ++m) acc[f][m] = floatx4{bias[f][0], bias[f][1], bias[f][2], bias[f][3]};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - ky * 3;
            const int pp0 = (1 + p.off + ky * p.kmul) * PW + (1 + p.off + kx * p.kmul) + (w * MF) * 16 + li;
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                half8 bf[MF];
#pragma unroll
                for (int m = 0; m < MF; ++m) bf[m] = ld8(P + ks * 2 * PLANE + (pp0 + m * 16) * 16);
#pragma unroll
                for (int m = 0; m < MF; ++m)
#pragma unroll
                    for (int f = 0; f < NF; ++f) acc[f][m] = mfma16(wf[tap][ks][f], bf[m], acc[f][m]);
            }
        }
        // rows of D = output channels 4 g + r of fragment f, column = pixel: four consecutive channels of one pixel per lane
        const size_t orow = ((size_t)img * p.H + y) * p.W + x0 + (w * MF) * 16 + li;
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int n = f * 16 + 4 * g;
                if (n < p.N) {
                    half4v o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)acc[f][m][r];
                    st4(p.C + (orow + m * 16) * p.ldc + n, o);
                }
            }
        CLORA_RAW_BARRIER();
    }
    CLORA_WAIT_VMCNT(0);
}

// The 8-channel case (the hint encoder's conv_in: 3 image channels zero-padded to 8, 16 bytes per pixel; 85 us on the implicit GEMM at
// 512^2 for a 16 MB input and a 67 MB output): K = 9 taps x 8 channels = 72 = three MFMA k-steps of FOUR taps each (the last one holds
// tap 8 and three zero taps).  Lane group g of k-step ks owns tap 4 ks + g: the B operand is one ds_read_b128 at THAT tap's patch offset
// (a per-lane constant), the A operand the matching 8 weights (zero for taps >= 9).  One 8-channel plane, one DMA chunk per pixel.
template <int NF, int TW>
__global__ __launch_bounds__(256) void conv3x3_strip8_kernel(StripArgs p) {
    constexpr int NW = 4, CIN = 8, KT = 3;
    constexpr int PW = TW + 2, PPIX = 3 * PW;
    constexpr int P_NI = (PPIX + 63) / 64, BUF = P_NI * 512;     // 64 pixels (chunks of 16 bytes) per DMA wave-instruction
    constexpr int NIW = (P_NI + NW - 1) / NW;
    constexpr int MF = TW / 16 / NW;
    __shared__ __attribute__((aligned(16))) half_t smem[2 * BUF + 512];
    const int t = threadIdx.x, w = t >> 6, l = t & 63, g = l >> 4, li = l & 15;
    const half_t* zero_page = reinterpret_cast<const half_t*>(g_clora_zero16);

    int bid = blockIdx.x;
    const int rc = bid % p.row_chunks; bid /= p.row_chunks;
    const int seg = bid % p.nseg, img = bid / p.nseg;
    const int x0 = seg * TW, y_beg = rc * p.rows_per_block;
    const int y_end = (y_beg + p.rows_per_block < p.H) ? y_beg + p.rows_per_block : p.H;

    half8 wf[KT][NF];
    int toff[KT];                                                // patch offset (pixels) of this lane group's tap in k-step ks
#pragma unroll
    for (int ks = 0; ks < KT; ++ks) {
        const int tap = ks * 4 + g;
        const int ky = tap / 3, kx = tap - ky * 3;
        toff[ks] = tap < 9 ? (1 + p.off + ky * p.kmul) * PW + (1 + p.off + kx * p.kmul) : 0;
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int n = f * 16 + li;
            wf[ks][f] = (tap < 9 && n < p.N) ? ld8(p.Wt + (size_t)n * (9 * CIN) + tap * CIN) : zero8();
        }
    }
    float bias[NF][4];
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = f * 16 + 4 * g + r;
            bias[f][r] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
        }

    int kind[NIW], soff[NIW], doff[NIW];
#pragma unroll
    for (int j = 0; j < NIW; ++j) {
        const int i = w + NW * j;
        kind[j] = -1; soff[j] = 0; doff[j] = 2 * BUF;
        if (i < P_NI) {
            const int pp = i * 64 + l;
            doff[j] = i * 512;
            if (pp < PPIX) {
                const int pr = pp / PW, pc = pp - pr * PW;
                const int x = x0 - 1 + pc;
                if (x >= 0 && x < p.W) { kind[j] = pr; soff[j] = ((pr - 1) * p.W + (pc - 1)) * CIN; }
            }
        }
    }
    auto issue_row = [&](int y, int buf) {
        half_t* dst = smem + buf * BUF;
        const bool live = y < y_end;
        const size_t row = ((size_t)img * p.H + (live ? y : y_beg)) * p.W + x0;
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const half_t* src = zero_page;
            if (live && kind[j] >= 0) {
                const int yy = y - 1 + kind[j];
                if (yy >= 0 && yy < p.H) src = p.X + (ptrdiff_t)row * CIN + soff[j];
            }
            half_t* d = (doff[j] == 2 * BUF) ? smem + 2 * BUF : dst + doff[j];
            CLORA_GLDS16(src, d);
        }
    };

    issue_row(y_beg, 0);
    for (int y = y_beg; y < y_end; ++y) {
        const int buf = (y - y_beg) & 1;
        issue_row(y + 1, buf ^ 1);
        CLORA_WAIT_VMCNT(NIW);
        CLORA_RAW_BARRIER();
        const half_t* P = smem + buf * BUF;
        floatx4 acc[NF][MF];
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int m = 0; m < MF; ++m) acc[f][m] = floatx4{bias[f][0], bias[f][1], bias[f][2], bias[f][3]};
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            half8 bf[MF];
#pragma unroll
            for (int m = 0; m < MF; ++m) bf[m] = ld8(P + (toff[ks] + (w * MF + m) * 16 + li) * CIN);
#pragma unroll
            for (int m = 0; m < MF; ++m)
#pragma unroll
                for (int f = 0; f < NF; ++f) acc[f][m] = mfma16(wf[ks][f], bf[m], acc[f][m]);
        }
        const size_t orow = ((size_t)img * p.H + y) * p.W + x0 + (w * MF) * 16 + li;
#pragma unroll
        for (int m = 0; m < MF; ++m)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                const int n = f * 16 + 4 * g;
                if (n < p.N) {
                    half4v o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)acc[f][m][r];
                    st4(p.C + (orow + m * 16) * p.ldc + n, o);
                }
            }
        CLORA_RAW_BARRIER();
    }
    CLORA_WAIT_VMCNT(0);
}

template <int CIN, int NF, int TW>
int launch_strip(StripArgs& a, hipStream_t s) {
    a.nseg = a.W / TW;
    const long strips = (long)a.nimg * a.nseg;
    const long target = clora_option(CLORA_OPT_STRIP_BLOCKS);    // blocks aimed at ("strip_blocks")
    long rpb = ((long)a.H * strips + target - 1) / target;
    if (rpb < 8) rpb = 8;                                        // (a block first loads the whole weight operand: at least eight rows of work;
    if (rpb > a.H) rpb = a.H;                                    //  sweep on MI355X, profiles/r06_conv_strip_bench.txt: 512^2 34 us at 512 blocks
    a.rows_per_block = (int)rpb;                                 //  against 48 / 39 / 45 at 256 / 1024 / 2048; 256^2 23 us at 8 rows against 29 at 4)
    a.row_chunks = clora_cdiv(a.H, rpb);
    hipLaunchKernelGGL((conv3x3_strip_kernel<CIN, NF, TW>), dim3((unsigned)(strips * a.row_chunks)), dim3(256), 0, s, a);
    return clora_check_launch();
}

template <int NF, int TW>
int launch_strip8(StripArgs& a, hipStream_t s) {
    a.nseg = a.W / TW;
    const long strips = (long)a.nimg * a.nseg;
    const long target = clora_option(CLORA_OPT_STRIP_BLOCKS);
    long rpb = ((long)a.H * strips + target - 1) / target;
    if (rpb < 8) rpb = 8;
    if (rpb > a.H) rpb = a.H;
    a.rows_per_block = (int)rpb;
    a.row_chunks = clora_cdiv(a.H, rpb);
    hipLaunchKernelGGL((conv3x3_strip8_kernel<NF, TW>), dim3((unsigned)(strips * a.row_chunks)), dim3(256), 0, s, a);
    return clora_check_launch();
}

// which instantiation (CIN * 100 + NF) takes this convolution, or 0 (everything else stays on the implicit GEMM / patch kernels)
int strip_plan(int M, int N, const clora_conv_t& c) {
    if (!c.enabled || c.ksize != 3 || c.mul != 1 || c.shift != 0 || c.need_even != 0 || c.kchunk != 0) return 0;
    if (!((c.kmul == 1 && c.off == -1) || (c.kmul == -1 && c.off == 1))) return 0;
    if (c.Hout != c.Hin || c.Wout != c.Win || c.lim_h != c.Hin || c.lim_w != c.Win || c.Hout <= 0 || (c.Wout % 128)) return 0;
    if ((M % (c.Hout * c.Wout)) || (long)M * 64 >= (1L << 31) || (N & 3)) return 0;
    if (c.Cin == 8 && N <= 32) return 802;
    if (c.Cin == 8 && N <= 64) return 804;
    if (c.Cin == 32 && N <= 32) return 3202;
    if (c.Cin == 32 && N <= 64) return 3204;
    if (c.Cin == 64 && N <= 32) return 6402;
    return 0;
}

// fp32 master weight [Co][Ci][ks][ks] of a trainable conv -> the two fp16 GEMM operands of this step in one launch:
// fwd [Co][ks*ks][Cip] (gather order ky,kx,ci; channels zero-padded to Cip) and, optionally, dgrad [Cip][ks*ks][Cop].
__global__ __launch_bounds__(256) void conv_weight_pack_kernel(const float* w, int Co, int Ci, int taps, int Cip, int Cop,
                                                               half_t* fwd, half_t* dgrad) {
    const int nf = Co * taps * Cip, nd = dgrad ? Cip * taps * Cop : 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nf + nd; i += gridDim.x * 256) {
        if (i < nf) {
            const int ci = i % Cip, tap = (i / Cip) % taps, co = i / (Cip * taps);
            fwd[i] = ci < Ci ? (half_t)w[((size_t)co * Ci + ci) * taps + tap] : (half_t)0.f;
        } else {
            const int q = i - nf;
            const int co = q % Cop, tap = (q / Cop) % taps, ci = q / (Cop * taps);
            dgrad[q] = (ci < Ci && co < Co) ? (half_t)w[((size_t)co * Ci + ci) * taps + tap] : (half_t)0.f;
        }
    }
}

// which gather the DMA main loop is instantiated for (see gemm_dma_kernel)
int conv_mode(const GemmArgs& a, int bk) {
    const clora_conv_t& c = a.conv;
    if (!c.enabled) return 0;
    const int kcs = c.kchunk > 0 ? c.kchunk : c.Cin;
    const bool fast = c.ksize == 3 && c.shift == 0 && c.need_even == 0 && (c.kmul == 1 || c.kmul == -1) &&
                      (kcs % bk) == 0 && (a.k_per_split % bk) == 0 && c.lim_h == c.Hin && c.lim_w == c.Win;
    return fast ? 2 : 1;
}

// Which XCD fetches what.  Workgroup L runs on XCD L % 8 (tools/probes/xcc_map_probe.hip) and the kernels hand every XCD a
// contiguous range of logical tiles.  A panel of A rows / B rows is fetched through the fabric once per XCD that touches it:
// sharers on one XCD are served by one fetch, sharers on different XCDs are not, and the MALL multiplies nothing
// (profiles/r02_l2_share_probe.txt).  m-major ranges (n fastest) fetch A about once and B up to eight times -- right for the
// 64x64 / 32x32 levels (activations 5-30 MB, weights 2-7 MB); at 16x16 / 8x8 the weights are 29-59 MB against 1-3 MB of
// activations and n-major ranges (m fastest) are the cheaper assignment.  The model counts, per XCD and split, the distinct
// panels of its range under either order.   "tile_order": 0 m-major always, 1 n-major always
// (tests), 2 pick by the model, 3 (default since round 5: live PMC traffic 1.69x -> 1.59x of the algorithmic bytes, bit-identical
// outputs) = 2 plus a per-XCD rectangle of tiles where whole divisors exist.
int g_opts[CLORA_OPT_COUNT] = {3, 1, 0, 0, 512, 1, 1, 0, 1, 1, 4, 1, 512, 2};    // tile_order, ln_rows, attn_fwd_waves (0 = auto), attn_bwd_waves (0 = auto), gn_blocks, epi_two_phase, lora_down_mode, gn_unroll, epi_hoist, gn_resident, defer_max_rows, wgrad_patch, strip_blocks, gn_team
int tile_order_mode() { return g_opts[CLORA_OPT_TILE_ORDER]; }

double fabric_model_bytes(int tiles_m, int tiles_n, int splits, double a_panel, double b_panel, bool n_major) {
    const int tiles = tiles_m * tiles_n, nwg = tiles * splits;
    const int inner = n_major ? tiles_m : tiles_n;
    const int xq = nwg >> 3, xr = nwg & 7;
    double bytes = 0.0;
    int beg = 0;
    for (int x = 0; x < 8; ++x) {
        const int end = beg + xq + (x < xr ? 1 : 0);
        for (int sp = beg / tiles; sp * tiles < end && sp < splits; ++sp) {
            const int t0 = (beg > sp * tiles ? beg : sp * tiles) - sp * tiles, t1 = (end < (sp + 1) * tiles ? end : (sp + 1) * tiles) - sp * tiles;
            if (t1 <= t0) continue;
            const int outer = (t1 - 1) / inner - t0 / inner + 1;
            const int inn = (t1 - t0 < inner) ? t1 - t0 : inner;
            bytes += n_major ? outer * b_panel + inn * a_panel : outer * a_panel + inn * b_panel;
        }
        beg = end;
    }
    return bytes;
}

void pick_tile_order(GemmArgs& a, int BM, int BN, int splits, bool patch) {
    a.tiles_m = clora_cdiv(a.M, BM);
    a.n_major = 0;
    a.xg_s = a.xg_m = a.xg_n = 0;
    const int mode = tile_order_mode();
    if (mode == 0 || a.tiles_m == 1 || a.tiles_n == 1) return;
    if (mode == 1) { a.n_major = 1; return; }
    // bytes of one panel per split: the implicit-GEMM conv re-reads its input rows per tap from L2, the fabric sees them once
    // (x ~1.5 halo); the patch kernel counts k_per_split in 64-channel slabs
    const double kper = patch ? (double)a.k_per_split * 576 : (double)a.k_per_split;
    const double a_panel = a.conv.enabled && a.conv.ksize == 3 ? BM * (kper / 9.0) * 2.0 * 1.5 : BM * kper * 2.0;
    const double b_panel = BN * kper * 2.0;
    const double m_cost = fabric_model_bytes(a.tiles_m, a.tiles_n, splits, a_panel, b_panel, false);
    const double n_cost = fabric_model_bytes(a.tiles_m, a.tiles_n, splits, a_panel, b_panel, true);
    a.n_major = n_cost < 0.85 * m_cost;
    if (mode == 3) {
        // 2-D / 3-D assignment: XCD x gets a rectangle of (splits / gs) x (tiles_m / gm) x (tiles_n / gn) tiles, gs * gm * gn = 8, whole
        // divisors only (every XCD then owns exactly nwg / 8 workgroups, which is what the hardware's round-robin hands it).  A panel is
        // fetched once per XCD that touches it: 8 * rs * (rm * a_panel + rn * b_panel) bytes.  Taken when it beats the better 1-D order by 5 %.
        double best = 0.95 * (a.n_major ? n_cost : m_cost);
        for (int gs = 1; gs <= 8; gs *= 2)
            for (int gm = 1; gs * gm <= 8; gm *= 2) {
                const int gn = 8 / (gs * gm);
                if (splits % gs || a.tiles_m % gm || a.tiles_n % gn) continue;
                const double c = 8.0 * (splits / gs) * ((a.tiles_m / gm) * a_panel + (a.tiles_n / gn) * b_panel);
                if (c < best) { best = c; a.xg_s = gs; a.xg_m = gm; a.xg_n = gn; }
            }
        if (a.xg_m) a.n_major = 0;
    }
}

template <int BM, int BN, int WM, int WN, int NST = 3, int BK = 32, int FLAGS = 0>
int launch_gemm(GemmArgs& a, int splits, hipStream_t s, bool dma) {
    a.tiles_n = clora_cdiv(a.N, BN);
    const int tiles_m = clora_cdiv(a.M, BM);
    pick_tile_order(a, BM, BN, splits, false);
    if (!dma) { a.n_major = 0; a.xg_s = a.xg_m = a.xg_n = 0; }      // the v1 loop decodes blockIdx directly
    const dim3 grid(tiles_m * a.tiles_n, splits);
    if (dma && a.epi.lora_dpack) {
        if constexpr (BK == 64 && ((WM * WN == 8 && BN == 320 && BM <= 128) || (WM * WN == 4 && BM * BN <= 128 * 128 && NST <= 3))) {
            hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, WM, WN, NST, 0, BK, FLAGS, 8>), grid, dim3(WM * WN * 64), 0, s, a);
            return clora_check_launch();
        } else {
            return CLORA_ERR_ARG;
        }
    }
    if (dma) {
        const int cm = conv_mode(a, BK);
        if (cm == 0) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, WM, WN, NST, 0, BK, FLAGS>), grid, dim3(WM * WN * 64), 0, s, a);
        else if (cm == 1) hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, WM, WN, NST, 1, BK, FLAGS>), grid, dim3(WM * WN * 64), 0, s, a);
        else hipLaunchKernelGGL((gemm_dma_kernel<BM, BN, WM, WN, NST, 2, BK, FLAGS>), grid, dim3(WM * WN * 64), 0, s, a);
    }
    else if constexpr (BM * BN <= 128 * 128 && WM * WN == 4)      // the register-staged v1 loop only exists for the round-1 tiles (A/B runs)
        hipLaunchKernelGGL((gemm_kernel<BM, BN, WM, WN>), dim3(tiles_m * a.tiles_n, splits), dim3(256), 0, s, a);
    else return CLORA_ERR_ARG;
    return clora_check_launch();
}

int launch_gemm_8p(GemmArgs& a, int splits, hipStream_t s) {
    a.tiles_n = clora_cdiv(a.N, 256);
    pick_tile_order(a, 256, 256, splits, false);
    const dim3 grid(clora_cdiv(a.M, 256) * a.tiles_n, splits);
    hipLaunchKernelGGL((gemm_8p_kernel<0>), grid, dim3(512), 0, s, a);
    return clora_check_launch();
}

// conv3x3_patch_kernel: what it can take (everything else stays on gemm_dma_kernel)
bool patch_eligible(const GemmArgs& a, int bm, int maxpp = 0) {
    const clora_conv_t& c = a.conv;
    if (!c.enabled || c.ksize != 3 || c.mul != 1 || c.need_even != 0 || c.kchunk != 64 || (c.Cin % 64)) return false;
    if (c.shift == 0) {
        if (!((c.kmul == 1 && c.off == -1) || (c.kmul == -1 && c.off == 1))) return false;    // forward pad 1 / its dgrad
        if (c.Hout != c.Hin || c.Wout != c.Win || c.lim_h != c.Hin || c.lim_w != c.Win) return false;
    } else {                                                                                   // forward of conv(nearest-2x(x)), pad 1
        if (c.shift != 1 || c.kmul != 1 || c.off != -1 || c.Hout != 2 * c.Hin || c.Wout != 2 * c.Win || c.lim_h != c.Hout || c.lim_w != c.Wout)
            return false;
    }
    const int W = c.Wout, HW = c.Hout * c.Wout;
    if (W <= 0 || (a.M % HW) || (long)a.M * c.Cin >= (1L << 31)) return false;
    if ((bm % W) && !(maxpp > 0 && (W % bm) == 0)) return false;                               // whole rows, or (wide patch) row segments
    if (!((HW % bm) == 0 || (bm % HW) == 0)) return false;                                     // ... of one image, or whole images
    const int nimg = HW >= bm ? 1 : bm / HW, rpi = HW >= bm ? (W >= bm ? 1 : bm / W) : c.Hout, tw = W > bm ? bm : W;
    return nimg * (rpi + 2) * (tw + 2) <= (maxpp > 0 ? maxpp : (bm == 256 ? 400 : 288));
}

template <int BM, int BN, int WM, int WN, int NST, int MAXPP_ = 0>
int launch_patch(GemmArgs& a, int splits, hipStream_t s) {
    a.tiles_n = clora_cdiv(a.N, BN);
    pick_tile_order(a, BM, BN, splits, true);
    const dim3 grid(clora_cdiv(a.M, BM) * a.tiles_n, splits);
    hipLaunchKernelGGL((conv3x3_patch_kernel<BM, BN, WM, WN, NST, MAXPP_>), grid, dim3(WM * WN * 64), 0, s, a);
    return clora_check_launch();
}

// split-K launches: the finish pass (slabs -> epilogue -> C), or -- clora_epilogue_t.defer -- its description for the consumer of C
int finish_or_defer(GemmArgs& a, int splits, clora_deferred_t* defer, hipStream_t s) {
    if (defer) {
        defer->partial = a.partial; defer->splits = splits; defer->M = a.M; defer->N = a.N; defer->C = (clora_half*)a.C; defer->ldc = a.ldc;
        defer->epi = a.epi;
        defer->epi.defer = nullptr;
        return CLORA_OK;
    }
    const size_t chunks = (size_t)a.M * (a.N / 8);
    int blocks = (int)((chunks + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(splitk_finish_kernel, dim3(blocks), dim3(256), 0, s, a, splits);
    return clora_check_launch();
}


}  // namespace

// ---- launch planning --------------------------------------------------------------------------------
// Tile shape and split-K come from a latency model calibrated with tools/kbench.py --sweep on MI355X
// (profiles/r01_kbench_sweep.json): one BK=32 step costs a roughly constant t_k per resident block (the
// LDS-DMA ring keeps two stages in flight, the rest of the global-load latency is exposed), and a CU keeps
// `conc` blocks of a given tile shape in flight (LDS: 3 x 48 KB, 4 x 36 KB, 6 x 24 KB), so
//     time = ceil(blocks / (256 * conc)) * (k_steps * t_k + t_fix)  +  split-K reduction traffic.
// Predictions are within ~15% of the sweep for the 15 UNet shapes; the model prefers split-K whenever
// M*N alone gives fewer than ~3 blocks per CU (the 16x16 / 8x8 UNet levels are weight-streaming problems).
namespace {
struct TileCfg { int bm, bn, conc; double t_k; };
const TileCfg kTiles[3] = {{128, 128, 3, 0.85e-6}, {128, 64, 4, 0.82e-6}, {64, 64, 6, 0.52e-6}};

void plan_gemm(int M, int N, int K, int max_split, int& tile, int& splits) {
    const int ksteps = clora_cdiv(K, 32);
    double best = 1e30;
    tile = 0; splits = 1;
    for (int c = 0; c < 3; ++c) {
        const TileCfg& tc = kTiles[c];
        const long tiles = (long)clora_cdiv(M, tc.bm) * clora_cdiv(N, tc.bn);
        for (int s = 1; s <= max_split && s <= 16; ++s) {
            if (s > 1 && ksteps / s < 4) break;
            const int kps = clora_cdiv(ksteps, s);
            const int real_s = clora_cdiv(ksteps, kps);
            const long blocks = tiles * real_s;
            const long rounds = (blocks + 256L * tc.conc - 1) / (256L * tc.conc);
            double t = (double)rounds * (kps * tc.t_k + 3.0e-6);
            if (real_s > 1) t += (double)M * N * (8.0 * real_s + 2.0) / 6.0e12 + 4.0e-6;
            if (t < best) { best = t; tile = c; splits = real_s; }
        }
    }
}
}  // namespace

extern "C" int clora_gemm_f16_ex(const clora_half* A, int lda, const clora_half* B, clora_half* C, int ldc, int M, int N,
                                 int K, const clora_conv_t* conv, const clora_epilogue_t* epi, int split_k, int tile_cfg,
                                 void* workspace, size_t workspace_bytes, void* stream) {
    if (!A || !B || (!C && !(epi && epi->geglu == 1)) || M <= 0 || N <= 0 || K <= 0 || (K & 7) || (N & 7) || (ldc & 7)) return CLORA_ERR_ARG;
    GemmArgs a;
    a.A = (const half_t*)A; a.B = (const half_t*)B; a.C = (half_t*)C; a.partial = nullptr;
    a.lda = lda; a.ldc = ldc; a.M = M; a.N = N; a.K = K;
    a.tiles_m = 0; a.n_major = 0; a.xg_s = a.xg_m = a.xg_n = 0;
    a.two_phase = g_opts[CLORA_OPT_EPI_TWO_PHASE];
    a.hoist_on = g_opts[CLORA_OPT_EPI_HOIST];
    if (conv && conv->enabled) {
        a.conv = *conv;
        if ((conv->Cin & 7) || K != conv->ksize * conv->ksize * conv->Cin) return CLORA_ERR_ARG;
        if (conv->kchunk < 0 || (conv->kchunk > 0 && ((conv->kchunk & 7) || conv->Cin % conv->kchunk))) return CLORA_ERR_ARG;
    } else {
        a.conv = clora_conv_t();
        a.conv.enabled = 0;
        if (lda & 7) return CLORA_ERR_ARG;
    }
    if (epi) a.epi = *epi; else a.epi = clora_epilogue_t();
    if (a.epi.ln_out && (!a.epi.ln_gamma || !a.epi.ln_beta || !clora_gemm_ln_fusable(M, N, K, tile_cfg, split_k) || a.epi.geglu ||
                         a.conv.enabled || ((uintptr_t)a.epi.ln_gamma & 15) || ((uintptr_t)a.epi.ln_beta & 15)))
        return CLORA_ERR_ARG;
    clora_deferred_t* const defer = a.epi.defer;             // host-side request: never travels in the kernel arguments
    a.epi.defer = nullptr;
    if (defer) {
        if (a.epi.geglu) return CLORA_ERR_ARG;
        defer->splits = 0;                                   // until a split-K launch below says otherwise
        defer->partial = nullptr;
    }
    if (a.epi.lora_t && (a.epi.lora_r <= 0 || a.epi.lora_seg <= 0 || (a.epi.lora_seg & 15))) return CLORA_ERR_ARG;
    if (a.epi.rowadd && a.epi.rows_per_batch <= 0) return CLORA_ERR_ARG;
    if (a.epi.residual && (a.epi.ldr & 7)) return CLORA_ERR_ARG;
    // compensated trunk: residual_lo rides on residual's pitch, c_lo on C's; not with GEGLU, never deferred
    if ((a.epi.residual_lo && !a.epi.residual) || ((a.epi.residual_lo || a.epi.c_lo) && (a.epi.geglu || !C || (ldc & 7))) || (a.epi.c_lo && defer))
        return CLORA_ERR_ARG;
    if (a.epi.geglu) {
        const int F = a.epi.geglu_f;
        if (a.epi.geglu < 0 || a.epi.geglu > 2 || F <= 0 || (F & 63) || a.epi.rowadd || a.epi.residual || a.epi.lora_t) return CLORA_ERR_ARG;
        if (a.epi.geglu == 1 && (N != 2 * F || !a.epi.geglu_y || (C && ldc < 2 * F))) return CLORA_ERR_ARG;
        if (a.epi.geglu == 2 && (N != F || !a.epi.geglu_h || !C || ldc < 2 * F)) return CLORA_ERR_ARG;
        if (split_k > 1) return CLORA_ERR_ARG;
        split_k = 1;                                     // the fused activation lives in the main kernel's epilogue only
    }
    hipStream_t s = (hipStream_t)stream;
    //   61 = conv3x3_strip_kernel (the hint encoder's large-map 3x3 convolutions and their dgrads: 32 / 64 input channels, <= 64 output
    //        channels, kchunk 0, bias-only epilogue, C rows 8-byte aligned); anything it cannot take falls back to the library's own choice
    if (tile_cfg == 61) {
        const int plan = (a.conv.enabled && split_k <= 1 && !defer && !a.epi.rowadd && !a.epi.residual && !a.epi.lora_t && !a.epi.geglu &&
                          !a.epi.ln_out && !a.epi.c_lo && C && (ldc & 3) == 0 && K == 9 * a.conv.Cin)
                             ? strip_plan(M, N, a.conv) : 0;
        if (plan) {
            StripArgs q;
            q.X = a.A; q.Wt = a.B; q.C = a.C; q.bias = a.epi.bias; q.N = N; q.ldc = ldc; q.H = a.conv.Hout; q.W = a.conv.Wout;
            q.nimg = M / (a.conv.Hout * a.conv.Wout); q.off = a.conv.off; q.kmul = a.conv.kmul;
            if (plan == 802) return launch_strip8<2, 128>(q, s);
            if (plan == 804) return launch_strip8<4, 128>(q, s);
            if (plan == 3202) return launch_strip<32, 2, 128>(q, s);
            if (plan == 3204) return launch_strip<32, 4, 128>(q, s);
            return launch_strip<64, 2, 128>(q, s);
        }
        tile_cfg = 0;
    }
    // split_k: 0 = automatic (bounded by the workspace the caller provided), >= 1 = forced
    int tile = 0, splits = 1;
    const int ws_cap = workspace ? (int)(workspace_bytes / ((size_t)M * N * sizeof(float))) : 1;
    plan_gemm(M, N, K, split_k >= 1 ? 1 : (ws_cap < 1 ? 1 : ws_cap), tile, splits);
    if (split_k >= 1) {
        const int ksteps = clora_cdiv(K, 32);
        splits = split_k > ksteps ? ksteps : split_k;
    }
    // tile_cfg: 0 = the latency model's choice among 1..3;
    //   1..3  = 128x128 / 128x64 / 64x64, BK 32, 3-stage ring        4..6 = the same tiles with the deep ring (5 / 6 / 8 stages)
    //   7, 8  = 256x128 (wave tile 128x64), BK 32, 3 stages; 8 reads its fragments before refilling the ring
    //   11..13 = the register-staged v1 main loop (A/B comparisons)
    //   9 = 1 with the round-1 (2-way conflicted) swizzle key, for A/B runs
    //   21, 22, 23, 26 = BK 64 (128-byte LDS rows, whole-line DMA): 128x128 x2 stages, 128x64 x3, 64x64 x3, 128x64 x2
    //   31..33 = 1..3 and 41..43 = 21..23 with fragment reads before the ring refill
    //   51..53 = 8-wave blocks (one per CU), BK 64: 128x320 x2 stages, 64x320 x3, 128x256 x3 (wave tiles 32x160 / 32x80 / 32x128) for the
    //            short-K projections -- with the whole of N = 320 in one tile A is fetched once instead of once per 64 / 128 columns;
    //            54..56 = the same with fragment reads before the ring refill; 57, 58 = 256x320 / 256x256 x2 stages (wave tiles 64x160 /
    //            64x128) for the M >= 32768 projections of the batch-32 inference forward
    bool dma = true;
    if (a.epi.lora_dpack) {
        // the adapter down-projection rides in this launch (clora_epilogue_t.lora_dpack): 8-wave 320-column tiles only
        const clora_epilogue_t& e = a.epi;
        if (a.conv.enabled || !e.lora_t || e.lora_r != 4 || e.geglu || (e.lora_seg % 64) || (N % e.lora_seg) || (K & 63) || split_k > 1 ||
            (e.ldt & 3) || (e.lora_u_tr ? (e.ldu & 3) != 0 : e.ldu != 4) || N / e.lora_seg > 32 ||
            (e.lora_t_in && ((e.ldt_in & 3) || e.lora_t_in_rows < 0)))
            return CLORA_ERR_ARG;
        // capable tiles: a tile must lie inside one column segment.  8-wave 320-column tiles (51 / 52 / 54 / 55), 64-column tiles
        // (22 / 42 / 26: 128x64; 23 / 43: 64x64), 128x128 (21 / 41); anything else is replaced by the library's choice
        const int bn = (tile_cfg == 51 || tile_cfg == 52 || tile_cfg == 54 || tile_cfg == 55) ? 320
                     : (tile_cfg == 22 || tile_cfg == 42 || tile_cfg == 26 || tile_cfg == 23 || tile_cfg == 43) ? 64
                     : (tile_cfg == 21 || tile_cfg == 41) ? 128 : 0;
        if (bn == 0 || (e.lora_seg % bn)) {
            if (e.lora_seg % 320 == 0) tile_cfg = M >= 32768 ? 54 : 55;
            else tile_cfg = 43;
        }
        split_k = 1; splits = 1;
    }
    if (tile_cfg >= 11 && tile_cfg <= 13) { dma = false; tile_cfg -= 10; }
    int cfg = tile_cfg > 0 ? tile_cfg : tile + 1;
    // untuned shape (no table entry: tile_cfg == 0) that the patch-staged conv kernel can take: it beat every implicit-GEMM
    // variant on all 46 tuned signatures (profiles/r02_tune_patch.log), so it is the default there too -- 128-pixel tiles, 160
    // columns when the width allows, split over slabs until the grid covers the chip (split_k == 0) or as forced
    // (shapes only the WIDE patch can take -- rows of 128 pixels and more: the VAE's levels -- stay on the implicit GEMM: on MI355X
    // tile_cfg 77 ties it at 128 x 128 maps and loses 10-25 % at 256^2 / 512^2, profiles/r03_vae_conv_ab.txt: with 128 output columns
    // per tile the weight stream, which the patch does not reduce, dominates the operand traffic)
    if (tile_cfg == 0 && dma && !a.epi.geglu && patch_eligible(a, 128)) {
        cfg = (N % 160 == 0) ? 76 : 72;
        if (split_k == 0) {
            const long blocks = (long)clora_cdiv(M, 128) * clora_cdiv(N, cfg == 76 ? 160 : 128);
            const int slabs = a.conv.Cin / 64;
            int want = (int)((256 + blocks - 1) / blocks);
            if (want > slabs) want = slabs;
            while (want > 1 && (!workspace || workspace_bytes < (size_t)want * M * N * sizeof(float))) --want;
            splits = want < 1 ? 1 : want;
        }
    }
    if (a.epi.geglu) {
        if (!dma) return CLORA_ERR_ARG;
        const bool wide = cfg == 1 || cfg == 4 || cfg == 7 || cfg == 8 || cfg == 9 || cfg == 21 || cfg == 31 || cfg == 41 || cfg == 53 || cfg == 56 || cfg == 58 || cfg == 59;
        if (a.epi.geglu == 1 && !wide) { if (tile_cfg > 0) return CLORA_ERR_ARG; cfg = 1; }
    }
    //   71..76 = conv3x3_patch_kernel (3x3 stride-1 pad-1 convs and their dgrads, input patch staged once per 64-channel slab):
    //            256x128 (8 waves of 64x64), 128x128 (64x32), 128x128 (32x64), 256x64 (32x64), 128x64 (32x32), 128x160 (32x80: the
    //            UNet widths 320 / 640 / 960 / 1280 are multiples of 160, not of 128); shapes it cannot take fall back to 21
    //            77, 78 = 128x128 / 128x64 with the 392-pixel patch (one 128-pixel row or row segment per tile: W = 128, 256, 512)
    //            79 = 256x160 (8 waves of 64x80; exactly 160 KB of LDS): batch >= 8 / inference shapes at N = 320 / 640 / 960 / 1280
    if (cfg >= 71 && cfg <= 79) {
        if (!dma || !patch_eligible(a, (cfg == 71 || cfg == 74 || cfg == 79) ? 256 : 128, (cfg == 77 || cfg == 78) ? kPatchWide : 0)) cfg = 21;
        else {
            const int slabs = a.conv.Cin / 64;
            if (splits > slabs) splits = slabs;
            a.k_per_split = clora_cdiv(slabs, splits);                 // SLABS per split for this kernel
            splits = clora_cdiv(slabs, a.k_per_split);
            if (splits > 1) {
                if (!workspace || workspace_bytes < (size_t)splits * M * N * sizeof(float)) return CLORA_ERR_WORKSPACE;
                a.partial = (float*)workspace;
            }
            int rc;
            switch (cfg) {
                case 71: rc = launch_patch<256, 128, 4, 2, 3>(a, splits, s); break;
                case 72: rc = launch_patch<128, 128, 2, 4, 3>(a, splits, s); break;
                case 73: rc = launch_patch<128, 128, 4, 2, 3>(a, splits, s); break;
                case 74: rc = launch_patch<256, 64, 8, 1, 4>(a, splits, s); break;
                case 75: rc = launch_patch<128, 64, 4, 2, 4>(a, splits, s); break;
                case 77: rc = launch_patch<128, 128, 2, 4, 3, kPatchWide>(a, splits, s); break;
                case 78: rc = launch_patch<128, 64, 4, 2, 4, kPatchWide>(a, splits, s); break;
                case 79: rc = launch_patch<256, 160, 4, 2, 3>(a, splits, s); break;
                default: rc = launch_patch<128, 160, 4, 2, 3>(a, splits, s); break;
            }
            if (rc != CLORA_OK) return rc;
            if (splits > 1) rc = finish_or_defer(a, splits, defer, s);
            return rc;
        }
    }
    // (round 5, measured and removed: conv3x3_patch_kernel with its two wave halves one barrier apart -- the ping-pong schedule of
    // gemm_8p_kernel, cfgs 84 / 86 / 89 at the time: parity green on hardware, 128x160 +0.5..1.5 % (noise level), the 256-row tiles
    // 25-60 % slower (spills); the lockstep tap loop is NOT what holds the patch convs at 43 % MFMA busy: profiles/r05_patch_pingpong_ab.txt)
    //   59 = gemm_8p_kernel: 256x256, 8 waves of 128x64, eight-phase ping-pong (plain GEMM rows only; convs fall back to 58)
    if (cfg == 59 && (a.conv.enabled || !dma)) cfg = 58;
    const int bk = ((cfg >= 21 && cfg <= 26) || (cfg >= 41 && cfg <= 43) || (cfg >= 51 && cfg <= 59) || (cfg >= 91 && cfg <= 96)) ? 64 : 32;
    a.k_per_split = clora_cdiv(clora_cdiv(K, bk), splits) * bk;
    splits = clora_cdiv(K, a.k_per_split);
    if (splits > 1) {
        if (!workspace || workspace_bytes < (size_t)splits * M * N * sizeof(float)) return CLORA_ERR_WORKSPACE;
        a.partial = (float*)workspace;
    }
    int rc;
    switch (cfg) {
        case 1: rc = launch_gemm<128, 128, 2, 2>(a, splits, s, dma); break;
        case 2: rc = launch_gemm<128, 64, 4, 1>(a, splits, s, dma); break;
        case 3: rc = launch_gemm<64, 64, 2, 2>(a, splits, s, dma); break;
        case 4: rc = launch_gemm<128, 128, 2, 2, 5>(a, splits, s, true); break;
        case 5: rc = launch_gemm<128, 64, 4, 1, 6>(a, splits, s, true); break;
        case 6: rc = launch_gemm<64, 64, 2, 2, 8>(a, splits, s, true); break;
        case 7: rc = launch_gemm<256, 128, 2, 2, 3, 32, 0>(a, splits, s, true); break;
        case 8: rc = launch_gemm<256, 128, 2, 2, 3, 32, 1>(a, splits, s, true); break;
        case 21: rc = launch_gemm<128, 128, 2, 2, 2, 64, 0>(a, splits, s, true); break;
        case 22: rc = launch_gemm<128, 64, 4, 1, 3, 64, 0>(a, splits, s, true); break;
        case 23: rc = launch_gemm<64, 64, 2, 2, 3, 64, 0>(a, splits, s, true); break;
        case 26: rc = launch_gemm<128, 64, 4, 1, 2, 64, 0>(a, splits, s, true); break;
        case 31: rc = launch_gemm<128, 128, 2, 2, 3, 32, 1>(a, splits, s, true); break;
        case 32: rc = launch_gemm<128, 64, 4, 1, 3, 32, 1>(a, splits, s, true); break;
        case 33: rc = launch_gemm<64, 64, 2, 2, 3, 32, 1>(a, splits, s, true); break;
        case 9: rc = launch_gemm<128, 128, 2, 2, 3, 32, 2>(a, splits, s, true); break;     // round-1 swizzle key (A/B)
        case 51: rc = launch_gemm<128, 320, 4, 2, 2, 64, 0>(a, splits, s, true); break;    // 8 waves, the whole of N = 320 per tile: A crosses L2 -> LDS once
        case 52: rc = launch_gemm<64, 320, 2, 4, 3, 64, 0>(a, splits, s, true); break;
        case 53: rc = launch_gemm<128, 256, 4, 2, 3, 64, 0>(a, splits, s, true); break;
        case 57: rc = launch_gemm<256, 320, 4, 2, 2, 64, 1>(a, splits, s, true); break;    // M >= 32768 (batch-32 inference): 142 flop per operand byte
        case 58: rc = launch_gemm<256, 256, 4, 2, 2, 64, 1>(a, splits, s, true); break;
        case 59: rc = launch_gemm_8p(a, splits, s); break;
        // (round 5, measured and removed: the 320-column tiles and the 128x160 / 128x128 patch tiles with FOUR waves = one wave per
        // SIMD, 512 registers, wave tiles 64x80 / 64x160 / 64x64 -- half the fragment reads per MFMA, but a lone wave per SIMD has
        // nobody to hide its LDS latency behind: 15..50 % SLOWER on every level-0 / level-1 shape, profiles/r05_onewave_ab.txt)
        case 54: rc = launch_gemm<128, 320, 4, 2, 2, 64, 1>(a, splits, s, true); break;
        case 55: rc = launch_gemm<64, 320, 2, 4, 3, 64, 1>(a, splits, s, true); break;
        case 56: rc = launch_gemm<128, 256, 4, 2, 3, 64, 1>(a, splits, s, true); break;
#ifdef CLORA_DMA_PROBE
        // timing probes (results are garbage): the A and / or B tiles are fetched from the 16-byte zero page -- the same DMA
        // instruction stream with no L2 -> LDS operand traffic.  Built only by tools/dma_probe.sh.
        case 91: rc = launch_gemm<128, 128, 2, 2, 2, 64, 4>(a, splits, s, true); break;
        case 92: rc = launch_gemm<128, 128, 2, 2, 2, 64, 8>(a, splits, s, true); break;
        case 93: rc = launch_gemm<128, 128, 2, 2, 2, 64, 12>(a, splits, s, true); break;
        case 94: rc = launch_gemm<128, 64, 4, 1, 2, 64, 4>(a, splits, s, true); break;
        case 95: rc = launch_gemm<128, 64, 4, 1, 2, 64, 8>(a, splits, s, true); break;
        case 96: rc = launch_gemm<128, 64, 4, 1, 2, 64, 12>(a, splits, s, true); break;
#endif
        case 41: rc = launch_gemm<128, 128, 2, 2, 2, 64, 1>(a, splits, s, true); break;
        case 42: rc = launch_gemm<128, 64, 4, 1, 3, 64, 1>(a, splits, s, true); break;
        case 43: rc = launch_gemm<64, 64, 2, 2, 3, 64, 1>(a, splits, s, true); break;
        default: return CLORA_ERR_ARG;
    }
    if (rc != CLORA_OK) return rc;
    if (splits > 1) rc = finish_or_defer(a, splits, defer, s);
    return rc;
}

extern "C" int clora_gemm_ln_fusable(int M, int N, int K, int tile_cfg, int split_k) {
    (void)K;
    return M > 0 && N == 320 && split_k == 1 && (tile_cfg == 51 || tile_cfg == 52 || tile_cfg == 54 || tile_cfg == 55) ? 1 : 0;
}

extern "C" int clora_finish_deferred(const clora_deferred_t* d, void* stream) {
    if (!d || !d->partial || !d->C || d->splits <= 1 || d->M <= 0 || d->N <= 0 || (d->N & 7) || (d->ldc & 7)) return CLORA_ERR_ARG;
    GemmArgs a = GemmArgs();
    a.partial = const_cast<float*>(d->partial); a.C = (half_t*)d->C; a.ldc = d->ldc; a.M = d->M; a.N = d->N; a.epi = d->epi;
    a.epi.defer = nullptr;
    return finish_or_defer(a, d->splits, nullptr, (hipStream_t)stream);
}

int clora_option(int id) { return (id >= 0 && id < CLORA_OPT_COUNT) ? g_opts[id] : 0; }

extern "C" int clora_set_option(const char* name, int value) {
    if (!name) return CLORA_ERR_ARG;
    struct Opt { const char* name; int id, lo, hi; };
    static const Opt kOpts[] = {{"tile_order", CLORA_OPT_TILE_ORDER, 0, 3}, {"ln_rows", CLORA_OPT_LN_ROWS, 0, 1},
                                {"attn_fwd_waves", CLORA_OPT_ATTN_FWD_WAVES, 0, 16}, {"attn_bwd_waves", CLORA_OPT_ATTN_BWD_WAVES, 0, 8},
                                {"gn_blocks", CLORA_OPT_GN_BLOCKS, 64, 1 << 16}, {"epi_two_phase", CLORA_OPT_EPI_TWO_PHASE, 0, 1},
                                {"lora_down_mode", CLORA_OPT_LORA_DOWN_MODE, 0, 2}, {"gn_unroll", CLORA_OPT_GN_UNROLL, 0, 1},
                                {"epi_hoist", CLORA_OPT_EPI_HOIST, 0, 1}, {"gn_resident", CLORA_OPT_GN_RESIDENT, 0, 1},
                                {"defer_max_rows", CLORA_OPT_DEFER_MAX_ROWS, 0, 16}, {"wgrad_patch", CLORA_OPT_WGRAD_PATCH, 0, 8192}, {"strip_blocks", CLORA_OPT_STRIP_BLOCKS, 64, 16384},
                                {"gn_team", CLORA_OPT_GN_TEAM, 0, 4}};
    for (const Opt& o : kOpts)
        if (!strcmp(name, o.name)) {
            if (value < o.lo || value > o.hi) return CLORA_ERR_ARG;
            if ((o.id == CLORA_OPT_ATTN_FWD_WAVES && value != 0 && value != 4 && value != 6 && value != 8 && value != 16) ||
                (o.id == CLORA_OPT_ATTN_BWD_WAVES && value != 0 && value != 4 && value != 8)) return CLORA_ERR_ARG;
            g_opts[o.id] = value;
            return CLORA_OK;
        }
    return CLORA_ERR_ARG;
}

extern "C" int clora_conv_strip_eligible(int M, int N, const clora_conv_t* conv) {
    return (conv && strip_plan(M, N, *conv)) ? 1 : 0;
}

extern "C" int clora_conv_patch_eligible(int M, const clora_conv_t* conv, int tile_cfg) {
    if (!conv || tile_cfg < 71 || tile_cfg > 79) return 0;
    GemmArgs a;
    a.M = M; a.conv = *conv;
    return patch_eligible(a, (tile_cfg == 71 || tile_cfg == 74 || tile_cfg == 79) ? 256 : 128, (tile_cfg == 77 || tile_cfg == 78) ? kPatchWide : 0) ? 1 : 0;   // same arguments as the launcher's switch
}

extern "C" int clora_gemm_f16(const clora_half* A, int lda, const clora_half* B, clora_half* C, int ldc, int M, int N,
                              int K, const clora_conv_t* conv, const clora_epilogue_t* epi, int split_k,
                              void* workspace, size_t workspace_bytes, void* stream) {
    return clora_gemm_f16_ex(A, lda, B, C, ldc, M, N, K, conv, epi, split_k, 0, workspace, workspace_bytes, stream);
}

// staging [Co][taps*Cip (+ bias column block)] in the gather's column order -> += into the OIHW parameter gradients; the
// staging buffer is left zeroed for the next step (it is persistent: no per-step fill launch).  The wgrad kernel keeps
// its atomics on the gather-ordered layout because OIHW puts consecutive channels 9 floats apart (measured: 3x slower).
__global__ __launch_bounds__(256) void conv_wgrad_unpack_kernel(float* stage, float* stage_b, float* gw, float* gb, int Co, int Ci,
                                                                int taps, int Cip) {
    const int nw = Co * taps * Cip;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nw + (gb ? Co : 0); i += gridDim.x * 256) {
        if (i < nw) {
            const int ci = i % Cip, tap = (i / Cip) % taps, co = i / (Cip * taps);
            const float v = stage[i];
            stage[i] = 0.f;
            if (ci < Ci) gw[((size_t)co * Ci + ci) * taps + tap] += v;
        } else {
            const int co = i - nw;
            gb[co] += stage_b[co];
            stage_b[co] = 0.f;
        }
    }
}

extern "C" int clora_conv_wgrad_unpack_f32(float* stage, float* stage_b, float* grad_w, float* grad_b, int Co, int Ci, int ksize,
                                           int Cip, void* stream) {
    if (!stage || !grad_w || Co <= 0 || Ci <= 0 || Cip < Ci || (ksize != 1 && ksize != 3) || ((grad_b == nullptr) != (stage_b == nullptr)))
        return CLORA_ERR_ARG;
    const int taps = ksize * ksize;
    const long total = (long)Co * taps * Cip + (grad_b ? Co : 0);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(conv_wgrad_unpack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, stage, stage_b, grad_w, grad_b,
                       Co, Ci, taps, Cip);
    return clora_check_launch();
}

extern "C" int clora_conv_weight_pack_f32(const float* w, int Co, int Ci, int ksize, int Cip, int Cop, clora_half* fwd,
                                          clora_half* dgrad, void* stream) {
    if (!w || !fwd || Co <= 0 || Ci <= 0 || (ksize != 1 && ksize != 3) || Cip < Ci || (Cip & 7) || (dgrad && (Cop < Co || (Cop & 7))))
        return CLORA_ERR_ARG;
    const int taps = ksize * ksize;
    const long total = (long)Co * taps * Cip + (dgrad ? (long)Cip * taps * Cop : 0);
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(conv_weight_pack_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, Co, Ci, taps, Cip, Cop,
                       (half_t*)fwd, (half_t*)dgrad);
    return clora_check_launch();
}

extern "C" int clora_conv_wgrad_f16(const clora_half* dY, int ldy, const clora_half* X, int ldx, float* dW, float* db,
                                    int M, int N, int K, const clora_conv_t* conv, int oihw_ci, void* stream) {
    if (!dY || !X || !dW || M <= 0 || N <= 0 || K <= 0 || (K & 7) || (N & 7) || (ldy & 7) || oihw_ci < 0) return CLORA_ERR_ARG;
    WgradArgs a;
    a.dY = (const half_t*)dY; a.X = (const half_t*)X; a.dW = dW; a.db = db;
    a.ldy = ldy; a.ldx = ldx; a.M = M; a.N = N; a.K = K; a.oihw_ci = oihw_ci;
    if (conv && conv->enabled) {
        a.conv = *conv;
        if ((conv->Cin & 7) || K != conv->ksize * conv->ksize * conv->Cin || conv->kchunk != 0) return CLORA_ERR_ARG;
    } else {
        a.conv = clora_conv_t();
        a.conv.enabled = 0;
        if (ldx & 7) return CLORA_ERR_ARG;
    }
    // the large-map 3x3 stride-1 pad-1 convolutions of the hint encoder: patch-staged kernel (conv_wgrad_patch_kernel)
    if (conv && conv->enabled && clora_option(CLORA_OPT_WGRAD_PATCH) && oihw_ci == 0 && ldy == N) {
        const clora_conv_t& c = *conv;
        const bool s1 = c.ksize == 3 && c.mul == 1 && c.kmul == 1 && c.off == -1 && c.shift == 0 && c.need_even == 0 && c.Hout == c.Hin &&
                        c.Wout == c.Win && c.lim_h == c.Hin && c.lim_w == c.Win && c.Hout > 0 && (M % (c.Hout * c.Wout)) == 0 &&
                        (long)M * (c.Cin > N ? c.Cin : N) < (1L << 31);
        // ... and its stride-2 downsamplers (F.pad(0, 1, 0, 1) then stride 2: conv_fwd_desc(asym_pad))
        const bool s2 = c.ksize == 3 && c.mul == 2 && c.kmul == 1 && c.off == 0 && c.shift == 0 && c.need_even == 0 && c.Hin == 2 * c.Hout &&
                        c.Win == 2 * c.Wout && c.lim_h == c.Hin && c.lim_w == c.Win && c.Hout > 0 && (M % (c.Hout * c.Wout)) == 0 &&
                        (long)M * 4 * (c.Cin > N ? c.Cin : N) < (1L << 31);
        if ((s1 || s2) && (c.Cin == 32 || c.Cin == 64) && N <= 64) {
            WgradPatchArgs q;
            q.dY = (const half_t*)dY; q.X = (const half_t*)X; q.dW = dW; q.db = db; q.N = N; q.K = K; q.H = c.Hout; q.W = c.Wout;
            q.Hin = c.Hin; q.Win = c.Win;
            q.nimg = M / (c.Hout * c.Wout);
            hipStream_t s = (hipStream_t)stream;
            if (s2) {
                if (c.Cin == 32 && N <= 32 && (c.Wout % 64) == 0) return launch_wgrad_patch<32, 1, 64, 2>(q, s);
                if (c.Cin == 32 && N <= 64 && (c.Wout % 64) == 0) return launch_wgrad_patch<32, 2, 64, 2>(q, s);
                if (c.Cin == 64 && N <= 64 && (c.Wout % 64) == 0) return launch_wgrad_patch<64, 2, 64, 2>(q, s);
            } else {
            if (c.Cin == 32 && N <= 32 && (c.Wout % 128) == 0) return launch_wgrad_patch<32, 1, 128>(q, s);
            if (c.Cin == 32 && N <= 64 && (c.Wout % 128) == 0) return launch_wgrad_patch<32, 2, 128>(q, s);
            if (c.Cin == 32 && N <= 64 && (c.Wout % 64) == 0) return launch_wgrad_patch<32, 2, 64>(q, s);
            if (c.Cin == 64 && N <= 64 && (c.Wout % 64) == 0) return launch_wgrad_patch<64, 2, 64>(q, s);
            // (64 -> 128 channels at 128^2: 74 K results per block; 56 us at 64 blocks against 57 for the gather kernel -- stays there)
            }
        }
    }
    a.tiles_n = clora_cdiv(N, 64);
    a.tiles_k = clora_cdiv(K + (db ? 8 : 0), 64);
    const int tiles = a.tiles_n * a.tiles_k;
    int chunks = clora_cdiv(2048, tiles);              // aim for ~2048 blocks
    int mpb = clora_cdiv(clora_cdiv(M, chunks), 64) * 64;
    if (mpb < 256) mpb = 256;
    a.m_per_block = mpb;
    hipLaunchKernelGGL(conv_wgrad_kernel, dim3(tiles, clora_cdiv(M, mpb)), dim3(256), 0, (hipStream_t)stream, a);
    return clora_check_launch();
}
