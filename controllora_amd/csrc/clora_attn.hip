// clora_attn.hip -- fused (flash-style) attention core for gfx950: forward, dQ, dK/dV.
//
// Replaces the materialised  baddbmm -> softmax -> bmm  of the reference processors
// (models.py:140-141, 270-271, 407-408; upstream CrossAttention.get_attention_scores):
// at 512^2, B=4 that score tensor is 1 GiB per self-attention site and is what autograd keeps for
// backward; here it never leaves registers.
//
// Layout: q/k/v/o are read and written IN PLACE in the [B, N, H*D] token-major activations (row
// strides ldq/ldk/...), so head_to_batch_dim / batch_to_head_dim permutes disappear.
//
// Register-level design (wave = 64 lanes, v_mfma_f32_16x16x32_f16):
//   * scores are computed TRANSPOSED, S^T = K . Q^T, so every lane owns one query column (lane&15):
//     softmax statistics (m, l), the LSE and the delta of the backward are lane-local scalars and the
//     row reductions need only two shuffles (across the 4 lane groups);
//   * a C-layout tile (lane: col = lane&15, rows 4*(lane>>4)+r) is fed back as the B operand of the
//     next MFMA WITHOUT any shuffle or LDS round trip: two 16-row tiles give the 8 k-slots of a lane,
//     and the A operand (V^T, K^T, Q^T, dO^T) is fetched with the same slot assignment from ROW-MAJOR LDS tiles by
//     the transpose read ds_read_b64_tr_b16 (frag_tr) -- the hardware only pairs slot e of lane-group g of A with
//     slot e of group g of B; no transposed copies, no 2-byte scatter stores;
//   * head dims 40 / 80 / 160 are zero-padded to 64 / 96 / 160 along the contraction dim only;
//   * LDS tile pitches are == 32 bytes (mod 64): pitch/4 == 8 (mod 16) banks, so the 8 rows one half-wave touches in a
//     transpose read start on 8 distinct multiples of 8 banks (conflict-free ds_read_b64_tr_b16) and the ds_read_b128
//     fragment reads are conflict-free for the hardware's real lane groups as well (MI355X_MICROARCH.md section LDS;
//     the round-1 pitch DP + 8 was 2-way conflicted for both).
#include <stdlib.h>
#include "clora_common.h"
#include "../../include/clora.h"

namespace {

struct AttnArgs {
    const half_t *q, *k, *v, *o, *dO;
    half_t *out, *dq, *dk, *dv;
    const float* lse_in;
    float *lse, *delta;
    int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    int B, H, Nq, Nk, D;
    float scale;
    int nsplit, q_per_split;   // dK/dV: split the query loop over grid.z (cross-attention: few keys, many queries)
    float* acc32;              // [2][B, Nk, H*D] fp32 accumulators for the split path
    int xcd_heads;             // block order: whole (batch, head) pairs per XCD (see attn_block_ids)
};

// The blocks of one (batch, head) read the same K / V (forward, dQ) or Q / dO (dK/dV) stream.  Workgroup L runs on XCD L % 8
// and a line is fetched through the fabric once per XCD that asks for it (tools/probes/l2_share_probe.hip): in launch order
// the blocks of a head sit on all eight XCDs.  With xcd_heads every XCD walks a contiguous range of (head, block) pairs, so
// a head's stream is fetched by one XCD (two at a range boundary).  Same remap as the GEMM tile order; results do not change.
__device__ __forceinline__ void attn_block_ids(const AttnArgs& p, int& bx, int& by) {
    bx = blockIdx.x; by = blockIdx.y;
    if (!p.xcd_heads || gridDim.z != 1) return;
    const int nx = gridDim.x, nwg = nx * gridDim.y;
    const int lin = by * nx + bx;
    const int xq = nwg >> 3, xr = nwg & 7, xcd = lin & 7;
    const int logical = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
    by = logical / nx;
    bx = logical - by * nx;
}

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr float kNegBig = -1.0e30f;

// A operand (k-slots = rows {pair*32 + 4g .. +3} and {pair*32 + 16 + 4g .. +3}, operand row = column col0 + li) straight
// from a ROW-MAJOR tile [row][col] with the gfx950 transpose read: the 16 lanes of group g
// address the 4 rows {pair*32 + 4g .. +3} x 16 columns {col0 ..} (4 lanes per row, 4 columns each) and lane li receives
// column col0 + li of those 4 rows; a second read 16 rows further down gives the other 4 k-slots.  The tile is written
// with 16-byte stores (no transposed 2-byte scatter) and needs no second, transposed copy in LDS.
template <int LD>
__device__ __forceinline__ half8 frag_tr(const half_t* tile, int col0, int pair, int g, int l) {
    const half_t* q = tile + (pair * 32 + 4 * g + ((l & 15) >> 2)) * LD + col0 + (l & 3) * 4;
    const half4v a = CLORA_DS_READ_TR16(q), b = CLORA_DS_READ_TR16(q + 16 * LD);
    half8 r;
    r[0] = a[0]; r[1] = a[1]; r[2] = a[2]; r[3] = a[3];
    r[4] = b[0]; r[5] = b[1]; r[6] = b[2]; r[7] = b[3];
    return r;
}
// B operand from two C-layout tiles (same slot assignment as frag_tr)
__device__ __forceinline__ half8 frag_from_acc(floatx4 lo, floatx4 hi) {
    half8 r;
    r[0] = (half_t)lo[0]; r[1] = (half_t)lo[1]; r[2] = (half_t)lo[2]; r[3] = (half_t)lo[3];
    r[4] = (half_t)hi[0]; r[5] = (half_t)hi[1]; r[6] = (half_t)hi[2]; r[7] = (half_t)hi[3];
    return r;
}

// K / V / Q / dO tiles go global -> LDS by LDS-DMA (`global_load_lds_dwordx4`: no staging registers, no ds_write pass) into a
// DOUBLE-BUFFERED tile pair: the copy of tile t+1 is issued before the MFMAs of tile t and only has to have landed at the
// single barrier that ends the iteration (s_waitcnt vmcnt(0) + s_barrier) -- one barrier per tile instead of two, no exposed
// LDS write phase, and 16-32 VGPRs of staging registers returned to the loop.
// A tile is ROWS x (LD/8) 16-byte chunks, row pitch LD halves (the conflict-free padded pitch of the fragment reads); the
// DMA image is lane-linear, so one wave-instruction fills 64 consecutive chunk slots; the lane that owns slot s fetches
// chunk s % (LD/8) of row s / (LD/8) -- or a 16-byte zero page for the head-dim / row padding, or (ONE) the "one page"
// for the chunk that starts at column D: the ones column of V that makes P.V accumulate rowsum(P) (attn_fwd_kernel).
__device__ __attribute__((aligned(16))) const unsigned g_attn_zero16[4] = {0u, 0u, 0u, 0u};
__device__ __attribute__((aligned(16))) const unsigned short g_attn_one16[8] = {0x3C00u, 0, 0, 0, 0, 0, 0, 0};

template <int ROWS, int LD, int NWV = 4>
struct TileDma {
    static constexpr int PCH = LD / 8;                 // chunks per padded row
    static constexpr int NI = ROWS * PCH / 64;         // wave-instructions per tile
    static constexpr int NIW = (NI + NWV - 1) / NWV;   // ... per wave (wave w of NWV issues instructions w, w+NWV, ...)
    static_assert((ROWS * PCH) % 64 == 0, "a tile must be a whole number of DMA wave-instructions");
    int row[NIW], col[NIW];                            // this lane's (row, first column) per instruction; col -1: padding, -2: ones chunk
    __device__ __forceinline__ void init(int w, int l, int D) {
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const int sl = (w + NWV * j) * 64 + l;
            const int r = sl / PCH, c = sl - r * PCH;
            row[j] = r;
            col[j] = (c * 8 < D) ? c * 8 : (c * 8 == D ? -2 : -1);
        }
    }
    template <bool ONE>
    __device__ __forceinline__ void issue(const half_t* base, int ld, int rows_valid, half_t* dst, int w) const {
        const half_t* zero_page = reinterpret_cast<const half_t*>(g_attn_zero16);
        const half_t* one_page = reinterpret_cast<const half_t*>(g_attn_one16);
#pragma unroll
        for (int j = 0; j < NIW; ++j) {
            const int i = w + NWV * j;
            if (i < NI) {
                const bool in = row[j] < rows_valid;       // selects, not branches: the issue path stays straight-line
                const half_t* src = (in && col[j] >= 0) ? base + row[j] * ld + col[j] : zero_page;
                if (ONE) src = (in && col[j] == -2) ? one_page : src;
                CLORA_GLDS16(src, dst + i * 512);
            }
        }
    }
};
__device__ __forceinline__ floatx4 splat4f(float x) {
    floatx4 z = {x, x, x, x};
    return z;
}
// ------------------------------------------------------------------------------------------ forward
// __launch_bounds__(256, 2) for head dims <= 64: with a 256-register budget the compiler keeps the MFMA accumulators in
// VGPRs; with the default (one block per CU, 512 registers) it parks them in AGPRs and the softmax / rescale VALU work
// pays ~145 v_accvgpr_read/write moves per KV tile (a third of the loop's VALU instructions).  Larger head dims would
// spill at 256 registers and keep the default.
//
// The loop is VALU-bound at head dim 40 (32 exponentials + ~135 plain VALU instructions against 28 MFMAs per KV tile and
// wave), so the softmax bookkeeping is kept off the vector pipe:
//   * the exponent reference is LAZY: the score accumulators are initialised to -mref (free: it replaces the zero
//     initialisation), so the MFMA delivers s - mref and the probabilities are one bare v_exp_f32 each -- no per-element
//     subtraction.  mref only follows the running maximum when a tile exceeds it by more than 2^kRebase (wave-uniform
//     branch; always on the first tile): probabilities are then bounded by 2^kRebase = 256 instead of 1, which changes
//     nothing for fp16's relative precision or the fp32 accumulators, and the common tile pays no rescale of O.
//   * ONES (head dims with D = 16 DT - 8, i.e. 40): V's zero-padding column D is staged as 1, so the P.V MFMAs that
//     multiply the padding anyway accumulate rowsum(P) in row D of O^T -- no per-element additions, and the running sum
//     is rescaled together with O.  (It is the sum of the fp16-rounded probabilities, i.e. exactly the weights that
//     multiplied V.)
constexpr float kRebase = 8.0f;

// NWV waves per block (32 queries each).  The loop is bound by the L2 -> LDS stream of K/V tiles (every block of NWV*32 queries
// re-streams all keys: 7.5 TB/s at B = 32, the same DMA ceiling the GEMMs hit), so more queries per block = less traffic per flop:
// 8 waves (256 queries, 126 registers: two blocks per CU = 16 waves) at head dims <= 64 when the grid stays full (-18 % at B = 4, N = 4096).
// CAUSAL (clora_attn_fwd_causal_f16: the CLIP text encoder's masked self-attention, 77 tokens): key j is visible to query i
// iff j <= i.  A separate instantiation, so the unmasked kernels of the UNet keep their exact code.
template <int DP, int DT, bool ONES, int NWV, bool CAUSAL = false>
__global__ __launch_bounds__(NWV * 64, (DP <= 64 ? (NWV == 16 ? 4 : (NWV == 4 ? 3 : 2)) : 1)) void attn_fwd_kernel(AttnArgs p) {
    constexpr int BKV = 64, LDK = DP + 16, DV = DT * 16, LDV = DV + ((DV % 32) == 16 ? 0 : 16), KS = DP / 32;
    constexpr int TILE = BKV * LDK + BKV * LDV;           // one K tile + one V tile; two of them: double buffer
    __shared__ __attribute__((aligned(16))) half_t smem[2 * TILE];
    const int t = threadIdx.x, w = t >> 6, l = t & 63, g = l >> 4, li = l & 15;
    int bx, by;
    attn_block_ids(p, bx, by);
    const int b = by / p.H, h = by % p.H;
    const int q0 = bx * (NWV * 32) + w * 32;
    const int D = p.D;
    const float c = p.scale * kLog2e;

    const half_t* kbase = p.k + (size_t)b * p.Nk * p.ldk + h * D;
    const half_t* vbase = p.v + (size_t)b * p.Nk * p.ldv + h * D;
    TileDma<BKV, LDK, NWV> dk_;
    TileDma<BKV, LDV, NWV> dv_;
    dk_.init(w, l, D);
    dv_.init(w, l, D);
    {
        const int rows0 = p.Nk < BKV ? p.Nk : BKV;
        dk_.template issue<false>(kbase, p.ldk, rows0, smem, w);
        dv_.template issue<ONES>(vbase, p.ldv, rows0, smem + BKV * LDK, w);
    }

    half8 qf[2][KS];       // Q pre-multiplied by scale*log2(e): scores come out of the MFMA ready for exp2
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int q = q0 + qg * 16 + li, d = ks * 32 + g * 8;
            half8 v = (q < p.Nq && d < D) ? ld8(p.q + ((size_t)b * p.Nq + q) * p.ldq + h * D + d) : zero8();
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * c);
            qf[qg][ks] = v;
        }
    floatx4 oacc[DT][2];
#pragma unroll
    for (int i = 0; i < DT; ++i) { oacc[i][0] = zero4f(); oacc[i][1] = zero4f(); }
    float mref[2] = {0.f, 0.f}, lrun[2] = {0.f, 0.f};
    bool first = true;

    CLORA_WAIT_VMCNT(0);
    __syncthreads();                                       // tile 0 has landed for every wave
    int cur = 0;
    for (int kv0 = 0; kv0 < p.Nk; kv0 += BKV) {
        const int rows = (p.Nk - kv0 < BKV) ? p.Nk - kv0 : BKV;
        const half_t* Ks = smem + cur * TILE;
        const half_t* Vs = Ks + BKV * LDK;                 // V row-major [key][d]; read transposed (frag_tr) for P.V
        if (kv0 + BKV < p.Nk) {                            // tile t+1 -> the other buffer (consumed in iteration t-1, barrier since)
            const int nrows = (p.Nk - kv0 - BKV < BKV) ? p.Nk - kv0 - BKV : BKV;
            half_t* nb = smem + (cur ^ 1) * TILE;
            dk_.template issue<false>(kbase + (size_t)(kv0 + BKV) * p.ldk, p.ldk, nrows, nb, w);
            dv_.template issue<ONES>(vbase + (size_t)(kv0 + BKV) * p.ldv, p.ldv, nrows, nb + BKV * LDK, w);
        }

        floatx4 s[4][2];
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) { s[kt][0] = splat4f(-mref[0]); s[kt][1] = splat4f(-mref[1]); }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt) {
                const half8 a = ld8(Ks + (kt * 16 + li) * LDK + ks * 32 + g * 8);
                s[kt][0] = mfma16(a, qf[0][ks], s[kt][0]);
                s[kt][1] = mfma16(a, qf[1][ks], s[kt][1]);
            }
        if (CAUSAL) {                                      // keys past the query (and past the end of a ragged tile)
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int kl = kt * 16 + 4 * g + r, key = kv0 + kl;
                    if (kl >= rows || key > q0 + li) s[kt][0][r] = kNegBig;
                    if (kl >= rows || key > q0 + 16 + li) s[kt][1][r] = kNegBig;
                }
        } else if (rows < BKV) {                           // ragged last tile only: mask the missing keys
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (kt * 16 + 4 * g + r >= rows) { s[kt][0][r] = kNegBig; s[kt][1][r] = kNegBig; }
        }
        float mx[2];
#pragma unroll
        for (int qg = 0; qg < 2; ++qg) {
            float m = kNegBig;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) m = fmaxf(m, s[kt][qg][r]);
            m = fmaxf(m, __shfl_xor(m, 16));
            m = fmaxf(m, __shfl_xor(m, 32));
            mx[qg] = m;                                    // tile maximum relative to mref, same in the 4 lanes of a query
        }
        if (__any(first || mx[0] > kRebase || mx[1] > kRebase)) {
#pragma unroll
            for (int qg = 0; qg < 2; ++qg) {
                const float d = (first || mx[qg] > kRebase) ? mx[qg] : 0.f;
                // first tile: O and l are still zero and d may be hugely negative (every logit of the tile below -88:
                // exp2(-d) = +inf and 0 * inf = NaN) -- nothing to rescale yet
                const float alpha = first ? 1.f : CLORA_EXP2(-d);
                mref[qg] += d;
                lrun[qg] *= alpha;
#pragma unroll
                for (int kt = 0; kt < 4; ++kt) s[kt][qg] -= d;
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) oacc[dt][qg] *= alpha;
            }
            first = false;
        }
#pragma unroll
        for (int qg = 0; qg < 2; ++qg) {
            float ps = 0.f;
#pragma unroll
            for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = CLORA_EXP2(s[kt][qg][r]);
                    s[kt][qg][r] = pv;
                    if (!ONES) ps += pv;
                }
            if (!ONES) lrun[qg] += ps;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const half8 pb0 = frag_from_acc(s[2 * j][0], s[2 * j + 1][0]);
            const half8 pb1 = frag_from_acc(s[2 * j][1], s[2 * j + 1][1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const half8 a = frag_tr<LDV>(Vs, dt * 16, j, g, l);
                oacc[dt][0] = mfma16(a, pb0, oacc[dt][0]);
                oacc[dt][1] = mfma16(a, pb1, oacc[dt][1]);
            }
        }
        CLORA_WAIT_VMCNT(0);                               // this wave's share of tile t+1 has landed ...
        __syncthreads();                                   // ... everyone's has, and tile t is fully consumed
        cur ^= 1;
    }
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
        float lt;
        if (ONES) {
            lt = __shfl(oacc[DT - 1][qg][0], 32 + li);     // row D = 16 (DT-1) + 8 of O^T: lane group 2, register 0
        } else {
            lt = lrun[qg];
            lt += __shfl_xor(lt, 16);
            lt += __shfl_xor(lt, 32);
        }
        const float inv = 1.0f / lt;
        const int q = q0 + qg * 16 + li;
        if (q < p.Nq) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = dt * 16 + 4 * g;
                if (d < D) {
                    half4v o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)(oacc[dt][qg][r] * inv);
                    st4(p.out + ((size_t)b * p.Nq + q) * p.ldo + h * D + d, o);
                }
            }
            if (g == 0 && p.lse) p.lse[((size_t)b * p.H + h) * p.Nq + q] = (mref[qg] + log2f(lt)) * kLn2;
        }
    }
}

// ------------------------------------------------------------------------------------------ dQ
template <int DP, int DT, int BKV, int NWV>
__global__ __launch_bounds__(NWV * 64, (DP <= 64 && NWV == 4 ? 2 : 1)) void attn_bwd_dq_kernel(AttnArgs p) {
    constexpr int LDK = DP + 16, KS = DP / 32, KT = BKV / 16, NP = BKV / 32;
    constexpr int TILE = 2 * BKV * LDK;                    // K tile + V tile; double buffered (see TileDma)
    __shared__ __attribute__((aligned(16))) half_t smem[2 * TILE];
    const int t = threadIdx.x, w = t >> 6, l = t & 63, g = l >> 4, li = l & 15;
    int bx, by;
    attn_block_ids(p, bx, by);
    const int b = by / p.H, h = by % p.H;
    const int q0 = bx * (NWV * 32) + w * 32;
    const int D = p.D;
    const half_t* kbase = p.k + (size_t)b * p.Nk * p.ldk + h * D;
    const half_t* vbase = p.v + (size_t)b * p.Nk * p.ldv + h * D;
    TileDma<BKV, LDK, NWV> dma;                            // K and V tiles share the slot -> (row, column) map
    dma.init(w, l, D);
    {
        const int rows0 = p.Nk < BKV ? p.Nk : BKV;
        dma.template issue<false>(kbase, p.ldk, rows0, smem, w);
        dma.template issue<false>(vbase, p.ldv, rows0, smem + BKV * LDK, w);
    }

    half8 qf[2][KS], dof[2][KS];
    float Lq[2], Dq[2];
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
        const int q = q0 + qg * 16 + li;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 32 + g * 8;
            const bool ok = q < p.Nq && d < D;
            qf[qg][ks] = ok ? ld8(p.q + ((size_t)b * p.Nq + q) * p.ldq + h * D + d) : zero8();
            dof[qg][ks] = ok ? ld8(p.dO + ((size_t)b * p.Nq + q) * p.lddo + h * D + d) : zero8();
        }
        const size_t si = ((size_t)b * p.H + h) * p.Nq + q;
        Lq[qg] = (q < p.Nq) ? p.lse_in[si] * kLog2e : 1.0e30f;
        // delta[q] = sum_d dO[q,d] * O[q,d], computed here from the dO fragment this lane already holds (the four
        // lanes li, li+16, li+32, li+48 own the four 8-wide d-groups of a row) and published for the dK/dV kernel,
        // which runs after this one on the same stream: no separate delta launch.
        float dl = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int d = ks * 32 + g * 8;
            if (q < p.Nq && d < D) {
                const half8 ov = ld8(p.o + ((size_t)b * p.Nq + q) * p.ldo + h * D + d);
#pragma unroll
                for (int e = 0; e < 8; ++e) dl += (float)dof[qg][ks][e] * (float)ov[e];
            }
        }
        dl += __shfl_xor(dl, 16);
        dl += __shfl_xor(dl, 32);
        Dq[qg] = dl;
        if (g == 0 && q < p.Nq) p.delta[si] = dl;
    }
    const float c = p.scale * kLog2e;
#pragma unroll
    for (int qg = 0; qg < 2; ++qg)          // scores leave the MFMA already multiplied by scale*log2(e)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) qf[qg][ks][e] = (half_t)((float)qf[qg][ks][e] * c);
    floatx4 acc[DT][2];
#pragma unroll
    for (int i = 0; i < DT; ++i) { acc[i][0] = zero4f(); acc[i][1] = zero4f(); }

    CLORA_WAIT_VMCNT(0);
    __syncthreads();                                       // tile 0 has landed for every wave
    int cur = 0;
    for (int kv0 = 0; kv0 < p.Nk; kv0 += BKV) {
        const int rows = (p.Nk - kv0 < BKV) ? p.Nk - kv0 : BKV;
        const half_t* Ks = smem + cur * TILE;              // K row-major: A operand of S^T as is, of dQ^T through frag_tr
        const half_t* Vs = Ks + BKV * LDK;
        if (kv0 + BKV < p.Nk) {                            // tile t+1 -> the other buffer
            const int nrows = (p.Nk - kv0 - BKV < BKV) ? p.Nk - kv0 - BKV : BKV;
            half_t* nb = smem + (cur ^ 1) * TILE;
            dma.template issue<false>(kbase + (size_t)(kv0 + BKV) * p.ldk, p.ldk, nrows, nb, w);
            dma.template issue<false>(vbase + (size_t)(kv0 + BKV) * p.ldv, p.ldv, nrows, nb + BKV * LDK, w);
        }

        // the accumulators start at -LSE / -delta of the lane's query column: the MFMAs deliver s - L and dP - delta
        // directly and dS^T is one exponential and one multiplication per element
        floatx4 s[KT][2], dp[KT][2];
#pragma unroll
        for (int kt = 0; kt < KT; ++kt) {
            s[kt][0] = splat4f(-Lq[0]); s[kt][1] = splat4f(-Lq[1]);
            dp[kt][0] = splat4f(-Dq[0]); dp[kt][1] = splat4f(-Dq[1]);
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                const half8 a = ld8(Ks + (kt * 16 + li) * LDK + ks * 32 + g * 8);
                const half8 av = ld8(Vs + (kt * 16 + li) * LDK + ks * 32 + g * 8);
                s[kt][0] = mfma16(a, qf[0][ks], s[kt][0]);
                s[kt][1] = mfma16(a, qf[1][ks], s[kt][1]);
                dp[kt][0] = mfma16(av, dof[0][ks], dp[kt][0]);
                dp[kt][1] = mfma16(av, dof[1][ks], dp[kt][1]);
            }
        if (rows == BKV) {                                 // full tile (all but the last one): no per-element key mask
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int qg = 0; qg < 2; ++qg)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        s[kt][qg][r] = CLORA_EXP2(s[kt][qg][r]) * dp[kt][qg][r];  // dS^T
        } else {
#pragma unroll
            for (int kt = 0; kt < KT; ++kt)
#pragma unroll
                for (int qg = 0; qg < 2; ++qg)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = (kt * 16 + 4 * g + r) < rows;
                        const float pv = ok ? CLORA_EXP2(s[kt][qg][r]) : 0.f;
                        s[kt][qg][r] = pv * dp[kt][qg][r];
                    }
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const half8 b0 = frag_from_acc(s[2 * j][0], s[2 * j + 1][0]);
            const half8 b1 = frag_from_acc(s[2 * j][1], s[2 * j + 1][1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const half8 a = frag_tr<LDK>(Ks, dt * 16, j, g, l);
                acc[dt][0] = mfma16(a, b0, acc[dt][0]);
                acc[dt][1] = mfma16(a, b1, acc[dt][1]);
            }
        }
        CLORA_WAIT_VMCNT(0);                               // tile t+1 has landed; tile t is fully consumed after the barrier
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int qg = 0; qg < 2; ++qg) {
        const int q = q0 + qg * 16 + li;
        if (q < p.Nq) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = dt * 16 + 4 * g;
                if (d < D) {
                    half4v o;
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = (half_t)(acc[dt][qg][r] * p.scale);
                    st4(p.dq + ((size_t)b * p.Nq + q) * p.lddq + h * D + d, o);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ dK, dV
// (256, 2): asking for 3 blocks per CU (168 registers) spills ~48 dwords of loop-invariant addresses into the loop --
// measured 1.75x SLOWER at d = 40 (profiles/r02_attn_ab.json note); with the DMA double buffer two blocks hide the latency.
template <int DP, int DT, int BQT, int NWV>
__global__ __launch_bounds__(NWV * 64, (DP <= 64 && NWV == 4 ? 2 : 1)) void attn_bwd_dkv_kernel(AttnArgs p) {
    constexpr int LDK = DP + 16, KS = DP / 32, QT = BQT / 16, NP = BQT / 32;
    constexpr int TILE = 2 * BQT * LDK;                    // Q tile + dO tile; double buffered (see TileDma)
    __shared__ __attribute__((aligned(16))) half_t smem[2 * TILE + 2 * 4 * BQT];
    float* LD_ = reinterpret_cast<float*>(smem + 2 * TILE);   // [2 buffers][-LSE*log2e (BQT) | -delta (BQT)]
    const int t = threadIdx.x, w = t >> 6, l = t & 63, g = l >> 4, li = l & 15;
    int bx, by;
    attn_block_ids(p, bx, by);
    const int b = by / p.H, h = by % p.H;
    const int k0 = bx * (NWV * 32) + w * 32;
    const int D = p.D;
    const float c = p.scale * kLog2e;

    const int q_beg = blockIdx.z * p.q_per_split;
    const int q_end = (q_beg + p.q_per_split < p.Nq) ? q_beg + p.q_per_split : p.Nq;
    const half_t* qbase = p.q + (size_t)b * p.Nq * p.ldq + h * D;
    const half_t* dobase = p.dO + (size_t)b * p.Nq * p.lddo + h * D;
    const size_t sbase = ((size_t)b * p.H + h) * p.Nq;
    TileDma<BQT, LDK, NWV> dma;                            // Q and dO tiles share the slot -> (row, column) map
    dma.init(w, l, D);
    // thread t < BQT carries -LSE*log2(e) of query row t of the tile in flight, thread BQT <= t < 2 BQT carries -delta:
    // they initialise the accumulators (below), so they are kept negated
    auto load_ld = [&](int qq, int rows) -> float {
        if (t >= 2 * BQT) return 0.f;
        const int r = t < BQT ? t : t - BQT;
        if (r >= rows) return t < BQT ? -1.0e30f : 0.f;
        return t < BQT ? -p.lse_in[sbase + qq + r] * kLog2e : -p.delta[sbase + qq + r];
    };
    float ldreg;
    {
        const int rows0 = (q_end - q_beg < BQT) ? q_end - q_beg : BQT;
        dma.template issue<false>(qbase + (size_t)q_beg * p.ldq, p.ldq, rows0, smem, w);
        dma.template issue<false>(dobase + (size_t)q_beg * p.lddo, p.lddo, rows0, smem + BQT * LDK, w);
        ldreg = load_ld(q_beg, rows0);
    }

    // K pre-multiplied by scale*log2(e) (as the forward does with Q): S leaves the MFMA ready for exp2 once the accumulator
    // carries -LSE; Q stays unscaled in LDS because dK = scale * dS^T Q contracts against it
    half8 kf[2][KS], vf[2][KS];
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int key = k0 + kg * 16 + li, d = ks * 32 + g * 8;
            const bool ok = key < p.Nk && d < D;
            half8 kv_ = ok ? ld8(p.k + ((size_t)b * p.Nk + key) * p.ldk + h * D + d) : zero8();
#pragma unroll
            for (int e = 0; e < 8; ++e) kv_[e] = (half_t)((float)kv_[e] * c);
            kf[kg][ks] = kv_;
            vf[kg][ks] = ok ? ld8(p.v + ((size_t)b * p.Nk + key) * p.ldv + h * D + d) : zero8();
        }
    floatx4 dkacc[DT][2], dvacc[DT][2];
#pragma unroll
    for (int i = 0; i < DT; ++i) { dkacc[i][0] = zero4f(); dkacc[i][1] = zero4f(); dvacc[i][0] = zero4f(); dvacc[i][1] = zero4f(); }

    CLORA_WAIT_VMCNT(0);
    if (t < 2 * BQT) LD_[t] = ldreg;
    __syncthreads();                                       // tile 0 (and its LSE / delta) is in LDS for every wave
    int cur = 0;
    for (int qq = q_beg; qq < q_end; qq += BQT) {
        const half_t* Qs = smem + cur * TILE;              // Q, dO row-major: A operands of S / dP as is, of dK / dV through frag_tr
        const half_t* dOs = Qs + BQT * LDK;
        const float* Ls = LD_ + cur * 2 * BQT;
        const float* Ds = Ls + BQT;
        const bool more = qq + BQT < q_end;
        if (more) {                                        // tile t+1 -> the other buffer
            const int nrows = (q_end - qq - BQT < BQT) ? q_end - qq - BQT : BQT;
            half_t* nb = smem + (cur ^ 1) * TILE;
            dma.template issue<false>(qbase + (size_t)(qq + BQT) * p.ldq, p.ldq, nrows, nb, w);
            dma.template issue<false>(dobase + (size_t)(qq + BQT) * p.lddo, p.lddo, nrows, nb + BQT * LDK, w);
            ldreg = load_ld(qq + BQT, nrows);
        }

        // accumulators start at -LSE / -delta of their query rows (kept negated in LDS; 4 consecutive rows per lane: one 16-byte
        // read each): the MFMAs deliver s*c - L and dP - delta, P and dS cost one v_exp_f32 and one v_mul_f32 per element
        floatx4 s[QT][2], dp[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            const floatx4 lv = *reinterpret_cast<const floatx4*>(Ls + qt * 16 + 4 * g);
            const floatx4 dv = *reinterpret_cast<const floatx4*>(Ds + qt * 16 + 4 * g);
            s[qt][0] = lv; s[qt][1] = lv; dp[qt][0] = dv; dp[qt][1] = dv;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                const half8 a = ld8(Qs + (qt * 16 + li) * LDK + ks * 32 + g * 8);
                const half8 ad = ld8(dOs + (qt * 16 + li) * LDK + ks * 32 + g * 8);
                s[qt][0] = mfma16(a, kf[0][ks], s[qt][0]);
                s[qt][1] = mfma16(a, kf[1][ks], s[qt][1]);
                dp[qt][0] = mfma16(ad, vf[0][ks], dp[qt][0]);
                dp[qt][1] = mfma16(ad, vf[1][ks], dp[qt][1]);
            }
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int kg = 0; kg < 2; ++kg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pv = CLORA_EXP2(s[qt][kg][r]);
                    s[qt][kg][r] = pv;                       // P
                    dp[qt][kg][r] *= pv;                     // dS
                }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const half8 p0 = frag_from_acc(s[2 * j][0], s[2 * j + 1][0]);
            const half8 p1 = frag_from_acc(s[2 * j][1], s[2 * j + 1][1]);
            const half8 d0 = frag_from_acc(dp[2 * j][0], dp[2 * j + 1][0]);
            const half8 d1 = frag_from_acc(dp[2 * j][1], dp[2 * j + 1][1]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const half8 ao = frag_tr<LDK>(dOs, dt * 16, j, g, l);
                const half8 aq = frag_tr<LDK>(Qs, dt * 16, j, g, l);
                dvacc[dt][0] = mfma16(ao, p0, dvacc[dt][0]);
                dvacc[dt][1] = mfma16(ao, p1, dvacc[dt][1]);
                dkacc[dt][0] = mfma16(aq, d0, dkacc[dt][0]);
                dkacc[dt][1] = mfma16(aq, d1, dkacc[dt][1]);
            }
        }
        CLORA_WAIT_VMCNT(0);                               // tile t+1 and its LSE / delta values have arrived
        if (more && t < 2 * BQT) LD_[(cur ^ 1) * 2 * BQT + t] = ldreg;
        __syncthreads();                                   // ... for every wave, and tile t is fully consumed
        cur ^= 1;
    }
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
        const int key = k0 + kg * 16 + li;
        if (key < p.Nk) {
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const int d = dt * 16 + 4 * g;
                if (d < D) {
                    if (p.nsplit > 1) {                    // this split's own fp32 slab (plain 16-byte stores: no atomics, deterministic)
                        const size_t o = ((size_t)b * p.Nk + key) * (p.H * D) + h * D + d;
                        const size_t plane = (size_t)p.B * p.Nk * p.H * D;
                        float* dst = p.acc32 + (size_t)blockIdx.z * 2 * plane + o;
                        *reinterpret_cast<floatx4*>(dst) = dkacc[dt][kg] * p.scale;
                        *reinterpret_cast<floatx4*>(dst + plane) = dvacc[dt][kg];
                    } else {
                        half4v ok_, ov;
#pragma unroll
                        for (int r = 0; r < 4; ++r) { ok_[r] = (half_t)(dkacc[dt][kg][r] * p.scale); ov[r] = (half_t)dvacc[dt][kg][r]; }
                        st4(p.dk + ((size_t)b * p.Nk + key) * p.lddk + h * D + d, ok_);
                        st4(p.dv + ((size_t)b * p.Nk + key) * p.lddv + h * D + d, ov);
                    }
                }
            }
        }
    }
}

// split path epilogue: fold the per-split fp32 slabs [nsplit][dk | dv][B, Nk, H*D] (fixed order) -> fp16 dk / dv (row-strided).
// (Round 1 combined the splits with fp32 atomics into one zeroed buffer: 3 M atomics on 0.2 M addresses made the cross-attention
// dK/dV launch 87 us at N = 4096 -- profiles/r02_attn_trace_by_grid.txt.)
__global__ __launch_bounds__(256) void attn_dkv_convert_kernel(AttnArgs p) {
    const int HD = p.H * p.D;
    const size_t rows = (size_t)p.B * p.Nk, total = rows * (HD / 4), plane = rows * HD;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / (HD / 4);
        const int c = (int)(i - row * (HD / 4)) * 4;
        floatx4 a = zero4f(), bb = zero4f();
        const float* q0 = p.acc32 + row * HD + c;
        int z = 0;
        for (; z + 4 <= p.nsplit; z += 4) {                      // four slab pairs in flight (was load, load, wait per split: up to 16
            floatx4 ta[4], tb[4];                                // dependent round trips per output chunk); same summation order
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float* q = q0 + (size_t)(z + u) * 2 * plane;
                ta[u] = *reinterpret_cast<const floatx4*>(q);
                tb[u] = *reinterpret_cast<const floatx4*>(q + plane);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) { a += ta[u]; bb += tb[u]; }
        }
        for (; z < p.nsplit; ++z) {
            const float* q = q0 + (size_t)z * 2 * plane;
            a += *reinterpret_cast<const floatx4*>(q);
            bb += *reinterpret_cast<const floatx4*>(q + plane);
        }
        half4v ka, va;
#pragma unroll
        for (int r = 0; r < 4; ++r) { ka[r] = (half_t)a[r]; va[r] = (half_t)bb[r]; }
        st4(p.dk + row * p.lddk + c, ka);
        st4(p.dv + row * p.lddv + c, va);
    }
}

constexpr bool kAttnFwd16 = false;      // the automatic choice of the 16-wave forward block: off until it has been measured
template <int DP, int DT>
int launch_fwd(const AttnArgs& a, hipStream_t s) {
    // wider blocks when there are enough queries to keep the grid full (A/B: clora_set_option("attn_fwd_waves", 4 | 6 | 8))
    const int forced = clora_option(CLORA_OPT_ATTN_FWD_WAVES);
    int nw = forced ? forced : ((DP <= 64 && (long)clora_cdiv(a.Nq, 256) * a.B * a.H >= 512) ? 8 : 4);   // measured: 6 waves lose, 8 win 10-18 %
    // 16 waves = 512 queries per block (half the K/V tile stream per flop again) once even those blocks fill the chip twice over: the
    // batch-32 sampler's level-0 self-attention (2,048 blocks); "attn_fwd_waves" = 16 forces it (A/B)
    if (!forced && DP <= 64 && a.Nk >= 1024 && (long)clora_cdiv(a.Nq, 512) * a.B * a.H >= 1024) nw = kAttnFwd16 ? 16 : nw;
    if (DP > 64 && nw != 4) nw = 4;                         // larger head dims keep the 4-wave block (LDS / registers)
    const bool ones = a.D == DT * 16 - 8;
    const dim3 grid(clora_cdiv(a.Nq, nw * 32), a.B * a.H);
#define CLORA_FWD_LAUNCH(NWV)                                                                                         \
    do {                                                                                                              \
        if (ones) hipLaunchKernelGGL((attn_fwd_kernel<DP, DT, true, NWV>), grid, dim3(NWV * 64), 0, s, a);            \
        else hipLaunchKernelGGL((attn_fwd_kernel<DP, DT, false, NWV>), grid, dim3(NWV * 64), 0, s, a);                \
    } while (0)
    if constexpr (DP <= 64) {
        if (nw == 16) CLORA_FWD_LAUNCH(16);
        else if (nw == 8) CLORA_FWD_LAUNCH(8);
        else if (nw == 6) CLORA_FWD_LAUNCH(6);
        else CLORA_FWD_LAUNCH(4);
    } else {
        CLORA_FWD_LAUNCH(4);
    }
#undef CLORA_FWD_LAUNCH
    return clora_check_launch();
}
template <int DP, int DT>
int launch_fwd_causal(const AttnArgs& a, hipStream_t s) {
    const dim3 grid(clora_cdiv(a.Nq, 128), a.B * a.H);
    if (a.D == DT * 16 - 8) hipLaunchKernelGGL((attn_fwd_kernel<DP, DT, true, 4, true>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((attn_fwd_kernel<DP, DT, false, 4, true>), grid, dim3(256), 0, s, a);
    return clora_check_launch();
}
template <int DP, int DT, int BT>
int launch_bwd(const AttnArgs& a, hipStream_t s) {
    // 8-wave blocks (one per CU: the same 8 waves as two 4-wave blocks, half the K/V resp. Q/dO stream per flop) when the grids stay
    // full; A/B: clora_set_option("attn_bwd_waves", 4 | 8)
    const int forced = clora_option(CLORA_OPT_ATTN_BWD_WAVES);
    const bool wide_q = forced ? forced == 8 : (DP <= 64 && (long)clora_cdiv(a.Nq, 256) * a.B * a.H >= 256);
    const bool wide_k = forced ? forced == 8 : (DP <= 64 && a.nsplit == 1 && (long)clora_cdiv(a.Nk, 256) * a.B * a.H >= 256);
    if constexpr (DP <= 64) {
        if (wide_q) hipLaunchKernelGGL((attn_bwd_dq_kernel<DP, DT, BT, 8>), dim3(clora_cdiv(a.Nq, 256), a.B * a.H), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((attn_bwd_dq_kernel<DP, DT, BT, 4>), dim3(clora_cdiv(a.Nq, 128), a.B * a.H), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((attn_bwd_dq_kernel<DP, DT, BT, 4>), dim3(clora_cdiv(a.Nq, 128), a.B * a.H), dim3(256), 0, s, a);
    }
    int rc = clora_check_launch();
    if (rc != CLORA_OK) return rc;
    if constexpr (DP <= 64) {
        if (wide_k && a.nsplit == 1) hipLaunchKernelGGL((attn_bwd_dkv_kernel<DP, DT, BT, 8>), dim3(clora_cdiv(a.Nk, 256), a.B * a.H, 1), dim3(512), 0, s, a);
        else hipLaunchKernelGGL((attn_bwd_dkv_kernel<DP, DT, BT, 4>), dim3(clora_cdiv(a.Nk, 128), a.B * a.H, a.nsplit), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((attn_bwd_dkv_kernel<DP, DT, BT, 4>), dim3(clora_cdiv(a.Nk, 128), a.B * a.H, a.nsplit), dim3(256), 0, s, a);
    }
    if (a.nsplit > 1) {
        const size_t total = (size_t)a.B * a.Nk * (a.H * a.D / 4);
        int blocks = (int)((total + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(attn_dkv_convert_kernel, dim3(blocks), dim3(256), 0, s, a);
    }
    return clora_check_launch();
}

bool bad_dims(int B, int H, int Nq, int Nk, int D) {
    return B <= 0 || H <= 0 || Nq <= 0 || Nk <= 0 || D <= 0 || (D & 7) || D > 160;
}

}  // namespace

#define CLORA_ATTN_DISPATCH(FN, ...)                                  \
    do {                                                              \
        const int dt_ = (a.D + 15) / 16;                              \
        if (a.D <= 32) return FN<32, 2 __VA_ARGS__>(a, s);            \
        if (a.D <= 64 && dt_ <= 3) return FN<64, 3 __VA_ARGS__>(a, s);\
        if (a.D <= 64) return FN<64, 4 __VA_ARGS__>(a, s);            \
        if (a.D <= 96 && dt_ <= 5) return FN<96, 5 __VA_ARGS__>(a, s);\
        if (a.D <= 96) return FN<96, 6 __VA_ARGS__>(a, s);            \
    } while (0)

extern "C" int clora_attn_fwd_f16(const clora_half* q, int ldq, const clora_half* k, int ldk, const clora_half* v,
                                  int ldv, clora_half* o, int ldo, float* lse, int B, int H, int Nq, int Nk, int D,
                                  float scale, void* stream) {
    if (!q || !k || !v || !o || bad_dims(B, H, Nq, Nk, D) || ((ldq | ldk | ldv | ldo) & 7)) return CLORA_ERR_ARG;
    AttnArgs a = AttnArgs();
    a.xcd_heads = clora_xcd_policy() != 0;
    a.q = (const half_t*)q; a.k = (const half_t*)k; a.v = (const half_t*)v; a.out = (half_t*)o; a.lse = lse;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.scale = scale;
    hipStream_t s = (hipStream_t)stream;
    CLORA_ATTN_DISPATCH(launch_fwd);
    if (D <= 128) return launch_fwd<128, 8>(a, s);
    return launch_fwd<160, 10>(a, s);
}

extern "C" int clora_attn_fwd_causal_f16(const clora_half* q, int ldq, const clora_half* k, int ldk, const clora_half* v,
                                         int ldv, clora_half* o, int ldo, int B, int H, int N, int D, float scale, void* stream) {
    if (!q || !k || !v || !o || bad_dims(B, H, N, N, D) || D > 64 || ((ldq | ldk | ldv | ldo) & 7)) return CLORA_ERR_ARG;
    AttnArgs a = AttnArgs();
    a.q = (const half_t*)q; a.k = (const half_t*)k; a.v = (const half_t*)v; a.out = (half_t*)o; a.lse = nullptr;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.B = B; a.H = H; a.Nq = N; a.Nk = N; a.D = D; a.scale = scale;
    hipStream_t s = (hipStream_t)stream;
    if (D <= 32) return launch_fwd_causal<32, 2>(a, s);
    if ((D + 15) / 16 <= 3) return launch_fwd_causal<64, 3>(a, s);
    return launch_fwd_causal<64, 4>(a, s);
}

extern "C" int clora_attn_bwd_f16(const clora_half* q, int ldq, const clora_half* k, int ldk, const clora_half* v,
                                  int ldv, const clora_half* o, int ldo, const clora_half* dO, int lddo,
                                  const float* lse, float* delta, clora_half* dq, int lddq, clora_half* dk, int lddk,
                                  clora_half* dv, int lddv, int B, int H, int Nq, int Nk, int D, float scale,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    if (!q || !k || !v || !o || !dO || !lse || !delta || !dq || !dk || !dv || bad_dims(B, H, Nq, Nk, D) ||
        ((ldq | ldk | ldv | ldo | lddo | lddq | lddk | lddv) & 7))
        return CLORA_ERR_ARG;
    AttnArgs a = AttnArgs();
    a.xcd_heads = clora_xcd_policy() != 0;
    a.q = (const half_t*)q; a.k = (const half_t*)k; a.v = (const half_t*)v; a.o = (const half_t*)o;
    a.dO = (const half_t*)dO; a.lse_in = lse; a.delta = delta;
    a.dq = (half_t*)dq; a.dk = (half_t*)dk; a.dv = (half_t*)dv;
    a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo; a.lddo = lddo; a.lddq = lddq; a.lddk = lddk; a.lddv = lddv;
    a.B = B; a.H = H; a.Nq = Nq; a.Nk = Nk; a.D = D; a.scale = scale;
    hipStream_t s = (hipStream_t)stream;
    // few key blocks (cross-attention, Nk = 77): split the query loop so the dK/dV kernel fills the chip
    a.nsplit = 1;
    a.q_per_split = Nq;
    const long kv_blocks = (long)clora_cdiv(Nk, 128) * B * H;
    if (kv_blocks < 256 && Nq >= 512) {
        int ns = (int)(512 / kv_blocks);
        if (ns > Nq / 128) ns = Nq / 128;
        const size_t slab = (size_t)2 * B * Nk * H * D * sizeof(float);             // one split's dK and dV
        if (workspace && ns > (int)(workspace_bytes / slab)) ns = (int)(workspace_bytes / slab);
        if (ns > 1 && workspace && (D & 3) == 0) {
            a.q_per_split = clora_cdiv(clora_cdiv(Nq, ns), 64) * 64;
            a.nsplit = clora_cdiv(Nq, a.q_per_split);
            a.acc32 = (float*)workspace;
        }
    }
    // delta = rowsum(dO * O) is produced by the dQ kernel's prologue and consumed by the dK/dV kernel launched after it
#define COMMA_64 , 64
#define COMMA_32 , 32
    CLORA_ATTN_DISPATCH(launch_bwd, COMMA_64);
    if (D <= 128) return launch_bwd<128, 8, 32>(a, s);
    return launch_bwd<160, 10, 32>(a, s);
}
