// clora_norm.hip -- GroupNorm(+SiLU) and LayerNorm, forward and backward, NHWC fp16 / fp32 statistics.
//
// HBM-bound kernels: every pass reads/writes whole 16-byte channel chunks of
// full rows, so global traffic is coalesced regardless of how channels split into groups
// (C/32 = 10 channels per group at C=320 is not a multiple of the 8-channel load width).
//   GroupNorm forward : partial (sum, sumsq per (b, row-chunk, group), deterministic block reduce) -> apply
//                       (each block re-folds the partials of its batch element and keeps its per-channel
//                       scale/shift in registers: two launches, no atomics, no coefficient table)
//   GroupNorm backward: partial (s1 = sum dy*gamma, s2 = sum dy*gamma*xhat) -> apply (dx = k1*dy' + k2*x + k3)
//   LayerNorm         : one wave per row, the row lives in registers (two-pass variance), no workspace.
#include <stdlib.h>
#include "clora_common.h"
#include "../../include/clora.h"
#include "clora_epilogue.h"

namespace {

constexpr int kMaxCols = 2;   // column chunks (of 8 channels) owned by one thread in the GroupNorm passes: C <= 4096
constexpr int kLnCols = 4;    // LayerNorm: chunks per lane, C <= 2048

struct GnArgs {
    const half_t* x;
    const half_t* dy;
    const half_t* dres;   // bwd, optional: a second gradient of x (residual / shortcut branch) added into dx
    half_t* y;      // fwd: output, bwd: dx
    const float* gamma;
    const float* beta;
    float* stats;   // [B, G, 2]  (mean, rstd)
    float* partial; // [B, nchunk, G, 2]
    float* chpart;  // bwd, trainable affine: [B, nchunk, C, 2] per-channel (sum dyp, sum dyp*xhat)
    float* dgamma;
    float* dbeta;
    int B, HW, C, G, nchunk, rows_per_chunk;
    int accumulate_params;   // dgamma / dbeta are += (they are the parameters' .grad) instead of written
    int params_in_apply;     // bwd, trainable affine: the first cdiv(C, 32) blocks of the apply launch also fold chpart into dgamma / dbeta
                             // (round 6: the hint encoder's 22 gn_bwd_params launches per step are gone)
    int nslab, CS;  // channel slabs (whole groups, multiple of 8 channels) = blockIdx.y: fills the chip at low resolution
    float eps;
    int fuse_silu;
    // round 6 (include/clora.h clora_groupnorm_*_ex)
    const half_t* x2;     // fwd: channels [Ca, C) of the input live in x2 [B*HW, C - Ca], x is [B*HW, Ca] (nullptr: x is [B*HW, C])
    int Ca;
    half_t* xcopy;        // fwd, optional: the raw input as one contiguous [B*HW, C] tensor (concatenated / finished)
    half_t* y2;           // bwd, optional: dx of channels [Ca, C) goes to y2 [B*HW, C - Ca], y holds [B*HW, Ca]
    const float* fin_partial;   // deferred split-K producer of x (fwd) / dy (bwd): slabs [fin_splits][B*HW][C] + its epilogue
    int fin_splits;
    clora_epilogue_t fin_epi;
    // team kernels (one launch for the 64x64 / 32x32 maps): the caller's persistent exchange state (clora_groupnorm_team_state_bytes)
    unsigned long long* tm_gran;   // [kTeamUnits][kTeamMembers][kTeamGran] 8-byte {epoch, fp32 bits} granules
    unsigned* tm_gen;              // [kTeamUnits] epoch of the last completed exchange of a unit
    unsigned* tm_err;              // sticky: an exchange gave up (a member never published)
    int tm_nb, tm_rpb;             // members per unit, rows per member
    int tm_probe;                  // emulator only (blocks run one after the other): publish, do not gather
    unsigned tm_spin;
};

constexpr int kTeamUnits = 32, kTeamMembers = 256, kTeamGran = 64, kTeamBlocks = 256, kTeamNT = 512;
constexpr size_t kTeamHeader = 4096;                     // bytes: error word at 0, generation words at 256
constexpr size_t kTeamStateBytes = kTeamHeader + (size_t)kTeamUnits * kTeamMembers * kTeamGran * 8;

// where a thread's 8-channel chunk (first channel ch0, a multiple of 8; Ca % 8 == 0) of the input lives: base pointer at row 0 of batch
// element 0 and the row pitch in halves -- one tensor, or the two halves of a channel concatenation
struct GnCol { const half_t* p; int pitch; };
__device__ __forceinline__ GnCol gn_in_col(const GnArgs& a, int ch0) {
    GnCol c;
    if (a.x2 && ch0 >= a.Ca) { c.p = a.x2 + (ch0 - a.Ca); c.pitch = a.C - a.Ca; }
    else { c.p = a.x + ch0; c.pitch = a.x2 ? a.Ca : a.C; }
    return c;
}
struct GnColW { half_t* p; int pitch; };
__device__ __forceinline__ GnColW gn_dx_col(const GnArgs& a, int ch0) {
    GnColW c;
    if (a.y2 && ch0 >= a.Ca) { c.p = a.y2 + (ch0 - a.Ca); c.pitch = a.C - a.Ca; }
    else { c.p = a.y + ch0; c.pitch = a.y2 ? a.Ca : a.C; }
    return c;
}

// thread -> (row lane, first column chunk, column stride)
__device__ __forceinline__ void gn_thread_map(int t, int CH, int& rl, int& nrl, int& c0, int& cstep, bool& active) {
    if (CH >= 256) { rl = 0; nrl = 1; c0 = t; cstep = 256; active = true; }
    else { nrl = 256 / CH; rl = t / CH; c0 = t - rl * CH; cstep = CH; active = rl < nrl; }
}

// Row loops of the four GroupNorm passes.  A thread owns rows r_beg + rl + it * nrl; written as `for (r = ...; r < r_end; r += nrl)`
// the trip count differs per thread, every unrolled copy gets its own exit branch and the compiler emits load -> s_waitcnt
// vmcnt(0) -> use per row: ONE 16-byte load in flight per thread (seen in the ISA of all four kernels; 6-10 dependent
// L2 / fabric round trips per launch).  Here the trip count nit is block-uniform, rows past r_end re-read row r_beg
// (a valid address) and are masked out of the sums / not stored, and kGnU rows are loaded before the first one is used.
// The per-thread order of the accumulations is unchanged, so the results are bit-identical to the branchy loops.
// NJ = column chunks per thread: 2 only for slabs wider than 2048 channels (gn_plan keeps slabs <= 2048 when it can).
constexpr int kGnU = 4;

// Deterministic block reduction: per-thread per-channel pairs -> LDS [nrl][C][2] -> per-channel totals in
// red[0][c][*] -> per-group sums (optionally weighted by gamma) written to out_group[g*2 + {0,1}].
template <int NJ>
__device__ __forceinline__ void gn_block_reduce(float* red, const float (&a)[NJ][8], const float (&b)[NJ][8],
                                                int t, int C, int G, int rl, int nrl, int c0, int cstep, bool active,
                                                const float* weight, float* out_group, float* out_channel) {
    const int CH = C / 8, cpg = C / G;
    if (active) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int cc = c0 + j * cstep;
            if (cc < CH && (j == 0 || cstep == 256)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[((size_t)rl * C + cc * 8 + e) * 2] = a[j][e];
                    red[((size_t)rl * C + cc * 8 + e) * 2 + 1] = b[j][e];
                }
            }
        }
    }
    __syncthreads();
    for (int c = t; c < C; c += 256) {   // fold the row lanes (fixed order)
        float sa = 0.f, sb = 0.f;
        for (int r = 0; r < nrl; ++r) { sa += red[((size_t)r * C + c) * 2]; sb += red[((size_t)r * C + c) * 2 + 1]; }
        red[c * 2] = sa; red[c * 2 + 1] = sb;
        if (out_channel) { out_channel[c * 2] = sa; out_channel[c * 2 + 1] = sb; }
    }
    __syncthreads();
    if (t < G) {
        float sa = 0.f, sb = 0.f;
        for (int c = t * cpg; c < (t + 1) * cpg; ++c) {
            const float w = weight ? weight[c] : 1.0f;
            sa += w * red[c * 2]; sb += w * red[c * 2 + 1];
        }
        out_group[t * 2] = sa; out_group[t * 2 + 1] = sb;
    }
}

constexpr int kRedFloats = 2 * 4096;   // LDS: max(nrl*C, C) * 2 floats with nrl*C <= 2048 for CH < 256

// ---- forward, pass 1
template <int NJ, int UF = 1>
__global__ __launch_bounds__(256) void gn_fwd_partial_kernel(GnArgs p) {
    constexpr int kGnU = ::kGnU * UF;                        // rows loaded before the first is used (option "gn_unroll": UF = 2)
    __shared__ float red[kRedFloats];
    const int t = threadIdx.x, b = blockIdx.z, chunk = blockIdx.x, cb = blockIdx.y * p.CS;
    const int CH = p.CS / 8, gps = p.G / p.nslab;
    int rl, nrl, c0, cstep; bool active;
    gn_thread_map(t, CH, rl, nrl, c0, cstep, active);
    const int r_beg = chunk * p.rows_per_chunk;
    const int r_end = (r_beg + p.rows_per_chunk < p.HW) ? r_beg + p.rows_per_chunk : p.HW;
    const int nit = (r_end - r_beg + nrl - 1) / nrl;              // block-uniform
    float s[NJ][8], q[NJ][8];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[j][e] = 0.f; q[j][e] = 0.f; }
    if (active) {
        GnCol col[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) { const int cc = c0 + j * cstep; col[j] = gn_in_col(p, cb + (cc < CH ? cc : c0) * 8); }
        const size_t brow = (size_t)b * p.HW;
        for (int it0 = 0; it0 < nit; it0 += kGnU) {
            half8 v[kGnU][NJ];
#pragma unroll
            for (int u = 0; u < kGnU; ++u) {                     // every load of the batch first ...
                const int r = r_beg + rl + (it0 + u) * nrl;
                const size_t row = brow + (size_t)(r < r_end ? r : r_beg);
#pragma unroll
                for (int j = 0; j < NJ; ++j) v[u][j] = ld8(col[j].p + row * col[j].pitch);
            }
#pragma unroll
            for (int u = 0; u < kGnU; ++u) {                     // ... then the sums, rows past the chunk masked to zero
                const bool ok = r_beg + rl + (it0 + u) * nrl < r_end;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    CLORA_KEEP(v[u][j]);
                    if (c0 + j * cstep < CH) {                   // thread-constant
#pragma unroll
                        for (int e = 0; e < 8; ++e) { const float f = ok ? (float)v[u][j][e] : 0.f; s[j][e] += f; q[j][e] += f * f; }
                    }
                }
            }
        }
    }
    gn_block_reduce<NJ>(red, s, q, t, p.CS, gps, rl, nrl, c0, cstep, active, nullptr,
                        p.partial + (((size_t)b * p.nchunk + chunk) * p.G + blockIdx.y * gps) * 2, nullptr);
}

// ---- backward, pass 1: per channel a1 = sum dyp, a2 = sum dyp*xhat over this block's rows
template <int NJ, int UF = 1>
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(GnArgs p) {
    __shared__ float red[kRedFloats];
    const int t = threadIdx.x, b = blockIdx.z, chunk = blockIdx.x, cb = blockIdx.y * p.CS;
    const int CH = p.CS / 8, cpg = p.C / p.G, gps = p.G / p.nslab;
    int rl, nrl, c0, cstep; bool active;
    gn_thread_map(t, CH, rl, nrl, c0, cstep, active);
    const int r_beg = chunk * p.rows_per_chunk;
    const int r_end = (r_beg + p.rows_per_chunk < p.HW) ? r_beg + p.rows_per_chunk : p.HW;
    const int nit = (r_end - r_beg + nrl - 1) / nrl;              // block-uniform
    float a1[NJ][8], a2[NJ][8];
    float kmean[NJ][8], krstd[NJ][8], kg[NJ][8], kb[NJ][8];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            a1[j][e] = 0.f; a2[j][e] = 0.f;
            const int cc = c0 + j * cstep;
            const int ch = cb + (cc < CH ? cc : 0) * 8 + e;
            const float* st = p.stats + ((size_t)b * p.G + ch / cpg) * 2;
            kmean[j][e] = st[0]; krstd[j][e] = st[1]; kg[j][e] = p.gamma[ch]; kb[j][e] = p.beta[ch];
        }
    if (active) {
        const size_t boff = (size_t)b * p.HW * p.C + cb;
        constexpr int U = 2 * UF;                                // two tensors per row: 4 * NJ * UF loads in flight
        for (int it0 = 0; it0 < nit; it0 += U) {
            half8 xv[U][NJ], gv[U][NJ];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int r = r_beg + rl + (it0 + u) * nrl;
                const size_t off = boff + (size_t)(r < r_end ? r : r_beg) * p.C;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    const int cc = c0 + j * cstep, cl = (cc < CH ? cc : c0) * 8;
                    xv[u][j] = ld8(p.x + off + cl);
                    gv[u][j] = ld8(p.dy + off + cl);
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool ok = r_beg + rl + (it0 + u) * nrl < r_end;
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    CLORA_KEEP(xv[u][j]);
                    CLORA_KEEP(gv[u][j]);
                    if (c0 + j * cstep < CH) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const float xh = ((float)xv[u][j][e] - kmean[j][e]) * krstd[j][e];
                            float d = ok ? (float)gv[u][j][e] : 0.f;
                            if (p.fuse_silu) d *= dsilu_f(xh * kg[j][e] + kb[j][e]);
                            a1[j][e] += d;
                            a2[j][e] += d * xh;
                        }
                    }
                }
            }
        }
    }
    gn_block_reduce<NJ>(red, a1, a2, t, p.CS, gps, rl, nrl, c0, cstep, active, p.gamma + cb,
                        p.partial + (((size_t)b * p.nchunk + chunk) * p.G + blockIdx.y * gps) * 2,
                        p.chpart ? p.chpart + (((size_t)b * p.nchunk + chunk) * p.C + cb) * 2 : nullptr);
}

// ---- fused finalize + apply (forward): every block re-folds the chunk partials of its batch element (a few KB
// from L2) into group statistics, derives the per-channel scale/shift of ITS OWN columns into registers and
// streams its rows: no coefficient table, no separate finalize launch.
__device__ __forceinline__ void gn_fold_groups(const GnArgs& p, int b, int t, float* out2, float inv_n) {
    // 8 threads per group (4 when G > 32), each folding every 8th (4th) chunk partial with EIGHT loads in flight: the partials sit
    // in L2 and a load -> wait -> add loop paid one L2 round trip per partial (32 of them at 64x64 maps: ~8 us of the apply
    // kernels' 10-22 us -- profiles/r02_step_trace_by_grid.txt).  Fixed order: still deterministic.
    const int sh = p.G > 32 ? 2 : 3, parts = 1 << sh;
    const int g = t >> sh, part = t & (parts - 1);
    float s = 0.f, q = 0.f;
    if (g < p.G) {
        const float* base = p.partial + ((size_t)b * p.nchunk * p.G + g) * 2;
        const size_t cs = (size_t)p.G * 2;                       // floats between consecutive chunks
        int c = part;
        for (; c + 7 * parts < p.nchunk; c += 8 * parts) {
            float v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float* pp = base + (size_t)(c + u * parts) * cs; v0[u] = pp[0]; v1[u] = pp[1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += v0[u]; q += v1[u]; }
        }
        for (; c < p.nchunk; c += parts) { const float* pp = base + (size_t)c * cs; s += pp[0]; q += pp[1]; }
    }
    s += __shfl_xor(s, 1); q += __shfl_xor(q, 1);
    s += __shfl_xor(s, 2); q += __shfl_xor(q, 2);
    if (sh == 3) { s += __shfl_xor(s, 4); q += __shfl_xor(q, 4); }
    if (g < p.G && part == 0) { out2[g * 2] = s * inv_n; out2[g * 2 + 1] = q * inv_n; }
}

template <int NJ, int UF = 1>
__global__ __launch_bounds__(256) void gn_fwd_apply2_kernel(GnArgs p) {
    constexpr int kGnU = ::kGnU * UF;
    __shared__ float mr[64 * 2];
    const int t = threadIdx.x, b = blockIdx.z, chunk = blockIdx.x, cb = blockIdx.y * p.CS;
    const int CH = p.CS / 8, cpg = p.C / p.G;
    gn_fold_groups(p, b, t, mr, 1.0f / ((float)p.HW * (float)cpg));
    __syncthreads();
    if (t < p.G) {                                  // (E[x], E[x^2]) -> (mean, rstd)
        const float mean = mr[t * 2];
        float var = mr[t * 2 + 1] - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = rsqrtf(var + p.eps);
        mr[t * 2 + 1] = rstd;
        if (chunk == 0 && blockIdx.y == 0) { p.stats[((size_t)b * p.G + t) * 2] = mean; p.stats[((size_t)b * p.G + t) * 2 + 1] = rstd; }
    }
    __syncthreads();
    int rl, nrl, c0, cstep; bool active;
    gn_thread_map(t, CH, rl, nrl, c0, cstep, active);
    float sc[NJ][8], sh[NJ][8];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cc = c0 + j * cstep;
            const int ch = cb + (cc < CH ? cc : 0) * 8 + e, g = ch / cpg;
            sc[j][e] = mr[g * 2 + 1] * p.gamma[ch];
            sh[j][e] = p.beta[ch] - mr[g * 2] * sc[j][e];
        }
    if (!active) return;
    const int r_beg = chunk * p.rows_per_chunk;
    const int r_end = (r_beg + p.rows_per_chunk < p.HW) ? r_beg + p.rows_per_chunk : p.HW;
    const int nit = (r_end - r_beg + nrl - 1) / nrl;              // block-uniform
    const size_t boff = (size_t)b * p.HW * p.C + cb;
    GnCol col[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) { const int cc = c0 + j * cstep; col[j] = gn_in_col(p, cb + (cc < CH ? cc : c0) * 8); }
    const size_t brow = (size_t)b * p.HW;
    for (int it0 = 0; it0 < nit; it0 += kGnU) {
        half8 v[kGnU][NJ];
#pragma unroll
        for (int u = 0; u < kGnU; ++u) {
            const int r = r_beg + rl + (it0 + u) * nrl;
            const size_t row = brow + (size_t)(r < r_end ? r : r_beg);
#pragma unroll
            for (int j = 0; j < NJ; ++j) v[u][j] = ld8(col[j].p + row * col[j].pitch);
        }
#pragma unroll
        for (int u = 0; u < kGnU; ++u) {
            const int r = r_beg + rl + (it0 + u) * nrl;
            const size_t off = boff + (size_t)(r < r_end ? r : r_beg) * p.C;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int cc = c0 + j * cstep;
                CLORA_KEEP(v[u][j]);
                half8 o;                                         // computed unconditionally (keeps the loads above the branch) ...
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float yv = (float)v[u][j][e] * sc[j][e] + sh[j][e];
                    if (p.fuse_silu) yv = silu_f(yv);
                    o[e] = (half_t)yv;
                }
                if (r < r_end && cc < CH) {                      // ... only the stores are conditional
                    st8(p.y + off + cc * 8, o);
                    if (p.xcopy) st8(p.xcopy + off + cc * 8, v[u][j]);      // the concatenated input, once, for the shortcut / backward
                }
            }
        }
    }
}

// ---- backward, trainable affine: dgamma[c] = sum_{b,chunk} a2, dbeta[c] = sum a1  (fixed order: 8 interleaved
// partial sums per channel folded through LDS; one block = 32 channels)
__device__ __forceinline__ void gn_bwd_params_block(const GnArgs& p, int blk, float* red) {     // red: 512 floats of LDS
    const int t = threadIdx.x, cl = t & 31, part = t >> 5;
    const int c = blk * 32 + cl;
    float sa = 0.f, sb = 0.f;
    if (c < p.C) {
        const int n = p.B * p.nchunk;
        int i = part;
        for (; i + 56 < n; i += 64) {                            // 8 loads in flight (was one L2 round trip per partial, 64 of them)
            float v0[8], v1[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float* pp = p.chpart + ((size_t)(i + 8 * u) * p.C + c) * 2; v0[u] = pp[0]; v1[u] = pp[1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) { sa += v0[u]; sb += v1[u]; }
        }
        for (; i < n; i += 8) {
            const float* pp = p.chpart + ((size_t)i * p.C + c) * 2;
            sa += pp[0]; sb += pp[1];
        }
    }
    red[t * 2] = sa; red[t * 2 + 1] = sb;
    __syncthreads();
    if (part == 0 && c < p.C) {
        for (int q = 1; q < 8; ++q) { sa += red[(q * 32 + cl) * 2]; sb += red[(q * 32 + cl) * 2 + 1]; }
        if (p.accumulate_params) { p.dbeta[c] += sa; p.dgamma[c] += sb; }
        else { p.dbeta[c] = sa; p.dgamma[c] = sb; }
    }
}

// ---- fused finalize + apply (backward): dx = k1*dyp + k2*x + k3 with per-channel coefficients in registers
template <int NJ, int UF = 1>
__global__ __launch_bounds__(256) void gn_bwd_apply2_kernel(GnArgs p) {
    __shared__ float gs[64 * 2];
    const int t = threadIdx.x, b = blockIdx.z, chunk = blockIdx.x, cb = blockIdx.y * p.CS;
    const int CH = p.CS / 8, cpg = p.C / p.G;
    if (p.params_in_apply) {                                     // chpart was written by the partial launch before this one
        __shared__ float pred[256 * 2];
        const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        if (lin * 32 < p.C) gn_bwd_params_block(p, lin, pred);   // block-uniform
    }
    gn_fold_groups(p, b, t, gs, 1.0f / ((float)p.HW * (float)cpg));
    __syncthreads();
    int rl, nrl, c0, cstep; bool active;
    gn_thread_map(t, CH, rl, nrl, c0, cstep, active);
    float sc[NJ][8], sh[NJ][8], k1[NJ][8], k2[NJ][8], k3[NJ][8];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int cc = c0 + j * cstep;
            const int ch = cb + (cc < CH ? cc : 0) * 8 + e, g = ch / cpg;
            const float mean = p.stats[((size_t)b * p.G + g) * 2], rstd = p.stats[((size_t)b * p.G + g) * 2 + 1];
            const float s_ = rstd * p.gamma[ch];
            sc[j][e] = s_;
            sh[j][e] = p.beta[ch] - mean * s_;
            k1[j][e] = s_;
            k2[j][e] = -rstd * rstd * gs[g * 2 + 1];
            k3[j][e] = -k2[j][e] * mean - rstd * gs[g * 2];
        }
    if (!active) return;
    const int r_beg = chunk * p.rows_per_chunk;
    const int r_end = (r_beg + p.rows_per_chunk < p.HW) ? r_beg + p.rows_per_chunk : p.HW;
    const int nit = (r_end - r_beg + nrl - 1) / nrl;              // block-uniform
    const size_t boff = (size_t)b * p.HW * p.C + cb;
    const bool has_res = p.dres != nullptr;                      // kernel-uniform
    const half_t* res = has_res ? p.dres : p.dy;                 // a valid address either way: the loads stay unconditional
    constexpr int U = 2 * UF;                                    // three tensors per row: 6 * NJ * UF loads in flight
    for (int it0 = 0; it0 < nit; it0 += U) {
        half8 xv[U][NJ], gv[U][NJ], rv[U][NJ];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r_beg + rl + (it0 + u) * nrl;
            const size_t off = boff + (size_t)(r < r_end ? r : r_beg) * p.C;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int cc = c0 + j * cstep, cl = (cc < CH ? cc : c0) * 8;
                xv[u][j] = ld8(p.x + off + cl);
                gv[u][j] = ld8(p.dy + off + cl);
                rv[u][j] = ld8(res + off + cl);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int r = r_beg + rl + (it0 + u) * nrl;
            const size_t off = boff + (size_t)(r < r_end ? r : r_beg) * p.C;
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                const int cc = c0 + j * cstep;
                CLORA_KEEP(xv[u][j]);
                CLORA_KEEP(gv[u][j]);
                CLORA_KEEP(rv[u][j]);
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xf = (float)xv[u][j][e];
                    float d = (float)gv[u][j][e];
                    if (p.fuse_silu) d *= dsilu_f(xf * sc[j][e] + sh[j][e]);
                    o[e] = (half_t)(k1[j][e] * d + k2[j][e] * xf + k3[j][e]);
                }
                if (has_res) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)o[e] + (float)rv[u][j][e]);
                }
                if (r < r_end && cc < CH) {
                    const GnColW dc = gn_dx_col(p, cb + cc * 8);
                    st8(dc.p + ((size_t)b * p.HW + r) * dc.pitch, o);
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void gn_bwd_params_kernel(GnArgs p) {
    __shared__ float red[256 * 2];
    gn_bwd_params_block(p, blockIdx.x, red);
}

// ------------------------------------------------------------------------------------------ GroupNorm in ONE launch
// The 16x16 / 8x8 (and, at 512 threads, 32x32) feature maps: a whole (batch element, channel slab) is 20-160 KB, the two-launch
// scheme above spends most of its 9-15 us on fixed cost (two launches, the partials' round trip through L2, two ramp-ups) and reads x
// twice.  Here a block owns (batch b, slab of whole groups, ALL rows) and keeps its part of x (backward: x and dy) in registers between
// the statistics and the apply: one launch, one read, one write, no workspace.  Thread map as above (fixed column chunk per thread, row
// lanes), so loads stay whole 16-byte chunks of contiguous row segments; the block reduction is per channel through LDS, a fixed-order
// pairwise tree over the row lanes (every thread folds; log2(nrl) steps), then per group (8 threads per group + shuffles).
// Frozen affine only (no dgamma / dbeta: the UNet's norms; the hint encoder's trainable norms are 256^2 / 512^2 maps and stay above).
template <int NT>
__device__ __forceinline__ void gn_res_reduce(float* red, float* chs, const float (&a)[8], const float (&b)[8], int t, int CS, int cpg, int gps,
                                              int rl, int nrl, int c0, bool active, const float* weight, float* out2, float inv_n) {
    const int W = CS * 2;
    if (active) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { red[rl * W + (c0 * 8 + e) * 2] = a[e]; red[rl * W + (c0 * 8 + e) * 2 + 1] = b[e]; }
    }
    __syncthreads();
    // per channel: `seg` adjacent threads each fold every seg-th row lane, then shuffles (fixed order); three barriers in all
    int seg = 1;
    while (seg < 8 && seg * 2 * CS <= NT) seg *= 2;
    {
        const int c = t / seg, sg = t - c * seg;
        float sa = 0.f, sb = 0.f;
        if (c < CS)
            for (int r = sg; r < nrl; r += seg) { sa += red[r * W + c * 2]; sb += red[r * W + c * 2 + 1]; }
        if (seg > 1) { sa += __shfl_xor(sa, 1); sb += __shfl_xor(sb, 1); }
        if (seg > 2) { sa += __shfl_xor(sa, 2); sb += __shfl_xor(sb, 2); }
        if (seg > 4) { sa += __shfl_xor(sa, 4); sb += __shfl_xor(sb, 4); }
        if (c < CS && sg == 0) { chs[c * 2] = sa; chs[c * 2 + 1] = sb; }
    }
    __syncthreads();
    // per group: 8 threads, each every 8th channel, then three shuffles
    const int g = t >> 3, part = t & 7;
    float sa = 0.f, sb = 0.f;
    if (g < gps) {
        for (int c = g * cpg + part; c < (g + 1) * cpg; c += 8) {
            const float w = weight ? weight[c] : 1.0f;
            sa += w * chs[c * 2]; sb += w * chs[c * 2 + 1];
        }
    }
    sa += __shfl_xor(sa, 1); sb += __shfl_xor(sb, 1);
    sa += __shfl_xor(sa, 2); sb += __shfl_xor(sb, 2);
    sa += __shfl_xor(sa, 4); sb += __shfl_xor(sb, 4);
    if (g < gps && part == 0) { out2[g * 2] = sa * inv_n; out2[g * 2 + 1] = sb * inv_n; }
    __syncthreads();
}

// DEF: x is a deferred split-K GEMM (GnArgs.fin_*): every chunk is folded from the slabs with the GEMM's own epilogue while it is
// loaded (finish_chunk8: the bits the finish kernel would have stored) and written back once through xcopy.
template <int NT, int NPT, bool DEF = false>
__global__ __launch_bounds__(NT) void gn_fwd_resident_kernel(GnArgs p) {
    __shared__ float red[NT * 16];                               // [nrl][CS][2] with nrl * CS <= NT * 8
    __shared__ float chs[256 * 8 * 2];                           // per-channel totals of the slab (CS <= 2048)
    __shared__ float mr[64 * 2];
    const int t = threadIdx.x, b = blockIdx.z, cb = blockIdx.y * p.CS;
    const int CH = p.CS / 8, cpg = p.C / p.G, gps = p.G / p.nslab;
    const int nrl = NT / CH, rl = t / CH, c0 = t - rl * CH;
    const bool active = rl < nrl;
    const int npt = (p.HW + nrl - 1) / nrl;                      // block-uniform, <= NPT
    const GnCol col = gn_in_col(p, cb + (active ? c0 : 0) * 8);
    const size_t brow = (size_t)b * p.HW;
    // the affine parameters of this thread's eight channels travel with the x loads (they were a second dependent round trip after
    // the reduction)
    const floatx4* gp = reinterpret_cast<const floatx4*>(p.gamma + cb + c0 * 8);
    const floatx4* bp = reinterpret_cast<const floatx4*>(p.beta + cb + c0 * 8);
    const floatx4 ga = gp[0], gb = gp[1], ba = bp[0], bb = bp[1];
    half8 v[NPT];
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (active) {
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int r = rl + k * nrl;
            if constexpr (DEF) {
                v[k] = zero8();
                if (k < npt) v[k] = finish_chunk8(p.fin_partial, p.fin_splits, p.B * p.HW, p.C, p.fin_epi, (int)(brow + (r < p.HW ? r : 0)), cb + c0 * 8);
            } else {
                v[k] = (k < npt) ? ld8(col.p + (brow + (size_t)(r < p.HW ? r : 0)) * col.pitch) : zero8();
            }
        }
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const bool ok = k < npt && rl + k * nrl < p.HW;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = ok ? (float)v[k][e] : 0.f; s[e] += f; q[e] += f * f; }
        }
    }
    gn_res_reduce<NT>(red, chs, s, q, t, p.CS, cpg, gps, rl, nrl, c0, active, nullptr, mr, 1.0f / ((float)p.HW * (float)cpg));
    if (t < gps) {                                               // (E[x], E[x^2]) -> (mean, rstd)
        const float mean = mr[t * 2];
        float var = mr[t * 2 + 1] - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = rsqrtf(var + p.eps);
        mr[t * 2 + 1] = rstd;
        float* st = p.stats + ((size_t)b * p.G + blockIdx.y * gps + t) * 2;
        st[0] = mean; st[1] = rstd;
    }
    __syncthreads();
    if (!active) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int cl = c0 * 8 + e, g = cl / cpg;
        sc[e] = mr[g * 2 + 1] * (e < 4 ? ga[e & 3] : gb[e & 3]);
        sh[e] = (e < 4 ? ba[e & 3] : bb[e & 3]) - mr[g * 2] * sc[e];
    }
    half_t* ybase = p.y + (size_t)b * p.HW * p.C + cb + c0 * 8;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int r = rl + k * nrl;
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float yv = (float)v[k][e] * sc[e] + sh[e];
            if (p.fuse_silu) yv = silu_f(yv);
            o[e] = (half_t)yv;
        }
        if (k < npt && r < p.HW) {
            st8(ybase + (size_t)r * p.C, o);
            if (p.xcopy) st8(p.xcopy + (size_t)b * p.HW * p.C + cb + c0 * 8 + (size_t)r * p.C, v[k]);
        }
    }
}

template <int NT, int NPT, bool DEF = false>
__global__ __launch_bounds__(NT) void gn_bwd_resident_kernel(GnArgs p) {
    __shared__ float red[NT * 16];
    __shared__ float chs[256 * 8 * 2];
    __shared__ float gs[64 * 2];
    const int t = threadIdx.x, b = blockIdx.z, cb = blockIdx.y * p.CS;
    const int CH = p.CS / 8, cpg = p.C / p.G, gps = p.G / p.nslab;
    const int nrl = NT / CH, rl = t / CH, c0 = t - rl * CH;
    const bool active = rl < nrl;
    const int npt = (p.HW + nrl - 1) / nrl;
    const size_t boff = (size_t)b * p.HW * p.C + cb + c0 * 8;
    float mean[8], rstd[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = cb + (active ? c0 : 0) * 8 + e;
        const float* st = p.stats + ((size_t)b * p.G + ch / cpg) * 2;
        mean[e] = st[0]; rstd[e] = st[1];
        sc[e] = rstd[e] * p.gamma[ch];
        sh[e] = p.beta[ch] - mean[e] * sc[e];
    }
    half8 xv[NPT], gv[NPT];
    float a1[8], a2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    if (active) {
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int r = rl + k * nrl;
            const size_t off = boff + (size_t)(r < p.HW ? r : 0) * p.C;
            xv[k] = (k < npt) ? ld8(p.x + off) : zero8();
            if constexpr (DEF) {
                gv[k] = zero8();
                if (k < npt) gv[k] = finish_chunk8(p.fin_partial, p.fin_splits, p.B * p.HW, p.C, p.fin_epi, b * p.HW + (r < p.HW ? r : 0), cb + c0 * 8);
            } else {
                gv[k] = (k < npt) ? ld8(p.dy + off) : zero8();
            }
        }
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const bool ok = k < npt && rl + k * nrl < p.HW;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xf = (float)xv[k][e];
                const float xh = (xf - mean[e]) * rstd[e];
                float d = ok ? (float)gv[k][e] : 0.f;
                if (p.fuse_silu) d *= dsilu_f(xf * sc[e] + sh[e]);
                a1[e] += d;
                a2[e] += d * xh;
            }
        }
    }
    gn_res_reduce<NT>(red, chs, a1, a2, t, p.CS, cpg, gps, rl, nrl, c0, active, p.gamma + cb, gs, 1.0f / ((float)p.HW * (float)cpg));
    if (!active) return;
    float k2[8], k3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c0 * 8 + e) / cpg;
        k2[e] = -rstd[e] * rstd[e] * gs[g * 2 + 1];
        k3[e] = -k2[e] * mean[e] - rstd[e] * gs[g * 2];
    }
    const bool has_res = p.dres != nullptr;
    const GnColW dxc = gn_dx_col(p, cb + c0 * 8);
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int r = rl + k * nrl;
        if (!(k < npt && r < p.HW)) continue;
        const size_t off = boff + (size_t)r * p.C;
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xf = (float)xv[k][e];
            float d = (float)gv[k][e];
            if (p.fuse_silu) d *= dsilu_f(xf * sc[e] + sh[e]);
            o[e] = (half_t)(sc[e] * d + k2[e] * xf + k3[e]);
        }
        if (has_res) {
            const half8 rv = ld8(p.dres + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)o[e] + (float)rv[e]);
        }
        st8(dxc.p + ((size_t)b * p.HW + r) * dxc.pitch, o);
    }
}

// ------------------------------------------------------------------------------------------ GroupNorm in ONE launch, large maps
// The 64x64 / 32x32 maps do not fit one block per (batch element, slab), and the two-launch scheme reads x twice and pays two ramp-ups
// (7.4 + 12.0 us forward, 10.7 + 12.7 us backward per 64x64x320 map inside the train step).  Here a TEAM of tm_nb blocks shares a
// (batch element, slab) UNIT, every member keeps its rows in registers (the resident kernels' thread map), and the members exchange
// their per-group partial sums INSIDE the launch: 2 * gps fp32 values per member, each published as one 8-byte {epoch, value} granule
// by a relaxed agent-scope (write-through, sc1) store and swept by every member with sc1 loads until all tags carry the epoch
// (cdna_hip_programming.md Guideline 16, form R2: the datum is its own flag -- no fence, no L2 write-back, placement independent).
// The epoch of a unit is its generation word + 1, read by every member BEFORE it publishes and advanced by the unit's last member
// AFTER it has seen every member's granules (so after every member has read it): tags of a slot only ever grow, nothing is zeroed
// between launches and nothing depends on a per-launch argument (graph replay freezes those).  The members are folded in a fixed
// order: results are deterministic.  256 blocks of 512 threads = one per CU must be co-resident (checked by the planner against the
// device's CU count); the sweep is bounded and a give-up is recorded in the state's error word (clora_groupnorm_team_errors).
template <int NT>
__device__ __forceinline__ bool gn_team_exchange(const GnArgs& p, int u, int m, unsigned epoch, int GK, float* vals, float* gat, float scale) {
    const int t = threadIdx.x;
    unsigned long long* ubase = p.tm_gran + (size_t)u * kTeamMembers * kTeamGran;
    if (t < GK) CLORA_ST_AGENT_U64(ubase + (size_t)m * kTeamGran + t, ((unsigned long long)epoch << 32) | __float_as_uint(vals[t]));
    if (p.tm_probe) return false;
    const int P = NT / GK, gk = t % GK, pi = t / GK;
    float acc = 0.f;
    bool fail = false;
    if (pi < P) {
        unsigned long long x[8];
        unsigned spins = 0;
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int mm = pi + k * P;
                x[k] = 0;
                if (mm < p.tm_nb) {
                    x[k] = CLORA_LD_AGENT_U64(ubase + (size_t)mm * kTeamGran + gk);
                    ok = ok && (unsigned)(x[k] >> 32) == epoch;
                }
            }
            if (ok) break;
            if (++spins > p.tm_spin) { fail = true; break; }
            CLORA_SLEEP();
        }
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (pi + k * P < p.tm_nb) acc += __uint_as_float((unsigned)x[k]);
    }
    gat[t] = acc;
    if (fail) atomicOr(p.tm_err, 1u);
    __syncthreads();
    if (t < GK) {
        float sacc = 0.f;
        for (int pp = 0; pp < P; ++pp) sacc += gat[pp * GK + t];
        vals[t] = sacc * scale;
    }
    __syncthreads();
    if (m == p.tm_nb - 1 && t == 0) CLORA_ST_AGENT_U32(p.tm_gen + u, epoch);
    return true;
}

template <int NT, int NPT>
__global__ __launch_bounds__(NT) void gn_fwd_team_kernel(GnArgs p) {
    __shared__ float red[NT * 16];
    __shared__ float chs[NT * 2];
    __shared__ float mr[64 * 2];
    __shared__ float gat[NT];
    __shared__ unsigned s_epoch;
    const int t = threadIdx.x;
    const int U = p.B * p.nslab, u = blockIdx.x % U, m = blockIdx.x / U;
    const int b = u / p.nslab, cb = (u - b * p.nslab) * p.CS;
    if (t == 0) s_epoch = CLORA_LD_AGENT_U32(p.tm_gen + u) + 1u;
    const int CH = p.CS / 8, cpg = p.C / p.G, gps = p.G / p.nslab;
    const int nrl = NT / CH, rl = t / CH, c0 = t - rl * CH;
    const bool active = rl < nrl;
    const int r0 = m * p.tm_rpb;
    const int r_end = (r0 + p.tm_rpb < p.HW) ? r0 + p.tm_rpb : p.HW;
    const GnCol col = gn_in_col(p, cb + (active ? c0 : 0) * 8);
    const size_t brow = (size_t)b * p.HW;
    const floatx4* gp = reinterpret_cast<const floatx4*>(p.gamma + cb + (active ? c0 : 0) * 8);
    const floatx4* bp = reinterpret_cast<const floatx4*>(p.beta + cb + (active ? c0 : 0) * 8);
    const floatx4 ga = gp[0], gb = gp[1], ba = bp[0], bb = bp[1];
    half8 v[NPT];
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
    if (active) {
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int r = r0 + rl + k * nrl;
            v[k] = ld8(col.p + (brow + (size_t)(r < r_end ? r : (r0 < p.HW ? r0 : 0))) * col.pitch);
        }
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const bool ok = r0 + rl + k * nrl < r_end;
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float f = ok ? (float)v[k][e] : 0.f; s[e] += f; q[e] += f * f; }
        }
    }
    gn_res_reduce<NT>(red, chs, s, q, t, p.CS, cpg, gps, rl, nrl, c0, active, nullptr, mr, 1.0f);
    if (!gn_team_exchange<NT>(p, u, m, s_epoch, gps * 2, mr, gat, 1.0f / ((float)p.HW * (float)cpg))) return;
    if (t < gps) {                                               // (E[x], E[x^2]) -> (mean, rstd)
        const float mean = mr[t * 2];
        float var = mr[t * 2 + 1] - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float rstd = rsqrtf(var + p.eps);
        mr[t * 2 + 1] = rstd;
        if (m == 0) {
            float* st = p.stats + ((size_t)b * p.G + (cb / cpg) + t) * 2;
            st[0] = mean; st[1] = rstd;
        }
    }
    __syncthreads();
    if (!active) return;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int cl = c0 * 8 + e, g = cl / cpg;
        sc[e] = mr[g * 2 + 1] * (e < 4 ? ga[e & 3] : gb[e & 3]);
        sh[e] = (e < 4 ? ba[e & 3] : bb[e & 3]) - mr[g * 2] * sc[e];
    }
    half_t* ybase = p.y + (size_t)b * p.HW * p.C + cb + c0 * 8;
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int r = r0 + rl + k * nrl;
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float yv = (float)v[k][e] * sc[e] + sh[e];
            if (p.fuse_silu) yv = silu_f(yv);
            o[e] = (half_t)yv;
        }
        if (r < r_end) {
            st8(ybase + (size_t)r * p.C, o);
            if (p.xcopy) st8(p.xcopy + (size_t)b * p.HW * p.C + cb + c0 * 8 + (size_t)r * p.C, v[k]);
        }
    }
}

template <int NT, int NPT>
__global__ __launch_bounds__(NT) void gn_bwd_team_kernel(GnArgs p) {
    __shared__ float red[NT * 16];
    __shared__ float chs[NT * 2];
    __shared__ float gs[64 * 2];
    __shared__ float gat[NT];
    __shared__ unsigned s_epoch;
    const int t = threadIdx.x;
    const int U = p.B * p.nslab, u = blockIdx.x % U, m = blockIdx.x / U;
    const int b = u / p.nslab, cb = (u - b * p.nslab) * p.CS;
    if (t == 0) s_epoch = CLORA_LD_AGENT_U32(p.tm_gen + u) + 1u;
    const int CH = p.CS / 8, cpg = p.C / p.G, gps = p.G / p.nslab;
    const int nrl = NT / CH, rl = t / CH, c0 = t - rl * CH;
    const bool active = rl < nrl;
    const int r0 = m * p.tm_rpb;
    const int r_end = (r0 + p.tm_rpb < p.HW) ? r0 + p.tm_rpb : p.HW;
    const size_t boff = (size_t)b * p.HW * p.C + cb + (active ? c0 : 0) * 8;
    float mean[8], rstd[8], sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = cb + (active ? c0 : 0) * 8 + e;
        const float* st = p.stats + ((size_t)b * p.G + ch / cpg) * 2;
        mean[e] = st[0]; rstd[e] = st[1];
        sc[e] = rstd[e] * p.gamma[ch];
        sh[e] = p.beta[ch] - mean[e] * sc[e];
    }
    half8 xv[NPT], gv[NPT];
    float a1[8], a2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { a1[e] = 0.f; a2[e] = 0.f; }
    if (active) {
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const int r = r0 + rl + k * nrl;
            const size_t off = boff + (size_t)(r < r_end ? r : (r0 < p.HW ? r0 : 0)) * p.C;
            xv[k] = ld8(p.x + off);
            gv[k] = ld8(p.dy + off);
        }
#pragma unroll
        for (int k = 0; k < NPT; ++k) {
            const bool ok = r0 + rl + k * nrl < r_end;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xf = (float)xv[k][e];
                const float xh = (xf - mean[e]) * rstd[e];
                float d = ok ? (float)gv[k][e] : 0.f;
                if (p.fuse_silu) d *= dsilu_f(xf * sc[e] + sh[e]);
                a1[e] += d;
                a2[e] += d * xh;
            }
        }
    }
    gn_res_reduce<NT>(red, chs, a1, a2, t, p.CS, cpg, gps, rl, nrl, c0, active, p.gamma + cb, gs, 1.0f);
    if (!gn_team_exchange<NT>(p, u, m, s_epoch, gps * 2, gs, gat, 1.0f / ((float)p.HW * (float)cpg))) return;
    if (!active) return;
    float k2[8], k3[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (c0 * 8 + e) / cpg;
        k2[e] = -rstd[e] * rstd[e] * gs[g * 2 + 1];
        k3[e] = -k2[e] * mean[e] - rstd[e] * gs[g * 2];
    }
    const bool has_res = p.dres != nullptr;
    const GnColW dxc = gn_dx_col(p, cb + c0 * 8);
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
        const int r = r0 + rl + k * nrl;
        if (!(r < r_end)) continue;
        const size_t off = boff + (size_t)r * p.C;
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float xf = (float)xv[k][e];
            float d = (float)gv[k][e];
            if (p.fuse_silu) d *= dsilu_f(xf * sc[e] + sh[e]);
            o[e] = (half_t)(sc[e] * d + k2[e] * xf + k3[e]);
        }
        if (has_res) {
            const half8 rv = ld8(p.dres + off);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)o[e] + (float)rv[e]);
        }
        st8(dxc.p + ((size_t)b * p.HW + r) * dxc.pitch, o);
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm
struct LnArgs {
    const half_t* x;
    const half_t* dy;
    const half_t* dres;   // bwd, optional: the residual branch's gradient of x, added into dx
    half_t* y;
    const float* gamma;
    const float* beta;
    int M, C;
    float eps;
    const float* fin_partial;   // bwd: dy is a deferred split-K GEMM (slabs [fin_splits][M][C] + its epilogue), see GnArgs
    int fin_splits;
    clora_epilogue_t fin_epi;
};

template <bool BWD>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs p) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + w;
    const bool rok = row < p.M;
    const int CH = p.C / 8;
    const size_t off = (size_t)(rok ? row : 0) * p.C;
    half8 xv[kLnCols], gv[kLnCols];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kLnCols; ++j) {
        const int cc = l + 64 * j;
        xv[j] = zero8(); gv[j] = zero8();
        if (cc < CH) {
            xv[j] = ld8(p.x + off + cc * 8);
            if (BWD) gv[j] = ld8(p.dy + off + cc * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) s += (float)xv[j][e];
        }
    }
    const float mean = wave_sum(s) / (float)p.C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < kLnCols; ++j)
        if (l + 64 * j < CH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)xv[j][e] - mean; q += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)p.C + p.eps);
    if (!BWD) {
#pragma unroll
        for (int j = 0; j < kLnCols; ++j) {
            const int cc = l + 64 * j;
            if (cc < CH && rok) {
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    o[e] = (half_t)(((float)xv[j][e] - mean) * rstd * p.gamma[cc * 8 + e] + p.beta[cc * 8 + e]);
                st8(p.y + off + cc * 8, o);
            }
        }
    } else {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < kLnCols; ++j) {
            const int cc = l + 64 * j;
            if (cc < CH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float gg = (float)gv[j][e] * p.gamma[cc * 8 + e];
                    s1 += gg;
                    s2 += gg * ((float)xv[j][e] - mean) * rstd;
                }
            }
        }
        s1 = wave_sum(s1) / (float)p.C;
        s2 = wave_sum(s2) / (float)p.C;
#pragma unroll
        for (int j = 0; j < kLnCols; ++j) {
            const int cc = l + 64 * j;
            if (cc < CH && rok) {
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)xv[j][e] - mean) * rstd;
                    o[e] = (half_t)(rstd * ((float)gv[j][e] * p.gamma[cc * 8 + e] - s1 - xh * s2));
                }
                if (p.dres) {
                    const half8 rr = ld8(p.dres + off + cc * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)o[e] + (float)rr[e]);
                }
                st8(p.y + off + cc * 8, o);
            }
        }
    }
}

// Several rows per wave.  One row of C = 320 is 640 bytes: with a single row per wave a CU has ~20 KB in flight and the
// kernel sits at a third of the fabric rate (profiles/r02_stream_rate_probe.txt vs r02_kernel_stats_final.json: 17 us for
// 42 MB).  Here a wave issues the loads of ROWS rows (and gamma / beta, once) before the first reduction; per row the
// arithmetic and its order are exactly those of layernorm_kernel, so the results are bit-identical.
//   NC = 16-byte chunks per lane (C <= 512: 1, <= 1024: 2, <= 1536: 3)
template <bool BWD, int NC, int ROWS, bool DEF = false>
__global__ __launch_bounds__(256) void layernorm_rows_kernel(LnArgs p) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + w) * ROWS;
    if (row0 >= p.M) return;                                     // wave-uniform
    const int CH = p.C / 8;
    half8 xv[ROWS][NC], gv[ROWS][NC], rv[ROWS][NC];
    float gam[NC][8], bet[NC][8];
    const bool has_res = BWD && p.dres != nullptr;               // kernel-uniform
    const half_t* res = has_res ? p.dres : p.x;                  // a valid address either way: the loads stay unconditional
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const bool rok = row0 + r < p.M;
        const size_t off = (size_t)(rok ? row0 + r : row0) * p.C;
#pragma unroll
        for (int j = 0; j < NC; ++j) {                           // branch-free: lanes past the row re-read chunk 0 (never used),
            const int cc = l + 64 * j, cl = cc < CH ? cc : 0;    // so every load of the wave is issued back to back
            xv[r][j] = ld8(p.x + off + cl * 8);
            if (BWD) {
                if constexpr (DEF) gv[r][j] = finish_chunk8(p.fin_partial, p.fin_splits, p.M, p.C, p.fin_epi, rok ? row0 + r : row0, cl * 8);
                else gv[r][j] = ld8(p.dy + off + cl * 8);
                rv[r][j] = ld8(res + off + cl * 8);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NC; ++j) {
        const int cc = l + 64 * j, cl = cc < CH ? cc : 0;
        const floatx4 g0 = *reinterpret_cast<const floatx4*>(p.gamma + cl * 8), g1 = *reinterpret_cast<const floatx4*>(p.gamma + cl * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { gam[j][e] = g0[e]; gam[j][4 + e] = g1[e]; bet[j][e] = 0.f; bet[j][4 + e] = 0.f; }
        if (!BWD) {
            const floatx4 b0 = *reinterpret_cast<const floatx4*>(p.beta + cl * 8), b1 = *reinterpret_cast<const floatx4*>(p.beta + cl * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { bet[j][e] = b0[e]; bet[j][4 + e] = b1[e]; }
        }
    }
    float mean[ROWS], rstd[ROWS];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (l + 64 * j < CH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) s += (float)xv[r][j][e];
            }
        mean[r] = wave_sum(s) / (float)p.C;
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NC; ++j)
            if (l + 64 * j < CH) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = (float)xv[r][j][e] - mean[r]; q += d * d; }
            }
        rstd[r] = rsqrtf(wave_sum(q) / (float)p.C + p.eps);
    }
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        const bool rok = row0 + r < p.M;
        const size_t off = (size_t)(rok ? row0 + r : row0) * p.C;
        if (!BWD) {
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int cc = l + 64 * j;
                if (cc < CH && rok) {
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = (half_t)(((float)xv[r][j][e] - mean[r]) * rstd[r] * gam[j][e] + bet[j][e]);
                    st8(p.y + off + cc * 8, o);
                }
            }
        } else {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int j = 0; j < NC; ++j)
                if (l + 64 * j < CH) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float gg = (float)gv[r][j][e] * gam[j][e];
                        s1 += gg;
                        s2 += gg * ((float)xv[r][j][e] - mean[r]) * rstd[r];
                    }
                }
            s1 = wave_sum(s1) / (float)p.C;
            s2 = wave_sum(s2) / (float)p.C;
#pragma unroll
            for (int j = 0; j < NC; ++j) {
                const int cc = l + 64 * j;
                if (cc < CH && rok) {
                    half8 o;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const float xh = ((float)xv[r][j][e] - mean[r]) * rstd[r];
                        o[e] = (half_t)(rstd[r] * ((float)gv[r][j][e] * gam[j][e] - s1 - xh * s2));
                    }
                    if (has_res) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)o[e] + (float)rv[r][j][e]);
                    }
                    st8(p.y + off + cc * 8, o);
                }
            }
        }
    }
}

// can the deferred-dy form run?  (the several-rows kernels only; the one-row fallback kernel has no deferred variant)
bool ln_rows_plan(const LnArgs& a, bool bwd) {
    const bool al16 = (((uintptr_t)a.gamma | (uintptr_t)(bwd ? a.gamma : a.beta)) & 15) == 0;
    return clora_ln_rows() && a.C / 8 <= 192 && al16;
}

template <bool BWD, bool DEF = false>
void launch_layernorm(const LnArgs& a, hipStream_t s) {
    const int CH = a.C / 8;
    if (ln_rows_plan(a, BWD)) {
        if (a.M >= 2048) {                                       // enough rows to keep the chip full with fewer, fatter waves
            if (CH <= 64) hipLaunchKernelGGL((layernorm_rows_kernel<BWD, 1, 4, DEF>), dim3(clora_cdiv(a.M, 16)), dim3(256), 0, s, a);
            else if (CH <= 128) hipLaunchKernelGGL((layernorm_rows_kernel<BWD, 2, 2, DEF>), dim3(clora_cdiv(a.M, 8)), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((layernorm_rows_kernel<BWD, 3, 2, DEF>), dim3(clora_cdiv(a.M, 8)), dim3(256), 0, s, a);
        } else {                                                 // few rows (16x16 level, text tokens): one row per wave, but its 1-3 chunk
            if (CH <= 64) hipLaunchKernelGGL((layernorm_rows_kernel<BWD, 1, 1, DEF>), dim3(clora_cdiv(a.M, 4)), dim3(256), 0, s, a);        // loads (and dy, dres, gamma)
            else if (CH <= 128) hipLaunchKernelGGL((layernorm_rows_kernel<BWD, 2, 1, DEF>), dim3(clora_cdiv(a.M, 4)), dim3(256), 0, s, a);  // issued together instead of
            else hipLaunchKernelGGL((layernorm_rows_kernel<BWD, 3, 1, DEF>), dim3(clora_cdiv(a.M, 4)), dim3(256), 0, s, a);                 // one block per chunk
        }
        return;
    }
    hipLaunchKernelGGL((layernorm_kernel<BWD>), dim3(clora_cdiv(a.M, 4)), dim3(256), 0, s, a);
}

// ------------------------------------------------------------------------------------------ row softmax
// y[r, :] = softmax(scale * x[r, :]) for the single-head d=512 attention of the VAE mid block (upstream
// AutoencoderKL AttentionBlock; reference call sites train_text_to_image_control_lora.py:403,753 and
// apps/gradio_canny2image.py decode), whose head dim is beyond the flash kernels' register budget: scores are
// materialised by the GEMM kernel and normalised here.  One wave per row, the row lives in registers
// (cols <= 64 lanes x kSmCols x 8), fp32 max / sum, in-place allowed.
constexpr int kSmCols = 16;
__global__ __launch_bounds__(256) void softmax_rows_kernel(const half_t* x, half_t* y, int rows, int cols, int ld, float scale) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + w;
    if (row >= rows) return;
    const int CH = cols / 8;
    const half_t* xr = x + (size_t)row * ld;
    half_t* yr = y + (size_t)row * ld;
    half8 v[kSmCols];
    float mx = -3.0e38f;
    // all chunk loads of the row first (lanes past the row re-read chunk l, unused): under `if (cc < CH)` each chunk was its own
    // load -> s_waitcnt vmcnt(0) block, eight dependent round trips per 4096-wide row
#pragma unroll
    for (int j = 0; j < kSmCols; ++j) {
        const int cc = l + 64 * j;
        v[j] = ld8(xr + (cc < CH ? cc : (l < CH ? l : 0)) * 8);
    }
#pragma unroll
    for (int j = 0; j < kSmCols; ++j) {
        CLORA_KEEP(v[j]);
        if (l + 64 * j < CH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) mx = fmaxf(mx, (float)v[j][e]);
        }
    }
    mx = wave_max(mx) * scale;
    float f[kSmCols][8];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < kSmCols; ++j)
        if (l + 64 * j < CH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { f[j][e] = CLORA_EXP2(((float)v[j][e] * scale - mx) * 1.4426950408889634f); sum += f[j][e]; }
        }
    const float inv = 1.0f / wave_sum(sum);
#pragma unroll
    for (int j = 0; j < kSmCols; ++j) {
        const int cc = l + 64 * j;
        if (cc < CH) {
            half8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)(f[j][e] * inv);
            st8(yr + cc * 8, o);
        }
    }
}

int gn_plan(GnArgs& a, void* ws, size_t ws_bytes, bool bwd, bool params) {
    if (a.B <= 0 || a.HW <= 0 || a.C <= 0 || a.G <= 0 || a.G > 64 || (a.C % a.G) || (a.C & 7) || a.C > 4096)
        return CLORA_ERR_ARG;
    const int CH = a.C / 8;
    const int nrl = CH >= 256 ? 1 : 256 / CH;
    const long kBlocks = clora_option(CLORA_OPT_GN_BLOCKS);
    long rpc = ((long)a.HW * a.B + kBlocks - 1) / kBlocks;   // ~512 blocks in flight (A/B: clora_set_option("gn_blocks"))
    if (rpc < 32) rpc = 32;
    if (rpc < 2 * nrl) rpc = 2 * nrl;
    if (rpc > a.HW) rpc = a.HW;
    a.rows_per_chunk = (int)rpc;
    a.nchunk = clora_cdiv(a.HW, rpc);
    // channel slabs: when the row chunks alone leave most of the 256 CUs idle (16x16 / 8x8 feature maps with
    // 1280..2560 channels) split the channels too, at whole-group and 8-channel granularity
    {
        const int cpg = a.C / a.G;
        int unit_groups = 1;                                   // groups per smallest slab: (unit_groups*cpg) % 8 == 0
        while ((unit_groups * cpg) & 7) ++unit_groups;
        int nslab = 1;
        if (a.G % unit_groups == 0) {
            const int nunits = a.G / unit_groups;
            const long want = (kBlocks + (long)a.B * a.nchunk - 1) / ((long)a.B * a.nchunk);
            for (int d = 1; d <= nunits; ++d) {
                if (nunits % d) continue;
                if (d <= want || a.C / nslab > 2048) nslab = d;   // largest divisor <= want, grown until a slab fits
                else break;
            }
        }
        a.nslab = nslab;
        a.CS = a.C / nslab;
    }
    size_t need = (size_t)a.B * a.nchunk * a.G * 2;
    if (params) need += (size_t)a.B * a.nchunk * a.C * 2;
    if (!ws || ws_bytes < need * sizeof(float)) return CLORA_ERR_WORKSPACE;
    a.partial = (float*)ws;
    a.chpart = params ? a.partial + (size_t)a.B * a.nchunk * a.G * 2 : nullptr;
    return CLORA_OK;
}

// One-launch variant: which (threads, rows per thread) instantiation takes this shape, or 0.  Slabs = the smallest whole-group,
// 8-channel-aligned unit that is at least 40 channels wide (80-byte row segments), one block per (slab, batch element).
struct GnResident { int nt, npt; };
GnResident gn_resident_plan(GnArgs& a, bool bwd, bool params) {
    GnResident none = {0, 0};
    if (!clora_option(CLORA_OPT_GN_RESIDENT) || params) return none;
    const int cpg = a.C / a.G;
    int unit = 1;
    while ((unit * cpg) & 7) ++unit;
    if (a.G % unit) return none;
    int groups = unit;
    while (groups * cpg < 40 && groups * 2 <= a.G && a.G % (groups * 2) == 0) groups *= 2;
    const int CS = groups * cpg, CH = CS / 8;
    if (groups > 64 || CH > 256) return none;
    const int max_npt = bwd ? 8 : 16;
    for (int nt = 256; nt <= 512; nt *= 2) {
        const int nrl = nt / CH, npt = (a.HW + nrl - 1) / nrl;
        // CS <= nt: stage 2 of gn_res_reduce maps one thread (group of `seg` threads) to one channel -- a slab with more channels than
        // the block has threads would leave chs[] partly unreduced (C = 1280, G = 4)
        if (npt <= max_npt && groups * 8 <= nt && CS <= nt) {
            a.nslab = a.G / groups; a.CS = CS; a.nchunk = 1; a.rows_per_chunk = a.HW;
            GnResident r = {nt, npt <= 4 ? 4 : (npt <= 8 ? 8 : 16)};
            return r;
        }
    }
    return none;
}

// Team variant (gn_*_team_kernel): how many slabs, members and rows per thread, or 0.  The unit count B * nslab must divide the 256
// blocks of the launch; slabs are whole groups, 8-channel aligned, at most 512 channels (one reduction thread per channel); among the
// feasible splits the one with the most slabs whose row segments are still >= 320 bytes (fewest granules to gather), else the fewest.
int gn_team_plan(GnArgs& a, bool bwd, bool params, void* state, size_t state_bytes) {
    // smallest map by option level (measured, profiles/r06_gn_team_bench.txt): 1, 2 = 32x32 forward / 16x16 backward (the backward gains
    // at 16x16 on every width, the forward only above 1280 channels); 3 = 16x16 both; 4 = 8x8 both (slower than one block per slab)
    const int level = clora_option(CLORA_OPT_GN_TEAM);
    const int min_hw = level >= 4 ? 64 : ((level == 3 || (bwd && level == 2)) ? 256 : 1024);
    if (!state || state_bytes < kTeamStateBytes || params || !level || a.HW < min_hw) return 0;
    static int cus = -1;
    if (cus < 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 0;
        cus = n;
    }
    if (cus < kTeamBlocks) return 0;                         // one block per CU must be resident: the members wait for each other
    const int cpg = a.C / a.G;
    int unit = 1;
    while ((unit * cpg) & 7) ++unit;
    if (a.G % unit) return 0;
    const int nunits = a.G / unit;
    int pick = 0, pick_npt = 0;
    for (int d = 1; d <= nunits; ++d) {
        if (nunits % d) continue;
        const int U = a.B * d;
        if (U > kTeamUnits || (kTeamBlocks % U)) continue;
        const int NB = kTeamBlocks / U, CS = a.C / d, CH = CS / 8, gps = a.G / d, GK = 2 * gps;
        if (NB < 2 || CS > kTeamNT || GK > kTeamGran || gps * 8 > kTeamNT) continue;
        if (NB > 8 * (kTeamNT / GK)) continue;             // the sweep holds at most 8 granules per thread
        const int nrl = kTeamNT / CH, rpb = clora_cdiv(a.HW, NB), npt = clora_cdiv(rpb, nrl);
        if (npt > (bwd ? 8 : 16)) continue;               // backward: x and dy in registers (16 rows of both spill at 256 VGPRs)
        if (!pick || CS >= 160) { pick = d; pick_npt = npt; }
        if (CS < 160) break;
    }
    if (!pick) return 0;
    a.nslab = pick; a.CS = a.C / pick; a.nchunk = 1; a.rows_per_chunk = a.HW;
    a.tm_nb = kTeamBlocks / (a.B * pick); a.tm_rpb = clora_cdiv(a.HW, a.tm_nb);
    a.tm_err = (unsigned*)state; a.tm_gen = (unsigned*)((char*)state + 256);
    a.tm_gran = (unsigned long long*)((char*)state + kTeamHeader);
    a.tm_spin = 1u << 16; a.tm_probe = 0;
    return pick_npt <= 4 ? 4 : (pick_npt <= 8 ? 8 : 16);
}

template <bool BWD>
void gn_team_launch(GnArgs& a, int npt, hipStream_t s) {
    const dim3 grid(kTeamBlocks), block(kTeamNT);
    for (int pass = CLORA_SEQUENTIAL_BLOCKS ? 0 : 1; pass < 2; ++pass) {
        a.tm_probe = pass == 0;
        if (BWD) {
            if (npt == 4) hipLaunchKernelGGL((gn_bwd_team_kernel<kTeamNT, 4>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((gn_bwd_team_kernel<kTeamNT, 8>), grid, block, 0, s, a);
        } else {
            if (npt == 4) hipLaunchKernelGGL((gn_fwd_team_kernel<kTeamNT, 4>), grid, block, 0, s, a);
            else if (npt == 8) hipLaunchKernelGGL((gn_fwd_team_kernel<kTeamNT, 8>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((gn_fwd_team_kernel<kTeamNT, 16>), grid, block, 0, s, a);
        }
    }
}

int ew_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" size_t clora_groupnorm_workspace_bytes(int B, int HW, int C, int G, int backward, int param_grads) {
    GnArgs a = GnArgs();
    a.B = B; a.HW = HW; a.C = C; a.G = G;
    static float dummy;
    if (gn_plan(a, &dummy, (size_t)-1, backward != 0, param_grads != 0) != CLORA_OK) return 0;
    size_t need = (size_t)B * a.nchunk * G * 2;
    if (param_grads) need += (size_t)B * a.nchunk * C * 2;
    return need * sizeof(float);
}

namespace {
// does the consumer take this deferred producer as it is (N == C, M == B*HW, contiguous C)?  else: plain finish first
bool deferred_fits(const clora_deferred_t* d, int M, int C) {
    return d && d->splits > 1 && d->partial && d->M == M && d->N == C && d->ldc == C && d->C;
}
}  // namespace

namespace {
int gn_fwd_impl(const clora_half* x, const clora_half* x2, int Ca, const clora_deferred_t* src, clora_half* xcopy,
                clora_half* y, const float* gamma, const float* beta, float* stats, int B, int HW, int C,
                int G, float eps, int fuse_silu, void* team_state, size_t team_state_bytes, void* workspace, size_t workspace_bytes,
                void* stream) {
    if (!y || !gamma || !beta || !stats) return CLORA_ERR_ARG;
    if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 64 || (C % G) || (C & 7) || C > 4096) return CLORA_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const bool deferred = src && src->splits > 0;
    if (deferred) {
        if (x2 || !deferred_fits(src, B * HW, C)) return CLORA_ERR_ARG;
        x = src->C;                                              // where the finished tensor lives / will live
    }
    if (!x || (x2 && (Ca <= 0 || Ca >= C || (Ca & 7)))) return CLORA_ERR_ARG;
    GnArgs a = GnArgs();
    a.x = (const half_t*)x; a.y = (half_t*)y; a.gamma = gamma; a.beta = beta; a.stats = stats;
    a.B = B; a.HW = HW; a.C = C; a.G = G; a.eps = eps; a.fuse_silu = fuse_silu;
    a.x2 = (const half_t*)x2; a.Ca = x2 ? Ca : C; a.xcopy = (half_t*)xcopy;
    const GnResident res = gn_resident_plan(a, false, false);
    // Folding pays only where a thread owns few rows (npt <= 4: the 8x8 / 16x16 maps): the one-launch kernels run 32-128 blocks and a
    // thread's rows are folded one after the other -- measured on MI355X (profiles/r06_deferred_ab.txt): 9.6 us against 6.6 + 6.0 for
    // finish + plain at npt 4, but 17.4 against 8.1 + 6.0 at npt 8 and 36.2 against 13.9 + 6.0 at npt 16.  Everything else: the
    // plain finish pass (2048 blocks) first.
    bool deferred_here = deferred && res.nt && res.npt <= clora_option(CLORA_OPT_DEFER_MAX_ROWS);
    if (deferred && !deferred_here) {                            // finish first, then the plain passes over src->C
        const int rc = clora_finish_deferred(src, stream);
        if (rc != CLORA_OK) return rc;
    }
    // large maps: one launch, a team per unit.  Shapes whose one-block-per-slab plan can fold a deferred producer keep that plan whether
    // or not this call's input is deferred (deferring never changes the bits)
    const bool fold_shape = res.nt && res.npt <= clora_option(CLORA_OPT_DEFER_MAX_ROWS);
    if (!fold_shape && (!res.nt || clora_option(CLORA_OPT_GN_TEAM) >= 2)) {
        GnArgs ta = a;
        const int npt = gn_team_plan(ta, false, false, team_state, team_state_bytes);
        if (npt) {
            gn_team_launch<false>(ta, npt, s);
            return clora_check_launch();
        }
    }
    if (res.nt) {
        const dim3 rgrid(1, a.nslab, B);
        if (deferred_here) {
            a.fin_partial = src->partial; a.fin_splits = src->splits; a.fin_epi = src->epi; a.xcopy = (half_t*)src->C;
            if (res.nt == 256 && res.npt == 4) hipLaunchKernelGGL((gn_fwd_resident_kernel<256, 4, true>), rgrid, dim3(256), 0, s, a);
            else if (res.nt == 256 && res.npt == 8) hipLaunchKernelGGL((gn_fwd_resident_kernel<256, 8, true>), rgrid, dim3(256), 0, s, a);
            else if (res.nt == 256) hipLaunchKernelGGL((gn_fwd_resident_kernel<256, 16, true>), rgrid, dim3(256), 0, s, a);
            else if (res.npt <= 8) hipLaunchKernelGGL((gn_fwd_resident_kernel<512, 8, true>), rgrid, dim3(512), 0, s, a);
            else hipLaunchKernelGGL((gn_fwd_resident_kernel<512, 16, true>), rgrid, dim3(512), 0, s, a);
            return clora_check_launch();
        }
        if (res.nt == 256 && res.npt == 4) hipLaunchKernelGGL((gn_fwd_resident_kernel<256, 4>), rgrid, dim3(256), 0, s, a);
        else if (res.nt == 256 && res.npt == 8) hipLaunchKernelGGL((gn_fwd_resident_kernel<256, 8>), rgrid, dim3(256), 0, s, a);
        else if (res.nt == 256) hipLaunchKernelGGL((gn_fwd_resident_kernel<256, 16>), rgrid, dim3(256), 0, s, a);
        else if (res.npt <= 8) hipLaunchKernelGGL((gn_fwd_resident_kernel<512, 8>), rgrid, dim3(512), 0, s, a);
        else hipLaunchKernelGGL((gn_fwd_resident_kernel<512, 16>), rgrid, dim3(512), 0, s, a);
        return clora_check_launch();
    }
    int rc = gn_plan(a, workspace, workspace_bytes, false, false);
    if (rc != CLORA_OK) return rc;
    const dim3 grid(a.nchunk, a.nslab, B);
    if (a.CS / 8 > 256) {                                        // slabs wider than 2048 channels: two column chunks per thread
        hipLaunchKernelGGL(gn_fwd_partial_kernel<2>, grid, dim3(256), 0, s, a);
        hipLaunchKernelGGL(gn_fwd_apply2_kernel<2>, grid, dim3(256), 0, s, a);
    } else {
        if (clora_option(CLORA_OPT_GN_UNROLL)) {                 // twice the rows in flight per thread (same accumulation order: same bits)
            hipLaunchKernelGGL((gn_fwd_partial_kernel<1, 2>), grid, dim3(256), 0, s, a);
            hipLaunchKernelGGL((gn_fwd_apply2_kernel<1, 2>), grid, dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL(gn_fwd_partial_kernel<1>, grid, dim3(256), 0, s, a);
            hipLaunchKernelGGL(gn_fwd_apply2_kernel<1>, grid, dim3(256), 0, s, a);
        }
    }
    return clora_check_launch();
}
}  // namespace

extern "C" size_t clora_groupnorm_team_state_bytes(void) { return kTeamStateBytes; }

extern "C" int clora_groupnorm_fwd_f16_ex(const clora_half* x, const clora_half* x2, int Ca, const clora_deferred_t* src, clora_half* xcopy,
                                          clora_half* y, const float* gamma, const float* beta, float* stats, int B, int HW, int C,
                                          int G, float eps, int fuse_silu, void* workspace, size_t workspace_bytes, void* stream) {
    return gn_fwd_impl(x, x2, Ca, src, xcopy, y, gamma, beta, stats, B, HW, C, G, eps, fuse_silu, nullptr, 0, workspace, workspace_bytes, stream);
}

extern "C" int clora_groupnorm_fwd_f16_team(const clora_half* x, const clora_half* x2, int Ca, const clora_deferred_t* src, clora_half* xcopy,
                                            clora_half* y, const float* gamma, const float* beta, float* stats, int B, int HW, int C,
                                            int G, float eps, int fuse_silu, void* team_state, size_t team_state_bytes, void* workspace,
                                            size_t workspace_bytes, void* stream) {
    return gn_fwd_impl(x, x2, Ca, src, xcopy, y, gamma, beta, stats, B, HW, C, G, eps, fuse_silu, team_state, team_state_bytes, workspace,
                       workspace_bytes, stream);
}

extern "C" int clora_groupnorm_fwd_f16(const clora_half* x, clora_half* y, const float* gamma, const float* beta,
                                       float* stats, int B, int HW, int C, int G, float eps, int fuse_silu,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    return clora_groupnorm_fwd_f16_ex(x, nullptr, 0, nullptr, nullptr, y, gamma, beta, stats, B, HW, C, G, eps, fuse_silu, workspace,
                                      workspace_bytes, stream);
}

namespace {
int gn_bwd_impl(const clora_half* x, const clora_half* dy, const clora_deferred_t* dy_src, const clora_half* dres,
                clora_half* dx, clora_half* dx2, int Ca, const float* gamma, const float* beta,
                const float* stats, float* dgamma, float* dbeta, int B, int HW, int C, int G, int fuse_silu,
                int accumulate_params, void* team_state, size_t team_state_bytes, void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !dx || !gamma || !beta || !stats || ((dgamma == nullptr) != (dbeta == nullptr))) return CLORA_ERR_ARG;
    if (B <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 64 || (C % G) || (C & 7) || C > 4096) return CLORA_ERR_ARG;
    if (dx2 && (Ca <= 0 || Ca >= C || (Ca & 7))) return CLORA_ERR_ARG;
    const bool deferred = dy_src && dy_src->splits > 0;
    if (deferred) {
        if (!deferred_fits(dy_src, B * HW, C)) return CLORA_ERR_ARG;
        dy = dy_src->C;
    }
    if (!dy) return CLORA_ERR_ARG;
    GnArgs a = GnArgs();
    a.x = (const half_t*)x; a.dy = (const half_t*)dy; a.dres = (const half_t*)dres; a.y = (half_t*)dx; a.gamma = gamma; a.beta = beta;
    a.stats = const_cast<float*>(stats); a.dgamma = dgamma; a.dbeta = dbeta;
    a.B = B; a.HW = HW; a.C = C; a.G = G; a.fuse_silu = fuse_silu; a.accumulate_params = accumulate_params;
    a.y2 = (half_t*)dx2; a.Ca = dx2 ? Ca : C;
    hipStream_t s = (hipStream_t)stream;
    const GnResident res = gn_resident_plan(a, true, dgamma != nullptr);
    const bool deferred_here = deferred && res.nt && res.npt <= clora_option(CLORA_OPT_DEFER_MAX_ROWS);      // see the forward
    if (deferred && !deferred_here) {
        const int rc = clora_finish_deferred(dy_src, stream);
        if (rc != CLORA_OK) return rc;
    }
    const bool fold_shape = res.nt && res.npt <= clora_option(CLORA_OPT_DEFER_MAX_ROWS);      // see the forward
    if (!fold_shape && (!res.nt || clora_option(CLORA_OPT_GN_TEAM) >= 2)) {
        GnArgs ta = a;
        const int npt = gn_team_plan(ta, true, dgamma != nullptr, team_state, team_state_bytes);
        if (npt) {
            gn_team_launch<true>(ta, npt, s);
            return clora_check_launch();
        }
    }
    if (res.nt) {
        const dim3 rgrid(1, a.nslab, B);
        if (deferred_here) {
            a.fin_partial = dy_src->partial; a.fin_splits = dy_src->splits; a.fin_epi = dy_src->epi;
            if (res.nt == 256 && res.npt == 4) hipLaunchKernelGGL((gn_bwd_resident_kernel<256, 4, true>), rgrid, dim3(256), 0, s, a);
            else if (res.nt == 256) hipLaunchKernelGGL((gn_bwd_resident_kernel<256, 8, true>), rgrid, dim3(256), 0, s, a);
            else hipLaunchKernelGGL((gn_bwd_resident_kernel<512, 8, true>), rgrid, dim3(512), 0, s, a);
            return clora_check_launch();
        }
        if (res.nt == 256 && res.npt == 4) hipLaunchKernelGGL((gn_bwd_resident_kernel<256, 4>), rgrid, dim3(256), 0, s, a);
        else if (res.nt == 256) hipLaunchKernelGGL((gn_bwd_resident_kernel<256, 8>), rgrid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((gn_bwd_resident_kernel<512, 8>), rgrid, dim3(512), 0, s, a);
        return clora_check_launch();
    }
    int rc = gn_plan(a, workspace, workspace_bytes, true, dgamma != nullptr);
    if (rc != CLORA_OK) return rc;
    const dim3 grid(a.nchunk, a.nslab, B);
    const bool two = a.CS / 8 > 256;                             // slabs wider than 2048 channels
    if (two) hipLaunchKernelGGL(gn_bwd_partial_kernel<2>, grid, dim3(256), 0, s, a);
    else if (clora_option(CLORA_OPT_GN_UNROLL)) hipLaunchKernelGGL((gn_bwd_partial_kernel<1, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gn_bwd_partial_kernel<1>, grid, dim3(256), 0, s, a);
    a.params_in_apply = (dgamma && (long)a.nchunk * a.nslab * B >= clora_cdiv(C, 32)) ? 1 : 0;
    if (dgamma && !a.params_in_apply) hipLaunchKernelGGL(gn_bwd_params_kernel, dim3(clora_cdiv(C, 32)), dim3(256), 0, s, a);
    if (two) hipLaunchKernelGGL(gn_bwd_apply2_kernel<2>, grid, dim3(256), 0, s, a);
    else if (clora_option(CLORA_OPT_GN_UNROLL)) hipLaunchKernelGGL((gn_bwd_apply2_kernel<1, 2>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(gn_bwd_apply2_kernel<1>, grid, dim3(256), 0, s, a);
    return clora_check_launch();
}
}  // namespace

extern "C" int clora_groupnorm_bwd_f16_ex(const clora_half* x, const clora_half* dy, const clora_deferred_t* dy_src, const clora_half* dres,
                                          clora_half* dx, clora_half* dx2, int Ca, const float* gamma, const float* beta,
                                          const float* stats, float* dgamma, float* dbeta, int B, int HW, int C, int G, int fuse_silu,
                                          int accumulate_params, void* workspace, size_t workspace_bytes, void* stream) {
    return gn_bwd_impl(x, dy, dy_src, dres, dx, dx2, Ca, gamma, beta, stats, dgamma, dbeta, B, HW, C, G, fuse_silu, accumulate_params, nullptr, 0,
                       workspace, workspace_bytes, stream);
}

extern "C" int clora_groupnorm_bwd_f16_team(const clora_half* x, const clora_half* dy, const clora_deferred_t* dy_src, const clora_half* dres,
                                            clora_half* dx, clora_half* dx2, int Ca, const float* gamma, const float* beta,
                                            const float* stats, float* dgamma, float* dbeta, int B, int HW, int C, int G, int fuse_silu,
                                            int accumulate_params, void* team_state, size_t team_state_bytes, void* workspace,
                                            size_t workspace_bytes, void* stream) {
    return gn_bwd_impl(x, dy, dy_src, dres, dx, dx2, Ca, gamma, beta, stats, dgamma, dbeta, B, HW, C, G, fuse_silu, accumulate_params, team_state,
                       team_state_bytes, workspace, workspace_bytes, stream);
}

extern "C" int clora_groupnorm_bwd_f16(const clora_half* x, const clora_half* dy, const clora_half* dres, clora_half* dx, const float* gamma,
                                       const float* beta, const float* stats, float* dgamma, float* dbeta, int B,
                                       int HW, int C, int G, int fuse_silu, int accumulate_params, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    return clora_groupnorm_bwd_f16_ex(x, dy, nullptr, dres, dx, nullptr, 0, gamma, beta, stats, dgamma, dbeta, B, HW, C, G, fuse_silu,
                                      accumulate_params, workspace, workspace_bytes, stream);
}

extern "C" int clora_layernorm_fwd_f16(const clora_half* x, clora_half* y, const float* gamma, const float* beta, int M,
                                       int C, float eps, void* stream) {
    if (!x || !y || !gamma || !beta || M <= 0 || C <= 0 || (C & 7) || C / 8 > 64 * kLnCols) return CLORA_ERR_ARG;
    LnArgs a = LnArgs();
    a.x = (const half_t*)x; a.y = (half_t*)y; a.gamma = gamma; a.beta = beta; a.M = M; a.C = C; a.eps = eps;
    launch_layernorm<false>(a, (hipStream_t)stream);
    return clora_check_launch();
}

extern "C" int clora_layernorm_bwd_f16_ex(const clora_half* x, const clora_half* dy, const clora_deferred_t* dy_src, const clora_half* dres,
                                          clora_half* dx, const float* gamma, int M, int C, float eps, void* stream) {
    if (!x || !dx || !gamma || M <= 0 || C <= 0 || (C & 7) || C / 8 > 64 * kLnCols) return CLORA_ERR_ARG;
    const bool deferred = dy_src && dy_src->splits > 0;
    if (deferred) {
        if (!deferred_fits(dy_src, M, C)) return CLORA_ERR_ARG;
        dy = dy_src->C;
    }
    if (!dy) return CLORA_ERR_ARG;
    LnArgs a = LnArgs();
    a.x = (const half_t*)x; a.dy = (const half_t*)dy; a.dres = (const half_t*)dres; a.y = (half_t*)dx; a.gamma = gamma; a.M = M; a.C = C; a.eps = eps;
    // (measured: the folded LayerNorm backward costs 13.0 / 18.4 us against 7.5 / 5.7 + 6.0 for finish + plain at C = 640 / 1280 -- a lane
    // folds its 2-3 chunks one after the other; it is taken only with the "defer_max_rows" knob at 16, the A/B setting)
    if (deferred && ln_rows_plan(a, true) && clora_option(CLORA_OPT_DEFER_MAX_ROWS) >= 16) {
        a.fin_partial = dy_src->partial; a.fin_splits = dy_src->splits; a.fin_epi = dy_src->epi;
        launch_layernorm<true, true>(a, (hipStream_t)stream);
        return clora_check_launch();
    }
    if (deferred) {
        const int rc = clora_finish_deferred(dy_src, stream);
        if (rc != CLORA_OK) return rc;
    }
    launch_layernorm<true>(a, (hipStream_t)stream);
    return clora_check_launch();
}

extern "C" int clora_layernorm_bwd_f16(const clora_half* x, const clora_half* dy, const clora_half* dres, clora_half* dx,
                                       const float* gamma, int M, int C, float eps, void* stream) {
    return clora_layernorm_bwd_f16_ex(x, dy, nullptr, dres, dx, gamma, M, C, eps, stream);
}

extern "C" int clora_softmax_rows_f16(const clora_half* x, clora_half* y, int rows, int cols, int ld, float scale,
                                      void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || (cols & 7) || (ld & 7) || ld < cols || cols / 8 > 64 * kSmCols || !(scale > 0.f))
        return CLORA_ERR_ARG;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(clora_cdiv(rows, 4)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)x, (half_t*)y, rows, cols, ld, scale);
    return clora_check_launch();
}
