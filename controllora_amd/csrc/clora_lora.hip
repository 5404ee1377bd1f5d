// clora_lora.hip -- the rank-r adapter pieces that are NOT folded into the GEMM epilogue.
//
// Reference semantics (upstream LoRALinearLayer): the adapter casts its fp16 input to
// fp32, multiplies by down[r,K] and up[N,r] in fp32 and casts back.  With r = 4 an unfused adapter is
// ~4 flop/byte (hopelessly HBM/launch bound: ~25-30 micro-kernels per attention site in the reference),
// so the work is split as
//   down : T[M,r] = X . D^T             one pass over X, fp32 VALU (this file) -- several adapters that
//                                        share X write disjoint column ranges of one T buffer
//   up   : acc += s * T . U^T            inside the projection GEMM epilogue (clora_gemm.hip)
//   up (explicit) for the control term   hidden + s*to_control(ctrl)     (this file)
//   wgrad: dU = s * dY^T . T,  dD = dT^T . X    skinny reductions over M, two deterministic stages, no atomics
#include "clora_common.h"
#include "../../include/clora.h"

namespace {

// T[m, toff+j] (+)= sum_k X[m,k] * D[j,k] on the matrix cores with fp32-equivalent accuracy:
// D (fp32) is split on the fly into hi + lo fp16 halves (|D - hi - lo| <= 2^-22 |D|), X is exact fp16, products
// are exact in the fp32 accumulator -- two v_mfma_f32_16x16x32_f16 per k-step.  One wave = 16 rows, fragments
// are loaded straight from global (16 B per lane, 64-B row segments), no LDS, no barriers.  The k loop is
// processed four k-steps at a time with ALL loads of the group issued before the first use: the kernel is a
// pure stream and would otherwise pay one full memory latency per k-step.
// d_kmajor: D[j][k] lives at D[k*ldd + j] (an up matrix [K, r] acting as U^T in the backward pass).
struct DownJobs {
    clora_lora_down_job_t j[CLORA_LORA_MAX_JOBS];
};

// KSPLIT (few rows: the 32x32 / 16x16 / 8x8 levels and the 77-token text context): one block = ONE 16-row group,
// its four waves each take a quarter of K and the partial tiles are folded through LDS -- 4x more blocks and a 4x
// shorter dependent load chain for what is a pure latency problem at those sizes.
// G = k-steps whose loads are in flight together (4 KB of X per wave at G = 4).  Bytes in flight are what bounds this kernel: at
// M = 16384 the non-split form has ONE block per CU = 16 KB in flight per CU = 2.6 TB/s by Little's law, exactly what it measured.
// NW = waves per block (KSPLIT: the waves of a block share one 16-row group and split K NW ways -- 16 at the 16x16 / 8x8 levels,
// where 64 / 16 row groups are all the parallelism M offers and a wave's K chain is the whole launch).
template <bool KSPLIT, int G = 4, int NW = 4>
__global__ __launch_bounds__(NW * 64) void lora_down_kernel(DownJobs jobs) {
    const clora_lora_down_job_t& p = jobs.j[blockIdx.y];
    const half_t* X = (const half_t*)p.X;
    const float* __restrict__ D = p.D;
    float* __restrict__ T = p.T;
    const int ldx = p.ldx, ldd = p.ldd, ldt = p.ldt, toff = p.toff, M = p.M, K = p.K, R = p.R;
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63, g = l >> 4, li = l & 15;
    const int m0 = KSPLIT ? blockIdx.x * 16 : (blockIdx.x * NW + w) * 16;
    if (m0 >= M) return;                                    // jobs of one launch may differ in M (block/wave-uniform exit)
    int kbeg = 0, kend = K;
    if (KSPLIT) {
        const int per = ((K + 31) / 32 + NW - 1) / NW * 32;
        kbeg = w * per;
        kend = (kbeg + per < K) ? kbeg + per : K;
    }
    const int m = m0 + li;
    const bool mok = m < M;
    size_t xoff = (size_t)(mok ? (p.x_rows > 0 ? m % p.x_rows : m) : 0) * ldx;
    bool jok = li < R;
    const int jj = jok ? li : 0;
    const int d_kmajor = p.d_kmajor;
    const float dscale = p.d_scale;
    floatx4 acc = zero4f();
    const int npass = p.X2 ? 2 : 1;                        // second input: (X + X2) . D^T by linearity, same accumulator
    for (int pass = 0; pass < npass; ++pass) {
    if (pass == 1) {
        X = (const half_t*)p.X2; xoff = (size_t)(mok ? (p.x2_rows > 0 ? m % p.x2_rows : m) : 0) * p.ldx2;
        if (p.r2 > 0) jok = li < p.r2;                      // stacked adapters: the second input feeds the first r2 rows of D only
    }
    for (int k0 = kbeg; k0 < kend; k0 += 32 * G) {
        half8 a[G];
        floatx4 d0[G], d1[G];
#pragma unroll
        for (int u = 0; u < G; ++u) {                      // issue every load of the group ...
            const int k = k0 + u * 32 + g * 8;
            a[u] = zero8(); d0[u] = zero4f(); d1[u] = zero4f();
            if (k < kend) {                                // K % 8 == 0: a chunk is all-valid or all-out
                if (mok) a[u] = ld8(X + xoff + k);
                if (jok) {
                    if (d_kmajor) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            d0[u][e] = D[(size_t)(k + e) * ldd + jj];
                            d1[u][e] = D[(size_t)(k + 4 + e) * ldd + jj];
                        }
                    } else {
                        d0[u] = *reinterpret_cast<const floatx4*>(D + (size_t)jj * ldd + k);
                        d1[u] = *reinterpret_cast<const floatx4*>(D + (size_t)jj * ldd + k + 4);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < G; ++u) {                      // ... then split D into hi/lo halves and multiply
            half8 bh, bl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float f0 = d0[u][e] * dscale, f1 = d1[u][e] * dscale;
                const half_t h0 = (half_t)f0, h1 = (half_t)f1;
                bh[e] = h0; bh[4 + e] = h1;
                bl[e] = (half_t)(f0 - (float)h0); bl[4 + e] = (half_t)(f1 - (float)h1);
            }
            acc = mfma16(a[u], bh, acc);
            acc = mfma16(a[u], bl, acc);
        }
    }
    }
    if (KSPLIT) {                                           // fold the NW K-slices (fixed order) into wave 0
        __shared__ floatx4 part[NW - 1][64];
        if (w > 0) part[w - 1][l] = acc;
        __syncthreads();
        if (w > 0) return;
#pragma unroll
        for (int q = 0; q < NW - 1; ++q) {
            const floatx4 o = part[q][l];
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += o[r];
        }
    }
    // C layout: lane holds T[m0 + 4g + r][toff + li]
    if (li < R) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mm = m0 + 4 * g + r;
            if (mm < M) {
                float* out = T + (size_t)mm * ldt + toff + li;
                float v = acc[r];
                if (p.T_in && li < p.t_in_r) v += p.T_in[(size_t)(p.t_in_rows > 0 ? mm % p.t_in_rows : mm) * p.ldt_in + li];
                *out = p.accumulate ? *out + v : v;
            }
        }
    }
}

struct UpArgs {
    const half_t* base;
    const float* T;
    const float* U;
    half_t* Y;
    int ldb, ldt, toff, ldu, ldy, M, N, R, u_tr;
    float scale;
};

struct UpJobs {
    UpArgs j[CLORA_LORA_MAX_JOBS];
};

__global__ __launch_bounds__(256) void lora_up_kernel(UpJobs jobs) {
    const UpArgs& p = jobs.j[blockIdx.y];
    const int NC = p.N / 8;
    const size_t total = (size_t)p.M * NC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t m = i / NC;
        const int n = (int)(i - m * NC) * 8;
        const float* tr = p.T + m * p.ldt + p.toff;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (p.u_tr && (p.ldu & 3) == 0) {                 // U^T rows are contiguous along n: two float4 per rank index
            for (int j = 0; j < p.R; ++j) {
                const float tv = tr[j];
                const floatx4 u0 = *reinterpret_cast<const floatx4*>(p.U + (size_t)j * p.ldu + n);
                const floatx4 u1 = *reinterpret_cast<const floatx4*>(p.U + (size_t)j * p.ldu + n + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { acc[e] += tv * u0[e]; acc[4 + e] += tv * u1[e]; }
            }
        } else if (!p.u_tr && p.R == 4 && p.ldu == 4 && ((p.ldt | p.toff) & 3) == 0) {   // the common rank-4 case
            const floatx4 t4 = *reinterpret_cast<const floatx4*>(tr);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const floatx4 u = *reinterpret_cast<const floatx4*>(p.U + (size_t)(n + e) * 4);
                acc[e] += t4[0] * u[0] + t4[1] * u[1] + t4[2] * u[2] + t4[3] * u[3];
            }
        } else {
            for (int j = 0; j < p.R; ++j) {
                const float tv = tr[j];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += tv * (p.u_tr ? p.U[(size_t)j * p.ldu + n + e] : p.U[(size_t)(n + e) * p.ldu + j]);
            }
        }
        half8 o;
        if (p.base) {
            const half8 bv = ld8(p.base + m * p.ldb + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
#ifdef CLORA_RES_ADD_TWICE
                const half_t c16 = (half_t)(p.scale * (float)(half_t)acc[e]);  // fp16(scale * fp16(up(down))): the reference's fp16 arithmetic
                o[e] = (half_t)((float)bv[e] + (float)c16);
#else
                o[e] = (half_t)((float)bv[e] + p.scale * acc[e]);              // round 6: base + update formed in fp32, ONE rounding (clora_epilogue.h CLORA_RES_ADD)
#endif
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)(p.scale * (float)(half_t)acc[e]);
        }
        st8(p.Y + m * p.ldy + n, o);
    }
}

// G[n*gs_n + j*gs_j] += scale * sum_m A[m, n] * T[m, toff + j]      (adapter weight gradients)
// Pure HBM stream over A, reduced in two deterministic stages (same-address fp32 atomics serialise at ~0.2 us
// each on this chip -- measured -- so they are avoided entirely):
//   stage 1: block = 4 waves x rows_per_block/4 rows; lane owns 8 columns (16-byte loads, a wave spans 512
//            columns), 16 rows in flight, wave-uniform T rows by batched scalar loads; the 4 waves fold through
//            LDS and the block writes its [N x RT] slab to the workspace;
//   stage 2: 64 outputs x 4 slab-lanes per block fold the slabs in a fixed order and add into G.
struct WgradJobs {
    clora_lora_wgrad_job_t j[CLORA_LORA_WGRAD_MAX_JOBS];
    float* part[CLORA_LORA_WGRAD_MAX_JOBS];       // slab area of each job
    int rpb[CLORA_LORA_WGRAD_MAX_JOBS], nblk[CLORA_LORA_WGRAD_MAX_JOBS];
};
static_assert(sizeof(WgradJobs) <= 4096, "kernel arguments are limited to 4 KB");

template <int RT>
__global__ __launch_bounds__(256) void lora_wgrad_kernel(WgradJobs jobs) {
    __shared__ float red[3 * 64 * 8 * RT / ((RT > 8) ? 2 : 1)];
    const int job = blockIdx.z;
    const clora_lora_wgrad_job_t& p = jobs.j[job];
    if ((int)blockIdx.y >= jobs.nblk[job] || (int)blockIdx.x * 512 >= p.N) return;     // block-uniform exit
    const half_t* __restrict__ A = (const half_t*)p.A;
    const half_t* __restrict__ A2 = (const half_t*)p.A2;
    const float* __restrict__ T = p.T;
    const int lda = p.lda, lda2 = p.lda2, ldt = p.ldt, toff = p.toff, M = p.M, N = p.N, R = p.R, a_rows = p.a_rows;
    const int rows_per_block = jobs.rpb[job];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int n = (blockIdx.x * 64 + l) * 8;
    const int rpw = rows_per_block / 4;
    const int m_beg = blockIdx.y * rows_per_block + w * rpw;
    int m_end = m_beg + rpw;
    if (m_end > M) m_end = M;
    const bool nok = n < N;  // N % 8 == 0
    float acc[8][RT];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int j = 0; j < RT; ++j) acc[e][j] = 0.f;
    constexpr int UN = (RT <= 8) ? 16 : 8;     // rows in flight per wave
    const int nc = nok ? n : 0;                                  // lanes past N re-read column 0 and are masked: branch-free loads
    for (int mb = m_beg; mb < m_end; mb += UN) {
        half8 a[UN];
        float tv[UN][RT];
        // Every load of the batch is issued before the first use.  (With `if (nok && m < m_end) a[u] = ld8(...)` the
        // compiler emitted load -> s_waitcnt vmcnt(0) per row pair: the "16 rows in flight" were 2; rows past m_end now
        // re-read row m_end - 1 and are zeroed afterwards, CLORA_KEEP_PURE keeps the loads out of the masked branch.)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int m = (mb + u < m_end) ? mb + u : m_end - 1;
            const int ar = a_rows > 0 ? m % a_rows : m;
            a[u] = ld8(A + (size_t)ar * lda + nc);
        }
        half8 a2[UN];
        if (A2) {                                                // adapter input = fp16(A + A2), as the reference forms it (job-uniform)
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const int m = (mb + u < m_end) ? mb + u : m_end - 1;
                a2[u] = ld8(A2 + (size_t)m * lda2 + nc);
            }
        }
        // wave-uniform T rows (scalar loads), issued while the vector loads are in flight
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int m = (mb + u < m_end) ? mb + u : m_end - 1;
            const float* tr = T + (size_t)m * ldt + toff;
#pragma unroll
            for (int j = 0; j < RT; ++j) tv[u][j] = (j < R && mb + u < m_end && m_end > m_beg) ? tr[j] : 0.f;
        }
        if (A2) {
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                CLORA_KEEP_PURE(a2[u]);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[u][e] = (half_t)((float)a[u][e] + (float)a2[u][e]);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            CLORA_KEEP_PURE(a[u]);
            if (!(nok && mb + u < m_end)) a[u] = zero8();
        }
#pragma unroll
        for (int u = 0; u < UN; ++u)
#pragma unroll
            for (int j = 0; j < RT; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e][j] += (float)a[u][e] * tv[u][j];
    }
    // fold the 4 waves (fixed order) -- in two halves of the 8 columns when RT = 16 to stay inside 48 KB of LDS
    constexpr int HALF = (RT > 8) ? 2 : 1, EC = 8 / HALF;
    float* out = jobs.part[job] + ((size_t)blockIdx.y * N + n) * RT;
#pragma unroll
    for (int hh = 0; hh < HALF; ++hh) {
        if (hh) __syncthreads();
        if (w > 0) {
#pragma unroll
            for (int e = 0; e < EC; ++e)
#pragma unroll
                for (int j = 0; j < RT; ++j) red[(((w - 1) * 64 + l) * EC + e) * RT + j] = acc[hh * EC + e][j];
        }
        __syncthreads();
        if (w == 0 && nok) {
#pragma unroll
            for (int e = 0; e < EC; ++e)
#pragma unroll
                for (int j = 0; j < RT; ++j) {
                    float sacc = acc[hh * EC + e][j];
#pragma unroll
                    for (int ww = 0; ww < 3; ++ww) sacc += red[((ww * 64 + l) * EC + e) * RT + j];
                    out[(hh * EC + e) * RT + j] = sacc;
                }
        }
    }
}

__global__ __launch_bounds__(256) void lora_wgrad_finish_kernel(WgradJobs jobs, int RT) {
    __shared__ float red[256];
    const int job = blockIdx.y;
    const clora_lora_wgrad_job_t& p = jobs.j[job];
    const int t = threadIdx.x, o = t & 63, q = t >> 6;
    const int i = blockIdx.x * 64 + o;
    const int total = p.N * RT, nblk = jobs.nblk[job];
    if ((int)blockIdx.x * 64 >= total) return;               // block-uniform exit
    const float* part = jobs.part[job];
    float s = 0.f;
    if (i < total) {
        const size_t stride = (size_t)total;
        int b = q;
        for (; b + 28 < nblk; b += 32) {        // 8 independent loads in flight
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + 4 * u) * stride + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < nblk; b += 4) s += part[(size_t)b * stride + i];
    }
    red[t] = s;
    __syncthreads();
    if (q == 0 && i < total) {
        const int n = i / RT, j = i - n * RT;
        if (j < p.R) {
            s = red[o] + red[64 + o] + red[128 + o] + red[192 + o];
            float* dst = p.G + (size_t)n * p.gs_n + (size_t)j * p.gs_j;
            *dst += p.scale * s;
        }
    }
}

}  // namespace

extern "C" int clora_lora_down_multi_f16(const clora_lora_down_job_t* jobs, int njobs, void* stream) {
    if (!jobs || njobs <= 0 || njobs > CLORA_LORA_MAX_JOBS) return CLORA_ERR_ARG;
    DownJobs dj;
    int maxM = 0;
    for (int i = 0; i < njobs; ++i) {
        const clora_lora_down_job_t& j = jobs[i];
        if (!j.X || !j.D || !j.T || j.M <= 0 || j.K <= 0 || j.R <= 0 || j.R > 16 || (j.K & 7) || (j.ldx & 7)) return CLORA_ERR_ARG;
        if (j.X2 && (j.ldx2 & 7)) return CLORA_ERR_ARG;
        if (j.T_in && (j.t_in_r <= 0 || j.t_in_r > j.R || j.t_in_rows < 0)) return CLORA_ERR_ARG;
        if (!j.d_kmajor && ((j.ldd & 3) || ((uintptr_t)j.D & 15))) return CLORA_ERR_ARG;
        dj.j[i] = j;
        if (j.M > maxM) maxM = j.M;
    }
    // option "lora_down_mode": 0 = K-split up to 4096 rows, one wave per 16 rows x all of K above (round-1/2 behaviour);
    // 1 (default) = K-split at every size (4x the waves, each a quarter of K: more bytes in flight per CU at M = 16384: 11.8 -> 9.5 us
    //     for the stacked q|k|v job, bench line -0.13 ms, profiles/r03_lora_down_ab.txt) and sixteen waves per row group up to 1024 rows;
    // 2 = as 0 with eight k-steps in flight per wave above 4096 rows
    const int mode = clora_option(CLORA_OPT_LORA_DOWN_MODE);
    int maxK = 0;
    for (int i = 0; i < njobs; ++i) if (jobs[i].K > maxK) maxK = jobs[i].K;
    if (mode == 1 && maxM <= 1024 && maxK >= 1024)          // 64 / 16 row groups: sixteen waves split K (each >= 64 of it)
        hipLaunchKernelGGL((lora_down_kernel<true, 4, 16>), dim3(clora_cdiv(maxM, 16), njobs), dim3(1024), 0, (hipStream_t)stream, dj);
    else if (maxM <= 4096 || mode == 1)
        hipLaunchKernelGGL((lora_down_kernel<true, 4>), dim3(clora_cdiv(maxM, 16), njobs), dim3(256), 0, (hipStream_t)stream, dj);
    else if (mode == 2)
        hipLaunchKernelGGL((lora_down_kernel<false, 8>), dim3(clora_cdiv(maxM, 64), njobs), dim3(256), 0, (hipStream_t)stream, dj);
    else
        hipLaunchKernelGGL((lora_down_kernel<false, 4>), dim3(clora_cdiv(maxM, 64), njobs), dim3(256), 0, (hipStream_t)stream, dj);
    return clora_check_launch();
}

extern "C" int clora_lora_down_f16(const clora_half* X, int ldx, const float* D, int ldd, float* T, int ldt, int toff,
                                   int M, int K, int R, int accumulate, int x_rows, int d_kmajor, float d_scale,
                                   void* stream) {
    if (R <= 0) return CLORA_ERR_ARG;
    for (int r0 = 0; r0 < R; r0 += 16) {  // ranks > 16 (danbooru-sketch control_rank 256) take several passes over X
        clora_lora_down_job_t j;
        j.X = X; j.ldx = ldx; j.D = d_kmajor ? D + r0 : (D ? D + (size_t)r0 * ldd : D); j.ldd = ldd; j.T = T; j.ldt = ldt;
        j.toff = toff + r0; j.M = M; j.K = K; j.R = (R - r0 < 16) ? R - r0 : 16; j.accumulate = accumulate;
        j.x_rows = x_rows; j.d_kmajor = d_kmajor; j.d_scale = d_scale; j.X2 = nullptr; j.ldx2 = 0; j.x2_rows = 0; j.r2 = 0;
        j.T_in = nullptr; j.ldt_in = 0; j.t_in_rows = 0; j.t_in_r = 0;
        const int rc = clora_lora_down_multi_f16(&j, 1, stream);
        if (rc != CLORA_OK) return rc;
    }
    return CLORA_OK;
}

extern "C" int clora_lora_up_f16(const clora_half* base, int ldb, const float* T, int ldt, int toff, const float* U,
                                 int ldu, int u_transposed, clora_half* Y, int ldy, int M, int N, int R, float scale,
                                 void* stream) {
    if (!T || !U || !Y || M <= 0 || N <= 0 || R <= 0 || (N & 7) || (ldy & 7) || (base && (ldb & 7))) return CLORA_ERR_ARG;
    UpArgs a;
    a.base = (const half_t*)base; a.T = T; a.U = U; a.Y = (half_t*)Y;
    a.ldb = ldb; a.ldt = ldt; a.toff = toff; a.ldu = ldu; a.ldy = ldy; a.M = M; a.N = N; a.R = R; a.scale = scale;
    a.u_tr = u_transposed;
    size_t blocks = ((size_t)M * (N / 8) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    UpJobs uj;
    uj.j[0] = a;
    hipLaunchKernelGGL(lora_up_kernel, dim3((unsigned)blocks, 1), dim3(256), 0, (hipStream_t)stream, uj);
    return clora_check_launch();
}

extern "C" int clora_lora_up_multi_f16(const clora_lora_up_job_t* jobs, int njobs, void* stream) {
    if (!jobs || njobs <= 0 || njobs > CLORA_LORA_MAX_JOBS) return CLORA_ERR_ARG;
    UpJobs uj;
    size_t blocks = 1;
    for (int i = 0; i < njobs; ++i) {
        const clora_lora_up_job_t& j = jobs[i];
        if (!j.T || !j.U || !j.Y || j.M <= 0 || j.N <= 0 || j.R <= 0 || (j.N & 7) || (j.ldy & 7) || (j.base && (j.ldb & 7))) return CLORA_ERR_ARG;
        UpArgs& a = uj.j[i];
        a.base = (const half_t*)j.base; a.T = j.T; a.U = j.U; a.Y = (half_t*)j.Y;
        a.ldb = j.ldb; a.ldt = j.ldt; a.toff = j.toff; a.ldu = j.ldu; a.ldy = j.ldy; a.M = j.M; a.N = j.N; a.R = j.R;
        a.scale = j.scale; a.u_tr = j.u_transposed;
        const size_t b = ((size_t)j.M * (j.N / 8) + 255) / 256;
        if (b > blocks) blocks = b;
    }
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(lora_up_kernel, dim3((unsigned)blocks, njobs), dim3(256), 0, (hipStream_t)stream, uj);
    return clora_check_launch();
}

namespace {
int wgrad_rows_per_block(int M, int N) {
    int rpb = 64;   // 4 waves x one 16-row batch; grow until the grid is at most ~256 blocks
    while ((long)clora_cdiv(N, 512) * clora_cdiv(M, rpb) > 256) rpb *= 2;
    return rpb;
}
int wgrad_rt(int R) { return R <= 4 ? 4 : (R <= 8 ? 8 : 16); }
}  // namespace

extern "C" size_t clora_lora_wgrad_workspace_bytes(int M, int N, int R) {
    return (size_t)clora_cdiv(M, wgrad_rows_per_block(M, N)) * N * wgrad_rt(R > 16 ? 16 : R) * sizeof(float);
}

/* all jobs of one launch must share the rank class (R <= 4, <= 8 or <= 16) */
extern "C" int clora_lora_wgrad_multi_f16(const clora_lora_wgrad_job_t* jobs, int njobs, void* workspace,
                                          size_t workspace_bytes, void* stream) {
    if (!jobs || njobs <= 0 || njobs > CLORA_LORA_WGRAD_MAX_JOBS) return CLORA_ERR_ARG;
    WgradJobs wj;
    const int rt = wgrad_rt(jobs[0].R);
    size_t off = 0;
    int gx = 0, gy = 0, maxNRT = 0;
    // a launch of many jobs already fills the chip: grow the row chunk (fewer, longer blocks: 4x less slab traffic and a deeper load
    // pipeline per wave) until the launch is down to ~1024 blocks; never below the per-job choice the workspace was sized for
    int grow = 1;
    {
        long total = 0;
        for (int i = 0; i < njobs; ++i) total += (long)clora_cdiv(jobs[i].N, 512) * clora_cdiv(jobs[i].M, wgrad_rows_per_block(jobs[i].M, jobs[i].N));
        while (grow < 8 && total / (grow * 2) >= 1024) grow *= 2;
    }
    for (int i = 0; i < njobs; ++i) {
        const clora_lora_wgrad_job_t& j = jobs[i];
        if (!j.A || !j.T || !j.G || j.M <= 0 || j.N <= 0 || j.R <= 0 || j.R > 16 || (j.N & 7) || (j.lda & 7) || wgrad_rt(j.R) != rt)
            return CLORA_ERR_ARG;
        wj.j[i] = j;
        wj.rpb[i] = wgrad_rows_per_block(j.M, j.N) * grow;
        wj.nblk[i] = clora_cdiv(j.M, wj.rpb[i]);
        wj.part[i] = (float*)workspace + off;
        off += (size_t)wj.nblk[i] * j.N * rt;
        if (clora_cdiv(j.N, 512) > gx) gx = clora_cdiv(j.N, 512);
        if (wj.nblk[i] > gy) gy = wj.nblk[i];
        if (j.N * rt > maxNRT) maxNRT = j.N * rt;
    }
    if (!workspace || workspace_bytes < off * sizeof(float)) return CLORA_ERR_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(gx, gy, njobs);
    if (rt == 4) hipLaunchKernelGGL((lora_wgrad_kernel<4>), grid, dim3(256), 0, s, wj);
    else if (rt == 8) hipLaunchKernelGGL((lora_wgrad_kernel<8>), grid, dim3(256), 0, s, wj);
    else hipLaunchKernelGGL((lora_wgrad_kernel<16>), grid, dim3(256), 0, s, wj);
    hipLaunchKernelGGL(lora_wgrad_finish_kernel, dim3(clora_cdiv(maxNRT, 64), njobs), dim3(256), 0, s, wj, rt);
    return clora_check_launch();
}

extern "C" int clora_lora_wgrad_f16(const clora_half* A, int lda, const float* T, int ldt, int toff, float* G, int gs_n,
                                    int gs_j, int M, int N, int R, float scale, int a_rows, void* workspace,
                                    size_t workspace_bytes, void* stream) {
    if (R <= 0) return CLORA_ERR_ARG;
    for (int r0 = 0; r0 < R; r0 += 16) {
        clora_lora_wgrad_job_t j;
        j.A = A; j.lda = lda; j.T = T; j.ldt = ldt; j.toff = toff + r0; j.G = G + (size_t)r0 * gs_j; j.gs_n = gs_n; j.gs_j = gs_j;
        j.M = M; j.N = N; j.R = (R - r0 < 16) ? R - r0 : 16; j.scale = scale; j.a_rows = a_rows; j.A2 = nullptr; j.lda2 = 0;
        const int rc = clora_lora_wgrad_multi_f16(&j, 1, workspace, workspace_bytes, stream);
        if (rc != CLORA_OK) return rc;
    }
    return CLORA_OK;
}

// ------------------------------------------------------------------------------------------------
// clora_lora_pack_f16: fp32 adapter matrices -> the 8-row fp16 operand blocks of clora_epilogue_t.lora_dpack (rows 0..3 fp16(v),
// rows 4..7 fp16(v - fp16(v)), v = scale * D; rank <= 4).  One launch for a whole device-resident job table: grid.y = job.
namespace {
__global__ __launch_bounds__(256) void lora_pack_kernel(const clora_lora_pack_job_t* table, int njobs) {
    const clora_lora_pack_job_t j = table[blockIdx.y];
    half_t* out = (half_t*)j.out;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < j.K; k += gridDim.x * 256) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = 0.f;
            if (r < j.R) v = j.scale * (j.kmajor ? j.D[(size_t)k * j.ldd + r] : j.D[(size_t)r * j.ldd + k]);
            const half_t hi = (half_t)v;
            out[(size_t)r * j.K + k] = hi;
            out[(size_t)(4 + r) * j.K + k] = (half_t)(v - (float)hi);
        }
    }
}
}  // namespace

extern "C" int clora_lora_pack_f16(const clora_lora_pack_job_t* table, int njobs, int max_k, void* stream) {
    if (!table || njobs <= 0 || njobs > 65535 || max_k <= 0) return CLORA_ERR_ARG;     // (job.R <= 4 is the caller's contract: device table)
    hipLaunchKernelGGL(lora_pack_kernel, dim3(clora_cdiv(max_k, 256), njobs), dim3(256), 0, (hipStream_t)stream, table, njobs);
    return clora_check_launch();
}

// ------------------------------------------------------------------------------------------------
// The v1 control term in rank space (include/clora.h, clora_rank_*): 4 x r_c matrices and [rows, 4] / [rows, r_c] tensors only.
namespace {
constexpr int kRankMaxSites = 16, kRankMixRows = 256;      // rows of one rank_mix block (= one partial Gram sum)
struct RankSites { clora_rank_site_t s[kRankMaxSites]; };

// M[j, i] = scale * sum_c Dq[j, c] * Uc[c, i]: one block per site, threads stride over c, fixed-order LDS fold
__global__ __launch_bounds__(256) void rank_compose_kernel(RankSites a) {
    const clora_rank_site_t& p = a.s[blockIdx.x];
    __shared__ float red[256][33];
    const int t = threadIdx.x, rc = p.rc;
    float acc[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) acc[q] = 0.f;
    for (int c = t; c < p.C; c += 256) {
        float d[4], u[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = p.Dq[(size_t)j * p.lddq + c];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = i < rc ? p.Uc[(size_t)c * p.lduc + i] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[j * 8 + i] += d[j] * u[i];
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) red[t][q] = acc[q];
    __syncthreads();
    if (t < 32) {
        float s = 0.f;
        for (int r = 0; r < 256; ++r) s += red[r][t];
        const int j = t >> 3, i = t & 7;
        if (i < rc) p.M[j * rc + i] = p.scale * s;
    }
}

// forward: Tq = Tc_l . M^T.  backward: dTc_l = dTq . M and this block's partial Gram sum G[j, i] = sum_m dTq[m, j] Tc[m, i].
template <bool BWD>
__global__ __launch_bounds__(256) void rank_mix_kernel(RankSites a, float* gram_ws, int nblk) {
    const clora_rank_site_t& p = a.s[blockIdx.y];
    __shared__ float red[256][33];
    const int t = threadIdx.x, rc = p.rc, m = blockIdx.x * kRankMixRows + t;
    float Mv[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) Mv[q] = ((q & 7) < rc) ? p.M[(q >> 3) * rc + (q & 7)] : 0.f;
    float g[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) g[q] = 0.f;
    if (m < p.rows) {
        float tc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) tc[i] = i < rc ? p.Tc[(size_t)m * p.ldtc + p.toff + i] : 0.f;
        if (!BWD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += tc[i] * Mv[j * 8 + i];
                p.Tq[(size_t)m * p.ldtq + j] = s;
            }
        } else {
            float dq[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) dq[j] = p.Tq[(size_t)m * p.ldtq + j];
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < rc) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) s += dq[j] * Mv[j * 8 + i];
                    p.dTc[(size_t)m * p.lddtc + p.toff + i] = s;
                }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 8; ++i) g[j * 8 + i] = dq[j] * tc[i];
        }
    }
    if (BWD) {
#pragma unroll
        for (int q = 0; q < 32; ++q) red[t][q] = g[q];
        __syncthreads();
        if (t < 32) {
            float s = 0.f;
            for (int r = 0; r < 256; ++r) s += red[r][t];          // fixed order: deterministic
            gram_ws[((size_t)blockIdx.y * nblk + blockIdx.x) * 32 + t] = s;
        }
    }
}

// fold the partial Gram sums of a site (fixed order) and accumulate gUc[c, i] += scale * sum_j Dq[j, c] G[j, i],
// gDq[j, c] += scale * sum_i G[j, i] Uc[c, i]
__global__ __launch_bounds__(256) void rank_compose_bwd_kernel(RankSites a, const float* gram_ws, int nblk) {
    const clora_rank_site_t& p = a.s[blockIdx.y];
    __shared__ float G[32];
    const int t = threadIdx.x, rc = p.rc;
    if (t < 32) {
        float s = 0.f;
        const int nb = (p.rows + kRankMixRows - 1) / kRankMixRows;
        for (int b = 0; b < nb; ++b) s += gram_ws[((size_t)blockIdx.y * nblk + b) * 32 + t];
        G[t] = s * p.scale;
    }
    __syncthreads();
    for (int c = blockIdx.x * 256 + t; c < p.C; c += gridDim.x * 256) {
        float d[4], u[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[j] = p.Dq[(size_t)j * p.lddq + c];
#pragma unroll
        for (int i = 0; i < 8; ++i) u[i] = i < rc ? p.Uc[(size_t)c * p.lduc + i] : 0.f;
        if (p.gUc) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (i < rc) {
                    float s = 0.f;
#pragma unroll
                    for (int j = 0; j < 4; ++j) s += d[j] * G[j * 8 + i];
                    p.gUc[(size_t)c * p.lduc + i] += s;
                }
        }
        if (p.gDq) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += G[j * 8 + i] * u[i];
                p.gDq[(size_t)j * p.lddq + c] += s;
            }
        }
    }
}

int rank_check(const clora_rank_site_t* sites, int n, RankSites& a) {
    if (!sites || n <= 0 || n > kRankMaxSites) return CLORA_ERR_ARG;
    for (int i = 0; i < n; ++i) {
        const clora_rank_site_t& p = sites[i];
        if (!p.Dq || !p.Uc || !p.M || p.rc <= 0 || p.rc > 8 || p.C <= 0 || p.rows <= 0) return CLORA_ERR_ARG;
        a.s[i] = p;
    }
    return CLORA_OK;
}
}  // namespace

extern "C" int clora_rank_compose_f32(const clora_rank_site_t* sites, int n, void* stream) {
    RankSites a;
    if (rank_check(sites, n, a) != CLORA_OK) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(rank_compose_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, a);
    return clora_check_launch();
}

extern "C" size_t clora_rank_gram_ws_bytes(int rows, int n) {
    if (rows <= 0 || n <= 0) return 0;
    return (size_t)n * ((rows + kRankMixRows - 1) / kRankMixRows) * 32 * sizeof(float);
}

extern "C" int clora_rank_mix_f32(const clora_rank_site_t* sites, int n, int backward, float* gram_ws, size_t gram_ws_bytes, void* stream) {
    RankSites a;
    if (rank_check(sites, n, a) != CLORA_OK) return CLORA_ERR_ARG;
    int rows = 0;
    for (int i = 0; i < n; ++i) {
        if (!sites[i].Tc || !sites[i].Tq || (backward && !sites[i].dTc)) return CLORA_ERR_ARG;
        rows = sites[i].rows > rows ? sites[i].rows : rows;
    }
    const int nblk = (rows + kRankMixRows - 1) / kRankMixRows;
    if (backward) {
        if (!gram_ws || gram_ws_bytes < clora_rank_gram_ws_bytes(rows, n)) return CLORA_ERR_WORKSPACE;
        hipLaunchKernelGGL(rank_mix_kernel<true>, dim3(nblk, n), dim3(256), 0, (hipStream_t)stream, a, gram_ws, nblk);
    } else {
        hipLaunchKernelGGL(rank_mix_kernel<false>, dim3(nblk, n), dim3(256), 0, (hipStream_t)stream, a, (float*)nullptr, nblk);
    }
    return clora_check_launch();
}

extern "C" int clora_rank_compose_bwd_f32(const clora_rank_site_t* sites, int n, const float* gram_ws, void* stream) {
    RankSites a;
    if (rank_check(sites, n, a) != CLORA_OK || !gram_ws) return CLORA_ERR_ARG;
    int rows = 0, C = 0;
    for (int i = 0; i < n; ++i) { rows = sites[i].rows > rows ? sites[i].rows : rows; C = sites[i].C > C ? sites[i].C : C; }
    const int nblk = (rows + kRankMixRows - 1) / kRankMixRows;
    hipLaunchKernelGGL(rank_compose_bwd_kernel, dim3(clora_cdiv(C, 256), n), dim3(256), 0, (hipStream_t)stream, a, gram_ws, nblk);
    return clora_check_launch();
}

