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
//   wgrad: dU = s * dY^T . T,  dD = dT^T . X    skinny reductions over M with fp32 atomics (this file)
#include "clora_common.h"
#include "../../include/clora.h"

namespace {

// one thread = one row m; X is staged through LDS in [256 x 64] tiles so global reads stay coalesced; the
// adapter matrix D is wave-uniform and read with SCALAR loads straight into SGPR operands of v_pk_fma_f32
// (the __restrict__ kernel arguments are what lets hipcc prove that) -- no LDS traffic for D at all.
template <int RT>
__global__ __launch_bounds__(256) void lora_down_kernel(const half_t* __restrict__ X, const float* __restrict__ D,
                                                        float* __restrict__ T, int ldx, int ldd, int ldt, int toff, int M,
                                                        int K, int R, int accumulate, int x_rows) {
    constexpr int BKC = 64, LDX = BKC + 8;
    __shared__ __attribute__((aligned(16))) half_t Xs[256 * LDX];
    const int t = threadIdx.x;
    const int m0 = blockIdx.x * 256;
    float acc[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) acc[j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += BKC) {
        __syncthreads();
        for (int c = t; c < 256 * 8; c += 256) {
            const int row = c >> 3, col = (c & 7) * 8;
            const int m = m0 + row;
            half8 v = zero8();
            if (m < M && k0 + col < K) {
                const int xr = x_rows > 0 ? m % x_rows : m;
                v = ld8(X + (size_t)xr * ldx + k0 + col);
            }
            st8(Xs + row * LDX + col, v);
        }
        __syncthreads();
        const int kc_end = (K - k0 < BKC) ? (K - k0) / 8 : 8;   // K % 8 == 0
        for (int c8 = 0; c8 < kc_end; ++c8) {
            const half8 v = ld8(Xs + t * LDX + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xf = (float)v[e];
#pragma unroll
                for (int j = 0; j < RT; ++j)
                    if (j < R) acc[j] += xf * D[(size_t)j * ldd + k0 + c8 * 8 + e];
            }
        }
    }
    const int m = m0 + t;
    if (m < M) {
        float* out = T + (size_t)m * ldt + toff;
#pragma unroll
        for (int j = 0; j < RT; ++j)
            if (j < R) out[j] = accumulate ? out[j] + acc[j] : acc[j];
    }
}

struct UpArgs {
    const half_t* base;
    const float* T;
    const float* U;
    half_t* Y;
    int ldb, ldt, toff, ldu, ldy, M, N, R;
    float scale;
};

__global__ __launch_bounds__(256) void lora_up_kernel(UpArgs p) {
    const int NC = p.N / 8;
    const size_t total = (size_t)p.M * NC;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t m = i / NC;
        const int n = (int)(i - m * NC) * 8;
        const float* tr = p.T + m * p.ldt + p.toff;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        for (int j = 0; j < p.R; ++j) {
            const float tv = tr[j];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += tv * p.U[(size_t)(n + e) * p.ldu + j];
        }
        half8 o;
        if (p.base) {
            const half8 bv = ld8(p.base + m * p.ldb + n);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const half_t c16 = (half_t)(p.scale * (float)(half_t)acc[e]);  // fp16(scale * fp16(up(down)))
                o[e] = (half_t)((float)bv[e] + (float)c16);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (half_t)(p.scale * (float)(half_t)acc[e]);
        }
        st8(p.Y + m * p.ldy + n, o);
    }
}

// G[n*gs_n + j*gs_j] += scale * sum_m A[m, n] * T[m, toff + j]
// block = 128 columns x rows_per_block rows; wave w takes rows == w (mod 4), 8 rows in flight per wave (memory
// level parallelism: these reductions are pure HBM streams); lane owns 2 columns; the T row is wave-uniform
// (scalar loads); the 4 waves are combined in LDS so each block issues one atomic per output element.
template <int RT>
__global__ __launch_bounds__(256) void lora_wgrad_kernel(const half_t* __restrict__ A, const float* __restrict__ T,
                                                         float* __restrict__ G, int lda, int ldt, int toff, int gs_n,
                                                         int gs_j, int M, int N, int R, int a_rows, int rows_per_block,
                                                         float scale) {
    __shared__ float red[3 * 64 * 2 * RT];
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l = threadIdx.x & 63;
    const int n = blockIdx.x * 128 + 2 * l;
    const int m_beg = blockIdx.y * rows_per_block;
    const int m_end = (m_beg + rows_per_block < M) ? m_beg + rows_per_block : M;
    const bool nok = n < N;  // N is even
    float acc0[RT], acc1[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
    constexpr int UN = 8;
    for (int mb = m_beg + w; mb < m_end; mb += 4 * UN) {
        half2v a[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int m = mb + 4 * u;
            half2v z = {0, 0};
            a[u] = z;
            if (nok && m < m_end) {
                const int ar = a_rows > 0 ? m % a_rows : m;
                a[u] = *reinterpret_cast<const half2v*>(A + (size_t)ar * lda + n);
            }
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int m = mb + 4 * u;
            if (m < m_end) {
                const float a0 = (float)a[u][0], a1 = (float)a[u][1];
                const float* tr = T + (size_t)m * ldt + toff;
#pragma unroll
                for (int j = 0; j < RT; ++j)
                    if (j < R) { const float tv = tr[j]; acc0[j] += a0 * tv; acc1[j] += a1 * tv; }
            }
        }
    }
    if (w > 0) {
#pragma unroll
        for (int j = 0; j < RT; ++j) {
            red[(((w - 1) * 64 + l) * 2 + 0) * RT + j] = acc0[j];
            red[(((w - 1) * 64 + l) * 2 + 1) * RT + j] = acc1[j];
        }
    }
    __syncthreads();
    if (w == 0 && nok) {
#pragma unroll
        for (int j = 0; j < RT; ++j)
            if (j < R) {
                float s0 = acc0[j], s1 = acc1[j];
#pragma unroll
                for (int ww = 0; ww < 3; ++ww) {
                    s0 += red[((ww * 64 + l) * 2 + 0) * RT + j];
                    s1 += red[((ww * 64 + l) * 2 + 1) * RT + j];
                }
                atomicAdd(G + (size_t)n * gs_n + (size_t)j * gs_j, scale * s0);
                atomicAdd(G + (size_t)(n + 1) * gs_n + (size_t)j * gs_j, scale * s1);
            }
    }
}

}  // namespace

extern "C" int clora_lora_down_f16(const clora_half* X, int ldx, const float* D, int ldd, float* T, int ldt, int toff,
                                   int M, int K, int R, int accumulate, int x_rows, void* stream) {
    if (!X || !D || !T || M <= 0 || K <= 0 || R <= 0 || (K & 7) || (ldx & 7)) return CLORA_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const half_t* Xh = (const half_t*)X;
    for (int r0 = 0; r0 < R; r0 += 16) {  // ranks > 16 (danbooru-sketch control_rank 256) take several passes over X
        const float* Dp = D + (size_t)r0 * ldd;
        const int Rp = (R - r0 < 16) ? R - r0 : 16, to = toff + r0;
        const dim3 grid(clora_cdiv(M, 256));
        if (Rp <= 4) hipLaunchKernelGGL((lora_down_kernel<4>), grid, dim3(256), 0, s, Xh, Dp, T, ldx, ldd, ldt, to, M, K, Rp, accumulate, x_rows);
        else if (Rp <= 8) hipLaunchKernelGGL((lora_down_kernel<8>), grid, dim3(256), 0, s, Xh, Dp, T, ldx, ldd, ldt, to, M, K, Rp, accumulate, x_rows);
        else hipLaunchKernelGGL((lora_down_kernel<16>), grid, dim3(256), 0, s, Xh, Dp, T, ldx, ldd, ldt, to, M, K, Rp, accumulate, x_rows);
    }
    return clora_check_launch();
}

extern "C" int clora_lora_up_f16(const clora_half* base, int ldb, const float* T, int ldt, int toff, const float* U,
                                 int ldu, clora_half* Y, int ldy, int M, int N, int R, float scale, void* stream) {
    if (!T || !U || !Y || M <= 0 || N <= 0 || R <= 0 || (N & 7) || (ldy & 7) || (base && (ldb & 7))) return CLORA_ERR_ARG;
    UpArgs a;
    a.base = (const half_t*)base; a.T = T; a.U = U; a.Y = (half_t*)Y;
    a.ldb = ldb; a.ldt = ldt; a.toff = toff; a.ldu = ldu; a.ldy = ldy; a.M = M; a.N = N; a.R = R; a.scale = scale;
    size_t blocks = ((size_t)M * (N / 8) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(lora_up_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    return clora_check_launch();
}

extern "C" int clora_lora_wgrad_f16(const clora_half* A, int lda, const float* T, int ldt, int toff, float* G, int gs_n,
                                    int gs_j, int M, int N, int R, float scale, int a_rows, void* stream) {
    if (!A || !T || !G || M <= 0 || N <= 0 || R <= 0 || (N & 7) || (lda & 1)) return CLORA_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    const half_t* Ah = (const half_t*)A;
    int rpb = 128;
    while (rpb < 1024 && (long)clora_cdiv(N, 128) * clora_cdiv(M, rpb) > 1024) rpb *= 2;
    for (int r0 = 0; r0 < R; r0 += 16) {
        float* Gp = G + (size_t)r0 * gs_j;
        const int Rp = (R - r0 < 16) ? R - r0 : 16, to = toff + r0;
        const dim3 grid(clora_cdiv(N, 128), clora_cdiv(M, rpb));
        if (Rp <= 4) hipLaunchKernelGGL((lora_wgrad_kernel<4>), grid, dim3(256), 0, s, Ah, T, Gp, lda, ldt, to, gs_n, gs_j, M, N, Rp, a_rows, rpb, scale);
        else if (Rp <= 8) hipLaunchKernelGGL((lora_wgrad_kernel<8>), grid, dim3(256), 0, s, Ah, T, Gp, lda, ldt, to, gs_n, gs_j, M, N, Rp, a_rows, rpb, scale);
        else hipLaunchKernelGGL((lora_wgrad_kernel<16>), grid, dim3(256), 0, s, Ah, T, Gp, lda, ldt, to, gs_n, gs_j, M, N, Rp, a_rows, rpb, scale);
    }
    return clora_check_launch();
}
