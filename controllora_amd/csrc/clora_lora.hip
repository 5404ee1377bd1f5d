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

struct DownArgs {
    const half_t* X;
    const float* D;
    float* T;
    int ldx, ldd, ldt, toff, M, K, R, accumulate, x_rows;
};

// one thread = one row m; X is staged through LDS in [256 x 64] tiles so global reads stay coalesced
template <int RT>
__global__ __launch_bounds__(256) void lora_down_kernel(DownArgs p) {
    constexpr int BKC = 64, LDX = BKC + 8;
    __shared__ __attribute__((aligned(16))) half_t Xs[256 * LDX];
    __shared__ float Ds[RT * BKC];
    const int t = threadIdx.x;
    const int m0 = blockIdx.x * 256;
    float acc[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) acc[j] = 0.f;
    for (int k0 = 0; k0 < p.K; k0 += BKC) {
        __syncthreads();
        for (int c = t; c < 256 * 8; c += 256) {
            const int row = c >> 3, col = (c & 7) * 8;
            const int m = m0 + row;
            half8 v = zero8();
            if (m < p.M && k0 + col < p.K) {
                const int xr = p.x_rows > 0 ? m % p.x_rows : m;
                v = ld8(p.X + (size_t)xr * p.ldx + k0 + col);
            }
            st8(Xs + row * LDX + col, v);
        }
        for (int c = t; c < RT * BKC; c += 256) {
            const int j = c / BKC, kk = c - j * BKC;
            Ds[c] = (j < p.R && k0 + kk < p.K) ? p.D[(size_t)j * p.ldd + k0 + kk] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int c8 = 0; c8 < 8; ++c8) {
            const half8 v = ld8(Xs + t * LDX + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float xf = (float)v[e];
#pragma unroll
                for (int j = 0; j < RT; ++j) acc[j] += xf * Ds[j * BKC + c8 * 8 + e];
            }
        }
    }
    const int m = m0 + t;
    if (m < p.M) {
        float* out = p.T + (size_t)m * p.ldt + p.toff;
#pragma unroll
        for (int j = 0; j < RT; ++j)
            if (j < p.R) out[j] = p.accumulate ? out[j] + acc[j] : acc[j];
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

struct WgArgs {
    const half_t* A;
    const float* T;
    float* G;
    int lda, ldt, toff, gs_n, gs_j, M, N, R, a_rows, rows_per_block;
    float scale;
};

// block = 128 columns x rows_per_block rows; wave w takes rows == w (mod 4); lane owns 2 columns
template <int RT>
__global__ __launch_bounds__(256) void lora_wgrad_kernel(WgArgs p) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = blockIdx.x * 128 + 2 * l;
    const int m_beg = blockIdx.y * p.rows_per_block;
    const int m_end = (m_beg + p.rows_per_block < p.M) ? m_beg + p.rows_per_block : p.M;
    const bool nok = n < p.N;  // N is even (multiple of 8)
    float acc0[RT], acc1[RT];
#pragma unroll
    for (int j = 0; j < RT; ++j) { acc0[j] = 0.f; acc1[j] = 0.f; }
    for (int m = m_beg + w; m < m_end; m += 4) {
        const int ar = p.a_rows > 0 ? m % p.a_rows : m;
        half2v a = {0, 0};
        if (nok) a = *reinterpret_cast<const half2v*>(p.A + (size_t)ar * p.lda + n);
        const float a0 = (float)a[0], a1 = (float)a[1];
        const float* tr = p.T + (size_t)m * p.ldt + p.toff;
#pragma unroll
        for (int j = 0; j < RT; ++j) {
            const float tv = (j < p.R) ? tr[j] : 0.f;
            acc0[j] += a0 * tv;
            acc1[j] += a1 * tv;
        }
    }
    if (nok) {
#pragma unroll
        for (int j = 0; j < RT; ++j)
            if (j < p.R) {
                atomicAdd(p.G + (size_t)n * p.gs_n + (size_t)j * p.gs_j, p.scale * acc0[j]);
                atomicAdd(p.G + (size_t)(n + 1) * p.gs_n + (size_t)j * p.gs_j, p.scale * acc1[j]);
            }
    }
}

}  // namespace

extern "C" int clora_lora_down_f16(const clora_half* X, int ldx, const float* D, int ldd, float* T, int ldt, int toff,
                                   int M, int K, int R, int accumulate, int x_rows, void* stream) {
    if (!X || !D || !T || M <= 0 || K <= 0 || R <= 0 || (K & 7) || (ldx & 7)) return CLORA_ERR_ARG;
    hipStream_t s = (hipStream_t)stream;
    for (int r0 = 0; r0 < R; r0 += 16) {  // ranks > 16 (danbooru-sketch control_rank 256) take several passes over X
        DownArgs a;
        a.X = (const half_t*)X; a.D = D + (size_t)r0 * ldd; a.T = T;
        a.ldx = ldx; a.ldd = ldd; a.ldt = ldt; a.toff = toff + r0; a.M = M; a.K = K;
        a.R = (R - r0 < 16) ? R - r0 : 16; a.accumulate = accumulate; a.x_rows = x_rows;
        const dim3 grid(clora_cdiv(M, 256));
        if (a.R <= 4) hipLaunchKernelGGL((lora_down_kernel<4>), grid, dim3(256), 0, s, a);
        else if (a.R <= 8) hipLaunchKernelGGL((lora_down_kernel<8>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((lora_down_kernel<16>), grid, dim3(256), 0, s, a);
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
    for (int r0 = 0; r0 < R; r0 += 16) {
        WgArgs a;
        a.A = (const half_t*)A; a.T = T; a.G = G + (size_t)r0 * gs_j;
        a.lda = lda; a.ldt = ldt; a.toff = toff + r0; a.gs_n = gs_n; a.gs_j = gs_j; a.M = M; a.N = N;
        a.R = (R - r0 < 16) ? R - r0 : 16; a.a_rows = a_rows; a.scale = scale;
        a.rows_per_block = 256;
        const dim3 grid(clora_cdiv(N, 128), clora_cdiv(M, a.rows_per_block));
        if (a.R <= 4) hipLaunchKernelGGL((lora_wgrad_kernel<4>), grid, dim3(256), 0, s, a);
        else if (a.R <= 8) hipLaunchKernelGGL((lora_wgrad_kernel<8>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((lora_wgrad_kernel<16>), grid, dim3(256), 0, s, a);
    }
    return clora_check_launch();
}
