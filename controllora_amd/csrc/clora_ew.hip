// clora_ew.hip -- the small HBM-bound kernels on the path: elementwise ops, strided copies (skip
// concat / split), 2x2 sum-pool (backward of nearest upsample), column sums (bias grads), the fp32
// MSE loss with its gradient, and the fused flat-buffer optimizer (unscale + global-norm clip + AdamW +
// GradScaler-style skip/rescale, all device resident: no host sync in the step).
// All fp16 traffic is 16 bytes per lane (cdna_hip_programming.md Guideline 13).
#include "clora_common.h"
#include "../../include/clora.h"

namespace {

int ew_grid(size_t work) {
    size_t b = (work + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

#define EW_LOOP(i, n) for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (n); i += (size_t)gridDim.x * 256)

__global__ __launch_bounds__(256) void add_kernel(const half_t* a, const half_t* b, half_t* y, size_t n8) {
    EW_LOOP(i, n8) {
        const half8 x = ld8(a + i * 8), z = ld8(b + i * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)x[e] + (float)z[e]);
        st8(y + i * 8, o);
    }
}

__global__ __launch_bounds__(256) void silu_kernel(const half_t* x, half_t* y, size_t n8) {
    EW_LOOP(i, n8) {
        const half8 v = ld8(x + i * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)silu_f((float)v[e]);
        st8(y + i * 8, o);
    }
}

// CLIP's activation (transformers `quick_gelu`): x * sigmoid(1.702 x)
__global__ __launch_bounds__(256) void quick_gelu_kernel(const half_t* x, half_t* y, size_t n8) {
    EW_LOOP(i, n8) {
        const half8 v = ld8(x + i * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)v[e] * sigmoid_f(1.702f * (float)v[e]));
        st8(y + i * 8, o);
    }
}

__global__ __launch_bounds__(256) void silu_bwd_kernel(const half_t* x, const half_t* dy, half_t* dx, size_t n8) {
    EW_LOOP(i, n8) {
        const half8 v = ld8(x + i * 8), g = ld8(dy + i * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)g[e] * dsilu_f((float)v[e]));
        st8(dx + i * 8, o);
    }
}

__global__ __launch_bounds__(256) void geglu_fwd_kernel(const half_t* h, half_t* y, int M, int F) {
    const int FC = F / 8;
    const size_t total = (size_t)M * FC;
    EW_LOOP(i, total) {
        const size_t m = i / FC;
        const int j = (int)(i - m * FC) * 8;
        const half8 a = ld8(h + m * 2 * F + j), g = ld8(h + m * 2 * F + F + j);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)a[e] * gelu_f((float)g[e]));
        st8(y + m * F + j, o);
    }
}

__global__ __launch_bounds__(256) void geglu_bwd_kernel(const half_t* h, const half_t* dy, half_t* dh, int M, int F) {
    const int FC = F / 8;
    const size_t total = (size_t)M * FC;
    EW_LOOP(i, total) {
        const size_t m = i / FC;
        const int j = (int)(i - m * FC) * 8;
        const half8 a = ld8(h + m * 2 * F + j), g = ld8(h + m * 2 * F + F + j), d = ld8(dy + m * F + j);
        half8 da, dg;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float gf = (float)g[e], df = (float)d[e];
            float cdf, pdf;
            gelu_parts(gf, cdf, pdf);
            da[e] = (half_t)(df * gf * cdf);
            dg[e] = (half_t)(df * (float)a[e] * (cdf + gf * pdf));
        }
        st8(dh + m * 2 * F + j, da);
        st8(dh + m * 2 * F + F + j, dg);
    }
}

__global__ __launch_bounds__(256) void copy2d_kernel(const half_t* src, int lds, half_t* dst, int ldd, size_t M, int N) {
    const int NC = N / 8;
    const size_t total = M * NC;
    EW_LOOP(i, total) {
        const size_t m = i / NC;
        const int n = (int)(i - m * NC) * 8;
        st8(dst + m * ldd + n, ld8(src + m * lds + n));
    }
}

__global__ __launch_bounds__(256) void pool2x2_sum_kernel(const half_t* dy, half_t* dx, int B, int H, int W, int C) {
    const int CC = C / 8;
    const size_t total = (size_t)B * H * W * CC;
    EW_LOOP(i, total) {
        const int cc = (int)(i % CC);
        size_t r = i / CC;
        const int x = (int)(r % W); r /= W;
        const int y = (int)(r % H);
        const int b = (int)(r / H);
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int dyy = 0; dyy < 2; ++dyy)
#pragma unroll
            for (int dxx = 0; dxx < 2; ++dxx) {
                const half8 v = ld8(dy + (((size_t)b * 2 * H + 2 * y + dyy) * 2 * W + 2 * x + dxx) * C + cc * 8);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += (float)v[e];
            }
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (half_t)acc[e];
        st8(dx + (((size_t)b * H + y) * W + x) * C + cc * 8, o);
    }
}

__global__ __launch_bounds__(256) void colsum_kernel(const half_t* A, int lda, float* out, int M, int N, int rows_per_block) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = blockIdx.x * 128 + 2 * l;
    const int m_beg = blockIdx.y * rows_per_block;
    const int m_end = (m_beg + rows_per_block < M) ? m_beg + rows_per_block : M;
    if (n >= N) return;
    float a0 = 0.f, a1 = 0.f;
    for (int m = m_beg + w; m < m_end; m += 4) {
        const half2v a = *reinterpret_cast<const half2v*>(A + (size_t)m * lda + n);
        a0 += (float)a[0];
        a1 += (float)a[1];
    }
    atomicAdd(out + n, a0);
    atomicAdd(out + n + 1, a1);
}

__global__ __launch_bounds__(256) void mse_kernel(const half_t* pred, const half_t* target, float* loss_sum, half_t* dpred,
                                                  size_t n8, float grad_scale, const float* loss_scale) {
    __shared__ float red[4];
    const float gs = loss_scale ? grad_scale * loss_scale[0] : grad_scale;
    float acc = 0.f;
    EW_LOOP(i, n8) {
        const half8 a = ld8(pred + i * 8), b = ld8(target + i * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = (float)a[e] - (float)b[e];
            acc += d * d;
            o[e] = (half_t)(gs * d);
        }
        if (dpred) st8(dpred + i * 8, o);
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss_sum, red[0] + red[1] + red[2] + red[3]);
}

__global__ __launch_bounds__(256) void cast_f32_f16_kernel(const float* x, half_t* y, size_t n) {
    EW_LOOP(i, n) y[i] = (half_t)x[i];
}
__global__ __launch_bounds__(256) void cast_f16_f32_kernel(const half_t* x, float* y, size_t n) {
    EW_LOOP(i, n) y[i] = (float)x[i];
}

// ---- optimizer -------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void grad_sumsq_kernel(const float* g, size_t n, float* state) {
    __shared__ float red[4];
    __shared__ int bad[4];
    float acc = 0.f;
    int nf = 0;
    EW_LOOP(i, n) {
        const float v = g[i];
        acc += v * v;
        nf |= !(fabsf(v) <= 3.0e38f);   // inf or nan
    }
    acc = wave_sum(acc);
    float nff = wave_sum((float)nf);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = acc; bad[threadIdx.x >> 6] = nff > 0.f; }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(state + 0, red[0] + red[1] + red[2] + red[3]);
        if (bad[0] | bad[1] | bad[2] | bad[3]) atomicAdd(state + 1, 1.0f);
    }
}

__global__ void optim_prep_kernel(float* st, float max_norm, float beta1, float beta2, int dynamic, float growth,
                                  float backoff, int interval) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float sumsq = st[0];
    const bool badv = st[1] != 0.f || !(sumsq <= 3.0e38f);
    const float inv = 1.0f / (st[3] * (st[11] > 0.f ? st[11] : 1.0f));   // loss scale x data-parallel world size
    if (!badv) {
        const float norm = sqrtf(sumsq) * inv;
        float coef = 1.0f;
        if (max_norm > 0.f) { coef = max_norm / (norm + 1e-6f); coef = coef > 1.0f ? 1.0f : coef; }
        const float step = st[2] + 1.0f;
        st[2] = step;
        st[5] = inv * coef;
        st[6] = 0.f;
        st[7] = 1.0f - powf(beta1, step);
        st[8] = 1.0f - powf(beta2, step);
        st[9] = norm;
        if (dynamic) {
            float tr = st[4] + 1.0f;
            if (tr >= (float)interval) { st[3] *= growth; tr = 0.f; }
            st[4] = tr;
        }
    } else {
        st[5] = 0.f;
        st[6] = 1.f;
        st[9] = -1.f;
        if (dynamic) { st[3] *= backoff; st[4] = 0.f; }
    }
    st[0] = 0.f;
    st[1] = 0.f;
}

__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, size_t n, const float* st,
                                                    float lr, float beta1, float beta2, float eps, float wd) {
    if (st[6] != 0.f) return;  // GradScaler semantics: skip the step on inf/nan
    const float gm = st[5], bc1 = st[7], rbc2 = rsqrtf(st[8]);
    lr *= (st[10] > 0.f ? st[10] : 1.0f);   // device-resident LR-schedule multiplier (0 = unset): graphs stay valid
    const float step_size = lr / bc1;
    EW_LOOP(i, n) {
        const float gr = g[i] * gm;
        float pv = p[i] * (1.0f - lr * wd);
        const float mv = m[i] + (1.0f - beta1) * (gr - m[i]);
        const float vv = beta2 * v[i] + (1.0f - beta2) * gr * gr;
        pv -= step_size * mv / (sqrtf(vv) * rbc2 + eps);
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}

// sinusoidal timestep embedding (upstream get_timestep_embedding with flip_sin_to_cos, shift 0: SURVEY.md U1): out[b, j] = cos(t_b f_j),
// out[b, half + j] = sin(t_b f_j), fp32 math, fp16 result; t is int64 or fp32, one value per batch element or one for all
__global__ __launch_bounds__(256) void timestep_embedding_kernel(const void* t, int t_is_i64, int t_count, const float* freq, half_t* out,
                                                                 int batch, int half) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= batch * half) return;
    const int b = i / half, j = i - b * half;
    const int ti = t_count == 1 ? 0 : b;
    const float tv = t_is_i64 ? (float)((const long long*)t)[ti] : ((const float*)t)[ti];
    const float arg = tv * freq[j];
    out[(size_t)b * 2 * half + j] = (half_t)cosf(arg);
    out[(size_t)b * 2 * half + half + j] = (half_t)sinf(arg);
}

}  // namespace

#define H(x) ((const half_t*)(x))
#define HM(x) ((half_t*)(x))

extern "C" int clora_add_f16(const clora_half* a, const clora_half* b, clora_half* y, size_t n, void* stream) {
    if (!a || !b || !y || (n & 7)) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(add_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, H(a), H(b), HM(y), n / 8);
    return clora_check_launch();
}
extern "C" int clora_timestep_embedding_f16(const void* t, int t_is_i64, int t_count, const float* freq, clora_half* out, int batch,
                                            int half, void* stream) {
    if (!t || !freq || !out || batch <= 0 || half <= 0 || (t_count != 1 && t_count != batch)) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(clora_cdiv((long)batch * half, 256)), dim3(256), 0, (hipStream_t)stream, t, t_is_i64,
                       t_count, freq, HM(out), batch, half);
    return clora_check_launch();
}
extern "C" int clora_silu_f16(const clora_half* x, clora_half* y, size_t n, void* stream) {
    if (!x || !y || (n & 7)) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(silu_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, H(x), HM(y), n / 8);
    return clora_check_launch();
}
extern "C" int clora_quick_gelu_f16(const clora_half* x, clora_half* y, size_t n, void* stream) {
    if (!x || !y || (n & 7)) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(quick_gelu_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, H(x), HM(y), n / 8);
    return clora_check_launch();
}
extern "C" int clora_silu_bwd_f16(const clora_half* x, const clora_half* dy, clora_half* dx, size_t n, void* stream) {
    if (!x || !dy || !dx || (n & 7)) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(silu_bwd_kernel, dim3(ew_grid(n / 8)), dim3(256), 0, (hipStream_t)stream, H(x), H(dy), HM(dx), n / 8);
    return clora_check_launch();
}
extern "C" int clora_geglu_fwd_f16(const clora_half* h, clora_half* y, int M, int F, void* stream) {
    if (!h || !y || M <= 0 || F <= 0 || (F & 7)) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(geglu_fwd_kernel, dim3(ew_grid((size_t)M * (F / 8))), dim3(256), 0, (hipStream_t)stream, H(h), HM(y), M, F);
    return clora_check_launch();
}
extern "C" int clora_geglu_bwd_f16(const clora_half* h, const clora_half* dy, clora_half* dh, int M, int F, void* stream) {
    if (!h || !dy || !dh || M <= 0 || F <= 0 || (F & 7)) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(geglu_bwd_kernel, dim3(ew_grid((size_t)M * (F / 8))), dim3(256), 0, (hipStream_t)stream, H(h), H(dy), HM(dh), M, F);
    return clora_check_launch();
}
extern "C" int clora_copy2d_f16(const clora_half* src, int lds, clora_half* dst, int ldd, size_t M, int N, void* stream) {
    if (!src || !dst || N <= 0 || (N & 7) || (lds & 7) || (ldd & 7)) return CLORA_ERR_ARG;
    if (M == 0) return CLORA_OK;
    hipLaunchKernelGGL(copy2d_kernel, dim3(ew_grid(M * (N / 8))), dim3(256), 0, (hipStream_t)stream, H(src), lds, HM(dst), ldd, M, N);
    return clora_check_launch();
}
extern "C" int clora_pool2x2_sum_f16(const clora_half* dy, clora_half* dx, int B, int Hh, int W, int C, void* stream) {
    if (!dy || !dx || B <= 0 || Hh <= 0 || W <= 0 || C <= 0 || (C & 7)) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(pool2x2_sum_kernel, dim3(ew_grid((size_t)B * Hh * W * (C / 8))), dim3(256), 0, (hipStream_t)stream,
                       H(dy), HM(dx), B, Hh, W, C);
    return clora_check_launch();
}
extern "C" int clora_colsum_f16(const clora_half* A, int lda, float* out, int M, int N, void* stream) {
    if (!A || !out || M <= 0 || N <= 0 || (N & 1) || (lda & 1)) return CLORA_ERR_ARG;
    const int rpb = 1024;
    hipLaunchKernelGGL(colsum_kernel, dim3(clora_cdiv(N, 128), clora_cdiv(M, rpb)), dim3(256), 0, (hipStream_t)stream, H(A),
                       lda, out, M, N, rpb);
    return clora_check_launch();
}
extern "C" int clora_mse_f16(const clora_half* pred, const clora_half* target, float* loss_sum, clora_half* dpred, size_t n,
                             float grad_scale, const float* loss_scale, void* stream) {
    if (!pred || !target || !loss_sum || (n & 7)) return CLORA_ERR_ARG;
    int blocks = ew_grid(n / 8);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(mse_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, H(pred), H(target), loss_sum, HM(dpred),
                       n / 8, grad_scale, loss_scale);
    return clora_check_launch();
}
extern "C" int clora_cast_f32_to_f16(const float* x, clora_half* y, size_t n, void* stream) {
    if (!x || !y) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, x, HM(y), n);
    return clora_check_launch();
}
extern "C" int clora_cast_f16_to_f32(const clora_half* x, float* y, size_t n, void* stream) {
    if (!x || !y) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(cast_f16_f32_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, H(x), y, n);
    return clora_check_launch();
}
extern "C" int clora_grad_sumsq_f32(const float* g, size_t n, float* state, void* stream) {
    if (!g || !state) return CLORA_ERR_ARG;
    int blocks = ew_grid(n);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(grad_sumsq_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, state);
    return clora_check_launch();
}
extern "C" int clora_optim_prep_f32(float* state, float max_norm, float beta1, float beta2, int dynamic_scale,
                                    float growth_factor, float backoff_factor, int growth_interval, void* stream) {
    if (!state) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(optim_prep_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, max_norm, beta1, beta2,
                       dynamic_scale, growth_factor, backoff_factor, growth_interval);
    return clora_check_launch();
}
extern "C" int clora_adamw_flat_f32(float* p, const float* g, float* m, float* v, size_t n, const float* state, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, void* stream) {
    if (!p || !g || !m || !v || !state) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, state, lr, beta1,
                       beta2, eps, weight_decay);
    return clora_check_launch();
}
// Box calibration (bench.py): what clock does this chip hold under a dense MFMA stream?  One 256-thread block per CU-slot, every wave issues
// `iters` x 8 independent v_mfma_f32_16x16x32_f16 on pseudo-random operands (zero operands clock higher: cdna_hip_programming.md
// rule 25) and reports shader cycles (s_memtime) and 100 MHz wall ticks (s_memrealtime) around the loop: effective clock =
// cycles / ticks x 100 MHz, sustained dense rate = blocks x 4 waves x iters x 8 x 16384 flop / wall time.  out[3b .. 3b+2] =
// {cycles, ticks, checksum bits} of block b's wave 0.
__global__ __launch_bounds__(256) void clock_probe_kernel(unsigned long long* out, int iters) {
    const int l = threadIdx.x & 63;
    half8 a, b;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        a[e] = (half_t)(((l * 37 + e * 11 + blockIdx.x) % 61) * (1.0f / 32.0f) - 0.95f);
        b[e] = (half_t)(((l * 53 + e * 29 + threadIdx.x) % 59) * (1.0f / 32.0f) - 0.9f);
    }
    floatx4 acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = zero4f();
    __syncthreads();
    const unsigned long long c0 = CLORA_CYCLES(), w0 = CLORA_WALL_TICKS();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q) CLORA_MFMA_INPLACE(acc[q], a, b);
    }
    CLORA_MFMA_DRAIN();                                        // the last MFMAs' results are read below (hipcc pads nothing after an asm statement)
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
    const unsigned long long c1 = CLORA_CYCLES(), w1 = CLORA_WALL_TICKS();
    if (threadIdx.x == 0) {
        out[3 * (size_t)blockIdx.x] = c1 - c0;
        out[3 * (size_t)blockIdx.x + 1] = w1 - w0;
        out[3 * (size_t)blockIdx.x + 2] = (unsigned long long)__float_as_uint(s);
    }
}
extern "C" int clora_clock_probe(unsigned long long* out, int blocks, int iters, void* stream) {
    if (!out || blocks <= 0 || iters <= 0) return CLORA_ERR_ARG;
    hipLaunchKernelGGL(clock_probe_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, out, iters);
    return clora_check_launch();
}
extern "C" int clora_abi_version(void) { return CLORA_ABI_VERSION; }   // include/clora.h (3: clora_epilogue_t.defer; 4: clora_groupnorm_*_team, round 6)
extern "C" const char* clora_build_info(void) { return "libclora gfx950 (v_mfma_f32_16x16x32_f16), ABI 2"; }

// ------------------------------------------------------------------------------------------------
// Multi-job forms of clora_conv_weight_pack_f32 / clora_conv_wgrad_unpack_f32 (clora_gemm.hip): every trainable convolution of the
// hint encoder in one launch (grid.y = job).  Same per-element arithmetic as the single-job kernels.
namespace {
struct ConvPackJobs { clora_conv_pack_job_t j[CLORA_CONV_MAX_JOBS]; };
struct ConvUnpackJobs { clora_conv_unpack_job_t j[CLORA_CONV_MAX_JOBS]; };

__global__ __launch_bounds__(256) void conv_weight_pack_multi_kernel(ConvPackJobs jobs) {
    const clora_conv_pack_job_t& p = jobs.j[blockIdx.y];
    const int taps = p.ksize * p.ksize, Co = p.Co, Ci = p.Ci, Cip = p.Cip, Cop = p.Cop;
    const float* w = p.w;
    half_t* fwd = (half_t*)p.fwd;
    half_t* dgrad = (half_t*)p.dgrad;
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

__global__ __launch_bounds__(256) void conv_wgrad_unpack_multi_kernel(ConvUnpackJobs jobs) {
    const clora_conv_unpack_job_t& p = jobs.j[blockIdx.y];
    const int taps = p.ksize * p.ksize, Co = p.Co, Ci = p.Ci, Cip = p.Cip;
    const int nw = Co * taps * Cip;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nw + (p.grad_b ? Co : 0); i += gridDim.x * 256) {
        if (i < nw) {
            const int ci = i % Cip, tap = (i / Cip) % taps, co = i / (Cip * taps);
            const float v = p.stage[i];
            p.stage[i] = 0.f;
            if (ci < Ci) p.grad_w[((size_t)co * Ci + ci) * taps + tap] += v;
        } else {
            const int co = i - nw;
            p.grad_b[co] += p.stage_b[co];
            p.stage_b[co] = 0.f;
        }
    }
}
}  // namespace

extern "C" int clora_conv_weight_pack_multi_f32(const clora_conv_pack_job_t* jobs, int njobs, void* stream) {
    if (!jobs || njobs <= 0 || njobs > CLORA_CONV_MAX_JOBS) return CLORA_ERR_ARG;
    ConvPackJobs cj;
    long most = 0;
    for (int i = 0; i < njobs; ++i) {
        const clora_conv_pack_job_t& j = jobs[i];
        if (!j.w || !j.fwd || j.Co <= 0 || j.Ci <= 0 || (j.ksize != 1 && j.ksize != 3) || j.Cip < j.Ci || (j.Cip & 7) ||
            (j.dgrad && (j.Cop < j.Co || (j.Cop & 7))))
            return CLORA_ERR_ARG;
        const int taps = j.ksize * j.ksize;
        const long total = (long)j.Co * taps * j.Cip + (j.dgrad ? (long)j.Cip * taps * j.Cop : 0);
        if (total > most) most = total;
        cj.j[i] = j;
    }
    int blocks = clora_cdiv(most, 256 * 4);                     // ~4 elements per thread of the largest job
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(conv_weight_pack_multi_kernel, dim3(blocks, njobs), dim3(256), 0, (hipStream_t)stream, cj);
    return clora_check_launch();
}

extern "C" int clora_conv_wgrad_unpack_multi_f32(const clora_conv_unpack_job_t* jobs, int njobs, void* stream) {
    if (!jobs || njobs <= 0 || njobs > CLORA_CONV_MAX_JOBS) return CLORA_ERR_ARG;
    ConvUnpackJobs cj;
    long most = 0;
    for (int i = 0; i < njobs; ++i) {
        const clora_conv_unpack_job_t& j = jobs[i];
        if (!j.stage || !j.grad_w || j.Co <= 0 || j.Ci <= 0 || j.Cip < j.Ci || (j.ksize != 1 && j.ksize != 3) ||
            ((j.grad_b == nullptr) != (j.stage_b == nullptr)))
            return CLORA_ERR_ARG;
        const long total = (long)j.Co * j.ksize * j.ksize * j.Cip + (j.grad_b ? j.Co : 0);
        if (total > most) most = total;
        cj.j[i] = j;
    }
    int blocks = clora_cdiv(most, 256 * 4);
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(conv_wgrad_unpack_multi_kernel, dim3(blocks, njobs), dim3(256), 0, (hipStream_t)stream, cj);
    return clora_check_launch();
}
