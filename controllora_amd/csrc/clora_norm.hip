// clora_norm.hip -- GroupNorm(+SiLU) and LayerNorm, forward and backward, NHWC fp16 / fp32 statistics.
//
// HBM-bound kernels: every pass reads/writes whole 16-byte channel chunks of
// full rows, so global traffic is coalesced regardless of how channels split into groups
// (C/32 = 10 channels per group at C=320 is not a multiple of the 8-channel load width).
//   GroupNorm forward : partial (sum, sumsq per (b, row-chunk, group)) -> finalize (mean, rstd) -> apply
//   GroupNorm backward: partial (s1 = sum dy*gamma, s2 = sum dy*gamma*xhat) -> finalize -> apply
//   LayerNorm         : one wave per row, the row lives in registers (two-pass variance), no workspace.
#include "clora_common.h"
#include "../../include/clora.h"

namespace {

constexpr int kMaxCols = 4;  // column chunks (of 8 channels) owned by one thread: C <= 8192

struct GnArgs {
    const half_t* x;
    const half_t* dy;
    half_t* y;      // fwd: output, bwd: dx
    const float* gamma;
    const float* beta;
    float* stats;   // [B, G, 2]  (mean, rstd)
    float* partial; // [B, nchunk, G, 2]
    float* gsum;    // bwd: [B, G, 2] (s1, s2)
    float* dgamma;
    float* dbeta;
    int B, HW, C, G, nchunk, rows_per_chunk;
    float eps;
    int fuse_silu;
};

// thread -> (row lane, first column chunk, column stride)
__device__ __forceinline__ void gn_thread_map(int t, int CH, int& rl, int& nrl, int& c0, int& cstep, bool& active) {
    if (CH >= 256) { rl = 0; nrl = 1; c0 = t; cstep = 256; active = true; }
    else { nrl = 256 / CH; rl = t / CH; c0 = t - rl * CH; cstep = CH; active = rl < nrl; }
}

// ---- forward, pass 1
__global__ __launch_bounds__(256) void gn_fwd_partial_kernel(GnArgs p) {
    __shared__ float gs[64 * 2];
    const int t = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    const int CH = p.C / 8, cpg = p.C / p.G;
    for (int i = t; i < p.G * 2; i += 256) gs[i] = 0.f;
    __syncthreads();
    int rl, nrl, c0, cstep; bool active;
    gn_thread_map(t, CH, rl, nrl, c0, cstep, active);
    const int r_beg = chunk * p.rows_per_chunk;
    const int r_end = (r_beg + p.rows_per_chunk < p.HW) ? r_beg + p.rows_per_chunk : p.HW;
    float s[kMaxCols][8], q[kMaxCols][8];
#pragma unroll
    for (int j = 0; j < kMaxCols; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[j][e] = 0.f; q[j][e] = 0.f; }
    if (active) {
        for (int r = r_beg + rl; r < r_end; r += nrl) {
            const half_t* row = p.x + ((size_t)b * p.HW + r) * p.C;
#pragma unroll
            for (int j = 0; j < kMaxCols; ++j) {
                const int cc = c0 + j * cstep;
                if (cc < CH && (j == 0 || cstep == 256)) {
                    const half8 v = ld8(row + cc * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float f = (float)v[e]; s[j][e] += f; q[j][e] += f * f; }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kMaxCols; ++j) {
            const int cc = c0 + j * cstep;
            if (cc < CH && (j == 0 || cstep == 256)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int grp = (cc * 8 + e) / cpg;
                    atomicAdd(&gs[grp * 2], s[j][e]);
                    atomicAdd(&gs[grp * 2 + 1], q[j][e]);
                }
            }
        }
    }
    __syncthreads();
    for (int i = t; i < p.G * 2; i += 256) p.partial[((size_t)b * p.nchunk + chunk) * p.G * 2 + i] = gs[i];
}

// ---- forward, pass 2: mean / rstd
__global__ __launch_bounds__(256) void gn_fwd_finalize_kernel(GnArgs p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.B * p.G) return;
    const int b = i / p.G, g = i - b * p.G;
    float s = 0.f, q = 0.f;
    for (int c = 0; c < p.nchunk; ++c) {
        const float* pp = p.partial + (((size_t)b * p.nchunk + c) * p.G + g) * 2;
        s += pp[0]; q += pp[1];
    }
    const float n = (float)p.HW * (float)(p.C / p.G);
    const float mean = s / n;
    float var = q / n - mean * mean;
    var = var < 0.f ? 0.f : var;
    p.stats[i * 2] = mean;
    p.stats[i * 2 + 1] = rsqrtf(var + p.eps);
}

// ---- forward, pass 3
__global__ __launch_bounds__(256) void gn_fwd_apply_kernel(GnArgs p) {
    const int CH = p.C / 8, cpg = p.C / p.G;
    const size_t total = (size_t)p.B * p.HW * CH;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / CH;
        const int cc = (int)(i - row * CH);
        const int b = (int)(row / p.HW);
        const half8 v = ld8(p.x + row * p.C + cc * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = cc * 8 + e;
            const float* st = p.stats + ((size_t)b * p.G + ch / cpg) * 2;
            float yv = ((float)v[e] - st[0]) * st[1] * p.gamma[ch] + p.beta[ch];
            if (p.fuse_silu) yv = silu_f(yv);
            o[e] = (half_t)yv;
        }
        st8(p.y + row * p.C + cc * 8, o);
    }
}

// ---- backward, pass 1: s1 = sum dyp*gamma, s2 = sum dyp*gamma*xhat per (b, group); dgamma/dbeta
__global__ __launch_bounds__(256) void gn_bwd_partial_kernel(GnArgs p) {
    __shared__ float gs[64 * 2];
    const int t = threadIdx.x, b = blockIdx.y, chunk = blockIdx.x;
    const int CH = p.C / 8, cpg = p.C / p.G;
    for (int i = t; i < p.G * 2; i += 256) gs[i] = 0.f;
    __syncthreads();
    int rl, nrl, c0, cstep; bool active;
    gn_thread_map(t, CH, rl, nrl, c0, cstep, active);
    const int r_beg = chunk * p.rows_per_chunk;
    const int r_end = (r_beg + p.rows_per_chunk < p.HW) ? r_beg + p.rows_per_chunk : p.HW;
    float a1[kMaxCols][8], a2[kMaxCols][8];  // per channel: sum dyp, sum dyp*xhat
#pragma unroll
    for (int j = 0; j < kMaxCols; ++j)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a1[j][e] = 0.f; a2[j][e] = 0.f; }
    if (active) {
        for (int r = r_beg + rl; r < r_end; r += nrl) {
            const size_t off = ((size_t)b * p.HW + r) * p.C;
#pragma unroll
            for (int j = 0; j < kMaxCols; ++j) {
                const int cc = c0 + j * cstep;
                if (cc < CH && (j == 0 || cstep == 256)) {
                    const half8 xv = ld8(p.x + off + cc * 8), gv = ld8(p.dy + off + cc * 8);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int ch = cc * 8 + e;
                        const float* st = p.stats + ((size_t)b * p.G + ch / cpg) * 2;
                        const float xh = ((float)xv[e] - st[0]) * st[1];
                        float d = (float)gv[e];
                        if (p.fuse_silu) d *= dsilu_f(xh * p.gamma[ch] + p.beta[ch]);
                        a1[j][e] += d;
                        a2[j][e] += d * xh;
                    }
                }
            }
        }
#pragma unroll
        for (int j = 0; j < kMaxCols; ++j) {
            const int cc = c0 + j * cstep;
            if (cc < CH && (j == 0 || cstep == 256)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ch = cc * 8 + e, grp = ch / cpg;
                    const float gm = p.gamma[ch];
                    atomicAdd(&gs[grp * 2], a1[j][e] * gm);
                    atomicAdd(&gs[grp * 2 + 1], a2[j][e] * gm);
                    if (p.dgamma) {
                        atomicAdd(p.dgamma + ch, a2[j][e]);
                        atomicAdd(p.dbeta + ch, a1[j][e]);
                    }
                }
            }
        }
    }
    __syncthreads();
    for (int i = t; i < p.G * 2; i += 256) p.partial[((size_t)b * p.nchunk + chunk) * p.G * 2 + i] = gs[i];
}

__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(GnArgs p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= p.B * p.G) return;
    const int b = i / p.G, g = i - b * p.G;
    float s = 0.f, q = 0.f;
    for (int c = 0; c < p.nchunk; ++c) {
        const float* pp = p.partial + (((size_t)b * p.nchunk + c) * p.G + g) * 2;
        s += pp[0]; q += pp[1];
    }
    const float n = (float)p.HW * (float)(p.C / p.G);
    p.gsum[i * 2] = s / n;
    p.gsum[i * 2 + 1] = q / n;
}

__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GnArgs p) {
    const int CH = p.C / 8, cpg = p.C / p.G;
    const size_t total = (size_t)p.B * p.HW * CH;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t row = i / CH;
        const int cc = (int)(i - row * CH);
        const int b = (int)(row / p.HW);
        const half8 xv = ld8(p.x + row * p.C + cc * 8), gv = ld8(p.dy + row * p.C + cc * 8);
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ch = cc * 8 + e, grp = ch / cpg;
            const float* st = p.stats + ((size_t)b * p.G + grp) * 2;
            const float* gsm = p.gsum + ((size_t)b * p.G + grp) * 2;
            const float xh = ((float)xv[e] - st[0]) * st[1];
            float d = (float)gv[e];
            if (p.fuse_silu) d *= dsilu_f(xh * p.gamma[ch] + p.beta[ch]);
            o[e] = (half_t)(st[1] * (d * p.gamma[ch] - gsm[0] - xh * gsm[1]));
        }
        st8(p.y + row * p.C + cc * 8, o);
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm
struct LnArgs {
    const half_t* x;
    const half_t* dy;
    half_t* y;
    const float* gamma;
    const float* beta;
    int M, C;
    float eps;
};

template <bool BWD>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs p) {
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + w;
    const bool rok = row < p.M;
    const int CH = p.C / 8;
    const size_t off = (size_t)(rok ? row : 0) * p.C;
    half8 xv[kMaxCols], gv[kMaxCols];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kMaxCols; ++j) {
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
    for (int j = 0; j < kMaxCols; ++j)
        if (l + 64 * j < CH) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = (float)xv[j][e] - mean; q += d * d; }
        }
    const float rstd = rsqrtf(wave_sum(q) / (float)p.C + p.eps);
    if (!BWD) {
#pragma unroll
        for (int j = 0; j < kMaxCols; ++j) {
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
        for (int j = 0; j < kMaxCols; ++j) {
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
        for (int j = 0; j < kMaxCols; ++j) {
            const int cc = l + 64 * j;
            if (cc < CH && rok) {
                half8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = ((float)xv[j][e] - mean) * rstd;
                    o[e] = (half_t)(rstd * ((float)gv[j][e] * p.gamma[cc * 8 + e] - s1 - xh * s2));
                }
                st8(p.y + off + cc * 8, o);
            }
        }
    }
}

int gn_plan(GnArgs& a, void* ws, size_t ws_bytes, bool bwd) {
    if (a.B <= 0 || a.HW <= 0 || a.C <= 0 || a.G <= 0 || a.G > 64 || (a.C % a.G) || (a.C & 7) || a.C / 8 > 256 * kMaxCols)
        return CLORA_ERR_ARG;
    int nchunk = 1024 / a.B;
    if (nchunk < 1) nchunk = 1;
    int rpc = clora_cdiv(a.HW, nchunk);
    if (rpc < 8) rpc = 8;
    nchunk = clora_cdiv(a.HW, rpc);
    a.nchunk = nchunk;
    a.rows_per_chunk = rpc;
    const size_t need = ((size_t)a.B * nchunk * a.G * 2 + (bwd ? (size_t)a.B * a.G * 2 : 0)) * sizeof(float);
    if (!ws || ws_bytes < need) return CLORA_ERR_WORKSPACE;
    a.partial = (float*)ws;
    a.gsum = a.partial + (size_t)a.B * nchunk * a.G * 2;
    return CLORA_OK;
}

int ew_blocks(size_t n) {
    size_t b = (n + 255) / 256;
    return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int clora_groupnorm_fwd_f16(const clora_half* x, clora_half* y, const float* gamma, const float* beta,
                                       float* stats, int B, int HW, int C, int G, float eps, int fuse_silu,
                                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !y || !gamma || !beta || !stats) return CLORA_ERR_ARG;
    GnArgs a = GnArgs();
    a.x = (const half_t*)x; a.y = (half_t*)y; a.gamma = gamma; a.beta = beta; a.stats = stats;
    a.B = B; a.HW = HW; a.C = C; a.G = G; a.eps = eps; a.fuse_silu = fuse_silu;
    int rc = gn_plan(a, workspace, workspace_bytes, false);
    if (rc != CLORA_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_fwd_partial_kernel, dim3(a.nchunk, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(gn_fwd_finalize_kernel, dim3(clora_cdiv(B * G, 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(gn_fwd_apply_kernel, dim3(ew_blocks((size_t)B * HW * (C / 8))), dim3(256), 0, s, a);
    return clora_check_launch();
}

extern "C" int clora_groupnorm_bwd_f16(const clora_half* x, const clora_half* dy, clora_half* dx, const float* gamma,
                                       const float* beta, const float* stats, float* dgamma, float* dbeta, int B,
                                       int HW, int C, int G, int fuse_silu, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    if (!x || !dy || !dx || !gamma || !beta || !stats || ((dgamma == nullptr) != (dbeta == nullptr))) return CLORA_ERR_ARG;
    GnArgs a = GnArgs();
    a.x = (const half_t*)x; a.dy = (const half_t*)dy; a.y = (half_t*)dx; a.gamma = gamma; a.beta = beta;
    a.stats = const_cast<float*>(stats); a.dgamma = dgamma; a.dbeta = dbeta;
    a.B = B; a.HW = HW; a.C = C; a.G = G; a.fuse_silu = fuse_silu;
    int rc = gn_plan(a, workspace, workspace_bytes, true);
    if (rc != CLORA_OK) return rc;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_bwd_partial_kernel, dim3(a.nchunk, B), dim3(256), 0, s, a);
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3(clora_cdiv(B * G, 256)), dim3(256), 0, s, a);
    hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(ew_blocks((size_t)B * HW * (C / 8))), dim3(256), 0, s, a);
    return clora_check_launch();
}

extern "C" int clora_layernorm_fwd_f16(const clora_half* x, clora_half* y, const float* gamma, const float* beta, int M,
                                       int C, float eps, void* stream) {
    if (!x || !y || !gamma || !beta || M <= 0 || C <= 0 || (C & 7) || C / 8 > 64 * kMaxCols) return CLORA_ERR_ARG;
    LnArgs a = LnArgs();
    a.x = (const half_t*)x; a.y = (half_t*)y; a.gamma = gamma; a.beta = beta; a.M = M; a.C = C; a.eps = eps;
    hipLaunchKernelGGL((layernorm_kernel<false>), dim3(clora_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return clora_check_launch();
}

extern "C" int clora_layernorm_bwd_f16(const clora_half* x, const clora_half* dy, clora_half* dx, const float* gamma,
                                       int M, int C, float eps, void* stream) {
    if (!x || !dy || !dx || !gamma || M <= 0 || C <= 0 || (C & 7) || C / 8 > 64 * kMaxCols) return CLORA_ERR_ARG;
    LnArgs a = LnArgs();
    a.x = (const half_t*)x; a.dy = (const half_t*)dy; a.y = (half_t*)dx; a.gamma = gamma; a.M = M; a.C = C; a.eps = eps;
    hipLaunchKernelGGL((layernorm_kernel<true>), dim3(clora_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, a);
    return clora_check_launch();
}
