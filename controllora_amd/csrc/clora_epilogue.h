// clora_epilogue.h -- the GEMM epilogue in its per-chunk forms, shared by the GEMM kernels (clora_gemm.hip) and by the consumers of a
// DEFERRED split-K GEMM (clora_norm.hip; include/clora.h clora_deferred_t): finish_chunk8 is the split-K finish pass for one 8-column
// chunk of one output row, the same loads and the same arithmetic in the same order as splitk_finish_kernel -- a consumer that
// folds the slabs itself gets bit-identical input.
#pragma once
#include "clora_common.h"
#include "../../include/clora.h"

// The residual add of the epilogue (x + f(x): ResnetBlock2D, every BasicTransformerBlock sub-layer, Transformer2DModel).
// Until round 6 the branch was rounded to fp16 FIRST and the sum again -- the reference's fp16 arithmetic (`hidden_states = attn(...) +
// hidden_states` on fp16 tensors).  Now the sum is formed from the fp32 accumulator and rounded once: one rounding fewer per residual
// site (~100 per UNet evaluation), which is what stands between the product and the fp32 oracle (parity target of north_star).
// -DCLORA_RES_ADD_TWICE restores the double rounding for A/B.
#ifdef CLORA_RES_ADD_TWICE
#define CLORA_RES_ADD(acc_f32, rounded_f16, res_f16) ((half_t)((float)(rounded_f16) + (float)(res_f16)))
#else
#define CLORA_RES_ADD(acc_f32, rounded_f16, res_f16) ((half_t)((acc_f32) + (float)(res_f16)))
#endif

namespace {

// everything of the epilogue that happens BEFORE the fp16 rounding (scalar form: split-K finish, v1 kernel)
__device__ __forceinline__ float epi_pre(float acc, int m, int n, const clora_epilogue_t& e) {
    if (e.bias) acc += e.bias[n];
    if (e.rowadd) acc += (float)((const half_t*)e.rowadd)[(size_t)(m / e.rows_per_batch) * e.ld_rowadd + n];
    if (e.lora_t) {
        const int r = e.lora_r;
        const float* t = e.lora_t + (size_t)m * e.ldt + (n / e.lora_seg) * r;
        float s = 0.f;
        if (e.lora_u_tr) {
            const float* u = e.lora_u + n;
            for (int j = 0; j < r; ++j) s += t[j] * u[(size_t)j * e.ldu];
        } else {
            const float* u = e.lora_u + (size_t)n * e.ldu;
            for (int j = 0; j < r; ++j) s += t[j] * u[j];
        }
        acc += e.lora_scale * s;
    }
    return acc;
}

// Row-chunk form: 8 consecutive columns n..n+7 of row m (n % 8 == 0, lora_seg % 16 == 0 so the chunk belongs to
// one adapter).  Bias, U and the rank-r row of T are fetched with float4 loads -- instead of 2r scalar loads
// per output element.  Used by the LDS-staged epilogue of the DMA kernel and by the split-K finish kernel.
__device__ __forceinline__ void epi_chunk8(float (&v)[8], int m, int n, const clora_epilogue_t& e) {
    if (e.bias) {
        const floatx4 b0 = *reinterpret_cast<const floatx4*>(e.bias + n), b1 = *reinterpret_cast<const floatx4*>(e.bias + n + 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] += b0[q]; v[4 + q] += b1[q]; }
    }
    if (e.rowadd) {
        const half8 ra = ld8((const half_t*)e.rowadd + (size_t)(m / e.rows_per_batch) * e.ld_rowadd + n);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += (float)ra[q];
    }
    if (e.lora_t) {
        const int R = e.lora_r;
        const int toff = (n / e.lora_seg) * R;
        const float* tp = e.lora_t + (size_t)m * e.ldt + toff;
        const bool tvec = ((e.ldt | toff) & 3) == 0;
        for (int j0 = 0; j0 < R; j0 += 4) {
            floatx4 t = zero4f();
            if (j0 + 4 <= R && tvec) t = *reinterpret_cast<const floatx4*>(tp + j0);
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (j0 + q < R) t[q] = tp[j0 + q];
            }
            t *= e.lora_scale;
            if (e.lora_u_tr) {                       // u(n, j) = U[j*ldu + n]: contiguous along n
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (j0 + q < R) {
                        const float* up = e.lora_u + (size_t)(j0 + q) * e.ldu + n;
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] += t[q] * up[c];
                    }
            } else if (j0 + 4 <= R && (e.ldu & 3) == 0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const floatx4 u = *reinterpret_cast<const floatx4*>(e.lora_u + (size_t)(n + c) * e.ldu + j0);
                    v[c] += t[0] * u[0] + t[1] * u[1] + t[2] * u[2] + t[3] * u[3];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    for (int q = 0; q < 4; ++q)
                        if (j0 + q < R) v[c] += t[q] * e.lora_u[(size_t)(n + c) * e.ldu + j0 + q];
            }
        }
    }
}

// Split-K finish kernel only.  Inside epi_chunk8 every `if (e.bias) / if (e.rowadd) / if (e.lora_t)` is its own basic block with
// load -> wait -> use, and the residual comes after all of them: four to five dependent L1 / L2 round trips per output chunk.
// Here the operands that do not depend on the accumulators are requested together, before the slab values are touched.
// (The tile epilogues keep epi_chunk8 exactly as it is: they are register-bound -- requesting the operands, or only the
// residual, at the top of a chunk spilled accumulators in the occupancy-bounded kernels (64x64 BK32: 0 -> 60..84 B,
// 128x128 BK32: 88 -> 184..216 B, patch 128x64: 116 -> 124..132 VGPRs), and even an unused extra parameter on epi_chunk8
// changed their allocation.)
struct EpiOps {
    floatx4 b0, b1, t4;
    half8 ra;
    bool have;
    bool have_t;        // t4 holds the first four T values of the row (rank >= 4, 16-byte aligned)
};

__device__ __forceinline__ void epi_issue(EpiOps& o, int m, int n, const clora_epilogue_t& e) {
    o.have = true;
    o.b0 = zero4f(); o.b1 = zero4f(); o.t4 = zero4f(); o.ra = zero8();
    if (e.bias) { o.b0 = *reinterpret_cast<const floatx4*>(e.bias + n); o.b1 = *reinterpret_cast<const floatx4*>(e.bias + n + 4); }
    if (e.rowadd) o.ra = ld8((const half_t*)e.rowadd + (size_t)(m / e.rows_per_batch) * e.ld_rowadd + n);
    o.have_t = e.lora_t && e.lora_r >= 4 && (e.ldt & 3) == 0 && (e.lora_r & 3) == 0;       // uniform: toff is a multiple of 4 then
    if (o.have_t) o.t4 = *reinterpret_cast<const floatx4*>(e.lora_t + (size_t)m * e.ldt + (n / e.lora_seg) * e.lora_r);
}

__device__ __forceinline__ void epi_chunk8_pre(float (&v)[8], int m, int n, const clora_epilogue_t& e, const EpiOps* pre) {
    if (e.bias) {
        floatx4 b0, b1;
        if (pre) { b0 = pre->b0; b1 = pre->b1; }
        else { b0 = *reinterpret_cast<const floatx4*>(e.bias + n); b1 = *reinterpret_cast<const floatx4*>(e.bias + n + 4); }
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[q] += b0[q]; v[4 + q] += b1[q]; }
    }
    if (e.rowadd) {
        const half8 ra = pre ? pre->ra : ld8((const half_t*)e.rowadd + (size_t)(m / e.rows_per_batch) * e.ld_rowadd + n);
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] += (float)ra[q];
    }
    if (e.lora_t) {
        const int R = e.lora_r;
        const int toff = (n / e.lora_seg) * R;
        const float* tp = e.lora_t + (size_t)m * e.ldt + toff;
        const bool tvec = ((e.ldt | toff) & 3) == 0;
        for (int j0 = 0; j0 < R; j0 += 4) {
            floatx4 t = zero4f();
            if (j0 == 0 && pre && pre->have_t) t = pre->t4;
            else if (j0 + 4 <= R && tvec) t = *reinterpret_cast<const floatx4*>(tp + j0);
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (j0 + q < R) t[q] = tp[j0 + q];
            }
            t *= e.lora_scale;
            if (e.lora_u_tr) {                       // u(n, j) = U[j*ldu + n]: contiguous along n
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (j0 + q < R) {
                        const float* up = e.lora_u + (size_t)(j0 + q) * e.ldu + n;
#pragma unroll
                        for (int c = 0; c < 8; ++c) v[c] += t[q] * up[c];
                    }
            } else if (j0 + 4 <= R && (e.ldu & 3) == 0) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const floatx4 u = *reinterpret_cast<const floatx4*>(e.lora_u + (size_t)(n + c) * e.ldu + j0);
                    v[c] += t[0] * u[0] + t[1] * u[1] + t[2] * u[2] + t[3] * u[3];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c)
                    for (int q = 0; q < 4; ++q)
                        if (j0 + q < R) v[c] += t[q] * e.lora_u[(size_t)(n + c) * e.ldu + j0 + q];
            }
        }
    }
}

// One 8-column chunk (n % 8 == 0) of row m of a split-K GEMM's output: fold the slabs (four slabs' loads in flight), epilogue, fp16,
// + residual.  partial = [splits][M][N] fp32.
__device__ __forceinline__ half8 finish_chunk8(const float* partial, int splits, int M, int N, const clora_epilogue_t& epi, int m, int n,
                                              half8* lo_out = nullptr) {
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
    const size_t zs = (size_t)M * N;                         // floats between consecutive slabs
    const float* q0 = partial + (size_t)m * N + n;
    // the chunk's epilogue operands and residual travel together with the first slab loads (they were four more dependent
    // round trips after the fold)
    EpiOps ops;
    epi_issue(ops, m, n, epi);
    half8 rr = zero8(), rl = zero8();
    if (epi.residual) rr = ld8((const half_t*)epi.residual + (size_t)m * epi.ldr + n);
    if (epi.residual_lo) rl = ld8((const half_t*)epi.residual_lo + (size_t)m * epi.ldr + n);
    int z = 0;
    for (; z + 4 <= splits; z += 4) {                        // four slabs' loads in flight (a load -> wait -> add loop paid one
        floatx4 a[4], b[4];                                  // L2 / HBM round trip per slab: up to 12 per output chunk)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            a[u] = *reinterpret_cast<const floatx4*>(q0 + (size_t)(z + u) * zs);
            b[u] = *reinterpret_cast<const floatx4*>(q0 + (size_t)(z + u) * zs + 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) { s[e] += a[u][e]; s[4 + e] += b[u][e]; }
    }
    for (; z < splits; ++z) {
        const floatx4 a = *reinterpret_cast<const floatx4*>(q0 + (size_t)z * zs);
        const floatx4 b = *reinterpret_cast<const floatx4*>(q0 + (size_t)z * zs + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { s[e] += a[e]; s[4 + e] += b[e]; }
    }
    epi_chunk8_pre(s, m, n, epi, &ops);
    half8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (half_t)s[e];
    if (epi.residual_lo || lo_out) {                          // compensated trunk (clora_epilogue_t.residual_lo / c_lo)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float sum = s[e] + (float)rr[e] + (float)rl[e];
            v[e] = (half_t)sum;
            if (lo_out) (*lo_out)[e] = (half_t)(sum - (float)v[e]);
        }
        return v;
    }
    if (epi.residual) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = CLORA_RES_ADD(s[e], v[e], rr[e]);
    }
    return v;
}

}  // namespace
