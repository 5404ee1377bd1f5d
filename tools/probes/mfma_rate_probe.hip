// Issue rate of the fp16 MFMA shapes on gfx950, one wave per SIMD (256 CUs x 4 waves), 8 independent accumulators:
//   v_mfma_f32_16x16x32_f16 (gfx950), v_mfma_f32_16x16x16_f16 (legacy shape), v_mfma_f32_32x32x16_f16, v_mfma_f32_32x32x8_f16
// prints ns per MFMA per SIMD and the implied TFLOP/s -- decides whether a 32+16 contraction (head dim 40 -> 48 instead
// of 64) is cheaper than two 16x16x32 steps.   Build: hipcc --offload-arch=gfx950 -O3 mfma_rate_probe.hip -o mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

// The MFMAs are inline asm on "+v" accumulators: the compiler can neither merge the identical chains nor move the
// accumulators between the VGPR and AGPR files inside the loop (both happened with the builtins and made the first
// version of this probe meaningless).
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    const int l = threadIdx.x;
    h8 a8, b8;
    h4 a4, b4;
    for (int e = 0; e < 8; ++e) { a8[e] = (_Float16)(0.001f * (l + e)); b8[e] = (_Float16)(0.002f * (l - e)); }
    for (int e = 0; e < 4; ++e) { a4[e] = a8[e]; b4[e] = b8[e]; }
    float r = 0.f;
    if (MODE == 0 || MODE == 1) {
        f4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f4{(float)i, 0.f, (float)l, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));
                else asm volatile("v_mfma_f32_16x16x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
            }
        }
        for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][3];
    } else {
        f16v acc[4];
        for (int i = 0; i < 4; ++i)
            for (int e = 0; e < 16; ++e) acc[i][e] = (float)(i + e);
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a8), "v"(b8));
                else asm volatile("v_mfma_f32_32x32x8_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a4), "v"(b4));
            }
        }
        for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][15];
    }
    out[blockIdx.x * 256 + l] = r;
}

template <int MODE>
void run(const char* name, double flop_per_mfma, int per_iter) {
    float* d;
    hipMalloc(&d, 256 * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double n = (double)iters * per_iter;                 // MFMAs per wave = per SIMD
    const double ns = ms * 1e6 / n;
    printf("%-28s %7.2f ns per MFMA per SIMD  (%5.1f cycles @2.4 GHz)  %8.1f TFLOP/s chip\n", name, ns, ns * 2.4,
           flop_per_mfma * n * 1024 / (ms * 1e-3) / 1e12);
    hipFree(d);
}

int main() {
    run<0>("v_mfma_f32_16x16x32_f16", 2.0 * 16 * 16 * 32, 8);
    run<1>("v_mfma_f32_16x16x16_f16", 2.0 * 16 * 16 * 16, 8);
    run<2>("v_mfma_f32_32x32x16_f16", 2.0 * 32 * 32 * 16, 4);
    run<3>("v_mfma_f32_32x32x8_f16", 2.0 * 32 * 32 * 8, 4);
    return 0;
}
