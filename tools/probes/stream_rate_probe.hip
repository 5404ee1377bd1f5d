// How fast can one CU pull operands from L2 / MALL / HBM, and does the LDS-DMA path (global_load_lds_dwordx4, what every
// GEMM / conv / attention main loop here uses) differ from plain global_load_dwordx4 into registers?  The GEMM family was
// found bound by the global->LDS operand stream at ~20 B/clk/CU (profiles/r02_dma_probe.txt, r02_gemm_pmc_sq_*.json); this
// probe measures the ceiling of that stream in isolation as a function of the working set (2 MB: L2 hits, 64 MB: MALL,
// 4 GB: HBM), the wave-instructions in flight per wave (DEPTH) and the resident waves per CU.
//   Every wave-instruction moves 1 KB (64 lanes x 16 B, contiguous = eight 128-byte lines, like a BK=64 tile row group).
//   Build: hipcc --offload-arch=gfx950 -O3 stream_rate_probe.hip -o stream_rate_probe     (runs ~2 s)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

template <int DMA, int DEPTH>
__global__ __launch_bounds__(256) void stream_k(const u4* __restrict__ src, size_t mask, int iters, u4* sink) {
    __shared__ u4 lds[DMA ? DEPTH * 4 * 64 : 64];
    const int w = threadIdx.x >> 6;
    // every wave walks consecutive 1-KB chunks from its own start (2 MB apart, plus a skew so that the starts stay distinct
    // modulo a small working set): no two waves of a CU touch the same line at the same time, nothing is served by the vL1D
    const size_t stride = 64;
    const size_t wave_id = (size_t)blockIdx.x * 4 + w;
    size_t idx = (wave_id * 2048 + wave_id * 37) * 64 + (threadIdx.x & 63);
    u4 acc = {0u, 0u, 0u, 0u};
    constexpr int H = DEPTH / 2;
    if (DMA) {
        // two half-batches alternate: H..DEPTH wave-instructions stay in flight, as in the kernels' counted-vmcnt rings
#pragma unroll
        for (int d = 0; d < H; ++d) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (idx & mask)),
                                             (__attribute__((address_space(3))) void*)(lds + (w * DEPTH + d) * 64), 16, 0, 0);
            idx += stride;
        }
        for (int it = 0; it < iters; ++it) {
            const int half = (it & 1) ? 0 : H;
#pragma unroll
            for (int d = 0; d < H; ++d) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (idx & mask)),
                                                 (__attribute__((address_space(3))) void*)(lds + (w * DEPTH + half + d) * 64), 16, 0, 0);
                idx += stride;
            }
            if (H == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (H == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            if (H == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc = lds[threadIdx.x & 63];
    } else {
        u4 a[H], b[H];
#pragma unroll
        for (int d = 0; d < H; ++d) { a[d] = src[idx & mask]; idx += stride; }
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int d = 0; d < H; ++d) { b[d] = src[idx & mask]; idx += stride; }
#pragma unroll
            for (int d = 0; d < H; ++d) acc ^= a[d];
#pragma unroll
            for (int d = 0; d < H; ++d) { a[d] = src[idx & mask]; idx += stride; }
#pragma unroll
            for (int d = 0; d < H; ++d) acc ^= b[d];
        }
#pragma unroll
        for (int d = 0; d < H; ++d) acc ^= a[d];
    }
    if (acc[0] == 0x12345678u && acc[1] == 0x9abcdef0u) sink[threadIdx.x] = acc;     // never true: keeps the loads alive
}

template <int DMA, int DEPTH>
void run(const u4* src, size_t bytes_ws, int blocks_per_cu, u4* sink, const char* ws_name) {
    const int grid = 256 * blocks_per_cu;
    const size_t per_block = (size_t)8 << 20;                         // 8 MB per CU in total
    const size_t per_wave_instr = per_block / blocks_per_cu / 4 / 1024;   // wave-instructions per wave
    int iters = (int)(per_wave_instr / (DEPTH / 2));
    iters &= ~1;
    const size_t mask = bytes_ws / 16 - 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((stream_k<DMA, DEPTH>), dim3(grid), dim3(256), 0, 0, src, mask, iters, sink);    // warm (fills L2 / MALL)
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_k<DMA, DEPTH>), dim3(grid), dim3(256), 0, 0, src, mask, iters, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * 4 * ((double)iters + 1) * (DEPTH / 2) * 1024.0;
    const double tbs = bytes / (ms * 1e-3) / 1e12;
    printf("%-4s %-5s depth %2d  waves/CU %2d  %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU @2.4GHz\n", DMA ? "dma" : "reg", ws_name, DEPTH,
           4 * blocks_per_cu, ms, tbs, tbs * 1e12 / 256 / 2.4e9);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const size_t big = (size_t)4 << 30;
    u4* src; u4* sink;
    if (hipMalloc(&src, big) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 4096);
    hipMemset(src, 1, big);
    hipDeviceSynchronize();
    const size_t ws[3] = {(size_t)2 << 20, (size_t)64 << 20, big};
    const char* names[3] = {"2MB", "64MB", "4GB"};
    for (int i = 0; i < 3; ++i) {
        for (int bpc = 1; bpc <= 4; bpc *= 2) {
            run<1, 4>(src, ws[i], bpc, sink, names[i]);
            run<0, 4>(src, ws[i], bpc, sink, names[i]);
            run<1, 8>(src, ws[i], bpc, sink, names[i]);
            run<0, 8>(src, ws[i], bpc, sink, names[i]);
            if (bpc <= 2) {
                run<1, 16>(src, ws[i], bpc, sink, names[i]);
                run<0, 16>(src, ws[i], bpc, sink, names[i]);
            }
        }
    }
    hipFree(src); hipFree(sink);
    return 0;
}
