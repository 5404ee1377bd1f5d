// Do workgroups that run at the same time on one XCD share a stream of operand lines through that XCD's L2?  (The conv /
// GEMM tiles (m, 0), (m, 1), ... of one row panel read the same activation rows; PMC FETCH_SIZE says they are fetched about
// once PER TILE, profiles/r02_pmc_traffic_final.json.)  Streams live in a 4 GB buffer (every line is an L2 miss the first
// time).  Logical bytes = what the workgroups asked for; if k sharers are served by one fetch the logical rate approaches
// k x the ~6 TB/s fabric rate of profiles/r02_stream_rate_probe.txt.
//   modes: sharers on the SAME XCD (workgroups lin, lin+8, ...), on DIFFERENT XCDs (lin, lin+1, ...), and same-XCD sharers
//   where each follower walks `lag` KB behind the previous one (how long does a line survive in L2?).
//   Build: hipcc --offload-arch=gfx950 -O3 l2_share_probe.hip -o l2_share_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

// mode 0: k sharers same XCD   mode 1: k sharers on consecutive XCDs
__global__ __launch_bounds__(256) void share_k(const u4* __restrict__ src, size_t mask, int iters, int k, int mode, int lag_chunks, u4* sink) {
    __shared__ u4 lds[8 * 4 * 64];
    const int w = threadIdx.x >> 6;
    const int lin = blockIdx.x;
    int stream, member;
    if (mode == 0) { const int grp = lin / (8 * k), r = lin % (8 * k); stream = grp * 8 + (r & 7); member = r >> 3; }
    else { stream = lin / k; member = lin % k; }
    const size_t wave_stream = (size_t)stream * 4 + w;
    size_t idx = ((wave_stream * 2048 + wave_stream * 37 + (size_t)(k - 1 - member) * lag_chunks) * 64 + (threadIdx.x & 63));
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (idx & mask)),
                                         (__attribute__((address_space(3))) void*)(lds + (w * 8 + d) * 64), 16, 0, 0);
        idx += 64;
    }
    for (int it = 0; it < iters; ++it) {
        const int half = (it & 1) ? 0 : 4;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (idx & mask)),
                                             (__attribute__((address_space(3))) void*)(lds + (w * 8 + half + d) * 64), 16, 0, 0);
            idx += 64;
        }
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const u4 a = lds[threadIdx.x & 63];
    if (a[0] == 0x12345678u && a[1] == 0x9abcdef0u) sink[threadIdx.x] = a;
}

static void run(const u4* src, size_t bytes, int k, int mode, int lag_kb, u4* sink) {
    const int grid = 512;                                   // two 4-wave workgroups per CU
    const int iters = 510;                                  // 2 MB per wave (511 x 4 KB), streams 2 MB apart: 4 GB per repetition
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        // a different quarter of the buffer per repetition: nothing is left in L2 / MALL from the previous one
        const u4* base = src + (size_t)rep * (bytes / 4 / 16);
        hipEventRecord(e0);
        hipLaunchKernelGGL(share_k, dim3(grid), dim3(256), 0, 0, base, bytes / 4 / 16 - 1, iters, k, mode, lag_kb, sink);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
    }
    const double bytes_logical = (double)grid * 4 * (iters + 1) * 4 * 1024.0;
    printf("%-14s sharers %d  lag %5d KB   %7.3f ms   logical %6.2f TB/s   (unique %6.2f TB/s)\n", mode == 0 ? "same XCD" : "across XCDs", k, lag_kb,
           best, bytes_logical / (best * 1e-3) / 1e12, bytes_logical / k / (best * 1e-3) / 1e12);
    hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
    const size_t big = (size_t)16 << 30;
    u4* src; u4* sink;
    if (hipMalloc(&src, big) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&sink, 4096);
    hipMemset(src, 1, big);
    hipDeviceSynchronize();
    run(src, big, 1, 0, 0, sink);
    for (int k = 2; k <= 8; k *= 2) { run(src, big, k, 0, 0, sink); run(src, big, k, 1, 0, sink); }
    for (int lag = 16; lag <= 4096; lag *= 4) run(src, big, 2, 0, lag, sink);
    for (int lag = 16; lag <= 1024; lag *= 4) run(src, big, 4, 0, lag, sink);
    hipFree(src); hipFree(sink);
    return 0;
}
