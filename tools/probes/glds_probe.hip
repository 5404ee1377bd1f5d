// probe: semantics of __builtin_amdgcn_global_load_lds (16 B) on gfx950 -- where does lane l's data land?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gvoid;
__global__ void k(const unsigned* src, unsigned* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4096];
    const int t = threadIdx.x, l = t & 63, w = t >> 6;
    for (int i = t; i < 4096; i += blockDim.x) lds[i] = 0xdeadbeef;
    __syncthreads();
    // each lane reads 16 B from a lane-specific (permuted) source; the LDS pointer is per-wave base
    const unsigned* g = src + ((l * 7) % 64) * 4 + w * 256;
    unsigned* base = lds + w * 512;            // 2 KB per wave
    if (mode == 0) __builtin_amdgcn_global_load_lds((gvoid*)g, (lds_void*)base, 16, 0, 0);
    else __builtin_amdgcn_global_load_lds((gvoid*)g, (lds_void*)(base + l * 4), 16, 0, 0);   // per-lane lptr?
    __builtin_amdgcn_global_load_lds((gvoid*)(g + 1024), (lds_void*)base, 16, 1024, 0);        // imm offset 1024 B
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int i = t; i < 4096; i += blockDim.x) out[i] = lds[i];
}
int main() {
    std::vector<unsigned> h(8192);
    for (int i = 0; i < 8192; ++i) h[i] = i;
    unsigned *d, *o;
    hipMalloc(&d, 8192 * 4); hipMalloc(&o, 4096 * 4);
    hipMemcpy(d, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(128), 0, 0, d, o, mode);
        std::vector<unsigned> r(4096);
        hipMemcpy(r.data(), o, 4096 * 4, hipMemcpyDeviceToHost);
        printf("mode %d (lptr %s)\n", mode, mode ? "per-lane base+l*16" : "wave-uniform base");
        for (int w = 0; w < 2; ++w) {
            int ok_linear = 0, ok2 = 0;
            for (int l = 0; l < 64; ++l) {
                unsigned expect = ((l * 7) % 64) * 4 + w * 256;
                if (r[w * 512 + l * 4] == expect && r[w * 512 + l * 4 + 3] == expect + 3) ok_linear++;
                if (r[w * 512 + 256 + l * 4] == expect + 1024) ok2++;
            }
            printf("  wave %d: lane-linear hits %d/64, imm-offset(1024B) hits %d/64, first words: %u %u %u %u | %u\n", w, ok_linear, ok2,
                   r[w * 512], r[w * 512 + 1], r[w * 512 + 4], r[w * 512 + 8], r[w*512+256]);
        }
    }
    return 0;
}
