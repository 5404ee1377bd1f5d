// Probe of ds_read_b64_tr_b16 on gfx950: LDS holds s[i] = i (as fp16 bit patterns via integers < 2048), every lane
// passes its own address; print what each lane receives.  Build: hipcc --offload-arch=gfx950 tr16_probe.hip -o tr16_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __fp16 fp16x4 __attribute__((__vector_size__(8)));
__global__ void k(int mode, unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short s[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) s[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    int off;
    if (mode == 0) off = (l & 3) * 16 + ((l & 15) >> 2) * 4 + (l >> 4) * 64;      // lane 4q+b -> row b, cols 4q..4q+3 of a [4][16] block
    else if (mode == 1) off = l * 4;                                               // lane-linear words
    else off = (l & 15) * 16 + (l >> 4) * 4;                                       // lane i -> row i, 4 cols of group
    fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(s + off));
    unsigned short r[4];
    __builtin_memcpy(r, &v, 8);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) { printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); if (l == 19 && mode != 1) { l = 63; } }
    }
    return 0;
}
