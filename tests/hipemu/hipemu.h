// hipemu.h -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// A tiny host-side emulator of the HIP subset used by controllora_amd/csrc so that the
// *logic* of every kernel (tile indexing, LDS staging, barriers, MFMA fragment layouts,
// wave shuffles, split-K, edge handling) can be exercised by the CPU test-suite in a
// container without a GPU.  Each GPU thread is a fiber; __syncthreads(), wave shuffles
// and MFMA are rendezvous points.  MFMA follows the gfx950 16x16x32 f16 register layout
// (A: row = lane&15, k-group = lane>>4; B: col = lane&15, k-group = lane>>4;
//  C/D: col = lane&15, row = 4*(lane>>4)+reg -- cdna_hip_programming.md section 3).
// It says nothing about performance and is not a fallback: the product binding
// (controllora_amd/capi.py) only ever loads the gfx950 library.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0
enum hipMemcpyKind { hipMemcpyDeviceToDevice = 3 };

typedef _Float16 hipemu_half8 __attribute__((ext_vector_type(8)));
typedef float hipemu_floatx4 __attribute__((ext_vector_type(4)));

namespace hipemu {
struct Fiber;
extern Fiber* cur;
extern dim3 g_blockIdx, g_blockDim, g_gridDim;
dim3& cur_tid();
void run_grid(dim3 grid, dim3 block, const std::function<void()>& body);
void sync_threads();
uint32_t wave_shfl(uint32_t v, int src_lane_or_mask, int mode);   // mode 0: idx, 1: xor, 2: down
hipemu_floatx4 mfma_16x16x32_f16(hipemu_half8 a, hipemu_half8 b, hipemu_floatx4 c);

template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t, hipStream_t, Args... args) {
    std::function<void()> body = [=]() { kernel(args...); };
    run_grid(grid, block, body);
}
}  // namespace hipemu

#define threadIdx (hipemu::cur_tid())
#define blockIdx (hipemu::g_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch(kernel, grid, block, shmem, stream, __VA_ARGS__)

// __syncthreads() is a fence + barrier: the compiler drains vmcnt (incl. LDS-DMA loads) before s_barrier; the raw barrier
// (CLORA_RAW_BARRIER) does not -- the emulator models both (see glds16 / wait_vmcnt below)
namespace hipemu { void wait_vmcnt(int n); }
static inline void __syncthreads() { hipemu::wait_vmcnt(0); hipemu::sync_threads(); }

template <typename T> static inline T hipemu_shfl(T v, int a, int mode) {
    static_assert(sizeof(T) == 4, "emulator shuffles 32-bit values");
    uint32_t u; memcpy(&u, &v, 4);
    u = hipemu::wave_shfl(u, a, mode);
    T r; memcpy(&r, &u, 4);
    return r;
}
template <typename T> static inline T __shfl(T v, int src, int = 64) { return hipemu_shfl(v, src, 0); }
template <typename T> static inline T __shfl_xor(T v, int mask, int = 64) { return hipemu_shfl(v, mask, 1); }
template <typename T> static inline T __shfl_down(T v, int d, int = 64) { return hipemu_shfl(v, d, 2); }

// wave vote: OR of the predicate over the live lanes (butterfly of xor shuffles; missing lanes read themselves)
static inline int __any(int pred) {
    int v = pred ? 1 : 0;
    for (int m = 32; m >= 1; m >>= 1) v |= hipemu_shfl(v, m, 1);
    return v;
}

static inline hipemu_floatx4 __builtin_amdgcn_mfma_f32_16x16x32_f16(hipemu_half8 a, hipemu_half8 b,
                                                                    hipemu_floatx4 c, int, int, int) {
    return hipemu::mfma_16x16x32_f16(a, b, c);
}

static inline int __builtin_amdgcn_readfirstlane(int x) { return x; }

// host stand-ins for the asynchronous-copy primitives of clora_common.h, PESSIMISTIC about latency: an LDS-DMA copy is queued
// per wave and only lands when that wave executes a counted wait that covers it (s_waitcnt vmcnt(n): all but the n newest
// land, oldest first) or a __syncthreads(); until then LDS keeps its old bytes.  A kernel that waits for too few loads, or
// reads a stage another wave has not waited for yet, computes garbage here -- on the hardware the same bug is an intermittent
// race.  Destination = first lane's LDS pointer + lane*16 like the hardware.
namespace hipemu { void glds16(const void* gptr, void* lptr); }
#define CLORA_ASYNC_PRIMS
#define CLORA_GLDS16(gptr, lptr) hipemu::glds16((const void*)(gptr), (void*)(lptr))
#define CLORA_WAIT_VMCNT(n) hipemu::wait_vmcnt(n)
#define CLORA_RAW_BARRIER() hipemu::sync_threads()
namespace hipemu { uint64_t ds_read_tr16_b64(const void* lptr); }
typedef _Float16 hipemu_half4 __attribute__((ext_vector_type(4)));
static inline hipemu_half4 hipemu_tr16(const void* p) {
    uint64_t u = hipemu::ds_read_tr16_b64(p);
    hipemu_half4 r; memcpy(&r, &u, 8);
    return r;
}
#define CLORA_DS_READ_TR16(lptr) hipemu_tr16((const void*)(lptr))
#define CLORA_EXP2(x) exp2f(x)
#define CLORA_RCP(x) (1.0f / (x))
#define CLORA_KEEP(x) ((void)0)
#define CLORA_KEEP_PURE(x) ((void)0)
#define CLORA_FMA_F32(acc, a, b) ((acc) = fmaf((a), (b), (acc)))
#define CLORA_CYCLES() (0ull)
#define CLORA_WALL_TICKS() (0ull)
#define CLORA_MFMA_INPLACE(acc, a, b) ((acc) = __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (acc), 0, 0, 0))
#define CLORA_MFMA_DRAIN() ((void)0)
#define CLORA_WAIT_LGKMCNT(n) ((void)0)
#define CLORA_SETPRIO(n) ((void)0)
#define CLORA_SCHED_BARRIER() ((void)0)
#define CLORA_LD_AGENT_U64(p) (*(const volatile unsigned long long*)(p))
#define CLORA_ST_AGENT_U64(p, v) (*(volatile unsigned long long*)(p) = (unsigned long long)(v))
#define CLORA_LD_AGENT_U32(p) (*(const volatile unsigned*)(p))
#define CLORA_ST_AGENT_U32(p, v) (*(volatile unsigned*)(p) = (unsigned)(v))
#define CLORA_SLEEP() ((void)0)
// the emulator runs the blocks of a grid one after the other: kernels whose blocks wait for each other (GroupNorm team kernels) are
// launched twice by their host code, the first time publish-only
#define CLORA_SEQUENTIAL_BLOCKS 1
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
static inline int hipGetDevice(int* d) { *d = 0; return 0; }
static inline int hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 256; return 0; }
static inline float atomicAdd(float* p, float v) { float o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline int atomicAdd(int* p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicMax(unsigned* p, unsigned v) { unsigned o = *p; if (v > o) *p = v; return o; }
static inline unsigned atomicOr(unsigned* p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }

static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return 0; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return 0; }
static inline hipError_t hipGetLastError() { return 0; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
