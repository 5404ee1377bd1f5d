// clora_common.h -- shared device helpers for the gfx950 (CDNA4 / MI355X) kernels.
//
// Conventions used by every kernel in this directory
//   * activations are fp16, "tokens x channels" row-major (NHWC for images): [B, H*W, C]
//   * accumulation is fp32 on the matrix cores: v_mfma_f32_16x16x32_f16
//       A fragment: lane l holds A[row = l&15][8 k-slots of group g = l>>4]
//       B fragment: lane l holds B[8 k-slots of group g = l>>4][col = l&15]
//       C/D       : lane l holds D[row = 4*(l>>4)+r][col = l&15], r = 0..3
//     (the hardware pairs k-slot e of group g of A with k-slot e of group g of B, so any
//      consistent assignment of logical k to (g, e) on both operands is valid -- the
//      attention kernels use this to feed a C-layout tile straight back in as a B operand)
//   * wave = 64 lanes, workgroups are 256 threads (4 waves) unless stated
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

#define CLORA_WAVE 64

// error codes of the C ABI (include/clora.h)
#define CLORA_OK 0
#define CLORA_ERR_ARG (-1)
#define CLORA_ERR_LAUNCH (-2)
#define CLORA_ERR_WORKSPACE (-3)

// ---- asynchronous global -> LDS copies (LDS-DMA, `global_load_lds_dwordx4`) and the counted waits that let
// them stay in flight across a workgroup barrier.  Measured semantics on gfx950 (tools/probes/glds_probe.hip):
// the LDS destination is the FIRST lane's pointer + lane*16 (wave-uniform base), the global source is per lane.
// (tests/hipemu/hipemu.h provides host stand-ins for these three when the sources are built for the CPU emulator)
#ifndef CLORA_ASYNC_PRIMS
#define CLORA_ASYNC_PRIMS
#define CLORA_GLDS16(gptr, lptr)                                                                      \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr),            \
                                     (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)
#define CLORA_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
#define CLORA_RAW_BARRIER() __builtin_amdgcn_s_barrier()
// gfx950 LDS transpose read (ds_read_b64_tr_b16), semantics measured with tools/probes/tr16_probe.hip: every lane
// passes the address of 4 contiguous fp16; within each 16-lane group lane i receives element (i & 3) of the words
// of lanes (i >> 2) + 4j, j = 0..3.  Pointing lane l at row (l & 15) >> 2, columns (l & 3)*4.. of a row-major
// [4][16] block gives lane i column (i & 15), rows 0..3: a row-major LDS tile feeds a k-strided MFMA operand without
// a transposing write pass.
typedef __fp16 clora_fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ half4v clora_ds_read_tr16(const half_t* lptr) {
    clora_fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) clora_fp16x4*)(lptr));
    half4v r;
    __builtin_memcpy(&r, &v, 8);
    return r;
}
#define CLORA_DS_READ_TR16(lptr) clora_ds_read_tr16(lptr)
// bare v_exp_f32 (2^x, ~1 ulp).  exp2f() wraps it in a denormal-range fix-up (v_cmp / 2x v_cndmask / v_add / v_ldexp:
// five extra VALU instructions per element); softmax probabilities below 2^-126 may flush to zero.
#define CLORA_EXP2(x) __builtin_amdgcn_exp2f(x)
#define CLORA_RCP(x) __builtin_amdgcn_rcpf(x)      // v_rcp_f32 (1 ulp) instead of the IEEE division sequence
// Pin a loaded vector register: the value is "used and redefined" here, so the load that produced it cannot be sunk into a
// later conditional block.  (CodeGenPrepare turns `ok ? load : 0` and loads whose only use is a conditional store into
// branch + load -- in a batch of loads that puts load -> s_waitcnt vmcnt(0) back into every row.)  Costs no instruction.
#define CLORA_KEEP(x) asm volatile("" : "+v"(x))
// The same without `volatile`: does not count as a memory clobber, so uniform loads that FOLLOW it can still be scalar
// (s_load needs a provably unclobbered address range; MemorySSA treats volatile asm as a write).  Only where the kept value
// feeds a select, not a conditional block (a pure asm may be sunk together with its load).
#define CLORA_KEEP_PURE(x) asm("" : "+v"(x))
// acc += a * b as ONE v_fma_f32 the compiler cannot merge into a packed v_pk_fma_f32 (see hoisted_rank4 in clora_gemm.hip: the
// packed form of that block produced sporadic wrong low-half results on MI355X when several workgroups shared a CU)
#define CLORA_FMA_F32(acc, a, b) asm("v_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
// shader-cycle counter (s_memtime) and the constant 100 MHz wall clock (s_memrealtime): clora_clock_probe
#define CLORA_CYCLES() ((unsigned long long)__builtin_readcyclecounter())
#define CLORA_WALL_TICKS() ((unsigned long long)wall_clock64())
// acc = a . b + acc as ONE instruction on a register quad the compiler cannot re-number (the probe loop written with the builtin
// compiled to 40 v_accvgpr moves per 8 MFMAs: 28 cycles per MFMA instead of 16)
#define CLORA_MFMA_INPLACE(acc, a, b) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define CLORA_MFMA_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")   // before compiler code reads what an asm MFMA wrote
// schedule pins of the 8-phase GEMM main loop (gemm_8p_kernel): LDS-read counter wait, issue priority, "nothing crosses this line"
#define CLORA_WAIT_LGKMCNT(n) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(n) : "memory")
#define CLORA_SETPRIO(n) __builtin_amdgcn_s_setprio(n)
#define CLORA_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// words that workgroups of ONE launch exchange (GroupNorm team kernels, clora_norm.hip): relaxed agent-scope accesses = sc1 global
// loads / write-through stores (bypass the per-CU L1; cdna_hip_programming.md Guideline 16 form R2: the datum carries its own tag)
typedef __attribute__((address_space(1))) unsigned long long clora_gu64;
typedef __attribute__((address_space(1))) unsigned clora_gu32;
#define CLORA_LD_AGENT_U64(p) __hip_atomic_load((clora_gu64*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CLORA_ST_AGENT_U64(p, v) __hip_atomic_store((clora_gu64*)(p), (unsigned long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CLORA_LD_AGENT_U32(p) __hip_atomic_load((clora_gu32*)(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CLORA_ST_AGENT_U32(p, v) __hip_atomic_store((clora_gu32*)(p), (unsigned)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define CLORA_SLEEP() __builtin_amdgcn_s_sleep(1)
#endif
#ifndef CLORA_SEQUENTIAL_BLOCKS
#define CLORA_SEQUENTIAL_BLOCKS 0
#endif

__device__ __forceinline__ floatx4 mfma16(half8 a, half8 b, floatx4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ half8 ld8(const half_t* p) { return *reinterpret_cast<const half8*>(p); }
__device__ __forceinline__ void st8(half_t* p, half8 v) { *reinterpret_cast<half8*>(p) = v; }
__device__ __forceinline__ half4v ld4(const half_t* p) { return *reinterpret_cast<const half4v*>(p); }
__device__ __forceinline__ void st4(half_t* p, half4v v) { *reinterpret_cast<half4v*>(p) = v; }
__device__ __forceinline__ half8 zero8() {
    half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
    return z;
}
__device__ __forceinline__ floatx4 zero4f() {
    floatx4 z = {0.f, 0.f, 0.f, 0.f};
    return z;
}

// sigmoid through the bare hardware exp2 / rcp (1 ulp each; results are rounded to fp16 anyway)
__device__ __forceinline__ float sigmoid_f(float x) { return CLORA_RCP(1.0f + CLORA_EXP2(-1.4426950408889634f * x)); }
__device__ __forceinline__ float silu_f(float x) { return x * sigmoid_f(x); }
// d/dx silu(x) = s + x*s*(1-s), s = sigmoid(x)
__device__ __forceinline__ float dsilu_f(float x) {
    float s = sigmoid_f(x);
    return s * (1.0f + x * (1.0f - s));
}
// exact-erf GELU (what F.gelu / diffusers' GEGLU computes) with erf from Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7,
// far below the fp16 rounding of the result) on the bare exp2 / rcp instructions: ~15 VALU ops instead of libm's erff.
// gelu_parts returns Phi(x) = 0.5 (1 + erf(x / sqrt 2)) and phi(x) = exp(-x^2 / 2) / sqrt(2 pi) from ONE exponential.
__device__ __forceinline__ void gelu_parts(float x, float& cdf, float& pdf) {
    const float ax = fabsf(x) * 0.70710678118654752f;
    const float t = CLORA_RCP(1.0f + 0.3275911f * ax);
    const float ex = CLORA_EXP2(-1.4426950408889634f * ax * ax);          // exp(-x^2 / 2)
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float erf_abs = 1.0f - poly * ex;
    cdf = 0.5f * (1.0f + (x < 0.f ? -erf_abs : erf_abs));
    pdf = 0.39894228040143268f * ex;
}
__device__ __forceinline__ float gelu_f(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return x * cdf;
}
__device__ __forceinline__ float dgelu_f(float x) {
    float cdf, pdf;
    gelu_parts(x, cdf, pdf);
    return cdf + x * pdf;
}

__device__ __forceinline__ float wave_sum(float v) {
    v += __shfl_xor(v, 32);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 32));
    v = fmaxf(v, __shfl_xor(v, 16));
    v = fmaxf(v, __shfl_xor(v, 8));
    v = fmaxf(v, __shfl_xor(v, 4));
    v = fmaxf(v, __shfl_xor(v, 2));
    v = fmaxf(v, __shfl_xor(v, 1));
    return v;
}

// The library's ONLY process-global state: the tuning knobs of clora_set_option (include/clora.h), one int each, defined in
// clora_gemm.hip.  Results never depend on them.  The library itself reads no environment variable: the host layer
// (controllora_amd/capi.py) forwards CLORA_* variables through clora_set_option when it loads the library.
enum { CLORA_OPT_TILE_ORDER = 0, CLORA_OPT_LN_ROWS, CLORA_OPT_ATTN_FWD_WAVES, CLORA_OPT_ATTN_BWD_WAVES, CLORA_OPT_GN_BLOCKS, CLORA_OPT_EPI_TWO_PHASE, CLORA_OPT_LORA_DOWN_MODE, CLORA_OPT_GN_UNROLL, CLORA_OPT_EPI_HOIST, CLORA_OPT_GN_RESIDENT, CLORA_OPT_DEFER_MAX_ROWS, CLORA_OPT_WGRAD_PATCH, CLORA_OPT_STRIP_BLOCKS, CLORA_OPT_GN_TEAM, CLORA_OPT_COUNT };
__attribute__((visibility("hidden"))) int clora_option(int id);
// XCD assignment policy of the launches that follow ("tile_order"): 0 = launch-order defaults, 1 = n-major GEMM tiles (tests),
// 2 = fewest distinct operand panels per XCD; non-zero also gives every XCD whole attention heads (clora_attn.hip attn_block_ids)
static inline int clora_xcd_policy() { return clora_option(CLORA_OPT_TILE_ORDER); }
// LayerNorm with several rows in flight per wave ("ln_rows", default 1; 0: one row per wave)
static inline int clora_ln_rows() { return clora_option(CLORA_OPT_LN_ROWS); }

static inline int clora_check_launch() { return hipGetLastError() == hipSuccess ? CLORA_OK : CLORA_ERR_LAUNCH; }
static inline int clora_cdiv(long a, long b) { return (int)((a + b - 1) / b); }
