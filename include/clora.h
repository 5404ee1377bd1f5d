/* clora.h -- C ABI of libclora (gfx950 / MI355X kernels for the ControlLoRA hot path).
 *
 * Drop-in boundary.  The reference (HighCWu/ControlLoRA) is pure Python
 * and delegates every operation below to PyTorch/diffusers ops; each entry point names the
 * reference call it replaces.  Rules of the ABI:
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller;
 *   - kernels are enqueued on `stream` (a hipStream_t passed as void*) and never synchronise,
 *     allocate, or keep global state, so they can be captured in a hipGraph;
 *   - return 0 on success, negative on error (CLORA_ERR_*), never throw;
 *   - activations are fp16 "tokens x channels" row-major (NHWC); statistics, adapter
 *     parameters, gradients of trainable parameters and optimizer state are fp32.
 * Reference-side binding: see INTEGRATION.md (ctypes stub used by controllora_amd/capi.py).
 */
#ifndef CLORA_H
#define CLORA_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CLORA_OK 0
#define CLORA_ERR_ARG (-1)
#define CLORA_ERR_LAUNCH (-2)
#define CLORA_ERR_WORKSPACE (-3)

typedef uint16_t clora_half; /* IEEE fp16 bits */

/* ---- implicit-GEMM convolution gather (NHWC).  Row m of the GEMM is output pixel
 * (b, yo, xo); column k is (tap = ky*ksize+kx, ci).  Source coordinate along y:
 *   t = yo*mul + ky*kmul + off ; valid iff 0 <= t < lim_h (and t even when need_even);
 *   ysrc = t >> shift.   (same for x with lim_w)
 * forward stride s, pad p      : mul=s, kmul=+1, off=-p, lim=Hin,      shift=0
 * forward nearest-x2 upsample  : mul=1, kmul=+1, off=-1, lim=2*Hin,    shift=1
 * dgrad of stride 1, pad p     : mul=1, kmul=-1, off=+p, lim=Hout_fwd, shift=0
 * dgrad of stride 2, pad p     : mul=1, kmul=-1, off=+p, lim=2*Hout_fwd, shift=1, need_even=1 */
typedef struct {
    int enabled;
    int Hin, Win, Cin;   /* dims of the tensor being gathered (A operand) */
    int Hout, Wout;      /* spatial dims that enumerate the GEMM rows */
    int ksize;           /* 1 or 3 */
    int mul, kmul, off, lim_h, lim_w, shift, need_even;
    int kchunk;          /* K order of the gather (and of B's columns): 0 = tap-major, k = tap*Cin + ci;
                          * c > 0 (Cin % c == 0) = channel-chunk-major, k = ((ci / c) * ksize^2 + tap) * c + ci % c:
                          * the ksize^2 taps of one c-channel slab are walked back to back, so the shifted re-reads of the
                          * same input pixels happen within a few k-steps (L1 / L2 hits instead of a pass over the whole
                          * activation per tap) */
} clora_conv_t;

struct clora_deferred;      /* below: a split-K GEMM whose reduction + epilogue is left to the consumer of its output */

/* ---- fused epilogue of clora_gemm_f16:
 *   acc[m,n] (+ bias[n]) (+ rowadd[m / rows_per_batch, n]) (+ lora_scale * sum_j T[m, toff+j] * U[n, j])
 *   -> fp16 -> (+ residual[m,n]) -> C[m,n]
 * with toff = (n / lora_seg) * lora_r  (several adapters sharing one concatenated GEMM). */
typedef struct {
    const float* bias;          /* [N] or NULL */
    const clora_half* rowadd;   /* [M / rows_per_batch, ld_rowadd] or NULL  (time-embedding add) */
    int rows_per_batch;
    int ld_rowadd;
    const clora_half* residual; /* [M, ldr] or NULL */
    int ldr;
    const float* lora_t;        /* [M, ldt] or NULL : X . D^T (fp32, from clora_lora_down) */
    int ldt;
    const float* lora_u;        /* adapter up weights u(n,j): lora_u[n*ldu + j], or lora_u[j*ldu + n] when lora_u_tr */
    int ldu;
    int lora_u_tr;
    int lora_r;
    int lora_seg;
    float lora_scale;
    /* GEGLU fused into the feed-forward GEMMs (upstream FeedForward / GEGLU, SURVEY.md U5: proj -> a * gelu_erf(g) -> out):
     * geglu = 1 (forward, the `proj` GEMM, N = 2F): B's rows (and bias) are packed in groups of 128 = [64 a-columns |
     *   the 64 matching g-columns]; the epilogue writes y[m, j] = fp16(a) * gelu(fp16(g)) to geglu_y [M, F] and, when C is
     *   not NULL, fp16(a) / fp16(g) to C[m, j] / C[m, F + j] in the standard layout (what the backward needs).
     * geglu = 2 (backward, the dgrad GEMM of `out`, N = F): the GEMM result is dy; with a, g read from geglu_h [M, 2F]
     *   the epilogue writes C[m, j] = dy * gelu(g) and C[m, F + j] = dy * a * gelu'(g)  (C is [M, 2F]).
     * Both need split_k = 1, tiles at least 128 columns wide and no rowadd / residual / adapter terms. */
    int geglu;
    int geglu_f;                 /* F */
    const clora_half* geglu_h;   /* mode 2 */
    clora_half* geglu_y;         /* mode 1 */
    /* Adapter down-projection evaluated BY this launch (round 4; reference models.py:232-282: every `attn.to_x(h) + scale *
     * to_x_lora(h)` pair reads h twice -- once for the frozen projection, once for LoRALinearLayer.down).  lora_dpack != NULL:
     * the launch computes T[m, s*lora_r + j] = sum_k A[m, k] * D_s[j, k] for the adapter of every column segment s from the A
     * rows it streams for the projection itself (8 extra operand rows per ring stage: fp16(D) and fp16(D - fp16(D)), i.e. fp32
     * weights to ~2^-22, as clora_lora_down computes it), WRITES it to lora_t (which must then be writable: the backward reads
     * it) and uses it in the epilogue without a global round trip.  lora_t_in (optional): a precomputed part that is ADDED to
     * the computed T of the segments whose bit is set in lora_t_in_mask -- the control term's share of the q adapter,
     * L_q(h + c) = L_q(h) + L_q(c) (reference models.py:237-238) -- row m reads row m % lora_t_in_rows when lora_t_in_rows > 0
     * (control batch 1 broadcast), same column layout as lora_t with row pitch ldt_in.
     * Needs: plain GEMM (no conv), lora_r == 4, lora_seg % 64 == 0, K % 64 == 0, split_k == 1, no GEGLU (else CLORA_ERR_ARG);
     * runs on the BK = 64 tiles that lie inside one column segment -- 8-wave 320-column tiles (tile_cfg 51, 52, 54, 55), 64-column
     * tiles (22, 23, 26, 42, 43), 128x128 (21, 41); any other tile_cfg is replaced by the library's choice (54 / 55, or 43 when
     * lora_seg is not a multiple of 320).  Every n-tile of a segment recomputes T; the first one writes it. */
    const clora_half* lora_dpack; /* [N / lora_seg][8][K] from clora_lora_pack_f16, or NULL */
    const float* lora_t_in;
    int ldt_in, lora_t_in_rows;
    unsigned lora_t_in_mask;
    /* round 6: `defer` != NULL asks the launch NOT to run its split-K finish pass: if the launch is split-K (the caller's or the
     * planner's choice) the slabs stay in `workspace`, *defer is filled in and C is left unwritten; otherwise defer->splits = 0 and C
     * is complete as usual.  The caller hands *defer to the consumer of C (clora_groupnorm_*_ex / clora_layernorm_bwd_f16_ex), or to
     * clora_finish_deferred, before anything else uses `workspace`.  Not with GEGLU. */
    struct clora_deferred* defer;
    /* round 6: LayerNorm of the OUTPUT rows evaluated by this launch (upstream BasicTransformerBlock: `norm2(attn1(...) + x)`,
     * `norm3(...)`, and norm1 of proj_in's output): ln_out[m, :] = LayerNorm(C[m, :]) * ln_gamma + ln_beta over the
     * N columns, from the fp16 values stored to C (the arithmetic of clora_layernorm_fwd_f16, same summation order), written next to
     * C (row pitch N).  Only where ONE tile spans the row: N == 320 on tile_cfg 51 / 52 / 54 / 55 (the level-0 projections of the
     * SD-1.5 UNet), split_k == 1, no GEGLU; anything else: CLORA_ERR_ARG (ask clora_gemm_ln_fusable first). */
    const float* ln_gamma;
    const float* ln_beta;
    clora_half* ln_out;
    float ln_eps;
    /* round 6 -- compensated residual trunk (the x + f(x) chain of upstream ResnetBlock2D / BasicTransformerBlock / Transformer2DModel,
     * SURVEY.md A5-A7): c_lo != NULL (row pitch ldc) receives fp16(v - fp16(v)) for every element of C, v being the fp32 value that
     * was rounded into C -- the rounding remainder; residual_lo != NULL (needs `residual`, row pitch ldr) is added to the residual in
     * fp32, so that (residual, residual_lo) = the (C, c_lo) pair of the launch that wrote the trunk tensor continues the sum from its
     * un-rounded value.  Every other consumer (norms, MFMA operands) reads C as before.  Not with GEGLU; a split-K launch that carries
     * c_lo is never deferred (CLORA_ERR_ARG); tile_cfg 61 (strip kernel) is not taken. */
    const clora_half* residual_lo;
    clora_half* c_lo;
} clora_epilogue_t;

/* A deferred split-K GEMM (clora_epilogue_t.defer).  element (m, n) of the GEMM's output is
 *   fp16( fp16( sum_z partial[z][m][n] + bias / rowadd / adapter terms of `epi` ) + residual )
 * -- exactly what the library's own finish pass stores -- evaluated by the consumer while it loads its input: 71 of the 108 finish
 * launches of a train step sit directly in front of a GroupNorm / LayerNorm launch that re-reads what they wrote. */
typedef struct clora_deferred {
    const float* partial;       /* [splits][M][N] fp32 */
    int splits, M, N;
    clora_half* C;              /* where the finished tensor belongs (the GEMM's own C / ldc) */
    int ldc;
    clora_epilogue_t epi;       /* the GEMM's epilogue (defer = NULL inside) */
} clora_deferred_t;
/* would clora_gemm_f16_ex(..., tile_cfg, split_k) take clora_epilogue_t.ln_out for this shape?  (host-only, no launch) */
int clora_gemm_ln_fusable(int M, int N, int K, int tile_cfg, int split_k);
/* the plain finish pass for a deferred GEMM (a consumer that cannot fold it, or no consumer at all) */
int clora_finish_deferred(const clora_deferred_t* d, void* stream);

/* Packs adapter matrices for `lora_dpack`: out[8][K] fp16, rows 0..R-1 = fp16(scale * D), rows 4..4+R-1 = fp16(scale * D -
 * fp16(scale * D)), other rows zero; R <= 4.  kmajor = 0: D is a down weight [R, K] with row pitch ldd (LoRALinearLayer.down,
 * reference models.py:40-44); kmajor = 1: D is an up weight [K, R] with row pitch ldd used as its own transpose (the backward's
 * dT = dY . (scale * U)).  `table` is a DEVICE array of njobs jobs (built once per model: the pointers are stable), one launch
 * repacks every adapter of a step -- what the per-call fp32 -> compute-dtype casts of LoRALinearLayer.forward do. */
typedef struct {
    const float* D;
    clora_half* out;
    int ldd, R, K, kmajor;
    float scale;
    int pad_;
} clora_lora_pack_job_t;
int clora_lora_pack_f16(const clora_lora_pack_job_t* table, int njobs, int max_k, void* stream);

/* C[M,N] = A[M,K] . B[N,K]^T  (fp16 in, fp32 accumulate on MFMA, fp16 out).
 * Replaces torch Linear / Conv2d forward AND dgrad of the frozen UNet layers
 * (reference models.py:124-147,231-282 `attn.to_q/to_k/to_v/to_out`; upstream ResnetBlock2D /
 * Transformer2DModel / FeedForward, SURVEY.md U1,U4,U5) and of the hint encoder (models.py:470,529,684).
 * `conv` may be NULL (plain GEMM, A row stride lda).  K must be a multiple of 8, N a multiple of 8.
 * split_k: 0 = chosen by the library's cost model (bounded by the workspace provided), >= 1 = forced;
 * split_k > 1 needs workspace >= split_k*M*N*4 bytes. */
int clora_gemm_f16(const clora_half* A, int lda, const clora_half* B, clora_half* C, int ldc,
                   int M, int N, int K, const clora_conv_t* conv, const clora_epilogue_t* epi,
                   int split_k, void* workspace, size_t workspace_bytes, void* stream);
/* same, with the main-loop variant forced -- tuning / tests (0 = automatic: the library's latency model, or the patch-staged conv
 * kernel for the 3x3 convs it can take).  tile_cfg:
 *   1-3   128x128 / 128x64 / 64x64 tiles, BK 32, 3-stage LDS-DMA ring        4-6  the same with the deep 5/6/8-stage ring
 *   7, 8  256x128 (wave tile 128x64)                                         9    1 with the round-1 swizzle key (A/B)
 *   11-13 register-staged round-1 loop (A/B)
 *   21, 22, 23, 26  BK 64 (128-byte LDS rows): 128x128 x2 stages, 128x64 x3, 64x64 x3, 128x64 x2
 *   31-33 = 1-3 and 41-43 = 21-23 with the fragment reads before the ring refill
 *   51-58 8-wave blocks (one per CU), BK 64: 128x320, 64x320, 128x256 (54-56: fragment-first), 256x320, 256x256
 *   59    gemm_8p_kernel: 256x256, 8 waves of 128x64, eight-phase ping-pong schedule (the two wave rows one barrier apart, 16-KB
 *         half-tiles by LDS-DMA seven ahead, one counted wait per K-tile); plain GEMM rows only -- a conv falls back to 58
 *   71-76 conv3x3_patch_kernel (3x3 stride-1 pad-1 convs, their dgrads, conv(nearest-2x(x))): 256x128, 128x128 (two wave layouts),
 *         256x64, 128x64, 128x160; 77, 78: 128x128 / 128x64 on a 392-pixel patch (one 128-pixel row or row segment: W = 128 .. 512);
 *         79: 256x160 (64x80 wave tiles, the CU's whole 160 KB of LDS); a shape it cannot take falls back to 21
 *         (clora_conv_patch_eligible tells) */
int clora_gemm_f16_ex(const clora_half* A, int lda, const clora_half* B, clora_half* C, int ldc,
                      int M, int N, int K, const clora_conv_t* conv, const clora_epilogue_t* epi,
                      int split_k, int tile_cfg, void* workspace, size_t workspace_bytes, void* stream);

/* 1 when clora_gemm_f16_ex(..., tile_cfg) would run `conv` (M output pixels) on the patch-staged 3x3 kernel (tile_cfg 71..79:
 * stride 1, pad 1, kchunk 64, whole image rows per tile), 0 when it would fall back to the implicit-GEMM main loop.  The
 * reference op is the same F.conv2d of upstream ResnetBlock2D; tuner / tests use this to know what they time. */
int clora_conv_patch_eligible(int M, const clora_conv_t* conv, int tile_cfg);

/* 1 when clora_gemm_f16_ex(..., tile_cfg = 61) can run `conv` with N output columns on the strip kernel (round 6): 3x3, stride 1, pad 1
 * or its dgrad, kchunk 0, 32 input channels and <= 64 output channels or 64 and <= 32, image rows a multiple of 128 pixels wide -- the
 * hint encoder's 512^2 / 256^2 convolutions (reference models.py:470-543: F.conv2d of ConvBlock2D / SimpleDownEncoderBlock2D and its
 * autograd dgrad); additionally the launch must be bias-only (no row add / residual / adapter / GEGLU / LayerNorm / split-K), else
 * tile_cfg 61 falls back to the library's own choice.  A column strip of 128 pixels walks down the image: the three input rows are
 * staged once per output row (not once per filter tap) and the whole weight operand stays in registers. */
int clora_conv_strip_eligible(int M, int N, const clora_conv_t* conv);

/* Tuning knobs, no reference counterpart: results never depend on them (bit-identical outputs, tests/test_kernels_*.py; the two
 * exceptions re-partition an fp32 sum: "lora_down_mode" (the sum over K) -- bit-identical per mode, equal to ~1e-7 relative across
 * modes, tests/test_kernels_gpu.py::test_lora_down_launch_modes -- and "gn_resident" (the GroupNorm statistics) -- bit-identical per
 * setting, a few fp16 ulps on a handful of outputs across settings, tests/kernel_cases.py::case_groupnorm).
 * This table is the ABI's ONLY process-global state (every other entry point is a pure function of its arguments and the
 * stream); the library reads no environment variable.  A knob takes effect for the launches that follow (a captured hipGraph
 * keeps what it was captured with).
 *   "tile_order"      how clora_gemm_f16[_ex] assigns output tiles -- and the attention kernels their (batch, head) blocks -- to
 *                     the eight XCDs.  0 = every XCD a contiguous range of tiles in m-major order, attention blocks in launch
 *                     order; 1 = n-major tile ranges; 2 = per launch the order that fetches fewer distinct A / B panels per XCD
 *                     (same-box A/B on MI355X 24.51 -> 24.35 ms/step); 3 (default) = as 2, plus per launch an (split, m, n) RECTANGLE of tiles
 *                     per XCD where whole divisors exist and the same panel count model prefers it (fabric traffic of the GEMM
 *                     family 1.69x -> 1.59x of its algorithmic bytes, outputs bit-identical to 2).  1, 2 and 3 also give every XCD whole
 *                     attention heads.
 *   "ln_rows"         1 = LayerNorm keeps several rows in flight per wave (default), 0 = one row per wave.
 *   "attn_fwd_waves"  0 = pick by grid size (default), 4 | 6 | 8 | 16 = waves per forward attention block (head dims <= 64).
 *   "attn_bwd_waves"  0 = pick by grid size (default), 4 | 8 = waves per backward attention block (head dims <= 64).
 *   "gn_blocks"       target number of GroupNorm row-chunk blocks in flight (default 512, >= 64).
 *   "epi_two_phase"   1 = the 8-wave GEMM tiles request every T row / residual chunk of a thread before using the first (default),
 *                     0 = one chunk at a time.
 *   "gn_unroll"       1 = the GroupNorm passes keep twice as many rows in flight per thread (8 forward, 4 backward); same bits.
 *   "lora_down_mode"  how clora_lora_down[_multi]_f16 spreads a job: 0 = four waves split K up to 4096 rows, one wave per 16 rows above;
 *                     1 = K-split at every size, sixteen waves per row group up to 1024 rows (default); 2 = as 0 with eight k-steps of
 *                     loads in flight above 4096 rows.
 *   "epi_hoist"       1 = the 8-wave GEMM tiles keep a thread's adapter up-matrix columns and bias in registers for the whole tile
 *                     (default), 0 = fetch them per output chunk like the 4-wave tiles do (a switch to take the hoisted epilogue out
 *                     of the path without a rebuild; launches with lora_dpack always hoist).
 *   "gn_resident"     1 = GroupNorm passes whose (batch element, channel slab) fits in a block's registers run as ONE launch (statistics
 *                     and apply from the same registers: the 8x8 / 16x16 / 32x32 feature maps of the UNet); 0 = always the two-launch
 *                     partial + apply scheme.  Frozen affine only; same arithmetic, a different (still fixed) summation order.
 *   "gn_team"         the one-launch GroupNorm of the large maps (clora_groupnorm_*_team, needs the caller's team state): 2 (default) =
 *                     wherever the team plan applies, forward at HW >= 1024, backward at HW >= 256; 1 = only where the two-launch
 *                     scheme would run otherwise (the one-block-per-slab kernels of "gn_resident" keep the 32x32 forward and the
 *                     16x16 maps); 3 / 4 = both directions down to HW >= 256 / 64 (A/B settings); 0 = never.  Shapes whose
 *                     one-block-per-slab plan may fold a deferred producer ("defer_max_rows") always keep that plan.  Same arithmetic,
 *                     a different (fixed) summation order.  Same-box A/B of the train step: 21.79 / 21.58 / 21.49 ms at 0 / 1 / 2.
 *   "defer_max_rows"  a deferred split-K GEMM (clora_deferred_t) is folded inside the one-launch GroupNorm kernels whose threads own at
 *                     most this many rows (default 4: the 8x8 / 16x16 maps, where it is faster than finish + plain); above it -- and in
 *                     the LayerNorm backward unless the value is 16 -- the library runs the plain finish pass first.  0 = never fold.
 *                     Bit-identical results at every setting.
 *   "wgrad_patch"     1 (default) = clora_conv_wgrad_f16 runs the large-map 3x3 convolutions with 32 or 64 input channels and at most 64
 *                     output channels (stride 1 pad 1, and stride 2 with the zero row / column at the bottom / right: the hint encoder's
 *                     512^2 / 256^2 / 128^2 stages; gather-ordered destination, oihw_ci == 0) on the patch-staged kernel: dY and the input
 *                     rows are staged once per output row instead of once per 64-column k tile; 0 = always the gather kernel; >= 64 = as 1
 *                     with that many blocks (A/B of the atomics volume).  fp32 atomics in both kernels: equal to ~1e-6 relative.
 *   "strip_blocks"    number of blocks the strip convolution kernel (tile_cfg 61, clora_conv_strip_eligible) aims at (default 512,
 *                     64 .. 16384): a block walks H * strips / blocks (at least 8) output rows of its column strip.
 * Unknown names / values: CLORA_ERR_ARG. */
int clora_set_option(const char* name, int value);

/* dW[N, K] += dY[M,N]^T . gather(X)[M,K] and (db != NULL) db[N] += column sums of dY  (fp32 atomics; caller
 * zeroes dW / db).
 * Weight gradient of the trainable hint-encoder convolutions (reference models.py:470,529,594-597,684:
 * autograd of F.conv2d).  `conv` as in the forward of that layer (NULL = 1x1 / linear). */
int clora_conv_wgrad_f16(const clora_half* dY, int ldy, const clora_half* X, int ldx, float* dW, float* db,
                         int M, int N, int K, const clora_conv_t* conv, int oihw_ci, void* stream);
/* oihw_ci > 0: dW is the parameter's own gradient, laid out [N][oihw_ci][ks][ks] like the Conv2d weight
 * (reference models.py:470); the kernel accumulates straight into it and drops the zero-padded input channels.
 * oihw_ci == 0: dW is [N, K] in the gather's (ky, kx, ci) column order. */

/* grad_w[co][ci][tap] += stage[co][tap*Cip + ci], grad_b += stage_b, and stage / stage_b are reset to zero: folds a
 * persistent gather-ordered wgrad staging buffer (clora_conv_wgrad_f16 with oihw_ci == 0) into the Conv2d parameter
 * gradients of the hint encoder (reference models.py:470,529,594-597) without per-step fill / permute / add launches.
 * (Direct OIHW atomics from the wgrad kernel put consecutive channels 36 bytes apart: measured 3x slower.) */
int clora_conv_wgrad_unpack_f32(float* stage, float* stage_b, float* grad_w, float* grad_b, int Co, int Ci, int ksize,
                                int Cip, void* stream);

/* fp32 master weight [Co][Ci][ks][ks] -> fp16 GEMM operands of the step: fwd [Co][ks*ks][Cip] and (dgrad != NULL)
 * dgrad [Cip][ks*ks][Cop], zero padded.  What autocast's per-step weight cast does around the hint encoder
 * (reference train...:683), fused with the layout change the implicit GEMM wants. */
int clora_conv_weight_pack_f32(const float* w, int Co, int Ci, int ksize, int Cip, int Cop, clora_half* fwd,
                               clora_half* dgrad, void* stream);

/* The two calls above for EVERY trainable convolution of the hint encoder in one launch each (reference models.py:684-808 builds
 * 18 Conv2d layers under configs/fill50k.json: until round 6 the step issued 18 pack and 14 unpack launches of 5-10 us).
 * `jobs` is a HOST array (copied into the kernel arguments, graph-capture safe), njobs <= CLORA_CONV_MAX_JOBS; per job the
 * arguments and results are exactly those of the single-job calls. */
#define CLORA_CONV_MAX_JOBS 32
typedef struct {
    const float* w;            /* fp32 master weight [Co][Ci][ks][ks] */
    clora_half* fwd;           /* [Co][ks*ks][Cip] */
    clora_half* dgrad;         /* [Cip][ks*ks][Cop] or NULL */
    int Co, Ci, ksize, Cip, Cop, pad_;
} clora_conv_pack_job_t;
int clora_conv_weight_pack_multi_f32(const clora_conv_pack_job_t* jobs, int njobs, void* stream);
typedef struct {
    float* stage;              /* [Co][ks*ks*Cip] gather-ordered staging, reset to zero */
    float* stage_b;            /* [Co] or NULL (together with grad_b) */
    float* grad_w;             /* [Co][Ci][ks][ks] += */
    float* grad_b;             /* [Co] += or NULL */
    int Co, Ci, ksize, Cip;
} clora_conv_unpack_job_t;
int clora_conv_wgrad_unpack_multi_f32(const clora_conv_unpack_job_t* jobs, int njobs, void* stream);

/* ---- attention core: O = softmax(Q K^T * scale) V per (batch, head), flash-style (never
 * materialises the [B*H, N, Nk] scores the reference builds at models.py:140-141, 270-271).
 * q: [B, Nq, H*D] with row stride ldq (elements), k/v: [B, Nk, H*D] strides ldk/ldv, o: ldo.
 * lse: [B, H, Nq] fp32 (natural-log sum-exp of scaled scores), needed by the backward. */
int clora_attn_fwd_f16(const clora_half* q, int ldq, const clora_half* k, int ldk, const clora_half* v, int ldv,
                       clora_half* o, int ldo, float* lse, int B, int H, int Nq, int Nk, int D, float scale,
                       void* stream);
/* Forward only, causal: key j is visible to query i iff j <= i (self-attention, Nq = Nk = N, head dims <= 64).  The masked
 * self-attention of the frozen CLIP text encoder (reference train_text_to_image_control_lora.py:768
 * `text_encoder(batch["input_ids"])[0]`; transformers CLIPTextTransformer builds the same mask additively). */
int clora_attn_fwd_causal_f16(const clora_half* q, int ldq, const clora_half* k, int ldk, const clora_half* v, int ldv,
                              clora_half* o, int ldo, int B, int H, int N, int D, float scale, void* stream);
/* dq/dk/dv given do (autograd of the same lines). delta: [B,H,Nq] fp32 scratch.  workspace (optional, one
 * 2*B*Nk*H*D*4-byte slab pair per query split, up to 16) lets the dK/dV kernel split its query loop when there are few keys
 * (cross-attention); the splits are folded in a fixed order (no atomics). */
int clora_attn_bwd_f16(const clora_half* q, int ldq, const clora_half* k, int ldk, const clora_half* v, int ldv,
                       const clora_half* o, int ldo, const clora_half* dO, int lddo, const float* lse,
                       float* delta, clora_half* dq, int lddq, clora_half* dk, int lddk, clora_half* dv, int lddv,
                       int B, int H, int Nq, int Nk, int D, float scale, void* workspace, size_t workspace_bytes,
                       void* stream);

/* ---- GroupNorm (+ optional SiLU), NHWC.  Replaces torch GroupNorm + F.silu pairs
 * (upstream ResnetBlock2D / Transformer2DModel.norm / conv_norm_out; reference models.py:515-516,537-543).
 * x,y: [B, HW, C]; gamma/beta fp32 [C]; stats: [B, G, 2] fp32 (mean, rstd) written by fwd. */
int clora_groupnorm_fwd_f16(const clora_half* x, clora_half* y, const float* gamma, const float* beta, float* stats,
                            int B, int HW, int C, int G, float eps, int fuse_silu, void* workspace,
                            size_t workspace_bytes, void* stream);
/* dx (and, when dgamma != NULL, dgamma/dbeta: deterministic two-stage reduction, no atomics; WRITTEN, or added to
 * when accumulate_params != 0 -- the trainable hint-encoder norms pass their .grad buffers). */
size_t clora_groupnorm_workspace_bytes(int B, int HW, int C, int G, int backward, int param_grads);
int clora_groupnorm_bwd_f16(const clora_half* x, const clora_half* dy, const clora_half* dres, clora_half* dx, const float* gamma,
                            const float* beta, const float* stats, float* dgamma, float* dbeta, int B, int HW, int C,
                            int G, int fuse_silu, int accumulate_params, void* workspace, size_t workspace_bytes,
                            void* stream);

/* Round 6 forms of the two calls above (they are `_ex` with the extra arguments NULL / 0):
 *  - x2 != NULL: the input is the channel CONCATENATION of x [B*HW, Ca] and x2 [B*HW, C - Ca] (upstream UNet up blocks:
 *    `torch.cat([hidden_states, res_hidden_states], dim=1)` feeding ResnetBlock2D.norm1) read in place; xcopy (optional,
 *    [B*HW, C]) receives the concatenated tensor for the block's 1x1 shortcut and for the backward.  Ca % 8 == 0.
 *  - src != NULL with src->splits > 0: the input is a deferred split-K GEMM (clora_deferred_t; src->N == C, src->M == B*HW,
 *    src->ldc == C): folded while loading where the one-launch plan applies, else finished first; either way src->C holds the
 *    finished tensor afterwards and x is ignored.
 *  backward: dy_src = deferred producer of dy (dy is then only a scratch buffer of the fallback); dx2 != NULL: dx is written as
 *  two tensors [B*HW, Ca] and [B*HW, C - Ca] (the gradients of the two concatenated inputs). */
int clora_groupnorm_fwd_f16_ex(const clora_half* x, const clora_half* x2, int Ca, const clora_deferred_t* src, clora_half* xcopy,
                               clora_half* y, const float* gamma, const float* beta, float* stats, int B, int HW, int C, int G,
                               float eps, int fuse_silu, void* workspace, size_t workspace_bytes, void* stream);
int clora_groupnorm_bwd_f16_ex(const clora_half* x, const clora_half* dy, const clora_deferred_t* dy_src, const clora_half* dres,
                               clora_half* dx, clora_half* dx2, int Ca, const float* gamma, const float* beta, const float* stats,
                               float* dgamma, float* dbeta, int B, int HW, int C, int G, int fuse_silu, int accumulate_params,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ABI 4 -- the same two calls with a TEAM STATE: a caller-owned device buffer of clora_groupnorm_team_state_bytes() bytes, zeroed ONCE
 * when it is allocated and then handed unchanged to every call on ONE stream (calls that may run concurrently need separate states).
 * With it the 64x64 / 32x32 maps of the UNet (HW >= 1024, frozen affine: dgamma == NULL; B * slabs dividing 256, see gn_team_plan)
 * run as ONE launch instead of statistics + apply: 256 workgroups, one per compute unit, each keeping its rows in registers; the
 * workgroups that share a (batch element, channel slab) exchange their partial group sums through the state inside the launch
 * (8-byte {epoch, value} words, write-through stores and L1-bypassing loads; no fences, nothing to zero between launches, safe
 * under hipGraph replay).  Shapes the plan does not take, a NULL / short state, a device with fewer than 256 compute units or
 * option "gn_team" = 0 fall through to the `_ex` behaviour.  Results are deterministic (members folded in a fixed order) and equal
 * the two-launch results up to the order of the fp32 sums.  The first 32-bit word of the state is a sticky error flag: non-zero
 * after a launch whose exchange gave up (a member was not resident within the bounded wait) -- its output is then invalid. */
size_t clora_groupnorm_team_state_bytes(void);
int clora_groupnorm_fwd_f16_team(const clora_half* x, const clora_half* x2, int Ca, const clora_deferred_t* src, clora_half* xcopy,
                                 clora_half* y, const float* gamma, const float* beta, float* stats, int B, int HW, int C, int G,
                                 float eps, int fuse_silu, void* team_state, size_t team_state_bytes, void* workspace,
                                 size_t workspace_bytes, void* stream);
int clora_groupnorm_bwd_f16_team(const clora_half* x, const clora_half* dy, const clora_deferred_t* dy_src, const clora_half* dres,
                                 clora_half* dx, clora_half* dx2, int Ca, const float* gamma, const float* beta, const float* stats,
                                 float* dgamma, float* dbeta, int B, int HW, int C, int G, int fuse_silu, int accumulate_params,
                                 void* team_state, size_t team_state_bytes, void* workspace, size_t workspace_bytes, void* stream);

/* ---- row softmax  y[r,:] = softmax(scale * x[r,:])  (fp32 max/sum; cols % 8 == 0, cols <= 8192, scale > 0; in place
 * allowed).  Normalises the materialised scores of the VAE's single-head d=512 attention (upstream AutoencoderKL
 * AttentionBlock, used at reference train_text_to_image_control_lora.py:753 / apps/gradio_canny2image.py). */
int clora_softmax_rows_f16(const clora_half* x, clora_half* y, int rows, int cols, int ld, float scale, void* stream);

/* ---- LayerNorm over the last dim (upstream BasicTransformerBlock.norm1/2/3, eps 1e-5). */
int clora_layernorm_fwd_f16(const clora_half* x, clora_half* y, const float* gamma, const float* beta, int M, int C,
                            float eps, void* stream);
int clora_layernorm_bwd_f16(const clora_half* x, const clora_half* dy, const clora_half* dres, clora_half* dx,
                            const float* gamma, int M, int C, float eps, void* stream);
/* dy_src != NULL with dy_src->splits > 0: dy is a deferred split-K GEMM (the dgrad of the projection that consumed LN(x)),
 * folded while loading (dy_src->N == C, dy_src->M == M); dy itself is not read. */
int clora_layernorm_bwd_f16_ex(const clora_half* x, const clora_half* dy, const clora_deferred_t* dy_src, const clora_half* dres,
                               clora_half* dx, const float* gamma, int M, int C, float eps, void* stream);
/* dres (both norm backwards, may be NULL): a second gradient of x -- the residual / shortcut branch that forked off
 * before the norm (x + attn(LN(x)), ResnetBlock2D's shortcut) -- added into dx, so autograd never issues the add. */

/* ---- GEGLU: y[m, j] = h[m, j] * gelu_erf(h[m, F + j])  (upstream FeedForward). */
int clora_geglu_fwd_f16(const clora_half* h, clora_half* y, int M, int F, void* stream);
int clora_geglu_bwd_f16(const clora_half* h, const clora_half* dy, clora_half* dh, int M, int F, void* stream);

/* ---- rank-r adapter pieces (upstream LoRALinearLayer; reference models.py:89-97,185-188,316-323).
 * T[m, toff+j] (+)= sum_k X[bmap(m), k] * D[j, k]   (fp32 math like the reference's x.float() @ down.T)
 * x_rows > 0 : X holds one batch element of x_rows rows that is broadcast over M (control batch 1).
 * d_kmajor  : D[j][k] is stored at D[k*ldd + j] (an up matrix [N, r] acting as U^T in the backward pass).
 * d_scale   : D is multiplied by d_scale on the fly (the LoRA `scale`). */
int clora_lora_down_f16(const clora_half* X, int ldx, const float* D, int ldd, float* T, int ldt, int toff,
                        int M, int K, int R, int accumulate, int x_rows, int d_kmajor, float d_scale, void* stream);
/* Several independent problems of the same kind in ONE launch (a v1 self-attention site has three adapter
 * down-projections in the forward pass and six weight-gradient reductions in the backward pass; each is
 * launch-latency bound on its own).  R <= 16 per job; wgrad jobs of one call share the rank class (<=4, <=8, <=16). */
#define CLORA_LORA_MAX_JOBS 16
#define CLORA_LORA_WGRAD_MAX_JOBS 32   /* clora_lora_wgrad_multi_f16 (round 6: the end-of-backward flush in half as many launch pairs) */
typedef struct {
    const clora_half* X; int ldx; const float* D; int ldd; float* T; int ldt; int toff;
    int M, K, R, accumulate, x_rows, d_kmajor; float d_scale;
    const clora_half* X2; int ldx2;   /* optional second input [M, K]: T = (X + X2) . D^T without forming the sum */
    int x2_rows;                      /* > 0: X2 has only x2_rows rows, row m reads X2[m % x2_rows] (control batch 1, quirk C6) */
    int r2;                           /* > 0: X2 only feeds rows 0..r2-1 of D -- several adapters that share X stacked into ONE job
                                       * (D = [D_q; D_k; D_v], one pass over X) while only the first of them also reads X2
                                       * (reference models.py:237-238: the control term enters the q adapter only); 0 = all R rows */
    const float* T_in; int ldt_in, t_in_rows, t_in_r;  /* optional: T_in[m % t_in_rows (or m), j] is added to output column j < t_in_r --
                                       * a share of the down-projection that was evaluated elsewhere (clora_rank_mix_f32: the control
                                       * term's part of the q adapter, in rank space) */
} clora_lora_down_job_t;
typedef struct {
    const clora_half* A; int lda; const float* T; int ldt; int toff; float* G; int gs_n, gs_j;
    int M, N, R; float scale; int a_rows;
    const clora_half* A2; int lda2;   /* optional: the reduction runs over fp16(A + A2) (an adapter fed by a sum of tensors) */
} clora_lora_wgrad_job_t;
int clora_lora_down_multi_f16(const clora_lora_down_job_t* jobs, int njobs, void* stream);
/* workspace >= sum over jobs of clora_lora_wgrad_workspace_bytes(M, N, R) */
int clora_lora_wgrad_multi_f16(const clora_lora_wgrad_job_t* jobs, int njobs, void* workspace, size_t workspace_bytes,
                               void* stream);

/* ---- the v1 control term in RANK SPACE (round 4).  Reference models.py:214-218, 237-238: every plain ControlLoRA site adds
 * c = scale * to_control(control) = scale * U_c (D_c control) to the hidden states that feed its q adapter, q += scale * U_q D_q (h + c).
 * By linearity D_q (h + c) = D_q h + (scale * D_q U_c) (D_c control): the [M, C] tensor c never has to exist -- its share of the
 * q adapter's down-projection is a 4 x r_c matrix M_l = scale * D_q U_c applied to the r_c numbers per row that D_c control yields,
 * and the whole backward of the term (d control, dU_c, dD_c and the control share of dD_q) runs on [M, 4] / [M, r_c] tensors and
 * 4 x r_c Gram matrices.  (fp32 throughout: the reference's two fp16 roundings of c are NOT reproduced -- the result is closer
 * to the reference's fp32 arithmetic than its own fp16 run; parity limits unchanged.)  One site = one clora_rank_site_t; the
 * sites of a UNet level (10) go in one launch each (n <= 16; the array is read on the host).
 *   clora_rank_compose_f32      M_l[j, i] = scale * sum_c D_q[j, c] U_c[c, i]                       (4 x r_c, r_c <= 8)
 *   clora_rank_mix_f32  fwd     Tq[m, j]  = sum_i Tc[m, toff + i] M_l[j, i]                          (rows x 4)
 *                       bwd     dTc[m, toff + i] = sum_j dTq[m, j] M_l[j, i]   and per-block partial Gram sums of
 *                               G_l[j, i] = sum_m dTq[m, j] Tc[m, toff + i]  into `gram_ws` (clora_rank_gram_ws_bytes)
 *   clora_rank_compose_bwd_f32  folds the partial Gram sums in a fixed order (deterministic) and accumulates
 *                               gUc[c, i] += scale * sum_j D_q[j, c] G[j, i]   gDq[j, c] += scale * sum_i G[j, i] U_c[c, i] */
typedef struct {
    const float* Dq; int lddq;        /* to_q_lora.down.weight [4, C] */
    const float* Uc; int lduc;        /* to_control.up.weight   [C, r_c] */
    float* M;                         /* [4 * r_c] work buffer (device) */
    const float* Tc; int ldtc, toff;  /* D_c control: [rows, ldtc], this site's r_c columns start at toff */
    float* Tq; int ldtq;              /* fwd: out [rows, >= 4]; bwd: the incoming gradient dTq (read) */
    float* dTc; int lddtc;            /* bwd: out, same layout as Tc */
    float* gDq; float* gUc;           /* bwd: parameter gradients to accumulate into (either may be NULL) */
    int rows, C, rc;
    float scale;
} clora_rank_site_t;
int clora_rank_compose_f32(const clora_rank_site_t* sites, int n, void* stream);
int clora_rank_mix_f32(const clora_rank_site_t* sites, int n, int backward, float* gram_ws, size_t gram_ws_bytes, void* stream);
size_t clora_rank_gram_ws_bytes(int rows, int n);
int clora_rank_compose_bwd_f32(const clora_rank_site_t* sites, int n, const float* gram_ws, void* stream);

/* base != NULL: Y[m,n] = fp16(base[m,n] + scale * sum_j T[m,toff+j] U[n,j])  -- the explicit "hidden + to_control(control)" of
 * models.py:214-218,237-238 and the V2 pre/post adds (:369,:415), the sum formed in fp32 and rounded once (round 6; until then the
 * update was rounded twice first, the reference's fp16 arithmetic);  base == NULL: Y = fp16(scale * fp16(sum_j ...)), the term alone. */
int clora_lora_up_f16(const clora_half* base, int ldb, const float* T, int ldt, int toff, const float* U, int ldu,
                      int u_transposed, clora_half* Y, int ldy, int M, int N, int R, float scale, void* stream);
/* several of them in one launch: the control terms of the 10 attention sites of a UNet level share one hint-encoder
 * feature map, so ControlLoRA.forward (models.py:810-835) evaluates them together right after the hint encoder */
typedef struct {
    const clora_half* base; int ldb; const float* T; int ldt; int toff; const float* U; int ldu; int u_transposed;
    clora_half* Y; int ldy; int M, N, R; float scale;
} clora_lora_up_job_t;
int clora_lora_up_multi_f16(const clora_lora_up_job_t* jobs, int njobs, void* stream);
/* G[n*gs_n + j*gs_j] += scale * sum_m A[m, n] * T[m, toff+j]  (adapter weight gradients; two-stage
 * deterministic reduction through `workspace` of clora_lora_wgrad_workspace_bytes(M, N, R) bytes, no atomics). */
size_t clora_lora_wgrad_workspace_bytes(int M, int N, int R);
int clora_lora_wgrad_f16(const clora_half* A, int lda, const float* T, int ldt, int toff, float* G, int gs_n, int gs_j,
                         int M, int N, int R, float scale, int a_rows, void* workspace, size_t workspace_bytes,
                         void* stream);

/* ---- data-parallel exchange: ONE in-place all-reduce (sum) of the flat fp32 adapter-gradient buffer per optimizer step, RCCL over
 * xGMI, one process per GPU.  Replaces the gradient all-reduce of accelerate's DDP wrapper around the prepared `control_lora`
 * (reference train_text_to_image_control_lora.py:683-685 `accelerator.prepare`, :790 `accelerator.backward`); the 1/N of the
 * mean is folded into clora_optim_prep_f32.  The communicator is a per-process handle kept by the library (librccl is opened
 * lazily with dlopen: no link-time dependency).
 *   clora_comm_unique_id   rank 0 only: 128 opaque bytes; the host hands them to every rank (any transport)
 *   clora_comm_init        every rank, collectively, after hipSetDevice: creates the communicator (once per process)
 *   clora_comm_world       number of ranks of the live communicator, 0 when there is none
 *   clora_comm_rank        this process's rank in it, -1 when there is none (a host re-using the process for another group
 *                          checks both before deciding to keep the communicator)
 *   clora_allreduce_flat_f32  buf[0..n) := sum over ranks, enqueued on `stream` (asynchronous like every other entry point)
 *   clora_comm_destroy     releases the communicator */
int clora_comm_unique_id(void* id128);
int clora_comm_init(const void* id128, int rank, int world);
int clora_comm_library(char* path, size_t n);   /* file that defines the ncclAllReduce the library is bound to (dladdr); loads librccl */
int clora_comm_world(void);
int clora_comm_rank(void);
int clora_allreduce_flat_f32(float* buf, size_t n, void* stream);
int clora_comm_destroy(void);

/* ---- small elementwise / data-movement kernels on the path */
int clora_add_f16(const clora_half* a, const clora_half* b, clora_half* y, size_t n, void* stream);
int clora_silu_f16(const clora_half* x, clora_half* y, size_t n, void* stream);
/* Sinusoidal timestep embedding of the UNet (upstream `Timesteps(320, flip_sin_to_cos=True, freq_shift=0)`; what the
 * reference's `unet(noisy_latents, timesteps, ...)` at train_text_to_image_control_lora.py:782 evaluates first): out[b, j] = cos(t_b f_j),
 * out[b, half + j] = sin(t_b f_j), fp32 math, fp16 result -- ONE launch for upstream's arange / exp / mul / cos / sin / cat / cast.
 * t: int64 (t_is_i64) or fp32, t_count = batch values or 1 broadcast value; freq = the `half` frequencies exp(-ln(10000) j / half). */
int clora_timestep_embedding_f16(const void* t, int t_is_i64, int t_count, const float* freq, clora_half* out, int batch, int half,
                                 void* stream);
int clora_silu_bwd_f16(const clora_half* x, const clora_half* dy, clora_half* dx, size_t n, void* stream);
/* y = x * sigmoid(1.702 x): the activation of the CLIP text encoder's MLP (transformers `quick_gelu`; the frozen text
 * encoder the reference calls at train_text_to_image_control_lora.py:768). */
int clora_quick_gelu_f16(const clora_half* x, clora_half* y, size_t n, void* stream);
/* dst[m, 0:N] = src[m, 0:N] with independent row strides (skip-connection concat = two copies,
 * its backward = two slice copies; upstream torch.cat in the up blocks) */
int clora_copy2d_f16(const clora_half* src, int lds, clora_half* dst, int ldd, size_t M, int N, void* stream);
/* dx[b, y, x, c] = sum of the 2x2 block of dy[b, 2y.., 2x.., c]  (backward of nearest x2 upsample) */
int clora_pool2x2_sum_f16(const clora_half* dy, clora_half* dx, int B, int H, int W, int C, void* stream);
/* out[n] += sum_m A[m, n]  (bias gradients of trainable convs; fp32 atomics) */
int clora_colsum_f16(const clora_half* A, int lda, float* out, int M, int N, void* stream);
/* loss_sum[0] += sum (pred - target)^2 ; dpred = grad_scale * (loss_scale ? loss_scale[0] : 1) * (pred - target)
 * (F.mse_loss in fp32 + the GradScaler-scaled backward seed, train...:783,790; loss_scale is device resident) */
int clora_mse_f16(const clora_half* pred, const clora_half* target, float* loss_sum, clora_half* dpred, size_t n,
                  float grad_scale, const float* loss_scale, void* stream);
int clora_cast_f32_to_f16(const float* x, clora_half* y, size_t n, void* stream);
int clora_cast_f16_to_f32(const clora_half* x, float* y, size_t n, void* stream);

/* ---- fused optimizer over ONE flat fp32 buffer (train...:790-796: GradScaler unscale, clip_grad_norm_(1.0),
 * AdamW step, skipped on inf/nan, loss-scale growth/backoff) -- entirely device resident, no host sync.
 * state (device fp32[16]): [0] sum of squares of the (still scaled) grads  [1] non-finite count
 *   [2] optimizer step count  [3] loss scale  [4] growth tracker  [5] grad multiplier = clip/scale (out)
 *   [6] skip flag (out)  [7] 1-beta1^t  [8] 1-beta2^t  [9] unscaled grad norm (out, -1 when skipped)
 *   [10] learning-rate multiplier written by the host LR schedule (lr_scheduler, train...:660-665); 0 = unset = 1
 *   [11] gradient divisor = data-parallel world size: the flat buffer holds the all-reduce SUM over ranks and the mean
 *        (train...:790 DDP semantics) is taken here, folded into the unscale factor; 0 = unset = 1 */
int clora_grad_sumsq_f32(const float* g, size_t n, float* state, void* stream);
int clora_optim_prep_f32(float* state, float max_norm, float beta1, float beta2, int dynamic_scale,
                         float growth_factor, float backoff_factor, int growth_interval, void* stream);
int clora_adamw_flat_f32(float* p, const float* g, float* m, float* v, size_t n, const float* state, float lr,
                         float beta1, float beta2, float eps, float weight_decay, void* stream);

/* Box calibration (bench.py "calibration"): `blocks` workgroups of 4 waves issue iters x 8 independent dense MFMAs each on
 * pseudo-random operands; out[3b .. 3b+2] = {shader cycles, 100 MHz wall ticks, checksum} of block b.  No reference
 * counterpart: it exists so that a bench line can be normalised by the clock the box actually held (DVFS). */
int clora_clock_probe(unsigned long long* out, int blocks, int iters, void* stream);

/* library info: clora_abi_version() changes whenever a struct of this header changes layout (2: round 4's clora_epilogue_t /
 * clora_lora_down_job_t fields; 3: round 6's clora_epilogue_t.defer); a host built against another version must refuse the library */
#define CLORA_ABI_VERSION 4
int clora_abi_version(void);
const char* clora_build_info(void);

#ifdef __cplusplus
}
#endif
#endif /* CLORA_H */
