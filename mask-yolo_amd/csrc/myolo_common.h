// Internal helpers shared by the kernel translation units of libmyolo_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/myolo_hip_internal.h"

extern "C" void myolo_set_error(const char* fmt, ...);

// Process-wide tuning switches, changed ONLY through myolo_set_option() (include/myolo_hip.h): the launch path reads
// plain ints, never the environment.  Tuning switches default to 0 = the shipped behaviour (tn_wgs: 224, see below).
struct MyoloOptions {
    int no_nt;            // gemm: never use streaming (non-temporal) stores for large outputs
    int gemm_generic;     // gemm: force the generic (guarded) kernels
    int no_splitk;        // gemm: never split K for under-filled grids
    int gemm_w256;        // gemm: 128x256 block tiles for big launches
    int wino_nt;          // winograd multiply: streaming stores of the product
    int wino_w256;        // winograd multiply: 128x256 block tiles
    int bf16_regstage;    // bf16 gemm: register-staged variant instead of LDS-DMA
    int bf16_no256;       // bf16 gemm: never the 256x256-tile kernel
    int bf16_no_loopn;    // bf16 deconv+mask: a workgroup per (row tile, tap) instead of one per row tile walking all four taps (ablation)
    int bf16_mask_valu;   // bf16 deconv+mask, 256-row kernels: 1 = the 1x1 mask conv on the VALU from the fp32 deconv output (rounds 3-5) instead of on the matrix pipe from its bf16 rounding
    int bf16_mask_nofin;  // bf16 deconv+mask, all-taps kernel at 256 channels: 1 = partial logits per 64-channel slab + the finish launch (ablation / test reference)
    int deconv_mask_legacy; // fp32 (bf16x6) deconv + mask forward: 1 = the untransposed tile with the per-class butterfly epilogue (rounds 3-5), 2 = the transposed tile with partial logits + the finish launch; 0 = transposed, sigmoid stored by the kernel at 256 channels
    int bf16_no_c3;       // bf16 3x3 conv: the nine-fetch implicit GEMM instead of the LDS-resident activation block (ablation)
    int bf16_force256;    // bf16 gemm: always the 256x256-tile kernel
    int crop_bf16_legacy; // bf16 ROIAlign forward: 1 = four corner loads per output element (rounds 2-5) instead of the column walk; the same bits (test reference)
    int crop_bwd_nolds;   // ROIAlign backward: per-box terms recomputed per thread instead of staged in LDS
    int tune0;            // scratch integer for kernel-tuning experiments (0 = off); never set by the product
    int dw_min_wg;        // depthwise forward (row-sliding kernel): workgroups wanted before rows stop being split into chunks (0 = default 1024)
    int dw_rows1;         // depthwise forward (round-3 kernel): one output row per thread (no vertical strip)
    int dw_bwd_legacy;    // depthwise data gradient, stride 1: the round-3 gather kernel instead of the row-sliding one (ablation)
    int dw_legacy;        // depthwise forward: the round-3 register-tiled kernel also where the row-sliding LDS-staged one applies (ablation)
    int wino_x6;          // winograd multiply on the bf16 matrix pipe: 6 piece products per fp32 product, fp32 accumulation (csrc/wino_mm.hip)
    int wino_no_bt;       // winograd multiply: gemm_nn_fast on [K][N] filters instead of wino_mm_kernel on transposed ones
    int wino_no_mixed;    // winograd: F(4,3) for every tile (no F(2,3) on the ragged last tile row / column)
    int x6_no_half_tiles; // bf16x6 plain products: 1 = keep 128 x 256 tiles when they do not fill the chip (default: 128 x 128 tiles then)
    int w63_order;        // wino63 boundary kernels: 1 = the previous workgroup order (all images of channel slice 0, then slice 1, ...)
    int w63_wgs;          // wino63 boundary kernels: persistent workgroups per CU (0 = default 1)
    int w63_legacy;       // wino63 boundary kernels: 1 = the round-5 kernel (scalar transforms); the packed form gives the same bits (test reference)
    int pw_x6_min_rows;   // pointwise convs: fewest rows for the bf16x6 kernels (0 = default 4096)
    int deconv_no_x6;     // deconv forward / data gradient: the fp32-MFMA kernels even when "wino_x6" is on (ablation)
    int dw_wgrad_generic; // depthwise weight gradient: the generic 9-accumulator column reduction instead of the tiled kernel (ablation)
    int pw_skinny_nw4;    // conv_23 (pw_skinny_fwd_kernel): four waves per workgroup also from K = 512 up (rounds 2-5; ablation -- another summation order)
    int pw_no_smallm;     // pointwise convs with few rows and K >= 256: the split-K pair of gemm_nn_fast launches instead of pw_smallm_kernel (ablation)
    int pw_no_x6;         // pointwise convs with >= 256 channels: the fp32-MFMA kernels even when "wino_x6" is on (ablation)
    int tn_no_x6;         // winograd weight gradient: gemm_tn_fast (fp32 MFMA) even when "wino_x6" is on (ablation of wino_tn_x6_kernel)
    int tn_wgs;           // wino_tn_x6_kernel: its work units go out in launches of at most this many workgroups (default 224; 0 = one launch).  112 KB of LDS = one workgroup per CU, so 224 leave four CUs of every XCD to the chains of small kernels that run beside a weight gradient (profiles/r4_notes.md section 8)
    int no_trunk_fusion;  // *_bnstats_fwd: the conv, then a separate statistics pass (ablation of the producer-fused BatchNorm statistics)
    // NOT a tuning switch -- which Keras/TF pair the BatchNorm moving-variance update restates (default 1):
    // 1 = Keras 2.2.x on TF-1.x through tf.nn.fused_batch_norm (Bessel-corrected batch variance, then Keras' n/(n-(1+eps)));
    // 0 = Keras' factor on the biased variance (non-fused backend path).
    int bn_fused_tf_variance;
};
extern MyoloOptions g_myolo_opt;

#define MYOLO_REQUIRE(cond, ...)                                   \
    do {                                                           \
        if (!(cond)) {                                             \
            myolo_set_error(__VA_ARGS__);                          \
            return MYOLO_EINVAL;                                   \
        }                                                          \
    } while (0)

#define MYOLO_NEED_WS(bytes)                                                         \
    do {                                                                             \
        if ((size_t)(bytes) > ws_bytes || (ws == nullptr && (bytes) > 0)) {          \
            myolo_set_error("%s: workspace too small (%zu needed, %zu given)",       \
                            __func__, (size_t)(bytes), ws_bytes);                    \
            return MYOLO_EWORKSPACE;                                                 \
        }                                                                            \
    } while (0)

#define MYOLO_CHECK_LAUNCH()                                                         \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            myolo_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return MYOLO_ELAUNCH;                                                    \
        }                                                                            \
    } while (0)

// ---------------------------------------------------------------------------------------
// Prepared-weights registry (csrc/mem_kernels.hip).  Many entry points start by re-laying a WEIGHT tensor out for their kernel (a transpose, the
// three-bf16-piece split of the bf16x6 products, a Winograd filter transform): 30-odd small launches per training step that depend on nothing
// but the weights, all of them on the step's critical chain.  With a registry active (myolo_wprep_activate) such a site asks
// myolo_wprep_resolve(): a known (weights, kind, dims) entry that was refreshed for the current weight generation returns its slot in the caller's
// arena -- no launch; an unknown one is recorded (slot reserved) and prepared into the fallback scratch exactly as without a registry.  The owner
// re-runs every recorded preparation with myolo_wprep_refresh() on a stream of its choice (the training step: a side stream at the step's start,
// under the first layers) and orders its consumers behind that; myolo_wprep_invalidate() (the weights changed) makes every entry miss again.
// One launch thread per process; no registry active = the previous behaviour, bit for bit (the prepared bytes are the same either way).
// ---------------------------------------------------------------------------------------
#include <functional>
enum { WP_TRANSPOSE = 1, WP_X6_SPLIT = 2, WP_WINO63_U = 3, WP_WINO43_U = 4 };
const void* myolo_wprep_resolve(const void* w, int kind, long long d0, long long d1, long long d2, size_t bytes, void* fallback, hipStream_t s,
                                const std::function<void(void*, hipStream_t)>& run);

// csrc/wino_mm.hip: several bf16x6 weight splits in one launch (used by myolo_wprep_refresh for runs of WP_X6_SPLIT entries)
int myolo_x6_split_batched(int n, const void* const* src, void* const* dst, const long long* K, const long long* N, const long long* kn, hipStream_t s);

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

#define BN_EPS_F 1e-3f
#define BN_MOMENTUM_F 0.99f

// internal (C++ linkage) batched GEMM launchers of gemm_kernels.hip, used by wino_kernels.hip
int myolo_gemm_nn_batched(const float* A, const float* B, float* C, long long M, int K, int N, int batch, hipStream_t s);
// csrc/wino_mm.hip: the same product with the second operand stored transposed ([N][K]); _ok says whether (K, N) qualifies
bool myolo_gemm_nt_batched_ok(int K, int N);
bool myolo_gemm_nt_batched_x6(int K, int N);
bool myolo_deconv_mask_mm_ok(int Cin, int Cout);
size_t myolo_deconv_mask_mm_split_bytes(int Cin, int Cout);
int myolo_deconv_mask_mm(const float* x, const float* w, const float* bias, const float* w2, float* part, void* split,
                         long long M, int H, int W, int Cin, int Cout, int ncls, hipStream_t s,
                         const int32_t* keep_inv = nullptr, float* keep_d = nullptr, int keep_cap = 0,
                         const float* b2 = nullptr, float* p_out = nullptr, int* finished = nullptr);
int myolo_gemm_nt_batched_runs(const float* A, const float* Bt, float* C, int nruns, const long long* rows, const long long* a_off,
                               const long long* b_off, const long long* c_off, const int* nq, int K, int N, hipStream_t s);
// csrc/wino_mm.hip: C[z] = A[z]^T B[z] with six exact bf16 piece products per fp32 product (the Winograd weight gradient under "wino_x6")
bool myolo_gemm_tn_x6_ok(int Ka, int N);
size_t myolo_gemm_tn_x6_ws_bytes(int nruns, const long long* rows, const int* nq, int Ka, int N);
int myolo_gemm_tn_x6_runs(const float* A, const float* B, float* C, int nruns, const long long* rows, const long long* a_off, const long long* b_off,
                          const int* nq, int Ka, int N, void* part, size_t part_bytes, hipStream_t s,
                          const float* a_scale = nullptr, const float* a_shift = nullptr, int a_act = 0);
// Conv2DTranspose 2x2/s2 forward (which = 0) / data gradient (which = 1) under "wino_x6" (csrc/wino_mm.hip)
bool myolo_deconv_x6_ok(int Cin, int Co, int which);
int myolo_deconv_x6_fwd(const float* x, const float* w, const float* bias, float* y, long long M, int H, int W, int Cin, int Co, int act,
                        void* split, hipStream_t s);
int myolo_deconv_x6_bwd_data(const float* dy, const float* w, float* dx, long long M, int H, int W, int Cin, int Co, void* split, hipStream_t s);
size_t myolo_deconv_x6_bwd_weight_ws_bytes(long long M, int Cin, int Co);
int myolo_deconv_x6_bwd_weight(const float* x, const float* dy, float* dw, long long M, int H, int W, int Cin, int Co, void* part, size_t part_bytes,
                               hipStream_t s);
int myolo_matmul_f32_impl(const float* A, const float* B, float* C, int64_t M, int K, int N, int b_is_nk, int products,
                          void* ws, size_t ws_bytes, void* stream, bool b_is_weight);
// pointwise convs with >= 256 channels under "wino_x6" (csrc/wino_mm.hip)
bool myolo_pw_x6_ok(int K, int N);
size_t myolo_pw_x6_split_bytes(int K, int N);
int myolo_pw_x6_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w, float* y, double* stat,
                    long long M, int K, int N, void* split, hipStream_t s);
size_t myolo_gemm_tn_batched_ws_bytes(long long M, int Ka, int N, int batch);
int myolo_gemm_tn_batched(const float* A, const float* B, float* C, long long M, int Ka, int N, int batch, void* ws, size_t ws_bytes,
                          hipStream_t s);
// p[pix][c] = sigmoid(b2[c] + sum_k part[k][pix][c]) over the column slabs of a fused deconv + 1x1 epilogue (gemm_kernels.hip)
void myolo_launch_deconv_mask_finish(const float* part, const float* b2, float* out, long long npix, int ncls, int nslabs, hipStream_t s);
// BN batch statistics (mean / var / folded scale, shift / moving averages) from [nblk][2*C] double partial sums (mem_kernels.hip)
void myolo_bn_stats_from_partials(const double* part, double* tot, int nblk, int C, double M, const float* gamma, const float* beta,
                                  float* mean, float* var, float* scale, float* shift, float* mmean, float* mvar, hipStream_t s);
