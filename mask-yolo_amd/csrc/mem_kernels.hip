// HBM-bound kernels of the Mask-YOLO hot path for gfx950: conv1 (Cin=3), depthwise 3x3,
// BatchNorm statistics / apply / backward, ROIAlign (crop_and_resize) gather + scatter-add,
// the final mask 1x1+sigmoid, mask BCE, Adam, small helpers.
//
// Layout rule everywhere: NHWC fp32, one lane owns 4 consecutive channels (16-byte loads/stores),
// consecutive lanes own consecutive channel quads, so every wave-level access is a run of
// contiguous 16 B segments.  Column reductions (BN statistics, bias / weight gradients) use
// one shape: a block owns a slab of rows, each thread accumulates its rows in registers,
// a shared-memory tree combines the row lanes, per-block partials go to the caller's workspace
// in double and a second tiny kernel finishes -- deterministic, no atomics.
#include "myolo_common.h"
#include <type_traits>
#include <stdlib.h>
#include <stdarg.h>
#include <string.h>

// ---------------------------------------------------------------------------------------
// error string (per-thread)
// ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
extern "C" void myolo_set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* myolo_last_error_string(void) { return g_err; }
extern "C" int myolo_version(void) { return 211; }

// ---------------------------------------------------------------------------------------
// tuning switches (myolo_set_option): plain process-wide ints, no environment reads anywhere
// ---------------------------------------------------------------------------------------
static MyoloOptions default_options() { MyoloOptions o = {}; o.bn_fused_tf_variance = 1; o.tn_wgs = 224; return o; }
MyoloOptions g_myolo_opt = default_options();
static int* option_slot(const char* name)
{
    static const struct { const char* n; int MyoloOptions::*m; } tab[] = {
        {"no_nt", &MyoloOptions::no_nt}, {"gemm_generic", &MyoloOptions::gemm_generic}, {"no_splitk", &MyoloOptions::no_splitk},
        {"gemm_w256", &MyoloOptions::gemm_w256}, {"wino_nt", &MyoloOptions::wino_nt}, {"wino_w256", &MyoloOptions::wino_w256},
        {"bf16_regstage", &MyoloOptions::bf16_regstage}, {"bf16_no256", &MyoloOptions::bf16_no256}, {"bf16_no_c3", &MyoloOptions::bf16_no_c3}, {"bf16_no_loopn", &MyoloOptions::bf16_no_loopn}, {"bf16_mask_valu", &MyoloOptions::bf16_mask_valu}, {"deconv_mask_legacy", &MyoloOptions::deconv_mask_legacy}, {"bf16_mask_nofin", &MyoloOptions::bf16_mask_nofin},
        {"bf16_force256", &MyoloOptions::bf16_force256}, {"crop_bwd_nolds", &MyoloOptions::crop_bwd_nolds}, {"crop_bf16_legacy", &MyoloOptions::crop_bf16_legacy},
        {"tune0", &MyoloOptions::tune0}, {"dw_rows1", &MyoloOptions::dw_rows1}, {"dw_legacy", &MyoloOptions::dw_legacy}, {"dw_bwd_legacy", &MyoloOptions::dw_bwd_legacy}, {"dw_min_wg", &MyoloOptions::dw_min_wg}, {"wino_no_mixed", &MyoloOptions::wino_no_mixed}, {"no_trunk_fusion", &MyoloOptions::no_trunk_fusion}, {"tn_no_x6", &MyoloOptions::tn_no_x6}, {"tn_wgs", &MyoloOptions::tn_wgs}, {"pw_no_x6", &MyoloOptions::pw_no_x6}, {"pw_no_smallm", &MyoloOptions::pw_no_smallm}, {"pw_skinny_nw4", &MyoloOptions::pw_skinny_nw4}, {"dw_wgrad_generic", &MyoloOptions::dw_wgrad_generic}, {"deconv_no_x6", &MyoloOptions::deconv_no_x6}, {"pw_x6_min_rows", &MyoloOptions::pw_x6_min_rows}, {"w63_order", &MyoloOptions::w63_order}, {"w63_legacy", &MyoloOptions::w63_legacy}, {"w63_wgs", &MyoloOptions::w63_wgs}, {"x6_no_half_tiles", &MyoloOptions::x6_no_half_tiles}, {"wino_no_bt", &MyoloOptions::wino_no_bt}, {"wino_x6", &MyoloOptions::wino_x6}, {"bn_fused_tf_variance", &MyoloOptions::bn_fused_tf_variance},
    };
    if (!name) return nullptr;
    for (const auto& e : tab)
        if (strcmp(e.n, name) == 0) return &(g_myolo_opt.*(e.m));
    return nullptr;
}
extern "C" int myolo_set_option(const char* name, int value)
{
    int* s = option_slot(name);
    MYOLO_REQUIRE(s, "set_option: unknown option '%s'", name ? name : "(null)");
    *s = value;
    return MYOLO_OK;
}
extern "C" int myolo_get_option(const char* name, int* value)
{
    int* s = option_slot(name);
    MYOLO_REQUIRE(s && value, "get_option: unknown option '%s'", name ? name : "(null)");
    *value = *s;
    return MYOLO_OK;
}

// ---------------------------------------------------------------------------------------
// prepared-weights registry (see myolo_common.h)
// ---------------------------------------------------------------------------------------
#include <vector>
struct WPrepEntry {
    const void* w; int kind; long long d0, d1, d2;
    size_t off, bytes;
    unsigned long long gen;            // weight generation its slot was refreshed for (0: never)
    unsigned long long last_use;       // generation of the last resolve() that asked for it
    std::function<void(void*, hipStream_t)> run;
};
struct WPrep {
    char* arena; size_t cap, used;
    unsigned long long gen;
    std::vector<WPrepEntry> e;
    long long hits, misses;
    long long overflows;               // resolve() calls of a NEW site that found no room in the arena (it stays unrecorded and is made in place every step)
};
static WPrep* g_wprep = nullptr;       // the active registry (one launch thread)

const void* myolo_wprep_resolve(const void* w, int kind, long long d0, long long d1, long long d2, size_t bytes, void* fallback, hipStream_t s,
                                const std::function<void(void*, hipStream_t)>& run)
{
    WPrep* r = g_wprep;
    if (r) {
        for (auto& en : r->e)
            if (en.w == w && en.kind == kind && en.d0 == d0 && en.d1 == d1 && en.d2 == d2 && en.bytes == bytes) {
                en.last_use = r->gen;
                if (en.gen == r->gen) { ++r->hits; return r->arena + en.off; }
                ++r->misses;
                run(fallback, s);
                return fallback;
            }
        const size_t need = align256(bytes);
        if (r->used + need <= r->cap) {            // new site: reserve a slot, prepared by the owner's next refresh
            r->e.push_back(WPrepEntry{w, kind, d0, d1, d2, r->used, bytes, 0ull, r->gen, run});
            r->used += need;
        } else
            ++r->overflows;
        ++r->misses;
    }
    run(fallback, s);
    return fallback;
}

extern "C" int myolo_wprep_create(void* arena, size_t arena_bytes, void** handle)
{
    MYOLO_REQUIRE(handle && arena && ((uintptr_t)arena & 255) == 0, "wprep_create: needs a 256-byte aligned device arena and a handle slot");
    WPrep* r = new WPrep();
    r->arena = (char*)arena; r->cap = arena_bytes; r->used = 0; r->gen = 1; r->hits = r->misses = r->overflows = 0;
    *handle = r;
    return MYOLO_OK;
}
extern "C" int myolo_wprep_destroy(void* h)
{
    if (g_wprep == (WPrep*)h) g_wprep = nullptr;
    delete (WPrep*)h;
    return MYOLO_OK;
}
extern "C" int myolo_wprep_activate(void* h) { g_wprep = (WPrep*)h; return MYOLO_OK; }
extern "C" int myolo_wprep_count(void* h) { return h ? (int)((WPrep*)h)->e.size() : 0; }
extern "C" int myolo_wprep_invalidate(void* h) { if (h) ++((WPrep*)h)->gen; return MYOLO_OK; }
extern "C" int myolo_wprep_stats(void* h, long long* hits, long long* misses, long long* bytes_used)
{
    MYOLO_REQUIRE(h, "wprep_stats: null registry");
    WPrep* r = (WPrep*)h;
    if (hits) *hits = r->hits;
    if (misses) *misses = r->misses;
    if (bytes_used) *bytes_used = (long long)r->used;
    return MYOLO_OK;
}
extern "C" int myolo_wprep_overflows(void* h, long long* n)
{
    MYOLO_REQUIRE(h && n, "wprep_overflows: null registry or result slot");
    *n = ((WPrep*)h)->overflows;
    return MYOLO_OK;
}
/* re-run the recorded preparations [first, last) into their slots on `stream` and mark them valid for the current weight generation; entries no
 * resolve() has asked for during the last `max_idle` generations are skipped (they miss, and are prepared in place, when they come back).
 * Returns the number of preparations launched (< 0: error). */
extern "C" int myolo_wprep_refresh(void* h, int first, int last, int max_idle, void* stream)
{
    if (!h) { myolo_set_error("wprep_refresh: null registry"); return -1; }
    WPrep* r = (WPrep*)h;
    if (last > (int)r->e.size()) last = (int)r->e.size();
    int n = 0;
    // the bf16x6 weight splits of the range go out in ONE launch (round 5: ~17 launches of a few microseconds each per training step before); every
    // other kind through its recorded closure.  Same bytes either way; the order between entries does not matter (distinct slots, read-only sources).
    std::vector<const void*> bs;
    std::vector<void*> bd;
    std::vector<long long> bk, bn_, bkn;
    for (int i = first < 0 ? 0 : first; i < last; ++i) {
        WPrepEntry& en = r->e[i];
        if (max_idle > 0 && r->gen - en.last_use > (unsigned long long)max_idle) continue;
        if (en.kind == WP_X6_SPLIT && !(g_myolo_opt.tune0 & 2097152)) {
            bs.push_back(en.w); bd.push_back(r->arena + en.off); bk.push_back(en.d0); bn_.push_back(en.d1); bkn.push_back(en.d2);
        } else
            en.run(r->arena + en.off, (hipStream_t)stream);
        en.gen = r->gen;
        ++n;
    }
    if (!bs.empty()) myolo_x6_split_batched((int)bs.size(), bs.data(), bd.data(), bk.data(), bn_.data(), bkn.data(), (hipStream_t)stream);
    if (hipGetLastError() != hipSuccess) { myolo_set_error("wprep_refresh: a launch failed"); return -1; }
    return n;
}

__device__ __forceinline__ float4 ld4g(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4g(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
typedef float f32x4n __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st4g_nt(float* p, float4 v)
{   // streaming store: the line is not kept in L2 (the consumer is a later kernel and the tensor is >> L2)
    f32x4n t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4n*>(p));
}
__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ float4 f4fma(float4 a, float4 b, float4 c)
{
    return make_float4(fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z), fmaf(a.w, b.w, c.w));
}
__device__ __forceinline__ float actf(float v, int act)
{
    if (act == MYOLO_ACT_RELU) return fmaxf(v, 0.f);
    if (act == MYOLO_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}
__device__ __forceinline__ float actmask(float v, int act)
{   // 1 where the activation passes gradient
    if (act == MYOLO_ACT_RELU) return v > 0.f ? 1.f : 0.f;
    if (act == MYOLO_ACT_RELU6) return (v > 0.f && v < 6.f) ? 1.f : 0.f;
    return 1.f;
}

// ---------------------------------------------------------------------------------------
// generic column reduction:  out[v][c] = sum_rows OP(row, c)[v]
// ---------------------------------------------------------------------------------------
struct ColGeom {
    int cl;          // channel-quad lanes per block
    int pl;          // row lanes per block
    int cgroups;     // ceil((C/4) / cl)
    int rblocks;     // number of row slabs
    long long rows_per_block;
    int unroll4;     // four rows per loop trip (ablation: option tune0 & 64 turns it off)
};

static ColGeom col_geom(long long M, int C)
{
    ColGeom g;
    const int q = C / 4;
    // at most 64 channel quads per workgroup (wide layers are split over blockIdx.y) and at least 32 rows per row slab: a slab's
    // partial sums are 2 C doubles, so 8-row slabs of a 1024-channel layer wrote (and the finish kernel re-read) as many bytes of
    // partials as the tensor itself holds
    const int cl_max = 64;
    int cl = 1;
    while (cl * 2 <= q && cl * 2 <= cl_max) cl *= 2;
    if (cl > q) cl = q;
    g.cl = cl;
    g.pl = 256 / cl;
    g.cgroups = (q + cl - 1) / cl;
    long long want = 1024 / g.cgroups;                // target ~1024 blocks in total (4 per CU)
    if (want < 1) want = 1;
    long long rpb = cdiv64(M, want);
    long long min_rows = (long long)g.pl * 4;         // at least 4 rows per thread (small layers: spread wide)
    if (min_rows < 32) min_rows = 32;
    if (rpb < min_rows) rpb = min_rows;
    g.rows_per_block = rpb;
    g.rblocks = (int)cdiv64(M, rpb);
    g.unroll4 = (g_myolo_opt.tune0 & 64) ? 0 : 1;
    return g;
}

static size_t col_ws_bytes(long long M, int C, int nv)
{
    ColGeom g = col_geom(M, C);
    return (size_t)g.rblocks * nv * C * sizeof(double);
}

template <class OP, class = void> struct colreduce_hoists : std::false_type {};
template <class OP> struct colreduce_hoists<OP, std::void_t<decltype(OP::HOIST)>> : std::true_type {};
template <class OP>
__global__ __launch_bounds__(256) void colreduce_kernel(OP op, long long M, int C, ColGeom g, double* __restrict__ part)
{
    constexpr int NV = OP::NV;
    __shared__ float4 red[256];
    const int tid = threadIdx.x;
    const int cl_i = tid % g.cl, pl_i = tid / g.cl;
    const int cq = blockIdx.y * g.cl + cl_i;        // channel quad
    const bool cok = cq < C / 4 && pl_i < g.pl;
    const long long r0 = (long long)blockIdx.x * g.rows_per_block;
    long long r1 = r0 + g.rows_per_block;
    if (r1 > M) r1 = M;
    float4 acc[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = f4zero();
    if (cok) {
        if constexpr (colreduce_hoists<OP>::value) op.init(cq * 4);
        // four rows per trip: their loads are independent, so 4x the bytes are in flight per thread (one row per trip ran at ~2 TB/s on the
        // BatchNorm-backward sums: a dependent load -> accumulate chain per row).  Same rows, same order of additions per accumulator.
        long long r = r0 + pl_i;
        const long long st = g.pl;
        if (g.unroll4)
        for (; r + 3 * st < r1; r += 4 * st) {
            op(r, cq * 4, acc);
            op(r + st, cq * 4, acc);
            op(r + 2 * st, cq * 4, acc);
            op(r + 3 * st, cq * 4, acc);
        }
        for (; r < r1; r += st) op(r, cq * 4, acc);
    }
    // combine row lanes
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        __syncthreads();
        red[tid] = acc[v];
        __syncthreads();
        if (pl_i == 0 && cq < C / 4) {
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int j = 0; j < g.pl; ++j) {
                const float4 t = red[j * g.cl + cl_i];
                s0 += t.x; s1 += t.y; s2 += t.z; s3 += t.w;
            }
            double* o = part + ((long long)blockIdx.x * NV + v) * C + cq * 4;
            o[0] = s0; o[1] = s1; o[2] = s2; o[3] = s3;
        }
    }
}

// sum partials over row blocks: tot[v*C + c] (double).  8 outputs x 32 block-lanes per workgroup.  The consumer's own
// finishing arithmetic (double -> float, BN coefficients, ...) runs in the same kernel through FIN, so a reduction costs
// two launches, not three.
struct FinNone {
    static constexpr int PAIR = 0;
    __device__ void operator()(int, double) const {}
};
struct FinD2F {                      // out[i] = (float)tot[i]
    static constexpr int PAIR = 0;
    float* out;
    __device__ void operator()(int i, double t) const { out[i] = (float)t; }
};

// BL = block lanes (rows of partials walked side by side): 32 (256 threads) everywhere except the mask head's bn1, whose 4704 per-ROI rows made
// every lane walk 147 rows in 18 dependent trips (34 us on the step's main stream): 128 lanes (1024 threads) there.
template <class FIN, int BL = 32>
__global__ __launch_bounds__(8 * BL) void colreduce_finish(const double* __restrict__ part, double* __restrict__ tot, int nblk, int nvc, int C,
                                                        FIN fin)
{
    __shared__ double red[BL][9];
    __shared__ double fin_tot[8];
    const int ol = threadIdx.x & 7, bl = threadIdx.x >> 3;
    // PAIR (NV == 2): a workgroup owns 4 channels x both sums, so FIN sees (sum0, sum1) of a channel together
    const int i = FIN::PAIR ? (ol >> 2) * C + blockIdx.x * 4 + (ol & 3) : blockIdx.x * 8 + ol;
    const bool ok = FIN::PAIR ? ((int)blockIdx.x * 4 + (ol & 3)) < C : i < nvc;
    double s0 = 0, s1 = 0;
    if (ok) {
        // 8 independent loads in flight per thread: the loop is pure load latency (a few hundred partial rows per thread-lane)
        double s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
        int b = bl;
        for (; b + 7 * BL < nblk; b += 8 * BL) {
            const double* q = part + (long long)b * nvc + i;
            const long long st = (long long)BL * nvc;
            const double v0 = q[0], v1 = q[st], v2 = q[2 * st], v3 = q[3 * st], v4 = q[4 * st], v5 = q[5 * st], v6 = q[6 * st], v7 = q[7 * st];
            s0 += v0; s1 += v1; s2 += v2; s3 += v3; s4 += v4; s5 += v5; s6 += v6; s7 += v7;
        }
        for (; b < nblk; b += BL) s0 += part[(long long)b * nvc + i];
        s0 = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
        s1 = 0;
    }
    red[bl][ol] = s0 + s1;
    __syncthreads();
    if (bl == 0) {
        double t = 0;
#pragma unroll
        for (int k = 0; k < BL; ++k) t += red[k][ol];
        if (ok) tot[i] = t;
        if constexpr (FIN::PAIR != 0) fin_tot[ol] = t;
        else if (ok) fin(i, t);
    }
    if constexpr (FIN::PAIR != 0) {
        __syncthreads();
        const int fc = (int)(blockIdx.x * 4 + threadIdx.x);
        if (threadIdx.x < 4 && fc < C) fin.pair(fc, fin_tot[threadIdx.x], fin_tot[4 + threadIdx.x]);
    }
}

template <class OP, class FIN>
static int run_colreduce(OP op, long long M, int C, double* part, double* tot, hipStream_t s, FIN fin)
{
    static_assert(!FIN::PAIR || OP::NV == 2, "paired finish needs exactly two sums per channel");
    ColGeom g = col_geom(M, C);
    hipLaunchKernelGGL((colreduce_kernel<OP>), dim3(g.rblocks, g.cgroups), dim3(256), 0, s, op, M, C, g, part);
    const int nvc = OP::NV * C;
    const int blocks = FIN::PAIR ? (C + 3) / 4 : (nvc + 7) / 8;
    hipLaunchKernelGGL((colreduce_finish<FIN>), dim3(blocks), dim3(256), 0, s, part, tot, g.rblocks, nvc, C, fin);
    return 0;
}
template <class OP>
static int run_colreduce(OP op, long long M, int C, double* part, double* tot, hipStream_t s)
{
    return run_colreduce(op, M, C, part, tot, s, FinNone{});
}

// ---------------------------------------------------------------------------------------
// colsum (bias gradients)
// ---------------------------------------------------------------------------------------
struct OpSum {
    static constexpr int NV = 1;
    const float* x;
    int C;
    __device__ void operator()(long long r, int c, float4* acc) const
    {
        const float4 v = ld4g(x + r * C + c);
        acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
    }
};
__global__ void d2f_kernel(const double* __restrict__ in, float* __restrict__ out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

// ---------------------------------------------------------------------------------------
// BatchNorm
// ---------------------------------------------------------------------------------------
struct OpStats {
    static constexpr int NV = 2;
    const float* x;
    int C;
    __device__ void operator()(long long r, int c, float4* acc) const
    {
        const float4 v = ld4g(x + r * C + c);
        acc[0].x += v.x; acc[0].y += v.y; acc[0].z += v.z; acc[0].w += v.w;
        acc[1] = f4fma(v, v, acc[1]);
    }
};

struct FinBnStats {                  // batch statistics -> mean / var / folded scale, shift / moving averages
    static constexpr int PAIR = 1;
    const float* gamma;
    const float* beta;
    float *mean, *var, *scale, *shift, *mmean, *mvar;
    double M;
    int fused_tf;       // moving variance fed with tf.nn.fused_batch_norm's Bessel-corrected batch variance (see below)
    __device__ void operator()(int, double) const {}
    __device__ void pair(int c, double sum, double sumsq) const
    {
        const double mu = sum / M;
        double vr = sumsq / M - mu * mu;
        if (vr < 0) vr = 0;
        const float rstd = (float)(1.0 / sqrt(vr + (double)BN_EPS_F));
        const float sc = gamma[c] * rstd;
        mean[c] = (float)mu;
        var[c] = (float)vr;
        scale[c] = sc;
        shift[c] = beta[c] - (float)mu * sc;
        if (mmean) {
            // Keras 2.2 BatchNormalization.call multiplies the batch variance it gets from the backend by n/(n-(1+eps))
            // before the moving-average update.  On the TensorFlow backend a 4-D NHWC input with axis=-1 (every BN of this
            // graph, the TimeDistributed ones included) goes through tf.nn.fused_batch_norm, whose batch_variance output is
            // already Bessel-corrected (variance * n/(n-1), n > 1): both factors apply (option bn_fused_tf_variance = 1,
            // default).  0 restates the non-fused backend path (Keras' factor on the biased variance only).
            float vb = (float)vr;
            if (fused_tf && M > 1.0) vb = vb * ((float)M / ((float)M - 1.0f));
            const float vu = vb * ((float)M / ((float)M - (1.0f + BN_EPS_F)));
            mmean[c] = mmean[c] * BN_MOMENTUM_F + (float)mu * (1.0f - BN_MOMENTUM_F);
            mvar[c] = mvar[c] * BN_MOMENTUM_F + vu * (1.0f - BN_MOMENTUM_F);
        }
    }
};

__global__ void bn_frozen_kernel(const float* gamma, const float* beta, const float* mm, const float* mv,
                                 float* scale, float* shift, int C)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float rstd = (float)(1.0 / sqrt((double)mv[c] + (double)BN_EPS_F));
    const float sc = gamma[c] * rstd;
    scale[c] = sc;
    shift[c] = beta[c] - mm[c] * sc;
}

// one workgroup per layer; table row = {gamma, beta offsets into params; mean, variance offsets into stats; output offset; C}
__global__ __launch_bounds__(256) void bn_frozen_batched_kernel(const float* __restrict__ params, const float* __restrict__ stats,
                                                                const long long* __restrict__ table, float* __restrict__ coeffs)
{
    const long long* t = table + 6ll * blockIdx.x;
    const float* gamma = params + t[0];
    const float* beta = params + t[1];
    const float* mm = stats + t[2];
    const float* mv = stats + t[3];
    const int C = (int)t[5];
    float* scale = coeffs + t[4];
    float* shift = scale + C;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float rstd = (float)(1.0 / sqrt((double)mv[c] + (double)BN_EPS_F));
        const float sc = gamma[c] * rstd;
        scale[c] = sc;
        shift[c] = beta[c] - mm[c] * sc;
    }
}

// (round 4) These streaming kernels are bound by their VALU work, not by HBM (a wave64 fp32 instruction takes 4 cycles: 16 lanes per cycle), so the
// per-element index arithmetic is taken out of the loop: the grid-stride is a multiple of 256 and C/4 divides 256 for every layer of this net, so a
// thread keeps ONE channel quad -- its coefficients are loaded once, no 64-bit modulo per element (HOIST; the general form stays for other C).
__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, float* __restrict__ y,
                                                       long long nquads, int C, int act)
{
    const bool nt = nquads > (4ll << 20);      // > 64 MB: stream past the L2
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int cq = C / 4;
    if ((256 % cq) == 0) {
        const int c = (int)((unsigned)(blockIdx.x * blockDim.x + threadIdx.x) % (unsigned)cq) * 4;
        const float4 sc = ld4g(scale + c), sh = ld4g(shift + c);
        auto one = [&](float4 v) {
            return make_float4(actf(fmaf(v.x, sc.x, sh.x), act), actf(fmaf(v.y, sc.y, sh.y), act), actf(fmaf(v.z, sc.z, sh.z), act),
                               actf(fmaf(v.w, sc.w, sh.w), act));
        };
        for (; i + 3 * stride < nquads; i += 4 * stride) {          // four independent quads per trip
            const float4 v0 = ld4g(x + i * 4), v1 = ld4g(x + (i + stride) * 4), v2 = ld4g(x + (i + 2 * stride) * 4), v3 = ld4g(x + (i + 3 * stride) * 4);
            const float4 o0 = one(v0), o1 = one(v1), o2 = one(v2), o3 = one(v3);
            if (nt) { st4g_nt(y + i * 4, o0); st4g_nt(y + (i + stride) * 4, o1); st4g_nt(y + (i + 2 * stride) * 4, o2); st4g_nt(y + (i + 3 * stride) * 4, o3); }
            else { st4g(y + i * 4, o0); st4g(y + (i + stride) * 4, o1); st4g(y + (i + 2 * stride) * 4, o2); st4g(y + (i + 3 * stride) * 4, o3); }
        }
        for (; i < nquads; i += stride) {
            const float4 o = one(ld4g(x + i * 4));
            if (nt) st4g_nt(y + i * 4, o); else st4g(y + i * 4, o);
        }
        return;
    }
    for (; i < nquads; i += stride) {
        const int c = (int)(i % cq) * 4;
        const float4 v = ld4g(x + i * 4);
        const float4 sc = ld4g(scale + c), sh = ld4g(shift + c);
        float4 o;
        o.x = actf(fmaf(v.x, sc.x, sh.x), act);
        o.y = actf(fmaf(v.y, sc.y, sh.y), act);
        o.z = actf(fmaf(v.z, sc.z, sh.z), act);
        o.w = actf(fmaf(v.w, sc.w, sh.w), act);
        if (nt) st4g_nt(y + i * 4, o); else st4g(y + i * 4, o);
    }
}

// frozen BatchNorm (moving statistics) + activation in ONE launch: bn_frozen_kernel's coefficients are formed by every thread for
// its own channel quad (constant over its grid-stride loop: the stride is a multiple of 256 and C/4 divides 256) and written out by
// the first workgroup, then bn_apply_kernel's loop.  Same expressions, same results as the two-kernel form.
__global__ __launch_bounds__(256) void bn_frozen_apply_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const float* __restrict__ mm,
                                                              const float* __restrict__ mv, float* __restrict__ scale,
                                                              float* __restrict__ shift, float* __restrict__ y, long long nquads, int C, int act)
{
    const bool nt = nquads > (4ll << 20);
    const int cq = C / 4;
    const int c = (int)(threadIdx.x % cq) * 4;
    float sc[4], sh[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float rstd = (float)(1.0 / sqrt((double)mv[c + k] + (double)BN_EPS_F));
        sc[k] = gamma[c + k] * rstd;
        sh[k] = beta[c + k] - mm[c + k] * sc[k];
    }
    if (blockIdx.x == 0 && (int)threadIdx.x < cq) {
        st4g(scale + c, make_float4(sc[0], sc[1], sc[2], sc[3]));
        st4g(shift + c, make_float4(sh[0], sh[1], sh[2], sh[3]));
    }
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < nquads; i += stride) {
        const float4 v = ld4g(x + i * 4);
        float4 o;
        o.x = actf(fmaf(v.x, sc[0], sh[0]), act);
        o.y = actf(fmaf(v.y, sc[1], sh[1]), act);
        o.z = actf(fmaf(v.z, sc[2], sh[2]), act);
        o.w = actf(fmaf(v.w, sc[3], sh[3]), act);
        if (nt) st4g_nt(y + i * 4, o); else st4g(y + i * 4, o);
    }
}

// backward pass 1: dbeta = sum dz, dgamma = sum dz*xhat, with dz = dy * actmask(x*scale+shift)
struct OpBnBwd {
    static constexpr int NV = 2;
    static constexpr int HOIST = 1;      // colreduce_kernel calls init(c) once per thread: the channel's terms stay in registers
    const float* dy;
    const float* x;
    const float* scale;
    const float* shift;
    const float* mean;
    const float* var;
    int C, act;
    float4 sc, sh, mu, rs;
    __device__ void init(int c)
    {
        sc = ld4g(scale + c); sh = ld4g(shift + c); mu = ld4g(mean + c);
        const float4 vr = ld4g(var + c);
        rs = make_float4(rsqrtf(vr.x + BN_EPS_F), rsqrtf(vr.y + BN_EPS_F), rsqrtf(vr.z + BN_EPS_F), rsqrtf(vr.w + BN_EPS_F));
    }
    __device__ void operator()(long long r, int c, float4* acc) const
    {
        const float4 g = ld4g(dy + r * C + c), v = ld4g(x + r * C + c);
        float dz, xh;
        dz = g.x * actmask(fmaf(v.x, sc.x, sh.x), act); xh = (v.x - mu.x) * rs.x; acc[0].x += dz; acc[1].x = fmaf(dz, xh, acc[1].x);
        dz = g.y * actmask(fmaf(v.y, sc.y, sh.y), act); xh = (v.y - mu.y) * rs.y; acc[0].y += dz; acc[1].y = fmaf(dz, xh, acc[1].y);
        dz = g.z * actmask(fmaf(v.z, sc.z, sh.z), act); xh = (v.z - mu.z) * rs.z; acc[0].z += dz; acc[1].z = fmaf(dz, xh, acc[1].z);
        dz = g.w * actmask(fmaf(v.w, sc.w, sh.w), act); xh = (v.w - mu.w) * rs.w; acc[0].w += dz; acc[1].w = fmaf(dz, xh, acc[1].w);
    }
};

// Frozen BatchNorm (+ activation) backward in ONE pass: dx = scale * dz does not depend on the sums, so the pass that forms the sums for
// dgamma / dbeta writes it as well (the two-kernel form read dy and x twice).  Same expressions as OpBnBwd + bn_bwd_dx_kernel(batch_stats = 0).
struct OpBnBwdFrozenDx {
    static constexpr int NV = 2;
    static constexpr int HOIST = 1;
    const float* dy;
    const float* x;
    const float* scale;
    const float* shift;
    const float* mean;
    const float* var;
    float* dx;
    int C, act;
    float4 sc, sh, mu, rs;
    __device__ void init(int c)
    {
        sc = ld4g(scale + c); sh = ld4g(shift + c); mu = ld4g(mean + c);
        const float4 vr = ld4g(var + c);
        rs = make_float4(rsqrtf(vr.x + BN_EPS_F), rsqrtf(vr.y + BN_EPS_F), rsqrtf(vr.z + BN_EPS_F), rsqrtf(vr.w + BN_EPS_F));
    }
    __device__ void operator()(long long r, int c, float4* acc) const
    {
        const float4 g = ld4g(dy + r * C + c), v = ld4g(x + r * C + c);
        float4 o;
        float dz, xh;
        dz = g.x * actmask(fmaf(v.x, sc.x, sh.x), act); xh = (v.x - mu.x) * rs.x; acc[0].x += dz; acc[1].x = fmaf(dz, xh, acc[1].x); o.x = sc.x * dz;
        dz = g.y * actmask(fmaf(v.y, sc.y, sh.y), act); xh = (v.y - mu.y) * rs.y; acc[0].y += dz; acc[1].y = fmaf(dz, xh, acc[1].y); o.y = sc.y * dz;
        dz = g.z * actmask(fmaf(v.z, sc.z, sh.z), act); xh = (v.z - mu.z) * rs.z; acc[0].z += dz; acc[1].z = fmaf(dz, xh, acc[1].z); o.z = sc.z * dz;
        dz = g.w * actmask(fmaf(v.w, sc.w, sh.w), act); xh = (v.w - mu.w) * rs.w; acc[0].w += dz; acc[1].w = fmaf(dz, xh, acc[1].w); o.w = sc.w * dz;
        st4g(dx + r * C + c, o);
    }
};

// tot = {dbeta[C], dgamma[C]} (double) -> write float grads
struct FinBnBwd {                    // dbeta = sum(dz), dgamma = sum(dz * xhat)
    static constexpr int PAIR = 1;
    float* dgamma;
    float* dbeta;
    __device__ void operator()(int, double) const {}
    __device__ void pair(int c, double t0, double t1) const
    {
        dbeta[c] = (float)t0;
        dgamma[c] = (float)t1;
    }
};

struct FinBnBwdCoef {                // FinBnBwd + the per-channel terms of dx = scale*dz + (ka + kb*x), see bn_bwd_dx_sparse_kernel
    static constexpr int PAIR = 1;
    float *dgamma, *dbeta, *ka, *kb;
    const float *scale, *mean, *var;
    float invM;
    __device__ void operator()(int, double) const {}
    __device__ void pair(int c, double t0, double t1) const
    {
        dbeta[c] = (float)t0;
        dgamma[c] = (float)t1;
        const float rstd = rsqrtf(var[c] + BN_EPS_F);
        const float b = -scale[c] * invM * (float)t1 * rstd;
        kb[c] = b;
        ka[c] = -scale[c] * invM * (float)t0 - b * mean[c];
    }
};

__device__ __forceinline__ float bn_dx_one(float g, float x, float sc, float sh, float mu, float rstd, float db, float dg, float invM, int act,
                                           int batch_stats)
{
    const float dz = g * actmask(fmaf(x, sc, sh), act);
    if (batch_stats) {
        const float xh = (x - mu) * rstd;
        return sc * (dz - (db + xh * dg) * invM);
    }
    return sc * dz;
}
__global__ __launch_bounds__(256) void bn_bwd_dx_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        const float* __restrict__ mean, const float* __restrict__ var,
                                                        const double* __restrict__ tot, float* __restrict__ dx,
                                                        long long nquads, int C, int act, int batch_stats, float invM)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int cq = C / 4;
    const bool nt = nquads > (4ll << 20);
    if ((256 % cq) == 0) {              // HOIST (see bn_apply_kernel): one channel quad per thread, its eight per-channel terms formed once
        const int c = (int)((unsigned)(blockIdx.x * blockDim.x + threadIdx.x) % (unsigned)cq) * 4;
        const float4 sc = ld4g(scale + c), sh = ld4g(shift + c);
        float4 mu = f4zero(), rs = f4zero(), db = f4zero(), dg = f4zero();
        if (batch_stats) {
            mu = ld4g(mean + c);
            const float4 vr = ld4g(var + c);
            rs = make_float4(rsqrtf(vr.x + BN_EPS_F), rsqrtf(vr.y + BN_EPS_F), rsqrtf(vr.z + BN_EPS_F), rsqrtf(vr.w + BN_EPS_F));
            db = make_float4((float)tot[c], (float)tot[c + 1], (float)tot[c + 2], (float)tot[c + 3]);
            dg = make_float4((float)tot[C + c], (float)tot[C + c + 1], (float)tot[C + c + 2], (float)tot[C + c + 3]);
        }
        auto one = [&](float4 g, float4 v) {
            return make_float4(bn_dx_one(g.x, v.x, sc.x, sh.x, mu.x, rs.x, db.x, dg.x, invM, act, batch_stats),
                               bn_dx_one(g.y, v.y, sc.y, sh.y, mu.y, rs.y, db.y, dg.y, invM, act, batch_stats),
                               bn_dx_one(g.z, v.z, sc.z, sh.z, mu.z, rs.z, db.z, dg.z, invM, act, batch_stats),
                               bn_dx_one(g.w, v.w, sc.w, sh.w, mu.w, rs.w, db.w, dg.w, invM, act, batch_stats));
        };
        for (; i + 3 * stride < nquads; i += 4 * stride) {          // four independent quads per trip: eight loads in flight per thread
            const float4 g0 = ld4g(dy + i * 4), v0 = ld4g(x + i * 4);
            const float4 g1 = ld4g(dy + (i + stride) * 4), v1 = ld4g(x + (i + stride) * 4);
            const float4 g2 = ld4g(dy + (i + 2 * stride) * 4), v2 = ld4g(x + (i + 2 * stride) * 4);
            const float4 g3 = ld4g(dy + (i + 3 * stride) * 4), v3 = ld4g(x + (i + 3 * stride) * 4);
            const float4 o0 = one(g0, v0), o1 = one(g1, v1), o2 = one(g2, v2), o3 = one(g3, v3);
            if (nt) { st4g_nt(dx + i * 4, o0); st4g_nt(dx + (i + stride) * 4, o1); st4g_nt(dx + (i + 2 * stride) * 4, o2); st4g_nt(dx + (i + 3 * stride) * 4, o3); }
            else { st4g(dx + i * 4, o0); st4g(dx + (i + stride) * 4, o1); st4g(dx + (i + 2 * stride) * 4, o2); st4g(dx + (i + 3 * stride) * 4, o3); }
        }
        for (; i < nquads; i += stride) {
            const float4 o = one(ld4g(dy + i * 4), ld4g(x + i * 4));
            if (nt) st4g_nt(dx + i * 4, o); else st4g(dx + i * 4, o);
        }
        return;
    }
    for (; i < nquads; i += stride) {
        const int c = (int)(i % cq) * 4;
        const float4 g = ld4g(dy + i * 4), v = ld4g(x + i * 4);
        const float4 sc = ld4g(scale + c), sh = ld4g(shift + c);
        float gv[4] = {g.x, g.y, g.z, g.w}, xv[4] = {v.x, v.y, v.z, v.w};
        float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float mu = batch_stats ? mean[c + k] : 0.f, rstd = batch_stats ? rsqrtf(var[c + k] + BN_EPS_F) : 0.f;
            const float db = batch_stats ? (float)tot[c + k] : 0.f, dg = batch_stats ? (float)tot[C + c + k] : 0.f;
            o[k] = bn_dx_one(gv[k], xv[k], scv[k], shv[k], mu, rstd, db, dg, invM, act, batch_stats);
        }
        if (nt) st4g_nt(dx + i * 4, make_float4(o[0], o[1], o[2], o[3]));
        else st4g(dx + i * 4, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// Frozen BatchNorm (+ activation) backward from the POST-activation tensor a = act(gamma * xhat + beta): where the activation
// passes gradient, xhat = (a - beta) / gamma, elsewhere dz = 0 -- the pre-BN tensor is not needed (the fused forward never
// stores it; the compacted mask-head backward used to re-run the convolution to get it back).
struct OpBnBwdPost {
    static constexpr int NV = 2;
    const float* dy;
    const float* a;
    const float* gamma;
    const float* beta;
    int C, act;
    __device__ void operator()(long long r, int c, float4* acc) const
    {
        const float4 g = ld4g(dy + r * C + c), v = ld4g(a + r * C + c);
        const float4 ga = ld4g(gamma + c), be = ld4g(beta + c);
        float dz, xh;
        dz = g.x * actmask(v.x, act); xh = ga.x != 0.f ? (v.x - be.x) / ga.x : 0.f; acc[0].x += dz; acc[1].x = fmaf(dz, xh, acc[1].x);
        dz = g.y * actmask(v.y, act); xh = ga.y != 0.f ? (v.y - be.y) / ga.y : 0.f; acc[0].y += dz; acc[1].y = fmaf(dz, xh, acc[1].y);
        dz = g.z * actmask(v.z, act); xh = ga.z != 0.f ? (v.z - be.z) / ga.z : 0.f; acc[0].z += dz; acc[1].z = fmaf(dz, xh, acc[1].z);
        dz = g.w * actmask(v.w, act); xh = ga.w != 0.f ? (v.w - be.w) / ga.w : 0.f; acc[0].w += dz; acc[1].w = fmaf(dz, xh, acc[1].w);
    }
};

__global__ __launch_bounds__(256) void bn_bwd_dx_post_kernel(const float* __restrict__ dy, const float* __restrict__ a,
                                                             const float* __restrict__ scale, float* __restrict__ dx, long long nquads,
                                                             int C, int act)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int cq = C / 4;
    for (; i < nquads; i += stride) {
        const int c = (int)(i % cq) * 4;
        const float4 g = ld4g(dy + i * 4), v = ld4g(a + i * 4), sc = ld4g(scale + c);
        st4g(dx + i * 4, make_float4(sc.x * g.x * actmask(v.x, act), sc.y * g.y * actmask(v.y, act), sc.z * g.z * actmask(v.z, act),
                                     sc.w * g.w * actmask(v.w, act)));
    }
}

// ---------------------------------------------------------------------------------------
// group gather: dst[i] = src[idx[i]] for groups of `gq` float4 (one group = one ROI's rows)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_groups_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                            float* __restrict__ dst, int n, long long gq)
{
    // blockIdx.y walks the groups, blockIdx.x / the loop the quads inside one: no 64-bit division per element (round 3: i / gq)
    for (int g = blockIdx.y; g < n; g += gridDim.y) {
        const float* sp = src + (long long)idx[g] * gq * 4;
        float* dp = dst + (long long)g * gq * 4;
        for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < gq; o += (long long)gridDim.x * blockDim.x)
            st4g(dp + o * 4, ld4g(sp + o * 4));
    }
}

// gather of row groups fused with the BatchNorm apply + activation that followed it in the compacted mask-head backward: dst_pre (optional) =
// the gathered rows as they are, dst_act = act(row * scale + shift) -- the expressions of bn_apply_kernel, one read of the source instead of two
__global__ __launch_bounds__(256) void gather_groups_affine_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                                   const float* __restrict__ scale, const float* __restrict__ shift, int act,
                                                                   float* __restrict__ dst_pre, float* __restrict__ dst_act, int n, long long gq, int cq)
{
    // gq (quads per group) is a multiple of cq (quads per row) and the x-stride of the loop a multiple of 256: with 256 % cq == 0 a thread keeps one channel quad
    const int c = (int)((unsigned)(blockIdx.x * blockDim.x + threadIdx.x) % (unsigned)cq) * 4;
    const float4 sc = ld4g(scale + c), sh = ld4g(shift + c);
    for (int g = blockIdx.y; g < n; g += gridDim.y) {
        const float* sp = src + (long long)idx[g] * gq * 4;
        const long long base = (long long)g * gq * 4;
        for (long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x; o < gq; o += (long long)gridDim.x * blockDim.x) {
            const float4 v = ld4g(sp + o * 4);
            if (dst_pre) st4g(dst_pre + base + o * 4, v);
            st4g(dst_act + base + o * 4, make_float4(actf(fmaf(v.x, sc.x, sh.x), act), actf(fmaf(v.y, sc.y, sh.y), act),
                                                     actf(fmaf(v.z, sc.z, sh.z), act), actf(fmaf(v.w, sc.w, sh.w), act)));
        }
    }
}

__global__ __launch_bounds__(256) void gather_groups_scalar_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                                   float* __restrict__ dst, int n, long long ge)
{
    const long long total = (long long)n * ge;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const long long g = i / ge, o = i - g * ge;
        dst[i] = src[(long long)idx[g] * ge + o];
    }
}

// Row-sparse training-mode BN backward: the upstream gradient is non-zero only in the row groups
// listed in idx (compact tensor dyc [n*grows, C]); x is dense [M, C].
struct OpBnBwdSparse {
    static constexpr int NV = 2;
    const float* dyc;
    const float* x;
    const int32_t* idx;
    const float* scale;
    const float* shift;
    const float* mean;
    const float* var;
    int C, act, grows;
    __device__ void operator()(long long r, int c, float4* acc) const
    {   // r indexes the COMPACT rows
        const long long g = r / grows;
        const long long xr = (long long)idx[g] * grows + (r - g * grows);
        const float4 gv = ld4g(dyc + r * C + c), v = ld4g(x + xr * C + c);
        const float4 sc = ld4g(scale + c), sh = ld4g(shift + c), mu = ld4g(mean + c), vr = ld4g(var + c);
        float dz, xh;
        dz = gv.x * actmask(fmaf(v.x, sc.x, sh.x), act); xh = (v.x - mu.x) * rsqrtf(vr.x + BN_EPS_F); acc[0].x += dz; acc[1].x = fmaf(dz, xh, acc[1].x);
        dz = gv.y * actmask(fmaf(v.y, sc.y, sh.y), act); xh = (v.y - mu.y) * rsqrtf(vr.y + BN_EPS_F); acc[0].y += dz; acc[1].y = fmaf(dz, xh, acc[1].y);
        dz = gv.z * actmask(fmaf(v.z, sc.z, sh.z), act); xh = (v.z - mu.z) * rsqrtf(vr.z + BN_EPS_F); acc[0].z += dz; acc[1].z = fmaf(dz, xh, acc[1].z);
        dz = gv.w * actmask(fmaf(v.w, sc.w, sh.w), act); xh = (v.w - mu.w) * rsqrtf(vr.w + BN_EPS_F); acc[0].w += dz; acc[1].w = fmaf(dz, xh, acc[1].w);
    }
};

__global__ __launch_bounds__(256) void bn_bwd_dx_sparse_kernel(const float* __restrict__ dyc, const float* __restrict__ x,
                                                               const int32_t* __restrict__ inv, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ mean,
                                                               const float* __restrict__ var, const double* __restrict__ tot,
                                                               float* __restrict__ dx, long long nquads, int C, int act, int grows,
                                                               float invM)
{
    // grid.x = row group (one ROI): the group's slot lookup is block-uniform and nothing is divided per element
    (void)nquads;
    const int cq = C / 4;
    const long long g = blockIdx.x;
    const int slot = inv[g];
    const unsigned gq = (unsigned)grows * (unsigned)cq;
    const float* xg = x + g * (long long)grows * C;
    float* dxg = dx + g * (long long)grows * C;
    const float* dyg = slot >= 0 ? dyc + (long long)slot * grows * C : nullptr;
    if (blockDim.x % (unsigned)cq == 0) {
        // a thread keeps its 4 channels for the whole group: per-channel terms are formed once.
        // dx = sc*dz - sc*(db + xh*dg)/M  with xh = (x - mean)*rstd   ==   sc*dz + (ka + kb*x)
        const int c = (int)(threadIdx.x % (unsigned)cq) * 4;
        const float4 sc = ld4g(scale + c), sh = ld4g(shift + c);
        const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
        float ka[4], kb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float rstd = rsqrtf(var[c + k] + BN_EPS_F);
            const float db = (float)tot[c + k], dg = (float)tot[C + c + k];
            kb[k] = -scv[k] * invM * dg * rstd;
            ka[k] = -scv[k] * invM * db - kb[k] * mean[c + k];
        }
        for (unsigned e = threadIdx.x; e < gq; e += blockDim.x) {
            const float4 v = ld4g(xg + (long long)e * 4);
            float4 gv4 = f4zero();
            if (dyg) gv4 = ld4g(dyg + (long long)e * 4);
            const float gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w}, xv[4] = {v.x, v.y, v.z, v.w};
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dz = gv[k] * actmask(fmaf(xv[k], scv[k], shv[k]), act);
                o[k] = fmaf(scv[k], dz, fmaf(kb[k], xv[k], ka[k]));
            }
            st4g_nt(dxg + (long long)e * 4, make_float4(o[0], o[1], o[2], o[3]));
        }
        return;
    }
    for (unsigned e = threadIdx.x; e < gq; e += blockDim.x) {
        const int c = (int)(e % (unsigned)cq) * 4;
        const float4 v = ld4g(xg + (long long)e * 4);
        float4 gv4 = f4zero();
        if (dyg) gv4 = ld4g(dyg + (long long)e * 4);
        const float4 sc = ld4g(scale + c), sh = ld4g(shift + c);
        float gv[4] = {gv4.x, gv4.y, gv4.z, gv4.w}, xv[4] = {v.x, v.y, v.z, v.w};
        float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float dz = gv[k] * actmask(fmaf(xv[k], scv[k], shv[k]), act);
            const float xh = (xv[k] - mean[c + k]) * rsqrtf(var[c + k] + BN_EPS_F);
            const float db = (float)tot[c + k], dg = (float)tot[C + c + k];
            o[k] = scv[k] * (dz - (db + xh * dg) * invM);
        }
        st4g_nt(dxg + (long long)e * 4, make_float4(o[0], o[1], o[2], o[3]));
    }
}

// ---------------------------------------------------------------------------------------
// conv1: pad(1,1) + 3x3 stride 2, Cin = 3
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void conv1_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        float* __restrict__ y, int N, int H, int W, int Co, double* __restrict__ stat)
{
    extern __shared__ __attribute__((aligned(16))) float ws[];     // [27][Co]
    for (int i = threadIdx.x; i < 27 * Co; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const int Ho = H / 2, Wo = W / 2, cq = Co / 4;
    const long long total = (long long)N * Ho * Wo * cq;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float4 s1 = f4zero(), s2 = f4zero();         // stat: this thread's channel quad is the same in every iteration (256 % cq == 0)
    for (; i < total; i += stride) {
        const int c = (int)(i % cq) * 4;
        long long pix = i / cq;
        const int ox = (int)(pix % Wo);
        pix /= Wo;
        const int oy = (int)(pix % Ho);
        const int n = (int)(pix / Ho);
        float4 acc = f4zero();
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky - 1;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox + kx - 1;
                if (ix < 0 || ix >= W) continue;
                const float* xp = x + (((long long)n * H + iy) * W + ix) * 3;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float a = xp[ci];
                    const float4 wv = *reinterpret_cast<const float4*>(&ws[((ky * 3 + kx) * 3 + ci) * Co + c]);
                    acc.x = fmaf(a, wv.x, acc.x); acc.y = fmaf(a, wv.y, acc.y);
                    acc.z = fmaf(a, wv.z, acc.z); acc.w = fmaf(a, wv.w, acc.w);
                }
            }
        }
        st4g(y + i * 4, acc);
        s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
        s2 = f4fma(acc, acc, s2);
    }
    if (stat) {          // per-workgroup partial sums of the output (BatchNorm statistics, finished by colreduce_finish<FinBnStats>)
        __shared__ float4 red[256];
        const int tid = threadIdx.x, cl = cq, pl = 256 / cl, cl_i = tid % cl;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            __syncthreads();
            red[tid] = v == 0 ? s1 : s2;
            __syncthreads();
            if (tid < cl) {
                double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                for (int j = 0; j < pl; ++j) {
                    const float4 t = red[j * cl + cl_i];
                    a0 += t.x; a1 += t.y; a2 += t.z; a3 += t.w;
                }
                double* o = stat + ((long long)blockIdx.x * 2 + v) * Co + cl_i * 4;
                o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
            }
        }
    }
}

// The same conv with its input through LDS (training forward at 224^2: 19 MB in, 51 MB out; conv1_fwd_kernel above took 72-80 us for it: 27
// scalar global loads per thread and three 64-bit divisions per output).  A workgroup owns C1F_STEPS pairs of output rows of one image; the five
// input rows of a pair (one contiguous span of the image) arrive as 16-byte loads, the next pair's while this one is computed; a row in LDS is
// [4 floats of left padding][3 W floats], so the nine values of one kernel row of a pixel are nine consecutive floats.  A thread keeps the 27
// filter values of its four channels in registers (its channel quad never changes: 256 % (Co / 4) == 0) and walks the 2 Wo pixels of the pair.
// Same order of the 27 products per output as conv1_fwd_kernel (the padded taps add 0 * w): bit-identical y.
#define C1F_STEPS 2                      // with statistics (training): the rows of partials and their scratch are sized for this; without, the launcher may pass 1
#define C1F_SMAX 8
__global__ __launch_bounds__(256) void conv1_fwd_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                             int H, int W, int Co, int cq_shift, int chunks, double* __restrict__ stat,
                                                             const float* __restrict__ osc = nullptr, const float* __restrict__ osh = nullptr, int oact = MYOLO_ACT_NONE,
                                                             int steps = C1F_STEPS)
{
    extern __shared__ __attribute__((aligned(16))) float c1f_lds[];     // [5][4 + 3 W]
    __shared__ float4 red[256];
    const int tid = threadIdx.x, Ho = H / 2, Wo = W / 2, cq = Co / 4;
    const int n = blockIdx.x / chunks, ch = blockIdx.x - n * chunks;
    const int row3 = 3 * W, rowf = 4 + row3, nf4 = 5 * row3 / 4;
    const int c4 = (tid & (cq - 1)) * 4;
    float4 wr[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) wr[k] = ld4g(w + k * Co + c4);
    // inference (osc): the frozen BatchNorm + activation behind the conv on the way out -- bn_apply_kernel's expressions; no statistics then
    const float4 osc4 = osc ? ld4g(osc + c4) : f4zero(), osh4 = osc ? ld4g(osh + c4) : f4zero();
    if (tid < 20) c1f_lds[(tid >> 2) * rowf + (tid & 3)] = 0.f;
    const int nsteps = (Ho + 1) / 2;
    const int st0 = ch * steps, st1 = st0 + steps < nsteps ? st0 + steps : nsteps;
    const float* xi = x + (long long)n * H * row3;
    float4 sv[C1F_SMAX];
    int soff[C1F_SMAX], srow[C1F_SMAX];
#pragma unroll
    for (int t = 0; t < C1F_SMAX; ++t) {
        const int f = 4 * (tid + t * 256);
        srow[t] = f < 5 * row3 ? f / row3 : -100000;
        soff[t] = f;
    }
    auto fetch = [&](int step) {
        const int iy0 = 4 * step - 1;
#pragma unroll
        for (int t = 0; t < C1F_SMAX; ++t) {
            const int iy = iy0 + srow[t];
            sv[t] = (iy >= 0 && iy < H) ? ld4g(xi + (long long)iy0 * row3 + soff[t]) : f4zero();
        }
    };
    float4 s1 = f4zero(), s2 = f4zero();
    if (st0 < st1) fetch(st0);
    for (int step = st0; step < st1; ++step) {
        __syncthreads();
#pragma unroll
        for (int t = 0; t < C1F_SMAX; ++t)
            if (srow[t] >= 0) *reinterpret_cast<float4*>(c1f_lds + srow[t] * rowf + 4 + (soff[t] - srow[t] * row3)) = sv[t];
        __syncthreads();
        if (step + 1 < st1) fetch(step + 1);
        const int oy0 = 2 * step;
        for (int it = tid; it < 2 * Wo * cq; it += 256) {
            const int p = it >> cq_shift;
            const int rr = p >= Wo ? 1 : 0, ox = p - rr * Wo, oy = oy0 + rr;
            if (oy >= Ho) break;
            const float* b = c1f_lds + 2 * rr * rowf + 1 + 6 * ox;
            float4 acc = f4zero();
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int j = 0; j < 9; ++j) {
                    const float a = b[ky * rowf + j];
                    const float4 wv = wr[ky * 9 + j];
                    acc.x = fmaf(a, wv.x, acc.x); acc.y = fmaf(a, wv.y, acc.y);
                    acc.z = fmaf(a, wv.z, acc.z); acc.w = fmaf(a, wv.w, acc.w);
                }
            if (osc)
                acc = make_float4(actf(fmaf(acc.x, osc4.x, osh4.x), oact), actf(fmaf(acc.y, osc4.y, osh4.y), oact), actf(fmaf(acc.z, osc4.z, osh4.z), oact),
                                  actf(fmaf(acc.w, osc4.w, osh4.w), oact));
            st4g(y + (((long long)n * Ho + oy) * Wo + ox) * Co + c4, acc);
            s1.x += acc.x; s1.y += acc.y; s1.z += acc.z; s1.w += acc.w;
            s2 = f4fma(acc, acc, s2);
        }
    }
    if (stat) {          // per-workgroup partial sums of the output, as conv1_fwd_kernel leaves them
        const int pl = 256 / cq, cl_i = tid & (cq - 1);
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            __syncthreads();
            red[tid] = v == 0 ? s1 : s2;
            __syncthreads();
            if (tid < cq) {
                double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                for (int j = 0; j < pl; ++j) {
                    const float4 t = red[j * cq + cl_i];
                    a0 += t.x; a1 += t.y; a2 += t.z; a3 += t.w;
                }
                double* o = stat + ((long long)blockIdx.x * 2 + v) * Co + cl_i * 4;
                o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
            }
        }
    }
}

// the LDS-staged forward takes: rows that are whole 16-byte groups (W % 4 == 0), at most C1F_SMAX * 256 groups per five rows, a power-of-two number of
// channel quads <= 64
static bool conv1_fwd_rows_ok(int H, int W, int Cout)
{
    const int cq = Cout / 4;
    return (H & 1) == 0 && (W & 3) == 0 && 5 * 3 * W / 4 <= C1F_SMAX * 256 && cq >= 1 && cq <= 64 && (cq & (cq - 1)) == 0 && !(g_myolo_opt.tune0 & 16384);
}
static int conv1_fwd_rows_chunks(int H, int steps = C1F_STEPS) { return ((H / 2 + 1) / 2 + steps - 1) / steps; }
static void conv1_fwd_rows_launch(const float* x, const float* w, float* y, int N, int H, int W, int Cout, double* stat, hipStream_t s,
                                  const float* osc = nullptr, const float* osh = nullptr, int oact = MYOLO_ACT_NONE)
{
    // one pair of rows per workgroup when two would leave the chip under-filled and nobody counts the rows of partials (an inference batch of four
    // 416 x 416 images: 208 workgroups -> 416, 17.3 -> 13.5 us; the training batch keeps two: 30.0 us either way, tools/experiments/conv1_steps.py)
    const int steps = (!stat && (long long)N * conv1_fwd_rows_chunks(H) < 512) ? 1 : C1F_STEPS;
    const int chunks = conv1_fwd_rows_chunks(H, steps);
    int sh = 0;
    while ((1 << sh) < Cout / 4) ++sh;
    hipLaunchKernelGGL(conv1_fwd_rows_kernel, dim3((unsigned)(N * chunks)), dim3(256), (size_t)5 * (4 + 3 * W) * sizeof(float), s, x, w, y, H, W, Cout, sh, chunks,
                       stat, osc, osh, oact, steps);
}

// dw[k][co] = sum_pixels patch[k] * dy[co]; rows = output pixels, "channels" = Co, 27 accumulators
struct OpConv1Dw {
    static constexpr int NV = 27;
    const float* x;
    const float* dy;
    int H, W, Co;
    __device__ void operator()(long long r, int c, float4* acc) const
    {
        const int Ho = H / 2, Wo = W / 2;
        const unsigned ru = (unsigned)r;                 // rows < 2^31 (checked by the launcher): 32-bit divisions
        const unsigned t = ru / (unsigned)Wo;
        const int ox = (int)(ru - t * (unsigned)Wo);
        const int n = (int)(t / (unsigned)Ho);
        const int oy = (int)(t - (unsigned)n * (unsigned)Ho);
        const float4 g = ld4g(dy + r * Co + c);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox + kx - 1;
                const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                const float* xp = x + (((long long)n * H + (ok ? iy : 0)) * W + (ok ? ix : 0)) * 3;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float a = ok ? xp[ci] : 0.f;
                    float4& A = acc[(ky * 3 + kx) * 3 + ci];
                    A.x = fmaf(a, g.x, A.x); A.y = fmaf(a, g.y, A.y); A.z = fmaf(a, g.z, A.z); A.w = fmaf(a, g.w, A.w);
                }
            }
        }
    }
};

// ---------------------------------------------------------------------------------------
// depthwise 3x3.  Each thread produces TW consecutive outputs along W for one channel quad,
// holding the (TW-1)*S+3 input columns of each of the 3 rows in registers (each input quad is
// loaded once per thread instead of up to 9 times).
// ---------------------------------------------------------------------------------------
struct DwAffine { const float* scale; const float* shift; int act; };   // scale == nullptr: none
// training-mode fusion (model.py:57-66 with BatchNormalization on batch statistics): `in` = the PRODUCING layer's BatchNorm apply +
// activation, done on the load of its pre-BN output (zero padding stays zero) -- the normalised tensor is never written;
// `stat` != nullptr: per-workgroup partial sums (sum, sum of squares) of THIS conv's output per channel, [workgroup][2][C] doubles,
// finished by colreduce_finish<FinBnStats> -- the statistics pass over the output disappears.  Needs 256 % (C/4) == 0.
// `bw.x` != nullptr (data-gradient use, MODE 3): the conv's OUTPUT is the gradient reaching a training-mode BatchNorm + activation whose pre-BN tensor is
// bw.x (same shape as the output): the partial sums of that BatchNorm's backward -- sum dz, sum dz * xhat, dz = out * actmask(x * scale + shift) --
// leave in bw.part [workgroup][2][C] doubles, finished by colreduce_finish<FinBnBwd>: colreduce_kernel<OpBnBwd>'s pass over (dy, x) disappears.
struct DwBnBwd { const float* x; const float* scale; const float* shift; const float* mean; const float* var; int act; double* part; };
struct DwFuse { DwAffine in; double* stat; DwBnBwd bw; };

template <int S, int TW, int TH>
__global__ __launch_bounds__(256) void dw_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     float* __restrict__ y, int N, int H, int W, int C, int Ho, int Wo, DwAffine af, DwFuse fu)
{
    // grid: x = (w-tile, channel quad) pairs, y = group of TH output rows, z = image: no 64-bit div/mod per thread.
    // The thread walks down the (TH-1)*S+3 input rows of its strip once (sliding window: every input row is loaded once per
    // strip and feeds up to 3 output rows), so an input element is fetched (TH+2)/TH * (TW+2)/TW times in all instead of 4.5.
    constexpr int NC = (TW - 1) * S + 3;
    constexpr int NR = (TH - 1) * S + 3;
    const int pt = (S == 1) ? 1 : 0, plft = (S == 1) ? 1 : 0;
    const int cq = C / 4;
    const int wtiles = (Wo + TW - 1) / TW;
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e < (unsigned)(wtiles * cq);
    if (!live && !fu.stat) return;                       // (with statistics every thread reaches the workgroup reduction)
    const int wt = live ? e / (unsigned)cq : 0;
    const int c = live ? (e - wt * cq) * 4 : 0;
    const int oy0 = blockIdx.y * TH, n = blockIdx.z;
    const int ox0 = wt * TW;
    float4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = ld4g(w + k * C + c);
    float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = f4zero();
    if (fu.in.scale) { isc = ld4g(fu.in.scale + c); ish = ld4g(fu.in.shift + c); }
    float4 acc[TH][TW];
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int j = 0; j < TW; ++j) acc[r][j] = f4zero();
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
        const int iy = oy0 * S + ri - pt;
        float4 col[NC];
        const bool rowin = iy >= 0 && iy < H;
        const float* rowp = x + (((long long)n * H + (rowin ? iy : 0)) * W) * C + c;
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int ix = ox0 * S + k - plft;
            const bool in = rowin && ix >= 0 && ix < W;
            col[k] = in ? ld4g(rowp + (long long)ix * C) : f4zero();
            if (fu.in.scale && in) {
                col[k].x = actf(fmaf(col[k].x, isc.x, ish.x), fu.in.act); col[k].y = actf(fmaf(col[k].y, isc.y, ish.y), fu.in.act);
                col[k].z = actf(fmaf(col[k].z, isc.z, ish.z), fu.in.act); col[k].w = actf(fmaf(col[k].w, isc.w, ish.w), fu.in.act);
            }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            // input row ri feeds output row r with r*S + ky == ri
            if ((ri - ky) >= 0 && ((ri - ky) % S) == 0 && (ri - ky) / S < TH) {
                constexpr int dummy = 0; (void)dummy;
                const int r = (ri - ky) / S;
#pragma unroll
                for (int j = 0; j < TW; ++j)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc[r][j] = f4fma(col[j * S + kx], wv[ky * 3 + kx], acc[r][j]);
            }
        }
    }
    // inference: the folded frozen BatchNorm + activation on the way out (same fma as bn_apply_kernel: bit-identical to the two-launch form)
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = f4zero();
    if (af.scale) { sc = ld4g(af.scale + c); sh = ld4g(af.shift + c); }
    float4 s1 = f4zero(), s2 = f4zero();
#pragma unroll
    for (int r = 0; r < TH; ++r) {
        const int oy = oy0 + r;
        if (oy >= Ho || !live) continue;
        float* yrow = y + (((long long)n * Ho + oy) * Wo) * C + c;
#pragma unroll
        for (int j = 0; j < TW; ++j)
            if (ox0 + j < Wo) {
                float4 o = acc[r][j];
                s1.x += o.x; s1.y += o.y; s1.z += o.z; s1.w += o.w;
                s2 = f4fma(o, o, s2);
                if (af.scale) {
                    o.x = actf(fmaf(o.x, sc.x, sh.x), af.act); o.y = actf(fmaf(o.y, sc.y, sh.y), af.act);
                    o.z = actf(fmaf(o.z, sc.z, sh.z), af.act); o.w = actf(fmaf(o.w, sc.w, sh.w), af.act);
                }
                st4g(yrow + (long long)(ox0 + j) * C, o);
            }
    }
    if (fu.stat) {
        // workgroup reduction over the threads that share a channel quad (tid % cq; 256 % cq == 0 -- checked by the launcher), in
        // double from here on; one row of partials per workgroup
        __shared__ float4 red[256];
        const int tid = threadIdx.x, cl = cq < 256 ? cq : 256, pl = 256 / cl, cl_i = tid % cl;
        const long long blk = blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * blockIdx.z);
        const int cq0 = (int)((blockIdx.x * 256u) % (unsigned)cq);           // first channel quad of this workgroup (cq > 256: a slice)
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            __syncthreads();
            red[tid] = v == 0 ? s1 : s2;
            __syncthreads();
            if (tid < cl) {
                double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
                for (int j = 0; j < pl; ++j) {
                    const float4 t = red[j * cl + cl_i];
                    a0 += t.x; a1 += t.y; a2 += t.z; a3 += t.w;
                }
                double* o = fu.stat + (blk * 2 + v) * C + (cq0 + cl_i) * 4;
                o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------
// depthwise 3x3, row-sliding LDS-staged form (round 4; shapes with C % 32 == 0 -- every layer of the alpha-1 net).
//
// A workgroup owns PX = 224 / CQB output columns x CQB channel quads (CQB x 16 B = 128 / 256 / 512 contiguous bytes per pixel) of ONE
// image and walks DOWN a chunk of rows.  Per input row: every thread loads ONE float4 (two for stride 2) -- issued PF rows ahead, so
// PF x 16 B per thread are in flight -- applies the producing layer's BatchNorm + ReLU6 to it once, puts it into a double-buffered LDS
// row and, behind ONE barrier, reads its left / right neighbours from there.  The row then feeds the three output rows it belongs to
// (rotating accumulators: ky = 2 of row iy-1 -> emitted, ky = 1 of row iy, ky = 0 of row iy+1; for stride 2 an even input row closes
// one output row and opens the next).  So an input element is fetched from global memory exactly once per workgroup (the round-3
// kernel: (TH+2)/TH x (TW+2)/TW = 2.25 times from L1/L2), the only re-fetch is the halo between neighbouring workgroups: one column
// either side of a strip (32 spare threads of the 256 load it) and the rows between two row chunks -- and neighbours are adjacent in
// the (XCD-contiguous) workgroup order, i.e. they find those lines in their XCD's L2.
// FMA order per output = the round-3 kernel's (ky outer, kx inner, from +0): bit-identical results.
// `fu.stat`: per-workgroup partial sums of the output, row = (image, strip, chunk), this workgroup's channel slice of it.
// ---------------------------------------------------------------------------------------
struct DwRowsGeom { int cqb, px, strips, chunks, rc, ncb, nblk; long long tiles; };

static bool dw_rows_ok(int H, int W, int C) { return (C % 32) == 0 && (long long)H * W * C * 4 < (1ll << 30) && !g_myolo_opt.dw_legacy; }

static DwRowsGeom dw_rows_geom(int N, int H, int W, int C, int S)
{
    DwRowsGeom g;
    const int Ho = H / S, Wo = W / S, cq = C / 4;
    g.cqb = (Wo <= 7 && (cq % 32) == 0) ? 32 : ((cq % 16) == 0 ? 16 : 8);
    g.px = 224 / g.cqb;
    g.strips = (Wo + g.px - 1) / g.px;
    g.ncb = cq / g.cqb;
    const long long base = (long long)N * g.ncb * g.strips;
    const long long want = g_myolo_opt.dw_min_wg ? g_myolo_opt.dw_min_wg : 1024;     // ~4 workgroups per CU, all resident at once
    int chunks = 1;
    while (base * chunks < want && Ho / (chunks * 2) >= 7) chunks *= 2;
    g.rc = (Ho + chunks - 1) / chunks;
    g.chunks = (Ho + g.rc - 1) / g.rc;
    g.tiles = base * g.chunks;
    g.nblk = N * g.strips * g.chunks;
    return g;
}

typedef unsigned int dw_u32x4 __attribute__((ext_vector_type(4)));
#define DW_OOB 0x7fffff00u           // buffer offset beyond every image (images are < 2^30 bytes, checked by the launcher): load gives 0, store is dropped
__device__ __forceinline__ float4 dw_bufld(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    const dw_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ void dw_bufst(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v)
{
    const dw_u32x4 t = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    __builtin_amdgcn_raw_buffer_store_b128(t, r, (int)off, 0, 0);
}
// branch-free act(v * sc + sh): lo / hi are the clamps of the activation (-inf / +inf where it has none), `none` keeps the unclamped value
__device__ __forceinline__ float4 dw_affine(float4 v, float4 sc, float4 sh, float lo, float hi, bool none)
{
    float4 a = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
    float4 c = make_float4(fminf(fmaxf(a.x, lo), hi), fminf(fmaxf(a.y, lo), hi), fminf(fmaxf(a.z, lo), hi), fminf(fmaxf(a.w, lo), hi));
    return none ? a : c;
}

typedef float dw_f2 __attribute__((ext_vector_type(2)));
struct dw_f4p { dw_f2 lo, hi; };                 // a float4 as two packed pairs: v_pk_fma_f32 / v_pk_add_f32 do two lanes' worth per instruction
__device__ __forceinline__ dw_f4p dw_pk(float4 v) { dw_f4p r; r.lo = dw_f2{v.x, v.y}; r.hi = dw_f2{v.z, v.w}; return r; }
__device__ __forceinline__ float4 dw_unpk(dw_f4p v) { return make_float4(v.lo.x, v.lo.y, v.hi.x, v.hi.y); }
__device__ __forceinline__ dw_f4p dw_fma(dw_f4p a, dw_f4p b, dw_f4p c)
{
    dw_f4p r;
    r.lo = __builtin_elementwise_fma(a.lo, b.lo, c.lo);
    r.hi = __builtin_elementwise_fma(a.hi, b.hi, c.hi);
    return r;
}
__device__ __forceinline__ dw_f4p dw_zero() { dw_f4p r; r.lo = dw_f2{0.f, 0.f}; r.hi = dw_f2{0.f, 0.f}; return r; }
// act(v * sc + sh): R6 = ReLU6 as one v_med3_f32 per element; otherwise the clamps lo / hi (-inf / +inf where the activation has none)
template <bool R6>
__device__ __forceinline__ dw_f4p dw_affine_pk(dw_f4p v, dw_f4p sc, dw_f4p sh, float lo, float hi)
{
    dw_f4p a = dw_fma(v, sc, sh);
    if (R6) {
        a.lo.x = __builtin_amdgcn_fmed3f(a.lo.x, 0.f, 6.f); a.lo.y = __builtin_amdgcn_fmed3f(a.lo.y, 0.f, 6.f);
        a.hi.x = __builtin_amdgcn_fmed3f(a.hi.x, 0.f, 6.f); a.hi.y = __builtin_amdgcn_fmed3f(a.hi.y, 0.f, 6.f);
    } else {
        a.lo.x = fminf(fmaxf(a.lo.x, lo), hi); a.lo.y = fminf(fmaxf(a.lo.y, lo), hi);
        a.hi.x = fminf(fmaxf(a.hi.x, lo), hi); a.hi.y = fminf(fmaxf(a.hi.y, lo), hi);
    }
    return a;
}

// What bounds this kernel is the VALU, not HBM: with every load and store pointed out of range (no memory traffic at all) the first form
// of it still took 21.7 of its 30.4 us on the 112x112x32 layer -- ~125 wave64 fp32 instructions per row at 4 cycles each (16 lanes per
// cycle; the 157 TFLOP/s vector peak is PACKED fp32).  Hence: packed FMAs / adds (v_pk_fma_f32: the 36 FMAs of a row are 18 instructions),
// ReLU6 as one v_med3_f32, the producing BatchNorm applied ONCE per element before it goes to LDS (not on each of its three reads), padding
// columns handled by a zero scale / shift instead of per-element selects, the first two rows of a chunk peeled so that no emitted row
// needs a mask, out-of-image rows by a uniform branch.
// Every global access is a raw buffer access on a per-image descriptor: rows above / below the image, padding columns, dead lanes and
// the rows a prefetch runs past its chunk are "offset out of range" -- the load returns 0 without a memory request, the store is dropped --
// so there is no branch around a load (a load inside an exec-masked branch makes hipcc wait vmcnt(0) at every use: an earlier form drained
// its prefetch queue once per row).
// MODE 0: plain; 1: the producing layer's BatchNorm + activation on load, statistics of the output when fu.stat; 2: folded frozen
// BatchNorm + activation on the way out (inference); 3 (stride 1, the data gradient): plain + the BatchNorm-backward sums of the output against
// fu.bw.x (see DwBnBwd) -- the pre-BN row of an output row is requested PF rows ahead, like the input rows.
template <int S, int CQB, int MODE, bool R6>
__global__ __launch_bounds__(256, 4) void dw_rows_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                         int H, int W, int C, int Ho, int Wo, int strips, int chunks, int rc, int ncb,
                                                         unsigned xcd_tiles, int flags, DwAffine af, DwFuse fu)
{
    constexpr int PX = 224 / CQB;                 // output columns of this workgroup
    constexpr int NE = (S == 1) ? PX + 2 : PX + 1;      // LDS row entries: S=1 input cols x0-1 .. x0+PX; S=2 even input cols 2*x0 .. 2*(x0+PX)
    constexpr int NHALO = (S == 1) ? 2 : 1;
    constexpr int NL = (S == 1) ? 1 : 2;           // float4 loads per thread and input row
    constexpr int PF = (S == 1) ? 3 : 2;           // input rows in flight per thread
    __shared__ float4 rowbuf[2][NE * CQB];
    __shared__ float4 red[224];
    const int tid = threadIdx.x;
    unsigned b = blockIdx.x;
    if (xcd_tiles) b = (b & 7u) * xcd_tiles + (b >> 3);          // workgroup b runs on XCD b % 8: give every XCD a contiguous run of tiles
    const int ch = b % (unsigned)chunks;
    unsigned t = b / (unsigned)chunks;
    const int sx = t % (unsigned)strips;
    t /= (unsigned)strips;
    const int cb = t % (unsigned)ncb;
    const int n = t / (unsigned)ncb;
    const int x0 = sx * PX;
    const int y0 = ch * rc, y1 = min(y0 + rc, Ho);
    // roles: threads 0..223 = (output column, channel quad); 224.. = halo columns
    const bool comp = tid < 224;
    int px, q, e_w, col;                           // LDS entry this thread writes, (first) input column it loads
    bool loader;
    if (comp) {
        px = tid / CQB; q = tid % CQB;
        if (S == 1) { e_w = px + 1; col = x0 + px; } else { e_w = px; col = 2 * (x0 + px); }
        loader = true;
    } else {
        const int k = tid - 224;
        px = 0; q = k % CQB;
        const int side = k / CQB;                  // S=1: 0 left, 1 right; S=2: 0 = the one (right) halo column
        loader = side < NHALO && CQB < 32;         // (CQB = 32: only chosen when the strip spans the row -- both halo columns are padding, zeroed once below)
        if (S == 1) { e_w = side ? PX + 1 : 0; col = side ? x0 + PX : x0 - 1; } else { e_w = PX; col = 2 * (x0 + PX); }
    }
    const int c = (cb * CQB + q) * 4;
    const bool colin = loader && col >= 0 && col < W;
    const bool col2in = comp && (S == 2) && (col + 1) < W;
    const int ox = x0 + px;
    const bool live = comp && ox < Wo;
    dw_f4p wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = dw_pk(ld4g(w + ((flags & 8) ? 8 - k : k) * C + c));     // flags & 8: the kernel rotated by 180 degrees (data gradient)
    // MODE 1: input map (scale / shift zeroed for a padding column: act(0 * v + 0) = 0); MODE 2: output map
    dw_f4p sc0 = dw_pk(make_float4(1.f, 1.f, 1.f, 1.f)), sh0 = dw_zero(), sc1 = sc0, sh1 = sh0;
    float lo = -INFINITY, hi = INFINITY;
    if (MODE != 0) {
        const DwAffine a = (MODE == 1) ? fu.in : af;
        if (a.scale) {                              // (MODE 1 without an input map: statistics only; the identity is exact)
            sc0 = dw_pk(ld4g(a.scale + c)); sh0 = dw_pk(ld4g(a.shift + c));
            lo = a.act == MYOLO_ACT_NONE ? -INFINITY : 0.f;
            hi = a.act == MYOLO_ACT_RELU6 ? 6.f : INFINITY;
        }
        sc1 = sc0; sh1 = sh0;
        if (MODE == 1) {
            if (!colin) { sc0 = dw_zero(); sh0 = dw_zero(); }
            if (!col2in) { sc1 = dw_zero(); sh1 = dw_zero(); }
        }
    }
    if (CQB == 32 && !comp) {                      // constant zero halo entries of both buffers
        const int k = tid - 224;
        rowbuf[0][k] = f4zero(); rowbuf[1][k] = f4zero();
        rowbuf[0][(NE - 1) * CQB + k] = f4zero(); rowbuf[1][(NE - 1) * CQB + k] = f4zero();
    }
    // input rows of this chunk: S=1: y0-1 .. y1 (pad 1 above / below); S=2: 2*y0 .. 2*y1 (pad below / right only)
    const int row0 = (S == 1) ? y0 - 1 : 2 * y0;
    const int nrows = (S == 1) ? (y1 - y0 + 2) : (2 * (y1 - y0) + 1);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long long)n * H * W * C), 0, H * W * C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + (long long)n * Ho * Wo * C), 0, Ho * Wo * C * 4, 0x00020000);
    const unsigned off0 = (colin && !(flags & 2)) ? (unsigned)(col * C + c) * 4u : DW_OOB;
    const unsigned off1 = col2in ? off0 + (unsigned)C * 4u : DW_OOB;
    const int rstride = W * C * 4;                 // bytes; a row index outside [0, H) drives the offset out of the descriptor's range
    // MODE 3: per-channel terms of the BatchNorm whose backward sums are formed, and the queue of its pre-BN rows (step r emits output row y0 + r - 2)
    dw_f4p bsc = dw_zero(), bsh = dw_zero(), bmu = dw_zero(), brs = dw_zero();
    float blo = -INFINITY, bhi = INFINITY;
    bool ball = true;                               // MODE 3: no activation -> dz = out whatever z is (actmask(): also for a non-finite z)
    float4 yq[PF];
    __amdgpu_buffer_rsrc_t rq = ry;
    if (MODE == 3) {
        static_assert(MODE != 3 || S == 1, "the fused BatchNorm-backward sums exist for the stride-1 row kernel");
        bsc = dw_pk(ld4g(fu.bw.scale + c)); bsh = dw_pk(ld4g(fu.bw.shift + c)); bmu = dw_pk(ld4g(fu.bw.mean + c));
        const float4 vr = ld4g(fu.bw.var + c);
        brs = dw_pk(make_float4(rsqrtf(vr.x + BN_EPS_F), rsqrtf(vr.y + BN_EPS_F), rsqrtf(vr.z + BN_EPS_F), rsqrtf(vr.w + BN_EPS_F)));
        ball = fu.bw.act != MYOLO_ACT_RELU && fu.bw.act != MYOLO_ACT_RELU6;
        blo = ball ? -INFINITY : 0.f;
        bhi = fu.bw.act == MYOLO_ACT_RELU6 ? 6.f : INFINITY;
        rq = __builtin_amdgcn_make_buffer_rsrc((void*)(fu.bw.x + (long long)n * Ho * Wo * C), 0, Ho * Wo * C * 4, 0x00020000);
    }
    float4 pf[PF][NL];
    auto fetch = [&](int r, float4* dst) {
        const int iy = row0 + r;
        const bool ok = r < nrows && iy >= 0 && iy < H;         // uniform
        const unsigned ro = ok ? (unsigned)(iy * rstride) : DW_OOB;     // (DW_OOB + DW_OOB wraps to 0xfffffe00: still out of range)
        dst[0] = dw_bufld(rx, off0 + ro);
        if (NL == 2) dst[NL - 1] = dw_bufld(rx, off1 + ro);
    };
#pragma unroll
    for (int j = 0; j < PF; ++j) fetch(j, pf[j]);
    dw_f4p a1 = dw_zero(), a2 = dw_zero();          // S=1: outputs iy / iy-1 in the making; S=2: a1 = current output row
    dw_f4p s1 = dw_zero(), s2 = dw_zero();
    const unsigned yoff = (live && !(flags & 4)) ? (unsigned)(ox * C + c) * 4u : DW_OOB;
    const unsigned qoff = live ? (unsigned)(ox * C + c) * 4u : DW_OOB;
    const int ystride = Wo * C * 4;
    auto fetchq = [&](int r, float4& dst) {         // pre-BN row of the output row that step r emits (rows outside this chunk: out of range, zeros)
        const int oy = y0 + r - 2;
        const bool ok = r >= 2 && oy < y1;          // uniform
        dst = dw_bufld(rq, qoff + (ok ? (unsigned)(oy * ystride) : DW_OOB));
    };
    if (MODE == 3) {
#pragma unroll
        for (int j = 0; j < PF; ++j) fetchq(2 + j, yq[j]);
    }
    dw_f4p qv = dw_zero();                          // MODE 3: the pre-BN values of the row being emitted
    auto emit = [&](dw_f4p o, int oy) {             // (only called for rows that exist: y0 <= oy < y1; dead lanes are zeroed out of the sums at the end)
        if (MODE == 1) {
            s1.lo += o.lo; s1.hi += o.hi;
            s2 = dw_fma(o, o, s2);
        }
        if (MODE == 3) {
            // dz = out * actmask(x * scale + shift); xhat = (x - mean) * rstd  (OpBnBwd's expressions)
            const dw_f4p z = dw_fma(qv, bsc, bsh);
            dw_f4p dz;
            dz.lo.x = (ball || (z.lo.x > blo && z.lo.x < bhi)) ? o.lo.x : 0.f; dz.lo.y = (ball || (z.lo.y > blo && z.lo.y < bhi)) ? o.lo.y : 0.f;
            dz.hi.x = (ball || (z.hi.x > blo && z.hi.x < bhi)) ? o.hi.x : 0.f; dz.hi.y = (ball || (z.hi.y > blo && z.hi.y < bhi)) ? o.hi.y : 0.f;
            dw_f4p xh;
            xh.lo = (qv.lo - bmu.lo) * brs.lo; xh.hi = (qv.hi - bmu.hi) * brs.hi;
            s1.lo += dz.lo; s1.hi += dz.hi;
            s2 = dw_fma(dz, xh, s2);
        }
        if (MODE == 2) o = dw_affine_pk<R6>(o, sc0, sh0, lo, hi);
        dw_bufst(ry, yoff + (unsigned)(oy * ystride), dw_unpk(o));
    };
    // one input row: value(s) -> (input map) -> LDS -> barrier -> neighbours; PEEL = 0: an ordinary row; 1 / 2: the first / second row of the
    // chunk, which only open accumulators (S=2: only 1 exists)
    auto row = [&](int r, float4* cur, auto peel, float4* qcur = nullptr) {
        constexpr int PEEL = decltype(peel)::value;
        dw_f4p v0 = dw_pk(cur[0]), v1 = dw_pk(cur[NL - 1]);
        fetch(r + PF, cur);
        if (MODE == 3 && PEEL == 0) {
            qv = dw_pk(*qcur);
            fetchq(r + PF, *qcur);
        }
        const int iy = row0 + r;
        if (MODE == 1) {
            if (iy >= 0 && iy < H) {                // uniform; a row outside the image stays zero (the load returned zeros)
                v0 = dw_affine_pk<R6>(v0, sc0, sh0, lo, hi);
                if (NL == 2) v1 = dw_affine_pk<R6>(v1, sc1, sh1, lo, hi);
            }
        }
        float4* buf = rowbuf[r & 1];
        if (loader) buf[e_w * CQB + q] = dw_unpk(v0);
        __syncthreads();
        if (S == 1) {
            const dw_f4p l = dw_pk(buf[px * CQB + q]), rt = dw_pk(buf[(px + 2) * CQB + q]);
            // this input row is ky = 2 of output iy-1 (a2, complete after it), ky = 1 of output iy (a1), ky = 0 of output iy+1
            if (PEEL == 0) {
                a2 = dw_fma(l, wv[6], a2); a2 = dw_fma(v0, wv[7], a2); a2 = dw_fma(rt, wv[8], a2);
                emit(a2, iy - 1);
            }
            if (PEEL != 1) { a1 = dw_fma(l, wv[3], a1); a1 = dw_fma(v0, wv[4], a1); a1 = dw_fma(rt, wv[5], a1); }
            dw_f4p a0 = dw_fma(l, wv[0], dw_zero()); a0 = dw_fma(v0, wv[1], a0); a0 = dw_fma(rt, wv[2], a0);
            a2 = a1; a1 = a0;
        } else {
            const dw_f4p rt = dw_pk(buf[(px + 1) * CQB + q]);
            if ((r & 1) == 0) {                     // even input row 2m: ky = 2 of output m-1 (then emitted), ky = 0 of output m
                if (PEEL == 0) {
                    a1 = dw_fma(v0, wv[6], a1); a1 = dw_fma(v1, wv[7], a1); a1 = dw_fma(rt, wv[8], a1);
                    emit(a1, (iy >> 1) - 1);
                }
                a1 = dw_fma(v0, wv[0], dw_zero()); a1 = dw_fma(v1, wv[1], a1); a1 = dw_fma(rt, wv[2], a1);
            } else {
                a1 = dw_fma(v0, wv[3], a1); a1 = dw_fma(v1, wv[4], a1); a1 = dw_fma(rt, wv[5], a1);
            }
        }
    };
    // rows 0 .. NPEEL-1 open the accumulators; the others each complete one output row (S=1) / one per pair (S=2).  The loop is unrolled by
    // the least common multiple of PF (static prefetch register) and 2 (static LDS buffer / S=2 row parity).
    constexpr int NPEEL = (S == 1) ? 2 : 1;
    constexpr int UN = (S == 1) ? 6 : 2;
    row(0, pf[0], std::integral_constant<int, 1>{});
    if (S == 1) row(1, pf[1], std::integral_constant<int, 2>{});
    for (int it = NPEEL; it < nrows; it += UN) {
#pragma unroll
        for (int j = 0; j < UN; ++j) {
            const int r = it + j;
            if (r < nrows) row(r, pf[(NPEEL + j) % PF], std::integral_constant<int, 0>{}, &yq[j % PF]);       // uniform
        }
    }
    if ((MODE == 1 && fu.stat) || MODE == 3) {
        // reduction over the PX threads that share a channel quad, in double; this workgroup's slice of partial row (n, strip, chunk)
        const long long blk = ((long long)n * strips + sx) * chunks + ch;
        if (!live) { s1 = dw_zero(); s2 = dw_zero(); }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            __syncthreads();
            if (comp) red[tid] = dw_unpk(v == 0 ? s1 : s2);
            __syncthreads();
            if (tid < CQB) {
                double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
                for (int j = 0; j < PX; ++j) {
                    const float4 tt = red[j * CQB + tid];
                    d0 += tt.x; d1 += tt.y; d2 += tt.z; d3 += tt.w;
                }
                double* o = (MODE == 3 ? fu.bw.part : fu.stat) + (blk * 2 + v) * C + (cb * CQB + tid) * 4;
                o[0] = d0; o[1] = d1; o[2] = d2; o[3] = d3;
            }
        }
    }
}

// Depthwise WEIGHT gradient in the same row-sliding form: dw[ky][kx][c] = sum over output pixels of x[oy*S+ky-pt][ox*S+kx-pl][c] * dy[oy][ox][c].
// A thread (output column, channel quad) walks down the input rows of its chunk; an input row meets the dy rows it feeds (S=1: oy = iy+1-ky,
// three of them, kept in a sliding register window; S=2: row 2m -> ky=0 of oy=m and ky=2 of oy=m-1, row 2m+1 -> ky=1 of oy=m), the
// neighbouring columns come through the LDS row.  Each chunk counts exactly its own output rows (dy of other rows is loaded as zero).
// The producing BatchNorm + activation is applied to x once per element on the way in (MODE-1 form of the forward).  Nine packed
// accumulators per thread, reduced over the workgroup's columns in double into ONE row of partials per (image, strip, chunk),
// finished by colreduce_finish<FinD2F> in a fixed order (deterministic).
template <int S, int CQB, bool R6>
__global__ __launch_bounds__(256, 4) void dw_rows_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, double* __restrict__ part,
                                                               int H, int W, int C, int Ho, int Wo, int strips, int chunks, int rc, int ncb,
                                                               unsigned xcd_tiles, DwAffine in)
{
    constexpr int PX = 224 / CQB;
    constexpr int NE = (S == 1) ? PX + 2 : PX + 1;
    constexpr int NHALO = (S == 1) ? 2 : 1;
    constexpr int NL = (S == 1) ? 1 : 2;
    constexpr int PF = 2;
    __shared__ float4 rowbuf[2][NE * CQB];
    __shared__ float4 red[224];
    const int tid = threadIdx.x;
    unsigned b = blockIdx.x;
    if (xcd_tiles) b = (b & 7u) * xcd_tiles + (b >> 3);
    const int ch = b % (unsigned)chunks;
    unsigned t = b / (unsigned)chunks;
    const int sx = t % (unsigned)strips;
    t /= (unsigned)strips;
    const int cb = t % (unsigned)ncb;
    const int n = t / (unsigned)ncb;
    const int x0 = sx * PX;
    const int y0 = ch * rc, y1 = min(y0 + rc, Ho);
    const bool comp = tid < 224;
    int px, q, e_w, col;
    bool loader;
    if (comp) {
        px = tid / CQB; q = tid % CQB;
        if (S == 1) { e_w = px + 1; col = x0 + px; } else { e_w = px; col = 2 * (x0 + px); }
        loader = true;
    } else {
        const int k = tid - 224;
        px = 0; q = k % CQB;
        const int side = k / CQB;
        loader = side < NHALO && CQB < 32;
        if (S == 1) { e_w = side ? PX + 1 : 0; col = side ? x0 + PX : x0 - 1; } else { e_w = PX; col = 2 * (x0 + PX); }
    }
    const int c = (cb * CQB + q) * 4;
    const bool colin = loader && col >= 0 && col < W;
    const bool col2in = comp && (S == 2) && (col + 1) < W;
    const int ox = x0 + px;
    const bool live = comp && ox < Wo;
    dw_f4p sc0 = dw_pk(make_float4(1.f, 1.f, 1.f, 1.f)), sh0 = dw_zero(), sc1 = sc0, sh1 = sh0;
    float lo = -INFINITY, hi = INFINITY;
    const bool inaff = in.scale != nullptr;
    if (inaff) {
        sc0 = dw_pk(ld4g(in.scale + c)); sh0 = dw_pk(ld4g(in.shift + c));
        lo = in.act == MYOLO_ACT_NONE ? -INFINITY : 0.f;
        hi = in.act == MYOLO_ACT_RELU6 ? 6.f : INFINITY;
        sc1 = sc0; sh1 = sh0;
        if (!colin) { sc0 = dw_zero(); sh0 = dw_zero(); }
        if (!col2in) { sc1 = dw_zero(); sh1 = dw_zero(); }
    }
    if (CQB == 32 && !comp) {
        const int k = tid - 224;
        rowbuf[0][k] = f4zero(); rowbuf[1][k] = f4zero();
        rowbuf[0][(NE - 1) * CQB + k] = f4zero(); rowbuf[1][(NE - 1) * CQB + k] = f4zero();
    }
    const int row0 = (S == 1) ? y0 - 1 : 2 * y0;
    const int nrows = (S == 1) ? (y1 - y0 + 2) : (2 * (y1 - y0) + 1);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc((void*)(x + (long long)n * H * W * C), 0, H * W * C * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc((void*)(dy + (long long)n * Ho * Wo * C), 0, Ho * Wo * C * 4, 0x00020000);
    const unsigned off0 = colin ? (unsigned)(col * C + c) * 4u : DW_OOB;
    const unsigned off1 = col2in ? off0 + (unsigned)C * 4u : DW_OOB;
    const unsigned goff = live ? (unsigned)(ox * C + c) * 4u : DW_OOB;
    const int rstride = W * C * 4, gstride = Wo * C * 4;
    // what input row r brings with it: its x value(s) and the ONE new dy row it needs (S=1: oy = iy+1; S=2: even rows 2m -> oy = m)
    auto fetch = [&](int r, float4* dx_, float4& dg) {
        const int iy = row0 + r;
        const bool ok = r < nrows && iy >= 0 && iy < H;
        const unsigned ro = ok ? (unsigned)(iy * rstride) : DW_OOB;
        dx_[0] = dw_bufld(rx, off0 + ro);
        if (NL == 2) dx_[NL - 1] = dw_bufld(rx, off1 + ro);
        const int oy = (S == 1) ? iy + 1 : (iy >> 1);
        const bool need = (S == 1) || ((r & 1) == 0);
        const bool gok = need && r < nrows && oy >= y0 && oy < y1;        // only this chunk's own output rows are counted
        dg = dw_bufld(rg, gok ? goff + (unsigned)(oy * gstride) : DW_OOB);
    };
    float4 pfx[PF][NL], pfg[PF];
#pragma unroll
    for (int j = 0; j < PF; ++j) fetch(j, pfx[j], pfg[j]);
    dw_f4p acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = dw_zero();
    dw_f4p gA = dw_zero(), gB = dw_zero();          // S=1: dy[iy-1], dy[iy] (own column); S=2: gB = dy[m] of the current pair, gA = dy[m-1]
    auto row = [&](int r, float4* curx, float4& curg) {
        dw_f4p v0 = dw_pk(curx[0]), v1 = dw_pk(curx[NL - 1]);
        const dw_f4p gC = dw_pk(curg);
        fetch(r + PF, curx, curg);
        const int iy = row0 + r;
        if (inaff) {
            if (iy >= 0 && iy < H) {
                v0 = dw_affine_pk<R6>(v0, sc0, sh0, lo, hi);
                if (NL == 2) v1 = dw_affine_pk<R6>(v1, sc1, sh1, lo, hi);
            }
        }
        float4* buf = rowbuf[r & 1];
        if (loader) buf[e_w * CQB + q] = dw_unpk(v0);
        __syncthreads();
        if (S == 1) {
            const dw_f4p l = dw_pk(buf[px * CQB + q]), rt = dw_pk(buf[(px + 2) * CQB + q]);
            // input row iy is tap ky of output row iy+1-ky: ky = 0 -> dy[iy+1] (just loaded), 1 -> dy[iy], 2 -> dy[iy-1]
            acc[0] = dw_fma(l, gC, acc[0]); acc[1] = dw_fma(v0, gC, acc[1]); acc[2] = dw_fma(rt, gC, acc[2]);
            acc[3] = dw_fma(l, gB, acc[3]); acc[4] = dw_fma(v0, gB, acc[4]); acc[5] = dw_fma(rt, gB, acc[5]);
            acc[6] = dw_fma(l, gA, acc[6]); acc[7] = dw_fma(v0, gA, acc[7]); acc[8] = dw_fma(rt, gA, acc[8]);
            gA = gB; gB = gC;
        } else {
            const dw_f4p rt = dw_pk(buf[(px + 1) * CQB + q]);
            if ((r & 1) == 0) {                     // row 2m: ky = 0 of output m (dy just loaded), ky = 2 of output m-1
                gA = gB; gB = gC;
                acc[0] = dw_fma(v0, gB, acc[0]); acc[1] = dw_fma(v1, gB, acc[1]); acc[2] = dw_fma(rt, gB, acc[2]);
                acc[6] = dw_fma(v0, gA, acc[6]); acc[7] = dw_fma(v1, gA, acc[7]); acc[8] = dw_fma(rt, gA, acc[8]);
            } else {                                // row 2m+1: ky = 1 of output m
                acc[3] = dw_fma(v0, gB, acc[3]); acc[4] = dw_fma(v1, gB, acc[4]); acc[5] = dw_fma(rt, gB, acc[5]);
            }
        }
    };
    for (int it = 0; it < nrows; it += PF) {
#pragma unroll
        for (int j = 0; j < PF; ++j) {
            const int r = it + j;
            if (r < nrows) row(r, pfx[j], pfg[j]);
        }
    }
    const long long blk = ((long long)n * strips + sx) * chunks + ch;
#pragma unroll
    for (int v = 0; v < 9; ++v) {
        __syncthreads();
        if (comp) red[tid] = dw_unpk(acc[v]);       // (dead and halo lanes only ever multiplied by dy = 0)
        __syncthreads();
        if (tid < CQB) {
            double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
            for (int j = 0; j < PX; ++j) {
                const float4 tt = red[j * CQB + tid];
                d0 += tt.x; d1 += tt.y; d2 += tt.z; d3 += tt.w;
            }
            double* o = part + (blk * 9 + v) * C + (cb * CQB + tid) * 4;
            o[0] = d0; o[1] = d1; o[2] = d2; o[3] = d3;
        }
    }
}

template <int S, int CQB>
static void dw_rows_launch(const DwRowsGeom& g, const float* x, const float* w, float* y, int H, int W, int C, DwAffine af, DwFuse fu, hipStream_t s,
                           bool flip = false)
{
    const unsigned xcd = (g.tiles % 8 == 0 && g.tiles >= 64 && !(g_myolo_opt.tune0 & 8)) ? (unsigned)(g.tiles / 8) : 0u;
    const int flags = ((g_myolo_opt.tune0 & 16) ? 2 : 0) | ((g_myolo_opt.tune0 & 32) ? 4 : 0) | (flip ? 8 : 0);      // 2 / 4: timing-only ablations (no loads / no stores)
#define DW_ROWS_GO(MODE, R6) hipLaunchKernelGGL((dw_rows_kernel<S, CQB, MODE, R6>), dim3((unsigned)g.tiles), dim3(256), 0, s, x, w, y, H, W, C, H / S, W / S, g.strips, g.chunks, g.rc, g.ncb, xcd, flags, af, fu)
    if (fu.bw.x) { if constexpr (S == 1) DW_ROWS_GO(3, false); }
    else if (fu.in.scale || fu.stat) { if (fu.in.scale && fu.in.act == MYOLO_ACT_RELU6) DW_ROWS_GO(1, true); else DW_ROWS_GO(1, false); }
    else if (af.scale) { if (af.act == MYOLO_ACT_RELU6) DW_ROWS_GO(2, true); else DW_ROWS_GO(2, false); }
    else DW_ROWS_GO(0, false);
#undef DW_ROWS_GO
}

// dx[iy,ix] = sum_{ky,kx} dy[(iy+pt-ky)/S, (ix+pl-kx)/S] * w[ky,kx]   (when divisible and in range)
template <int S>
__global__ __launch_bounds__(256) void dw_bwd_data_kernel(const float* __restrict__ dy, const float* __restrict__ w,
                                                          float* __restrict__ dx, int N, int H, int W, int C, int Ho, int Wo)
{
    const int pt = (S == 1) ? 1 : 0;
    const int cq = C / 4;
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (unsigned)(W * cq)) return;
    const int ix = e / (unsigned)cq;
    const int c = (e - ix * cq) * 4;
    const int iy = blockIdx.y, n = blockIdx.z;
    float4 acc = f4zero();
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = iy + pt - ky;
        if (ty < 0 || (ty % S) != 0) continue;
        const int oy = ty / S;
        if (oy >= Ho) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = ix + pt - kx;
            if (tx < 0 || (tx % S) != 0) continue;
            const int ox = tx / S;
            if (ox >= Wo) continue;
            acc = f4fma(ld4g(dy + ((((long long)n * Ho + oy) * Wo) + ox) * C + c), ld4g(w + (ky * 3 + kx) * C + c), acc);
        }
    }
    st4g(dx + ((((long long)n * H + iy) * W) + ix) * C + c, acc);
}

// stride 2, even H and W (padding 0 before / 1 after, as TF SAME gives): the thread of output-gradient pixel (r, q) writes the 2 x 2 block of dx at
// (2r.., 2q..) from dy[r-1..r][q-1..q] -- one 16-byte load per store instead of 2.25 with a branch ladder (dw_bwd_data_kernel<2>: 2.0 TB/s).
template <bool BW>
__global__ __launch_bounds__(256) void dw_bwd_data_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                             int Ho, int Wo, int C, long long total, DwBnBwd bw)
{
    __shared__ float4 red[256];
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    const bool in = e < total;
    if (!BW && !in) return;
    const int cq = C >> 2;
    float4 s1 = f4zero(), s2 = f4zero();
    if (in) {
    const unsigned eu = (unsigned)(e % ((long long)Wo * cq));
    const long long row = e / ((long long)Wo * cq);            // n * Ho + r
    const int q = eu / (unsigned)cq, c = (eu - q * cq) * 4;
    const int r = (int)(row % Ho);
    const float* p = dy + (row * Wo + q) * C + c;
    const float4 z = f4zero();
    const float4 g11 = ld4g(p);
    const float4 g10 = q > 0 ? ld4g(p - C) : z;
    const float4 g01 = r > 0 ? ld4g(p - (long long)Wo * C) : z;
    const float4 g00 = (r > 0 && q > 0) ? ld4g(p - (long long)Wo * C - C) : z;
    float4 wk[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wk[k] = ld4g(w + k * C + c);
    const int W = 2 * Wo;
    const long long o0 = ((row * 2) * W + 2 * q) * (long long)C + c;
    float* o = dx + o0;
    // same order of accumulation as dw_bwd_data_kernel<2> (ky = 0, 1, 2; kx = 0, 1, 2): bit-identical results
    const float4 d00 = f4fma(g00, wk[8], f4fma(g01, wk[6], f4fma(g10, wk[2], f4fma(g11, wk[0], z))));
    const float4 d01 = f4fma(g01, wk[7], f4fma(g11, wk[1], z));
    const float4 d10 = f4fma(g10, wk[5], f4fma(g11, wk[3], z));
    const float4 d11 = f4fma(g11, wk[4], z);
    st4g(o, d00);
    st4g(o + C, d01);
    st4g(o + (long long)W * C, d10);
    st4g(o + (long long)W * C + C, d11);
    if (BW) {
        // the 2 x 2 block of dx is the gradient reaching a training-mode BatchNorm whose pre-BN tensor is bw.x (DwBnBwd): its backward sums
        const float4 sc = ld4g(bw.scale + c), sh = ld4g(bw.shift + c), mu = ld4g(bw.mean + c), vr = ld4g(bw.var + c);
        const float4 rs = make_float4(rsqrtf(vr.x + BN_EPS_F), rsqrtf(vr.y + BN_EPS_F), rsqrtf(vr.z + BN_EPS_F), rsqrtf(vr.w + BN_EPS_F));
        const float* xq = bw.x + o0;
        const float4 x00 = ld4g(xq), x01 = ld4g(xq + C), x10 = ld4g(xq + (long long)W * C), x11 = ld4g(xq + (long long)W * C + C);
        auto add = [&](float4 g, float4 v) {
            float dz, xh;
            dz = g.x * actmask(fmaf(v.x, sc.x, sh.x), bw.act); xh = (v.x - mu.x) * rs.x; s1.x += dz; s2.x = fmaf(dz, xh, s2.x);
            dz = g.y * actmask(fmaf(v.y, sc.y, sh.y), bw.act); xh = (v.y - mu.y) * rs.y; s1.y += dz; s2.y = fmaf(dz, xh, s2.y);
            dz = g.z * actmask(fmaf(v.z, sc.z, sh.z), bw.act); xh = (v.z - mu.z) * rs.z; s1.z += dz; s2.z = fmaf(dz, xh, s2.z);
            dz = g.w * actmask(fmaf(v.w, sc.w, sh.w), bw.act); xh = (v.w - mu.w) * rs.w; s1.w += dz; s2.w = fmaf(dz, xh, s2.w);
        };
        add(d00, x00); add(d01, x01); add(d10, x10); add(d11, x11);
    }
    }
    if (BW) {
        // 256 consecutive elements = 256 / cq pixels x cq channel quads (256 % cq == 0, checked by the launcher): thread t's quad is t % cq
        const int tid = threadIdx.x;
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            __syncthreads();
            red[tid] = v == 0 ? s1 : s2;
            __syncthreads();
            if (tid < cq) {
                double d0 = 0, d1 = 0, d2 = 0, d3 = 0;
                for (int j = tid; j < 256; j += cq) {
                    const float4 tt = red[j];
                    d0 += tt.x; d1 += tt.y; d2 += tt.z; d3 += tt.w;
                }
                double* o = bw.part + ((long long)blockIdx.x * 2 + v) * C + tid * 4;
                o[0] = d0; o[1] = d1; o[2] = d2; o[3] = d3;
            }
        }
    }
}

// dw[k][c] = sum over output pixels of x[shifted] * dy
struct OpDwDw {
    static constexpr int NV = 9;
    const float* x;
    const float* dy;
    int H, W, C, Ho, Wo, S;
    DwAffine in;         // scale != nullptr: x is the producing layer's pre-BN output, normalised + activated on load (see DwFuse)
    __device__ void operator()(long long r, int c, float4* acc) const
    {
        float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = f4zero();
        if (in.scale) { isc = ld4g(in.scale + c); ish = ld4g(in.shift + c); }
        const int pt = (S == 1) ? 1 : 0;
        const unsigned ru = (unsigned)r;                 // rows < 2^31 (checked by the launcher)
        const unsigned t = ru / (unsigned)Wo;
        const int ox = (int)(ru - t * Wo);
        const int n = (int)(t / (unsigned)Ho);
        const int oy = (int)(t - (unsigned)n * Ho);
        const float4 g = ld4g(dy + r * C + c);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * S + ky - pt;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * S + kx - pt;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) {
                    float4 v = ld4g(x + (((long long)n * H + iy) * W + ix) * C + c);
                    if (in.scale) {
                        v.x = actf(fmaf(v.x, isc.x, ish.x), in.act); v.y = actf(fmaf(v.y, isc.y, ish.y), in.act);
                        v.z = actf(fmaf(v.z, isc.z, ish.z), in.act); v.w = actf(fmaf(v.w, isc.w, ish.w), in.act);
                    }
                    acc[ky * 3 + kx] = f4fma(v, g, acc[ky * 3 + kx]);
                }
            }
        }
    }
};


// Depthwise weight gradient with the forward kernel's access pattern: a thread owns a TH x TW block of OUTPUT pixels of one channel
// quad, keeps their dy in registers and walks down the (TH-1)*S+3 input rows of the block once (each x element is loaded once per
// block instead of once per tap, as the generic OpDwDw column reduction does): acc[ky][kx] += x[r*S+ky][j*S+kx] * dy[r][j].
// The 9 partial sums of a workgroup's threads that share a channel quad are combined through LDS (double from there on) into one row
// of partials per workgroup, finished by colreduce_finish<FinD2F> in a fixed order.  `in` != none: x is the producing layer's pre-BN
// output, normalised + activated on load (zero padding stays zero).  Needs 256 % (C/4) == 0.
template <int S, int TW, int TH>
__global__ __launch_bounds__(256, 2) void dw_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, double* __restrict__ part,
                                                       int N, int H, int W, int C, int Ho, int Wo, DwAffine in, int zb)
{
    constexpr int NC = (TW - 1) * S + 3;
    constexpr int NR = (TH - 1) * S + 3;
    const int pt = (S == 1) ? 1 : 0, plft = (S == 1) ? 1 : 0;
    const int cq = C / 4;
    const int wtiles = (Wo + TW - 1) / TW;
    const unsigned e = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = e < (unsigned)(wtiles * cq);
    const int wt = live ? e / (unsigned)cq : 0;
    const int c = live ? (e - wt * cq) * 4 : 0;
    const int ox0 = wt * TW;
    float4 isc = make_float4(1.f, 1.f, 1.f, 1.f), ish = f4zero();
    if (in.scale) { isc = ld4g(in.scale + c); ish = ld4g(in.shift + c); }
    float4 acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = f4zero();
    // a workgroup walks (image, row group) units with stride gridDim.y (zb = number of units): the workgroup-level reduction of the 9 sums
    // and its row of partials (9 C doubles) are paid once per workgroup, not once per 8 output pixels
    const int rgroups = (Ho + TH - 1) / TH;
#pragma unroll 1
    for (int u = blockIdx.y; u < zb; u += gridDim.y) {
    const int n = u / rgroups;
    const int oy0 = (u - n * rgroups) * TH;
    float4 g[TH][TW];
#pragma unroll
    for (int r = 0; r < TH; ++r)
#pragma unroll
        for (int j = 0; j < TW; ++j) {
            const bool ok = live && oy0 + r < Ho && ox0 + j < Wo;
            g[r][j] = ok ? ld4g(dy + ((((long long)n * Ho + oy0 + r) * Wo) + ox0 + j) * C + c) : f4zero();
        }
#pragma unroll
    for (int ri = 0; ri < NR; ++ri) {
        const int iy = oy0 * S + ri - pt;
        const bool rowin = live && iy >= 0 && iy < H;
        const float* rowp = x + (((long long)n * H + (rowin ? iy : 0)) * W) * C + c;
        float4 col[NC];
#pragma unroll
        for (int k = 0; k < NC; ++k) {
            const int ix = ox0 * S + k - plft;
            const bool inb = rowin && ix >= 0 && ix < W;
            col[k] = inb ? ld4g(rowp + (long long)ix * C) : f4zero();
            if (in.scale && inb) {
                col[k].x = actf(fmaf(col[k].x, isc.x, ish.x), in.act); col[k].y = actf(fmaf(col[k].y, isc.y, ish.y), in.act);
                col[k].z = actf(fmaf(col[k].z, isc.z, ish.z), in.act); col[k].w = actf(fmaf(col[k].w, isc.w, ish.w), in.act);
            }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            if ((ri - ky) >= 0 && ((ri - ky) % S) == 0 && (ri - ky) / S < TH) {
                const int r = (ri - ky) / S;
#pragma unroll
                for (int j = 0; j < TW; ++j)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) acc[ky * 3 + kx] = f4fma(col[j * S + kx], g[r][j], acc[ky * 3 + kx]);
            }
        }
    }
    }
    __shared__ float4 red[256];
    const int tid = threadIdx.x, cl = cq < 256 ? cq : 256, pl = 256 / cl, cl_i = tid % cl;
    const long long blk = blockIdx.x + (long long)gridDim.x * blockIdx.y;
#pragma unroll
    for (int v = 0; v < 9; ++v) {
        __syncthreads();
        red[tid] = acc[v];
        __syncthreads();
        if (tid < cl) {
            double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int j = 0; j < pl; ++j) {
                const float4 t = red[j * cl + cl_i];
                a0 += t.x; a1 += t.y; a2 += t.z; a3 += t.w;
            }
            double* o = part + (blk * 9 + v) * C + cl_i * 4;
            o[0] = a0; o[1] = a1; o[2] = a2; o[3] = a3;
        }
    }
}

// ---------------------------------------------------------------------------------------
// crop_and_resize (ROIAlign).  One channel-quad lane per output element quad; the 64 lanes of a
// wave cover 256 channels of one output pixel, so the four corner reads and the store are each
// one contiguous 1 KiB wave access.  Same float op order as the TF kernel.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ bool crop_coord(float lo, float hi, int size, int crop, int idx, float& in)
{
    if (crop > 1) {
        const float scale = (hi - lo) * (float)(size - 1) / (float)(crop - 1);
        in = lo * (float)(size - 1) + (float)idx * scale;
    } else {
        in = 0.5f * (lo + hi) * (float)(size - 1);
    }
    return !(in < 0.f || in > (float)(size - 1));
}

__global__ __launch_bounds__(256) void crop_fwd_kernel(const float* __restrict__ img, const float* __restrict__ boxes,
                                                       const int32_t* __restrict__ bind, float* __restrict__ out,
                                                       int H, int W, int C, int nb, int ch, int cw)
{
    // grid: x = crop row py, y = box.  The y coordinate is block-uniform, the x coordinate is uniform per
    // group of C/4 lanes; no integer division by runtime 64-bit values.
    const int b = blockIdx.y, py = blockIdx.x;
    const int cq = C / 4;
    const float4 bx = ld4g(boxes + (long long)b * 4);      // y1,x1,y2,x2
    float iny;
    const bool vy = crop_coord(bx.x, bx.z, H, ch, py, iny);
    const int ty = (int)floorf(iny), by = (int)ceilf(iny);
    const float wy = iny - (float)ty;
    const float* base = img + (long long)bind[b] * H * W * C;
    float* orow = out + ((long long)b * ch + py) * cw * C;
    const unsigned total = (unsigned)(cw * cq);
    for (unsigned e = threadIdx.x; e < total; e += blockDim.x) {
        const int px = e / (unsigned)cq;
        const int c = (e - px * cq) * 4;
        float inx;
        const bool vx = crop_coord(bx.y, bx.w, W, cw, px, inx);
        float4 o = f4zero();
        if (vy && vx) {
            const int lx = (int)floorf(inx), rx = (int)ceilf(inx);
            const float wx = inx - (float)lx;
            const float4 tl = ld4g(base + ((long long)ty * W + lx) * C + c), tr = ld4g(base + ((long long)ty * W + rx) * C + c);
            const float4 bl = ld4g(base + ((long long)by * W + lx) * C + c), br = ld4g(base + ((long long)by * W + rx) * C + c);
            float top, bot;
            top = tl.x + (tr.x - tl.x) * wx; bot = bl.x + (br.x - bl.x) * wx; o.x = top + (bot - top) * wy;
            top = tl.y + (tr.y - tl.y) * wx; bot = bl.y + (br.y - bl.y) * wx; o.y = top + (bot - top) * wy;
            top = tl.z + (tr.z - tl.z) * wx; bot = bl.z + (br.z - bl.z) * wx; o.z = top + (bot - top) * wy;
            top = tl.w + (tr.w - tl.w) * wx; bot = bl.w + (br.w - bl.w) * wx; o.w = top + (bot - top) * wy;
        }
        st4g_nt(orow + (long long)px * C + c, o);
    }
}

__device__ __forceinline__ void atomic_add4(float* p, float4 v)
{
    atomicAdd(p + 0, v.x); atomicAdd(p + 1, v.y); atomicAdd(p + 2, v.z); atomicAdd(p + 3, v.w);
}

__global__ __launch_bounds__(256) void crop_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ boxes,
                                                       const int32_t* __restrict__ bind, float* __restrict__ dimg,
                                                       int H, int W, int C, int nb, int ch, int cw)
{
    const int cq = C / 4;
    const long long total = (long long)nb * ch * cw * cq;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % cq) * 4;
        long long t = i / cq;
        const int px = (int)(t % cw);
        t /= cw;
        const int py = (int)(t % ch);
        const int b = (int)(t / ch);
        const float4 bx = ld4g(boxes + (long long)b * 4);
        float iny, inx;
        const bool vy = crop_coord(bx.x, bx.z, H, ch, py, iny);
        const bool vx = crop_coord(bx.y, bx.w, W, cw, px, inx);
        if (!(vy && vx)) continue;
        const float4 g = ld4g(dout + i * 4);
        const int ty = (int)floorf(iny), by = (int)ceilf(iny);
        const int lx = (int)floorf(inx), rx = (int)ceilf(inx);
        const float wy = iny - (float)ty, wx = inx - (float)lx;
        float* base = dimg + (long long)bind[b] * H * W * C + c;
        const float a = (1.f - wy) * (1.f - wx), bq = (1.f - wy) * wx, cc = wy * (1.f - wx), d = wy * wx;
        atomic_add4(base + ((long long)ty * W + lx) * C, make_float4(g.x * a, g.y * a, g.z * a, g.w * a));
        atomic_add4(base + ((long long)ty * W + rx) * C, make_float4(g.x * bq, g.y * bq, g.z * bq, g.w * bq));
        atomic_add4(base + ((long long)by * W + lx) * C, make_float4(g.x * cc, g.y * cc, g.z * cc, g.w * cc));
        atomic_add4(base + ((long long)by * W + rx) * C, make_float4(g.x * d, g.y * d, g.z * d, g.w * d));
    }
}

// Gather form of the ROIAlign backward for the layout the mask head uses: boxes grouped by image,
// R per image (box b*R+r belongs to image b).  One lane per (pixel, channel quad); it walks the R boxes
// of its image and, for each, the crop samples whose top/bottom (left/right) neighbour is this pixel,
// summing weight * dout in a fixed order -- no atomics, no memset, bit-reproducible.  With C = 256 a
// wave is exactly one pixel, so the box walk is wave-uniform.
__global__ __launch_bounds__(256) void crop_bwd_grouped_kernel(const float* __restrict__ dout, const float* __restrict__ boxes,
                                                               float* __restrict__ dimg, int B, int H, int W, int C, int R,
                                                               int ch, int cw)
{
    const int cq = C / 4;
    const long long total = (long long)B * H * W * cq;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < total; i += stride) {
        const int c = (int)(i % cq) * 4;
        long long t = i / cq;
        const int x = (int)(t % W);
        t /= W;
        const int y = (int)(t % H);
        const int b = (int)(t / H);
        float4 acc = f4zero();
        const float fy = (float)y, fx = (float)x;
        for (int r = 0; r < R; ++r) {
            const long long bi = (long long)b * R + r;
            const float4 bx = ld4g(boxes + bi * 4);          // y1,x1,y2,x2
            // sample k sits at c0 + k*s (same float expressions as the forward kernel); only samples within
            // one pixel of (y, x) can touch it.  Conservative index window first, exact test inside.
            const float sy = (ch > 1) ? (bx.z - bx.x) * (float)(H - 1) / (float)(ch - 1) : 0.f;
            const float sx = (cw > 1) ? (bx.w - bx.y) * (float)(W - 1) / (float)(cw - 1) : 0.f;
            const float y0 = (ch > 1) ? bx.x * (float)(H - 1) : 0.5f * (bx.x + bx.z) * (float)(H - 1);
            const float x0 = (cw > 1) ? bx.y * (float)(W - 1) : 0.5f * (bx.y + bx.w) * (float)(W - 1);
            int pya = 0, pyb = ch - 1, pxa = 0, pxb = cw - 1;
            if (fabsf(sy) > 1e-6f) {
                const float a = (fy - 1.f - y0) / sy, c = (fy + 1.f - y0) / sy;
                pya = max(0, (int)floorf(fminf(a, c)) - 1);
                pyb = min(ch - 1, (int)ceilf(fmaxf(a, c)) + 1);
            } else if (fabsf(y0 - fy) > 1.5f) continue;
            if (fabsf(sx) > 1e-6f) {
                const float a = (fx - 1.f - x0) / sx, c = (fx + 1.f - x0) / sx;
                pxa = max(0, (int)floorf(fminf(a, c)) - 1);
                pxb = min(cw - 1, (int)ceilf(fmaxf(a, c)) + 1);
            } else if (fabsf(x0 - fx) > 1.5f) continue;
            for (int py = pya; py <= pyb; ++py) {
                float iny;
                if (!crop_coord(bx.x, bx.z, H, ch, py, iny)) continue;
                const int ty = (int)floorf(iny), by = (int)ceilf(iny);
                if (ty != y && by != y) continue;
                const float ly = iny - (float)ty;
                const float wyv = (ty == y ? (1.f - ly) : 0.f) + (by == y ? ly : 0.f);
                for (int px = pxa; px <= pxb; ++px) {
                    float inx;
                    if (!crop_coord(bx.y, bx.w, W, cw, px, inx)) continue;
                    const int lx = (int)floorf(inx), rx = (int)ceilf(inx);
                    if (lx != x && rx != x) continue;
                    const float lxw = inx - (float)lx;
                    const float wxv = (lx == x ? (1.f - lxw) : 0.f) + (rx == x ? lxw : 0.f);
                    const float wgt = wyv * wxv;
                    const float4 g = ld4g(dout + ((bi * ch + py) * cw + px) * C + c);
                    acc.x = fmaf(g.x, wgt, acc.x); acc.y = fmaf(g.y, wgt, acc.y);
                    acc.z = fmaf(g.z, wgt, acc.z); acc.w = fmaf(g.w, wgt, acc.w);
                }
            }
        }
        st4g(dimg + i * 4, acc);
    }
}

// Same gather with the per-box terms (origin, step, reciprocal step) formed once per workgroup in LDS instead of once
// per (pixel, box): with C = 256 a 256-thread workgroup is 4 consecutive pixels of ONE image (H*W % 4 == 0), so its
// R boxes are shared.  The candidate window only has to be conservative (the exact floor/ceil test inside uses the
// forward kernel's expressions), so multiplying by a reciprocal instead of dividing changes no result.
template <int TP>        // the workgroup's pixel tile is TP x TP (one wave per pixel): 2 -> 256 threads, 4 -> 1024 threads
__global__ __launch_bounds__(64 * TP * TP) void crop_bwd_grouped_lds_kernel(const float* __restrict__ dout, const float* __restrict__ boxes,
                                                                   float* __restrict__ dimg, int H, int W, int R, int ch, int cw,
                                                                   unsigned xcd_tiles, int quad)
{
    extern __shared__ __attribute__((aligned(16))) float sp[];      // [R][8]: y1 x1 y2 x2 | y0 1/sy x0 1/sx  (1/s = 0: degenerate)
    constexpr int C = 256, cq = 64;
    // Which four pixels a workgroup owns decides how often a crop sample is fetched: a sample touches up to 2 x 2 feature pixels.  With four
    // CONSECUTIVE pixels of a row per workgroup and workgroup b on XCD b % 8, the two feature rows of a sample were served by different
    // XCDs' L2s: rocprofv3 counted 2.6 GB of fetches for 0.94 GB of gradient crops (profiles/r4_pmc_trunk.json).  Now: a 2 x 2 pixel quad
    // per workgroup (the four waves share most samples through L1) and XCD-contiguous workgroup order (every XCD owns whole images, so
    // vertically adjacent quads share its L2).  The sums themselves are unchanged (each wave still walks its pixel's boxes in order).
    unsigned bid = blockIdx.x;
    if (xcd_tiles) bid = (bid & 7u) * xcd_tiles + (bid >> 3);
    long long pix;
    int b, y, x;
    if (quad) {
        const unsigned qw = (unsigned)W / TP, qpi = ((unsigned)H / TP) * qw;
        b = (int)(bid / qpi);
        const unsigned qr = bid - (unsigned)b * qpi;
        const unsigned qy = qr / qw, qx = qr - qy * qw;
        const int wv = threadIdx.x >> 6;
        y = TP * (int)qy + (wv / TP);
        x = TP * (int)qx + (wv % TP);
        pix = ((long long)b * H + y) * W + x;
    } else {
        pix = (long long)bid * 4 + (threadIdx.x >> 6);
        b = (int)(pix / ((long long)H * W));
        const int rem = (int)(pix - (long long)b * H * W);
        y = rem / W; x = rem - y * W;
    }
    const int c = (threadIdx.x & 63) * 4;
    for (int r = threadIdx.x; r < R; r += 64 * TP * TP) {
        const float4 bx = ld4g(boxes + ((long long)b * R + r) * 4);
        const float sy = (ch > 1) ? (bx.z - bx.x) * (float)(H - 1) / (float)(ch - 1) : 0.f;
        const float sx = (cw > 1) ? (bx.w - bx.y) * (float)(W - 1) / (float)(cw - 1) : 0.f;
        const float y0 = (ch > 1) ? bx.x * (float)(H - 1) : 0.5f * (bx.x + bx.z) * (float)(H - 1);
        const float x0 = (cw > 1) ? bx.y * (float)(W - 1) : 0.5f * (bx.y + bx.w) * (float)(W - 1);
        *reinterpret_cast<float4*>(&sp[r * 8]) = bx;
        *reinterpret_cast<float4*>(&sp[r * 8 + 4]) =
            make_float4(y0, fabsf(sy) > 1e-6f ? 1.f / sy : 0.f, x0, fabsf(sx) > 1e-6f ? 1.f / sx : 0.f);
    }
    __syncthreads();
    float4 acc = f4zero();
    const float fy = (float)y, fx = (float)x;
    // A wave is ONE feature pixel (64 lanes = the 256 channels), so the candidate test of a box is the same in every lane:
    // the lanes test 64 different boxes at once, a ballot collects the boxes whose sample window can touch this pixel, and only
    // those are walked (ascending box order: the summation order, hence the result, does not depend on the lane assignment).
    const int lane = threadIdx.x & 63;
    for (int r0 = 0; r0 < R; r0 += 64) {
        const int rt = r0 + lane;
        bool hit = false;
        if (rt < R) {
            const float4 q = *reinterpret_cast<const float4*>(&sp[rt * 8 + 4]);      // y0, 1/sy, x0, 1/sx
            int pya = 0, pyb = ch - 1, pxa = 0, pxb = cw - 1;
            hit = true;
            if (q.y != 0.f) {
                const float a = (fy - 1.f - q.x) * q.y, cc = (fy + 1.f - q.x) * q.y;
                pya = max(0, (int)floorf(fminf(a, cc)) - 2);
                pyb = min(ch - 1, (int)ceilf(fmaxf(a, cc)) + 2);
            } else if (fabsf(q.x - fy) > 1.5f) hit = false;
            if (q.w != 0.f) {
                const float a = (fx - 1.f - q.z) * q.w, cc = (fx + 1.f - q.z) * q.w;
                pxa = max(0, (int)floorf(fminf(a, cc)) - 2);
                pxb = min(cw - 1, (int)ceilf(fmaxf(a, cc)) + 2);
            } else if (fabsf(q.z - fx) > 1.5f) hit = false;
            if (pya > pyb || pxa > pxb) hit = false;
        }
        unsigned long long todo = __ballot(hit);
        while (todo) {
            const int r = r0 + __builtin_ctzll(todo);
            todo &= todo - 1;
            const float4 bx = *reinterpret_cast<const float4*>(&sp[r * 8]);
            const float4 q = *reinterpret_cast<const float4*>(&sp[r * 8 + 4]);
            int pya = 0, pyb = ch - 1, pxa = 0, pxb = cw - 1;
            if (q.y != 0.f) {
                const float a = (fy - 1.f - q.x) * q.y, cc = (fy + 1.f - q.x) * q.y;
                pya = max(0, (int)floorf(fminf(a, cc)) - 2);
                pyb = min(ch - 1, (int)ceilf(fmaxf(a, cc)) + 2);
            }
            if (q.w != 0.f) {
                const float a = (fx - 1.f - q.z) * q.w, cc = (fx + 1.f - q.z) * q.w;
                pxa = max(0, (int)floorf(fminf(a, cc)) - 2);
                pxb = min(cw - 1, (int)ceilf(fmaxf(a, cc)) + 2);
            }
            const long long bi = (long long)b * R + r;
            // the candidate samples (py, px) of the window are tested one per lane; the ones that touch this pixel are then
            // accumulated in ascending (py, px) order -- the order of the plain double loop, so results are bit-identical
            const int wxn = pxb - pxa + 1;
            const int ncand = (pyb - pya + 1) * wxn;
            for (int k0 = 0; k0 < ncand; k0 += 64) {
                const int k = k0 + lane;
                float wgt = 0.f;
                int off = 0;
                bool use = false;
                if (k < ncand) {
                    const int py = pya + k / wxn, px = pxa + k % wxn;
                    float iny, inx;
                    if (crop_coord(bx.x, bx.z, H, ch, py, iny) && crop_coord(bx.y, bx.w, W, cw, px, inx)) {
                        const int ty = (int)floorf(iny), by = (int)ceilf(iny);
                        const int lx = (int)floorf(inx), rx = (int)ceilf(inx);
                        if ((ty == y || by == y) && (lx == x || rx == x)) {
                            const float ly = iny - (float)ty, lxw = inx - (float)lx;
                            const float wyv = (ty == y ? (1.f - ly) : 0.f) + (by == y ? ly : 0.f);
                            const float wxv = (lx == x ? (1.f - lxw) : 0.f) + (rx == x ? lxw : 0.f);
                            wgt = wyv * wxv;
                            off = py * cw + px;
                            use = true;
                        }
                    }
                }
                unsigned long long cm = __ballot(use);
                while (cm) {                 // four samples per trip: their loads are in flight together, the sums stay in order
                    float wg[4];
                    float4 g[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool any = cm != 0;
                        const int src = any ? __builtin_ctzll(cm) : 0;
                        cm &= cm - 1;                                   // 0 stays 0
                        wg[u] = any ? __shfl(wgt, src, 64) : 0.f;
                        const int o = __shfl(off, src, 64);
                        g[u] = any ? ld4g(dout + (bi * ch * cw + o) * C + c) : f4zero();
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {                       // a padded slot adds +0 * 0: exact no-op
                        acc.x = fmaf(g[u].x, wg[u], acc.x); acc.y = fmaf(g[u].y, wg[u], acc.y);
                        acc.z = fmaf(g[u].z, wg[u], acc.z); acc.w = fmaf(g[u].w, wg[u], acc.w);
                    }
                }
            }
        }
    }
    st4g(dimg + (pix * cq + (threadIdx.x & 63)) * 4, acc);
}

// The same gather with a 2 x 2 pixel quad per WAVE (a workgroup = a 4 x 4 pixel tile): a crop sample touches up to 2 x 2 feature pixels, and with one
// pixel per wave its four readers were four waves (or workgroups) whose progress through the boxes drifts apart -- 1.72x the algorithmic bytes fetched
// into L2 (profiles/r4_pmc_trunk.json).  Here a sample that touches the quad is loaded ONCE and added to each of its pixels in turn.  Per pixel the samples
// still arrive in ascending (box, py, px) order with the same weights (wy * wx, one rounding, as above), and a pixel the sample does not touch is skipped,
// so the sums are bit-identical to crop_bwd_grouped_lds_kernel's.
__global__ __launch_bounds__(256) void crop_bwd_quadwave_kernel(const float* __restrict__ dout, const float* __restrict__ boxes,
                                                                float* __restrict__ dimg, int H, int W, int R, int ch, int cw, unsigned xcd_tiles)
{
    extern __shared__ __attribute__((aligned(16))) float sp[];      // [R][8]: y1 x1 y2 x2 | y0 1/sy x0 1/sx  (1/s = 0: degenerate)
    constexpr int C = 256, cq = 64;
    unsigned bid = blockIdx.x;
    if (xcd_tiles) bid = (bid & 7u) * xcd_tiles + (bid >> 3);
    const unsigned tw = (unsigned)W / 4, tpi = ((unsigned)H / 4) * tw;
    const int b = (int)(bid / tpi);
    const unsigned tr = bid - (unsigned)b * tpi;
    const unsigned ty4 = tr / tw, tx4 = tr - ty4 * tw;
    const int wv = threadIdx.x >> 6;
    const int y0 = 4 * (int)ty4 + 2 * (wv >> 1), x0 = 4 * (int)tx4 + 2 * (wv & 1);       // the wave's quad: rows y0, y0 + 1, columns x0, x0 + 1
    const int c = (threadIdx.x & 63) * 4;
    for (int r = threadIdx.x; r < R; r += 256) {
        const float4 bx = ld4g(boxes + ((long long)b * R + r) * 4);
        const float sy = (ch > 1) ? (bx.z - bx.x) * (float)(H - 1) / (float)(ch - 1) : 0.f;
        const float sx = (cw > 1) ? (bx.w - bx.y) * (float)(W - 1) / (float)(cw - 1) : 0.f;
        const float yo = (ch > 1) ? bx.x * (float)(H - 1) : 0.5f * (bx.x + bx.z) * (float)(H - 1);
        const float xo = (cw > 1) ? bx.y * (float)(W - 1) : 0.5f * (bx.y + bx.w) * (float)(W - 1);
        *reinterpret_cast<float4*>(&sp[r * 8]) = bx;
        *reinterpret_cast<float4*>(&sp[r * 8 + 4]) =
            make_float4(yo, fabsf(sy) > 1e-6f ? 1.f / sy : 0.f, xo, fabsf(sx) > 1e-6f ? 1.f / sx : 0.f);
    }
    __syncthreads();
    float4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f4zero();
    const float fya = (float)y0 - 1.f, fyb = (float)y0 + 2.f, fxa = (float)x0 - 1.f, fxb = (float)x0 + 2.f;     // what can touch rows y0 .. y0 + 1
    const int lane = threadIdx.x & 63;
    for (int r0 = 0; r0 < R; r0 += 64) {
        const int rt = r0 + lane;
        bool hit = false;
        if (rt < R) {
            const float4 q = *reinterpret_cast<const float4*>(&sp[rt * 8 + 4]);      // y0, 1/sy, x0, 1/sx
            int pya = 0, pyb = ch - 1, pxa = 0, pxb = cw - 1;
            hit = true;
            if (q.y != 0.f) {
                const float a = (fya - q.x) * q.y, cc = (fyb - q.x) * q.y;
                pya = max(0, (int)floorf(fminf(a, cc)) - 2);
                pyb = min(ch - 1, (int)ceilf(fmaxf(a, cc)) + 2);
            } else if (q.x < fya - 0.5f || q.x > fyb + 0.5f) hit = false;
            if (q.w != 0.f) {
                const float a = (fxa - q.z) * q.w, cc = (fxb - q.z) * q.w;
                pxa = max(0, (int)floorf(fminf(a, cc)) - 2);
                pxb = min(cw - 1, (int)ceilf(fmaxf(a, cc)) + 2);
            } else if (q.z < fxa - 0.5f || q.z > fxb + 0.5f) hit = false;
            if (pya > pyb || pxa > pxb) hit = false;
        }
        unsigned long long todo = __ballot(hit);
        while (todo) {
            const int r = r0 + __builtin_ctzll(todo);
            todo &= todo - 1;
            const float4 bx = *reinterpret_cast<const float4*>(&sp[r * 8]);
            const float4 q = *reinterpret_cast<const float4*>(&sp[r * 8 + 4]);
            int pya = 0, pyb = ch - 1, pxa = 0, pxb = cw - 1;
            if (q.y != 0.f) {
                const float a = (fya - q.x) * q.y, cc = (fyb - q.x) * q.y;
                pya = max(0, (int)floorf(fminf(a, cc)) - 2);
                pyb = min(ch - 1, (int)ceilf(fmaxf(a, cc)) + 2);
            }
            if (q.w != 0.f) {
                const float a = (fxa - q.z) * q.w, cc = (fxb - q.z) * q.w;
                pxa = max(0, (int)floorf(fminf(a, cc)) - 2);
                pxb = min(cw - 1, (int)ceilf(fmaxf(a, cc)) + 2);
            }
            const long long bi = (long long)b * R + r;
            const int wxn = pxb - pxa + 1;
            const int ncand = (pyb - pya + 1) * wxn;
            for (int k0 = 0; k0 < ncand; k0 += 64) {
                const int k = k0 + lane;
                float wy0 = 0.f, wy1 = 0.f, wx0 = 0.f, wx1 = 0.f;         // weights of this sample on rows y0 / y0 + 1 and on columns x0 / x0 + 1
                int off = 0;
                bool use = false;
                if (k < ncand) {
                    const int py = pya + k / wxn, px = pxa + k % wxn;
                    float iny, inx;
                    if (crop_coord(bx.x, bx.z, H, ch, py, iny) && crop_coord(bx.y, bx.w, W, cw, px, inx)) {
                        const int ty = (int)floorf(iny), by = (int)ceilf(iny);
                        const int lx = (int)floorf(inx), rx = (int)ceilf(inx);
                        const float ly = iny - (float)ty, lxw = inx - (float)lx;
                        const bool ry0 = ty == y0 || by == y0, ry1 = ty == y0 + 1 || by == y0 + 1;
                        const bool cx0 = lx == x0 || rx == x0, cx1 = lx == x0 + 1 || rx == x0 + 1;
                        if ((ry0 || ry1) && (cx0 || cx1)) {
                            wy0 = (ty == y0 ? (1.f - ly) : 0.f) + (by == y0 ? ly : 0.f);
                            wy1 = (ty == y0 + 1 ? (1.f - ly) : 0.f) + (by == y0 + 1 ? ly : 0.f);
                            wx0 = (lx == x0 ? (1.f - lxw) : 0.f) + (rx == x0 ? lxw : 0.f);
                            wx1 = (lx == x0 + 1 ? (1.f - lxw) : 0.f) + (rx == x0 + 1 ? lxw : 0.f);
                            // which of the four pixels the sample touches (the test of the one-pixel kernel, per pixel): bits 0..3 = (row, column)
                            off = ((py * cw + px) << 4) | (ry0 && cx0 ? 1 : 0) | (ry0 && cx1 ? 2 : 0) | (ry1 && cx0 ? 4 : 0) | (ry1 && cx1 ? 8 : 0);
                            use = true;
                        }
                    }
                }
                unsigned long long cm = __ballot(use);
                while (cm) {                 // four samples per trip: their loads are in flight together, the sums stay in order
                    float a0[4], a1[4], b0[4], b1[4];
                    int tm[4];
                    float4 g[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool any = cm != 0;
                        const int src = any ? __builtin_ctzll(cm) : 0;
                        cm &= cm - 1;                                   // 0 stays 0
                        const int o = __shfl(off, src, 64);
                        tm[u] = any ? (o & 15) : 0;
                        a0[u] = __shfl(wy0, src, 64); a1[u] = __shfl(wy1, src, 64);
                        b0[u] = __shfl(wx0, src, 64); b1[u] = __shfl(wx1, src, 64);
                        g[u] = any ? ld4g(dout + (bi * ch * cw + (o >> 4)) * C + c) : f4zero();
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (tm[u] & 1) { const float w = a0[u] * b0[u]; acc[0][0].x = fmaf(g[u].x, w, acc[0][0].x); acc[0][0].y = fmaf(g[u].y, w, acc[0][0].y); acc[0][0].z = fmaf(g[u].z, w, acc[0][0].z); acc[0][0].w = fmaf(g[u].w, w, acc[0][0].w); }
                        if (tm[u] & 2) { const float w = a0[u] * b1[u]; acc[0][1].x = fmaf(g[u].x, w, acc[0][1].x); acc[0][1].y = fmaf(g[u].y, w, acc[0][1].y); acc[0][1].z = fmaf(g[u].z, w, acc[0][1].z); acc[0][1].w = fmaf(g[u].w, w, acc[0][1].w); }
                        if (tm[u] & 4) { const float w = a1[u] * b0[u]; acc[1][0].x = fmaf(g[u].x, w, acc[1][0].x); acc[1][0].y = fmaf(g[u].y, w, acc[1][0].y); acc[1][0].z = fmaf(g[u].z, w, acc[1][0].z); acc[1][0].w = fmaf(g[u].w, w, acc[1][0].w); }
                        if (tm[u] & 8) { const float w = a1[u] * b1[u]; acc[1][1].x = fmaf(g[u].x, w, acc[1][1].x); acc[1][1].y = fmaf(g[u].y, w, acc[1][1].y); acc[1][1].z = fmaf(g[u].z, w, acc[1][1].z); acc[1][1].w = fmaf(g[u].w, w, acc[1][1].w); }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
            st4g(dimg + ((((long long)b * H + y0 + i) * W + x0 + j) * cq + (threadIdx.x & 63)) * 4, acc[i][j]);
}

// ---------------------------------------------------------------------------------------
// final mask conv 1x1 (Cin -> C<=8) + bias + sigmoid.  One wave per row: each lane owns
// channel quads {lane, lane+64, ...}, partial dots are combined with a wave butterfly.
// ---------------------------------------------------------------------------------------
#define MASK_MAXC 8
template <int CC>
__global__ __launch_bounds__(256) void mask_out_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ p,
                                                           long long M, int Cin)
{
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int cq = Cin / 4;
    // weights of this lane's first channel quad stay in registers (covers Cin <= 256 entirely)
    float wr[4][CC];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int k = 0; k < CC; ++k) wr[e][k] = lane < cq ? w[(lane * 4 + e) * CC + k] : 0.f;
    const float bl = lane < CC ? bias[lane] : 0.f;
    for (long long r = wave0; r < M; r += nwaves) {
        float acc[CC];
#pragma unroll
        for (int k = 0; k < CC; ++k) acc[k] = 0.f;
        if (lane < cq) {
            const float4 v = ld4g(x + r * Cin + lane * 4);
            const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < CC; ++k) acc[k] = fmaf(xv[e], wr[e][k], acc[k]);
        }
        for (int q = lane + 64; q < cq; q += 64) {
            const float4 v = ld4g(x + r * Cin + q * 4);
            const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int k = 0; k < CC; ++k) acc[k] = fmaf(xv[e], w[(q * 4 + e) * CC + k], acc[k]);
        }
#pragma unroll
        for (int k = 0; k < CC; ++k) {
            float s = acc[k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
            acc[k] = s;
        }
        if (lane < CC) {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < CC; ++k) if (lane == k) s = acc[k];
            s += bl;
            p[r * CC + lane] = 1.f / (1.f + expf(-s));
        }
    }
}

// backward: dx = (dz w^T) * (x > 0);  dw[ci][k] = sum_m x[m,ci] dz[m,k]
template <int CC>
struct OpMaskOutBwd {
    static constexpr int NV = CC + 2;   // CC weight-gradient rows + 2 quads of bias gradient (column quad 0 only)
    const float* x;
    const float* w;     // [Cin][CC]
    const float* dz;    // [M][CC]
    float* dx;
    int Cin;
    __device__ void operator()(long long r, int c, float4* acc) const
    {
        const float4 v = ld4g(x + r * Cin + c);
        float g[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) g[k] = k < CC ? dz[r * CC + k] : 0.f;
        float4 o = f4zero();
#pragma unroll
        for (int k = 0; k < CC; ++k) {
            acc[k].x = fmaf(v.x, g[k], acc[k].x); acc[k].y = fmaf(v.y, g[k], acc[k].y);
            acc[k].z = fmaf(v.z, g[k], acc[k].z); acc[k].w = fmaf(v.w, g[k], acc[k].w);
            o.x = fmaf(g[k], w[(c + 0) * CC + k], o.x); o.y = fmaf(g[k], w[(c + 1) * CC + k], o.y);
            o.z = fmaf(g[k], w[(c + 2) * CC + k], o.z); o.w = fmaf(g[k], w[(c + 3) * CC + k], o.w);
        }
        if (c == 0) {
            acc[CC].x += g[0]; acc[CC].y += g[1]; acc[CC].z += g[2]; acc[CC].w += g[3];
            acc[CC + 1].x += g[4]; acc[CC + 1].y += g[5]; acc[CC + 1].z += g[6]; acc[CC + 1].w += g[7];
        }
        o.x = v.x > 0.f ? o.x : 0.f; o.y = v.y > 0.f ? o.y : 0.f;
        o.z = v.z > 0.f ? o.z : 0.f; o.w = v.w > 0.f ? o.w : 0.f;
        st4g(dx + r * Cin + c, o);
    }
};
// tot[k][ci] (double) -> dw[ci][k]
__global__ void mask_out_dw_finish(const double* __restrict__ tot, float* dw, float* db, int Cin, int CC)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < CC) db[i] = (float)tot[(CC + (i >> 2)) * Cin + (i & 3)];
    if (i >= Cin * CC) return;
    const int ci = i / CC, k = i % CC;
    dw[i] = (float)tot[k * Cin + ci];
}
// db[k] = sum_m dz[m][k]  (single block; M*CC is small relative to everything else)
__global__ void small_colsum_kernel(const float* __restrict__ dz, float* __restrict__ db, long long M, int CC)
{
    __shared__ double red[256];
    for (int k = 0; k < CC; ++k) {
        double s = 0;
        for (long long r = threadIdx.x; r < M; r += blockDim.x) s += dz[r * CC + k];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) db[k] = (float)red[0];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------
// mask BCE (K.binary_crossentropy on post-sigmoid p), forward + d/dlogit
// ---------------------------------------------------------------------------------------
__global__ void bce_count_kernel(const int32_t* __restrict__ ids, int NR, int* __restrict__ npos)
{
    __shared__ int red[256];
    int s = 0;
    for (int i = threadIdx.x; i < NR; i += blockDim.x) s += ids[i] > 0;
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) *npos = red[0];
}

__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ tm, const int32_t* __restrict__ ids,
                                                  const float* __restrict__ pred, const int* __restrict__ npos_p,
                                                  float lw, double* __restrict__ part, float* __restrict__ dz,
                                                  int NR, int hw, int C)
{
    __shared__ double red[256];
    const long long total = (long long)NR * hw;
    const int npos = *npos_p;
    const float invn = npos > 0 ? 1.f / ((float)npos * (float)hw) : 0.f;
    double lsum = 0;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float eps = 1e-7f;
    for (; i < total; i += stride) {
        const int roi = (int)(i / hw);
        const int id = ids[roi];
        for (int k = 0; k < C; ++k) {
            float g = 0.f;
            if (id > 0 && k == id) {
                const float p = pred[i * C + k], t = tm[i];
                const float pc = fminf(fmaxf(p, eps), 1.f - eps);
                const float z = logf(pc / (1.f - pc));
                const float l = fmaxf(z, 0.f) - z * t + log1pf(expf(-fabsf(z)));
                lsum += (double)l;
                const bool inside = (p >= eps) && (p <= 1.f - eps);
                g = inside ? (pc - t) * invn * lw : 0.f;
            }
            dz[i * C + k] = g;
        }
    }
    red[threadIdx.x] = lsum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void bce_finish_kernel(const double* __restrict__ part, int nblk, const int* __restrict__ npos_p, int hw,
                                                         float* __restrict__ out)
{
    // (one thread adding the 1024 partials one dependent load after the other held the step's main stream for 50 us)
    __shared__ double red[256];
    double s = 0;
    for (int b = threadIdx.x; b < nblk; b += 256) s += part[b];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    const int npos = *npos_p;
    out[0] = npos > 0 ? (float)(red[0] / ((double)npos * hw)) : 0.f;
    out[1] = (float)npos;
}

// ---------------------------------------------------------------------------------------
// Adam, helpers
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float lr_t, float b1, float b2,
                                                   float eps, float gs)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        const float gi = g[i] * gs;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
    }
}
// ---------------------------------------------------------------------------------------
// HBM stream-copy microbenchmark ("verify on the box with a stream-copy microbench and report the measured copy
// bandwidth alongside the nominal peak").  dst[i] = src[i] over float4 elements; each thread keeps UNR independent 16-byte loads in
// flight; variant 0: default cache policy, 1: non-temporal stores, 2: non-temporal loads and stores, 3: read only (sum into one
// store per thread: the read-side ceiling), 4: write only.
// ---------------------------------------------------------------------------------------
template <int VARIANT>
__global__ __launch_bounds__(256) void stream_copy_kernel(const f32x4n* __restrict__ src, f32x4n* __restrict__ dst, long long n4)
{
    constexpr int UNR = 4;
    const long long stride = (long long)gridDim.x * 256;
    long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    f32x4n acc = {0.f, 0.f, 0.f, 0.f};
    for (; i + (UNR - 1) * stride < n4; i += UNR * stride) {
        f32x4n v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (VARIANT == 4) v[u] = acc;
            else if (VARIANT == 2) v[u] = __builtin_nontemporal_load(src + i + u * stride);
            else v[u] = src[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            if (VARIANT == 3) acc += v[u];
            else if (VARIANT == 1 || VARIANT == 2) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) {
        if (VARIANT == 3) acc += src[i];
        else dst[i] = (VARIANT == 4) ? acc : src[i];
    }
    if (VARIANT == 3 && acc.x + acc.y + acc.z + acc.w == 12345.678f) dst[0] = acc;      // keeps the loads alive; practically never taken
}

// ---------------------------------------------------------------------------------------
// Matrix-pipe ceiling probe (the measured counterpart of the nominal MFMA peaks, as stream_copy_kernel is for HBM): every wave issues
// `iters` rounds of eight INDEPENDENT accumulator blocks of one MFMA shape from constant register operands -- no memory traffic, no LDS,
// no dependent-issue stalls -- so the rate it reaches is what the clock the power management grants under that load allows.
//   kind 0: v_mfma_f32_32x32x16_bf16 (32768 flop per instruction)   kind 1: v_mfma_f32_32x32x2_f32 (4096 flop per instruction)
// ---------------------------------------------------------------------------------------
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(256, 2) void mfma_probe_kernel(float* __restrict__ out, int iters, float seed)
{
    probe_f32x16 acc[8];
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    const float av = seed + (float)(threadIdx.x & 7) * 0.125f, bv = seed * 0.5f + (float)(threadIdx.x & 3) * 0.25f;
    probe_bf16x8 a8, b8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a8[e] = (__bf16)av; b8[e] = (__bf16)bv; }
    for (int it = 0; it < iters; ++it) {
        if (KIND == 2 || KIND == 3) {        // bf16, dependent chains: KIND 2 = two accumulator blocks alternating (a dependent MFMA every second issue), 3 = one block
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const int k = KIND == 2 ? (b & 1) : 0;
                acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[k], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            continue;
        }
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            if (KIND == 0) acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[b], 0, 0, 0);
            else acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[b], 0, 0, 0);
        }
    }
    float t = 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[b][r];
    if (t == 12345.678f) out[blockIdx.x * 256 + threadIdx.x] = t;       // keeps the accumulators alive; practically never taken
}

__global__ void add_kernel(float* __restrict__ a, const float* __restrict__ b, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] += b[i];
}
__global__ __launch_bounds__(256) void u8_unit_kernel(const uint8_t* __restrict__ x, float* __restrict__ y, long long n)
{
    __shared__ float lut[256];
    lut[threadIdx.x] = (float)((double)threadIdx.x / 255.0);      // correctly rounded float32 of the float64 quotient
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i * 4 < n; i += stride) {
        if (i * 4 + 4 <= n && ((((size_t)x) | ((size_t)y)) & 15) == 0) {
            const unsigned v = *reinterpret_cast<const unsigned*>(x + i * 4);
            st4g(y + i * 4, make_float4(lut[v & 255u], lut[(v >> 8) & 255u], lut[(v >> 16) & 255u], lut[v >> 24]));
        } else {
            for (long long j = i * 4; j < n && j < i * 4 + 4; ++j) y[j] = lut[x[j]];
        }
    }
}
__global__ void fill_kernel(float* __restrict__ a, float v, long long n)
{
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (; i < n; i += stride) a[i] = v;
}

static inline int ew_blocks(long long n)
{
    long long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (int)b;
}
// for the kernels whose loop takes four quads per trip and keeps per-thread channel terms: at least four quads per thread, at most 16 workgroups per CU
static inline int ew_blocks4(long long nquads)
{
    if (g_myolo_opt.tune0 & 128) return ew_blocks(nquads);       // ablation: round-3 launch shape (one quad per thread up to 2 M threads)
    long long b = (nquads + 1023) / 1024;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

template <int CC>
static int mask_out_bwd_impl(const float* x, const float* w, const float* dz, float* dx, float* dw, float* db, int64_t M, int Cin,
                             void* ws, size_t ws_bytes, hipStream_t s)
{
    const size_t pb = col_ws_bytes(M, Cin, CC + 2);
    MYOLO_NEED_WS(align256(pb) + (size_t)(CC + 2) * Cin * sizeof(double));
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + align256(pb));
    OpMaskOutBwd<CC> op{x, w, dz, dx, Cin};
    run_colreduce(op, M, Cin, part, tot, s);
    hipLaunchKernelGGL(mask_out_dw_finish, dim3((Cin * CC + 255) / 256), dim3(256), 0, s, tot, dw, db, Cin, CC);
    return MYOLO_OK;
}

// BN batch statistics from per-workgroup partial sums produced by another translation unit (wino_kernels.hip's
// output transform): part [nblk][2*C] doubles (sum, then sum of squares).
void myolo_bn_stats_from_partials(const double* part, double* tot, int nblk, int C, double M, const float* gamma, const float* beta,
                                  float* mean, float* var, float* scale, float* shift, float* mmean, float* mvar, hipStream_t s)
{
    const FinBnStats fin{gamma, beta, mean, var, scale, shift, mmean, mvar, M, g_myolo_opt.bn_fused_tf_variance};
    if (nblk >= 2048) hipLaunchKernelGGL((colreduce_finish<FinBnStats, 128>), dim3((C + 3) / 4), dim3(1024), 0, s, part, tot, nblk, 2 * C, C, fin);
    else hipLaunchKernelGGL((colreduce_finish<FinBnStats>), dim3((C + 3) / 4), dim3(256), 0, s, part, tot, nblk, 2 * C, C, fin);
}

// statistics pass over x [M][C] + finish, for other translation units (gemm_kernels.hip: the split-K pointwise layers)
int myolo_bn_stats_launch(const float* x, const float* gamma, const float* beta, float* mean, float* var, float* scale, float* shift,
                          float* moving_mean, float* moving_var, long long M, int C, void* ws, size_t ws_bytes, hipStream_t s)
{
    const size_t pb = col_ws_bytes(M, C, 2);
    MYOLO_NEED_WS(align256(pb) + 2 * C * sizeof(double));
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + align256(pb));
    OpStats op{x, C};
    run_colreduce(op, M, C, part, tot, s, FinBnStats{gamma, beta, mean, var, scale, shift, moving_mean, moving_var, (double)M, g_myolo_opt.bn_fused_tf_variance});
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

// =======================================================================================
extern "C" {

int myolo_colsum(const float* x, float* out, int64_t M, int C, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && out && M > 0 && C > 0, "colsum: bad arguments");
    if ((C & 3) != 0) {   // narrow odd-width case (conv_23 bias, C = N_BOX*(5+classes)): one block
        hipLaunchKernelGGL(small_colsum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, out, (long long)M, C);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    const size_t pb = col_ws_bytes(M, C, 1);
    MYOLO_NEED_WS(align256(pb) + C * sizeof(double));
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + align256(pb));
    hipStream_t s = (hipStream_t)stream;
    OpSum op{x, C};
    run_colreduce(op, M, C, part, tot, s, FinD2F{out});
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_bn_stats(const float* x, const float* gamma, const float* beta, float* mean, float* var, float* scale,
                   float* shift, float* moving_mean, float* moving_var, int64_t M, int C, void* ws, size_t ws_bytes,
                   void* stream)
{
    MYOLO_REQUIRE(x && gamma && beta && mean && var && scale && shift && M > 0 && (C & 3) == 0, "bn_stats: bad arguments");
    const size_t pb = col_ws_bytes(M, C, 2);
    MYOLO_NEED_WS(align256(pb) + 2 * C * sizeof(double));
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + align256(pb));
    hipStream_t s = (hipStream_t)stream;
    OpStats op{x, C};
    run_colreduce(op, M, C, part, tot, s, FinBnStats{gamma, beta, mean, var, scale, shift, moving_mean, moving_var, (double)M, g_myolo_opt.bn_fused_tf_variance});
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_bn_frozen_coeffs(const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                           float* scale, float* shift, int C, void* stream)
{
    MYOLO_REQUIRE(gamma && beta && moving_mean && moving_var && scale && shift && C > 0, "bn_frozen_coeffs: bad arguments");
    hipLaunchKernelGGL(bn_frozen_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma, beta,
                       moving_mean, moving_var, scale, shift, C);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* scale / shift of MANY frozen BatchNorm layers in one launch (an inference forward folds them into its conv epilogues).
 * table [nlayers][6] int64 on the device: element offsets of gamma, beta (into params), moving mean, moving variance (into stats), of the
 * layer's 2*C outputs (into coeffs: scale then shift), and C.  Same expressions as myolo_bn_frozen_coeffs. */
int myolo_bn_frozen_coeffs_batched(const float* params, const float* stats, const int64_t* table, int nlayers, float* coeffs, void* stream)
{
    MYOLO_REQUIRE(params && stats && table && coeffs && nlayers > 0, "bn_frozen_coeffs_batched: bad arguments");
    hipLaunchKernelGGL(bn_frozen_batched_kernel, dim3(nlayers), dim3(256), 0, (hipStream_t)stream, params, stats, (const long long*)table, coeffs);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_bn_apply_act(const float* x, const float* scale, const float* shift, float* y, int64_t M, int C, int act,
                       void* stream)
{
    MYOLO_REQUIRE(x && scale && shift && y && M > 0 && (C & 3) == 0, "bn_apply_act: bad arguments");
    const long long nq = (long long)M * C / 4;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(ew_blocks4(nq)), dim3(256), 0, (hipStream_t)stream, x, scale, shift, y, nq, C, act);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_bn_frozen_apply_act(const float* x, const float* gamma, const float* beta, const float* moving_mean, const float* moving_var,
                              float* scale, float* shift, float* y, int64_t M, int C, int act, void* stream)
{
    MYOLO_REQUIRE(x && gamma && beta && moving_mean && moving_var && scale && shift && y && M > 0, "bn_frozen_apply_act: bad arguments");
    MYOLO_REQUIRE((C & 3) == 0 && C >= 4 && 256 % (C / 4) == 0, "bn_frozen_apply_act: C/4 must divide 256 (got C=%d)", C);
    const long long nq = (long long)M * C / 4;
    hipLaunchKernelGGL(bn_frozen_apply_kernel, dim3(ew_blocks(nq)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, moving_mean,
                       moving_var, scale, shift, y, nq, C, act);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_bn_act_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* var,
                     const float* scale, const float* shift, float* dx, float* dgamma, float* dbeta, int64_t M, int C,
                     int act, int batch_stats, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(dy && x && mean && var && scale && shift && dx && dgamma && dbeta && M > 0 && (C & 3) == 0,
                  "bn_act_bwd: bad arguments");
    (void)gamma;
    const size_t pb = col_ws_bytes(M, C, 2);
    MYOLO_NEED_WS(align256(pb) + 2 * C * sizeof(double));
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + align256(pb));
    hipStream_t s = (hipStream_t)stream;
    if (!batch_stats && !(g_myolo_opt.tune0 & 256)) {           // frozen: one pass (tune0 & 256: the two-kernel form, ablation)
        OpBnBwdFrozenDx op{dy, x, scale, shift, mean, var, dx, C, act};
        run_colreduce(op, M, C, part, tot, s, FinBnBwd{dgamma, dbeta});
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    OpBnBwd op{dy, x, scale, shift, mean, var, C, act};
    run_colreduce(op, M, C, part, tot, s, FinBnBwd{dgamma, dbeta});
    const long long nq = (long long)M * C / 4;
    hipLaunchKernelGGL(bn_bwd_dx_kernel, dim3(ew_blocks4(nq)), dim3(256), 0, s, dy, x, scale, shift, mean, var, tot, dx, nq,
                       C, act, batch_stats, 1.0f / (float)M);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

// ---------------------------------------------------------------------------------------
// Training-mode BatchNorm (+ activation) backward in ONE launch: the sums, a grid-wide barrier per channel group, dx.
// The three-launch form (colreduce_kernel<OpBnBwd>, colreduce_finish<FinBnBwd>, bn_bwd_dx_kernel) is 26 x 3 short dependent kernels on the
// training step's backward chain.  Here a workgroup owns (row slab s, channel group): it forms its slab's partial sums
// (the same expressions and the same in-thread order as OpBnBwd), publishes them, waits until the S slabs of ITS channel group have arrived,
// adds the S partials in slab order (fixed order: bit-reproducible) and writes dx for its slab -- whose dy / x it read a moment ago.
// Barrier: sync[group] is a counter private to the calling stream, zero before its first use and never reset: a launch adds exactly 256 to it
// (each of the S = 2^k <= 256 workgroups adds 256 / S), so a workgroup that saw `old` before its own add waits for (old / 256 + 1) * 256.
// Every workgroup must become resident for the barrier to open: the grid is <= 256 workgroups of 512 threads (a CU holds four), and kernels of
// other streams only delay that, they cannot depend on this one.
// ---------------------------------------------------------------------------------------
struct BnFusedGeom { int cl, pl, cgroups, S; long long rps; };
static bool bn_fused_geom(long long M, int C, BnFusedGeom* g)
{
    const int q = C / 4;
    if (q < 1 || (C & 3)) return false;
    int cl = 1;
    while (cl * 2 <= q && cl * 2 <= 64) cl *= 2;
    g->cl = cl; g->pl = 512 / cl; g->cgroups = (q + cl - 1) / cl;
    if (g->cgroups > 64) return false;
    int S = 1;
    while (S * 2 * g->cgroups <= 256 && (long long)S * 2 * g->pl * 4 <= M) S *= 2;
    g->S = S;
    g->rps = (cdiv64(M, S) + g->pl - 1) / g->pl * g->pl;
    return true;                                      // (a slab past the last row has no rows: it still arrives, with zero sums)
}

__global__ __launch_bounds__(512) void bn_bwd_fused_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ scale,
                                                           const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ var,
                                                           float* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                           double* part, unsigned* sync, long long M, int C, int act, float invM, BnFusedGeom g)
{
    __shared__ float4 red[512];
    __shared__ float tots[2][256];
    const int tid = threadIdx.x;
    const int cl_i = tid % g.cl, pl_i = tid / g.cl;
    const int cgid = blockIdx.y, sl = blockIdx.x;
    const int cq = cgid * g.cl + cl_i;
    const bool cok = cq < C / 4;
    const int c = cq * 4;
    const long long r0 = (long long)sl * g.rps;
    long long r1 = r0 + g.rps;
    if (r1 > M) r1 = M;
    const long long st = g.pl;
    float4 sc = f4zero(), sh = f4zero(), mu = f4zero(), rs = f4zero();
    float4 acc0 = f4zero(), acc1 = f4zero();
    if (cok) {
        sc = ld4g(scale + c); sh = ld4g(shift + c); mu = ld4g(mean + c);
        const float4 vr = ld4g(var + c);
        rs = make_float4(rsqrtf(vr.x + BN_EPS_F), rsqrtf(vr.y + BN_EPS_F), rsqrtf(vr.z + BN_EPS_F), rsqrtf(vr.w + BN_EPS_F));
        auto op = [&](float4 gq, float4 v) {
            float dz, xh;
            dz = gq.x * actmask(fmaf(v.x, sc.x, sh.x), act); xh = (v.x - mu.x) * rs.x; acc0.x += dz; acc1.x = fmaf(dz, xh, acc1.x);
            dz = gq.y * actmask(fmaf(v.y, sc.y, sh.y), act); xh = (v.y - mu.y) * rs.y; acc0.y += dz; acc1.y = fmaf(dz, xh, acc1.y);
            dz = gq.z * actmask(fmaf(v.z, sc.z, sh.z), act); xh = (v.z - mu.z) * rs.z; acc0.z += dz; acc1.z = fmaf(dz, xh, acc1.z);
            dz = gq.w * actmask(fmaf(v.w, sc.w, sh.w), act); xh = (v.w - mu.w) * rs.w; acc0.w += dz; acc1.w = fmaf(dz, xh, acc1.w);
        };
        long long r = r0 + pl_i;
        for (; r + 3 * st < r1; r += 4 * st) {
            const float4 g0 = ld4g(dy + r * C + c), v0 = ld4g(x + r * C + c);
            const float4 g1 = ld4g(dy + (r + st) * C + c), v1 = ld4g(x + (r + st) * C + c);
            const float4 g2 = ld4g(dy + (r + 2 * st) * C + c), v2 = ld4g(x + (r + 2 * st) * C + c);
            const float4 g3 = ld4g(dy + (r + 3 * st) * C + c), v3 = ld4g(x + (r + 3 * st) * C + c);
            op(g0, v0); op(g1, v1); op(g2, v2); op(g3, v3);
        }
        for (; r < r1; r += st) op(ld4g(dy + r * C + c), ld4g(x + r * C + c));
    }
    // the slab's partial sums: row lanes combined in lane order, in double
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        __syncthreads();
        red[tid] = v ? acc1 : acc0;
        __syncthreads();
        if (pl_i == 0 && cok) {
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int j = 0; j < g.pl; ++j) {
                const float4 t = red[j * g.cl + cl_i];
                s0 += t.x; s1 += t.y; s2 += t.z; s3 += t.w;
            }
            double* o = part + ((long long)sl * 2 + v) * C + c;
            __builtin_nontemporal_store(s0, o); __builtin_nontemporal_store(s1, o + 1); __builtin_nontemporal_store(s2, o + 2); __builtin_nontemporal_store(s3, o + 3);
        }
    }
    // ---- barrier over the S workgroups of this channel group
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const unsigned add = 256u / (unsigned)g.S;
        const unsigned old = __hip_atomic_fetch_add(sync + cgid, add, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned target = (old / 256u + 1u) * 256u;
        while ((int)(__hip_atomic_load(sync + cgid, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) __builtin_amdgcn_s_sleep(4);
    }
    __syncthreads();
    __threadfence();
    // ---- totals of this thread's channels: S partials in slab order
    if (pl_i == 0 && cok) {
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            double t0 = 0, t1 = 0, t2 = 0, t3 = 0;
            for (int ss = 0; ss < g.S; ++ss) {
                const double* o = part + ((long long)ss * 2 + v) * C + c;
                t0 += __hip_atomic_load(o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); t1 += __hip_atomic_load(o + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                t2 += __hip_atomic_load(o + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); t3 += __hip_atomic_load(o + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            tots[v][cl_i * 4 + 0] = (float)t0; tots[v][cl_i * 4 + 1] = (float)t1; tots[v][cl_i * 4 + 2] = (float)t2; tots[v][cl_i * 4 + 3] = (float)t3;
            if (sl == 0) {
                float* o = v ? dgamma : dbeta;
                o[c] = (float)t0; o[c + 1] = (float)t1; o[c + 2] = (float)t2; o[c + 3] = (float)t3;
            }
        }
    }
    __syncthreads();
    if (!cok) return;
    const float4 db = make_float4(tots[0][cl_i * 4], tots[0][cl_i * 4 + 1], tots[0][cl_i * 4 + 2], tots[0][cl_i * 4 + 3]);
    const float4 dg = make_float4(tots[1][cl_i * 4], tots[1][cl_i * 4 + 1], tots[1][cl_i * 4 + 2], tots[1][cl_i * 4 + 3]);
    auto one = [&](float4 gq, float4 v) {
        return make_float4(bn_dx_one(gq.x, v.x, sc.x, sh.x, mu.x, rs.x, db.x, dg.x, invM, act, 1), bn_dx_one(gq.y, v.y, sc.y, sh.y, mu.y, rs.y, db.y, dg.y, invM, act, 1),
                           bn_dx_one(gq.z, v.z, sc.z, sh.z, mu.z, rs.z, db.z, dg.z, invM, act, 1), bn_dx_one(gq.w, v.w, sc.w, sh.w, mu.w, rs.w, db.w, dg.w, invM, act, 1));
    };
    long long r = r0 + pl_i;
    for (; r + 3 * st < r1; r += 4 * st) {
        const float4 g0 = ld4g(dy + r * C + c), v0 = ld4g(x + r * C + c);
        const float4 g1 = ld4g(dy + (r + st) * C + c), v1 = ld4g(x + (r + st) * C + c);
        const float4 g2 = ld4g(dy + (r + 2 * st) * C + c), v2 = ld4g(x + (r + 2 * st) * C + c);
        const float4 g3 = ld4g(dy + (r + 3 * st) * C + c), v3 = ld4g(x + (r + 3 * st) * C + c);
        st4g(dx + r * C + c, one(g0, v0)); st4g(dx + (r + st) * C + c, one(g1, v1));
        st4g(dx + (r + 2 * st) * C + c, one(g2, v2)); st4g(dx + (r + 3 * st) * C + c, one(g3, v3));
    }
    for (; r < r1; r += st) st4g(dx + r * C + c, one(ld4g(dy + r * C + c), ld4g(x + r * C + c)));
}

size_t myolo_bn_act_bwd_fused_ws_bytes(int64_t M, int C)
{
    BnFusedGeom g;
    if (!bn_fused_geom(M, C, &g)) return 0;
    return align256((size_t)g.S * 2 * C * sizeof(double));
}

int myolo_bn_act_bwd_fused(const float* dy, const float* x, const float* mean, const float* var, const float* scale, const float* shift, float* dx,
                           float* dgamma, float* dbeta, int64_t M, int C, int act, int32_t* sync, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(dy && x && mean && var && scale && shift && dx && dgamma && dbeta && sync && M > 0 && (C & 3) == 0, "bn_act_bwd_fused: bad arguments");
    BnFusedGeom g;
    MYOLO_REQUIRE(bn_fused_geom(M, C, &g), "bn_act_bwd_fused: unsupported shape (M = %lld, C = %d)", (long long)M, C);
    MYOLO_NEED_WS(myolo_bn_act_bwd_fused_ws_bytes(M, C));
    hipLaunchKernelGGL(bn_bwd_fused_kernel, dim3(g.S, g.cgroups), dim3(512), 0, (hipStream_t)stream, dy, x, scale, shift, mean, var, dx, dgamma, dbeta,
                       (double*)ws, (unsigned*)sync, (long long)M, C, act, 1.0f / (float)M, g);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_bn_act_bwd_frozen_post(const float* dy, const float* a_post, const float* gamma, const float* beta, const float* scale,
                                 float* dx, float* dgamma, float* dbeta, int64_t M, int C, int act, void* ws, size_t ws_bytes,
                                 void* stream)
{
    MYOLO_REQUIRE(dy && a_post && gamma && beta && scale && dx && dgamma && dbeta && M > 0 && (C & 3) == 0,
                  "bn_act_bwd_frozen_post: bad arguments");
    MYOLO_REQUIRE(act == MYOLO_ACT_RELU || act == MYOLO_ACT_RELU6, "bn_act_bwd_frozen_post: needs a ReLU / ReLU6 (the mask is read off the output)");
    const size_t pb = col_ws_bytes(M, C, 2);
    MYOLO_NEED_WS(align256(pb) + 2 * C * sizeof(double));
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + align256(pb));
    hipStream_t s = (hipStream_t)stream;
    OpBnBwdPost op{dy, a_post, gamma, beta, C, act};
    run_colreduce(op, M, C, part, tot, s, FinBnBwd{dgamma, dbeta});
    const long long nq = (long long)M * C / 4;
    hipLaunchKernelGGL(bn_bwd_dx_post_kernel, dim3(ew_blocks(nq)), dim3(256), 0, s, dy, a_post, scale, dx, nq, C, act);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

// Index of the positive ROIs built on the device (detect_mask_target_graph puts an image's positives first, model.py:593): ROI (b, r) is positive iff
// r < n_pos[b]; its compact slot = (positives of images < b) + r.  One workgroup: an exclusive scan of the B counts in LDS, then every ROI's three entries.
__global__ __launch_bounds__(256) void positive_index_kernel(const int32_t* __restrict__ npos, int B, int R, int32_t* __restrict__ flags,
                                                             int32_t* __restrict__ idx, int32_t* __restrict__ inv, int32_t* __restrict__ total)
{
    extern __shared__ int32_t pfx[];              // [B + 1]
    if (threadIdx.x == 0) {
        int32_t run = 0;
        for (int b = 0; b < B; ++b) {
            int32_t c = npos[b];
            c = c < 0 ? 0 : c > R ? R : c;
            pfx[b] = run;
            run += c;
        }
        pfx[B] = run;
        if (total) *total = run;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < B * R; i += 256) {
        const int b = i / R, r = i - b * R;
        const bool pos = r < pfx[b + 1] - pfx[b];
        if (flags) flags[i] = pos ? 1 : 0;
        inv[i] = pos ? pfx[b] + r : -1;
        if (pos) idx[pfx[b] + r] = i;
    }
}

int myolo_positive_index(const int32_t* n_pos, int B, int R, int32_t* flags, int32_t* idx, int32_t* inv, int32_t* total, void* stream)
{
    MYOLO_REQUIRE(n_pos && idx && inv && B > 0 && R > 0 && B <= 8192, "positive_index: bad arguments (1 <= B <= 8192)");
    hipLaunchKernelGGL(positive_index_kernel, dim3(1), dim3(256), (size_t)(B + 1) * sizeof(int32_t), (hipStream_t)stream, n_pos, B, R, flags, idx, inv, total);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_gather_groups(const float* src, const int32_t* idx, float* dst, int n, int64_t group_elems, void* stream)
{
    MYOLO_REQUIRE(src && idx && dst && n > 0 && group_elems > 0, "gather_groups: bad arguments");
    if (group_elems & 3) {           // 4-byte elements of any type, e.g. one class id per ROI
        const long long tot1 = (long long)n * group_elems;
        hipLaunchKernelGGL(gather_groups_scalar_kernel, dim3(ew_blocks(tot1)), dim3(256), 0, (hipStream_t)stream, src, idx, dst, n,
                           (long long)group_elems);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    const long long total = (long long)n * (group_elems / 4);
    {
        const long long gq = (long long)(group_elems / 4);
        long long bx = (gq + 255) / 256;
        if (bx > 64) bx = 64;
        long long by = n;
        if (by * bx > 16384) by = 16384 / bx;
        if (by < 1) by = 1;
        (void)total;
        hipLaunchKernelGGL(gather_groups_kernel, dim3((unsigned)bx, (unsigned)by), dim3(256), 0, (hipStream_t)stream, src, idx, dst, n, gq);
    }
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_gather_groups_affine_act(const float* src, const int32_t* idx, const float* scale, const float* shift, int act, float* dst_pre,
                                   float* dst_act, int n, int64_t group_rows, int C, void* stream)
{
    MYOLO_REQUIRE(src && idx && scale && shift && dst_act && n > 0 && group_rows > 0 && C > 0 && (C & 3) == 0 && (256 % (C / 4)) == 0,
                  "gather_groups_affine_act: bad arguments (C / 4 must divide 256)");
    const long long gq = (long long)group_rows * (C / 4);
    long long bx = (gq + 255) / 256;
    if (bx > 64) bx = 64;
    long long by = n;
    if (by * bx > 16384) by = 16384 / bx;
    if (by < 1) by = 1;
    hipLaunchKernelGGL(gather_groups_affine_kernel, dim3((unsigned)bx, (unsigned)by), dim3(256), 0, (hipStream_t)stream, src, idx, scale, shift, act,
                       dst_pre, dst_act, n, gq, C / 4);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_bn_act_bwd_rowsparse(const float* dy_compact, const float* x, const int32_t* idx, const int32_t* inv, const float* mean,
                               const float* var, const float* scale, const float* shift, float* dx, float* dgamma, float* dbeta,
                               int64_t M, int C, int n_groups, int group_rows, int act, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(dy_compact && x && idx && inv && mean && var && scale && shift && dx && dgamma && dbeta, "bn_act_bwd_rowsparse: null pointer");
    MYOLO_REQUIRE(M > 0 && (C & 3) == 0 && n_groups > 0 && group_rows > 0 && M % group_rows == 0, "bn_act_bwd_rowsparse: bad sizes");
    const long long Mc = (long long)n_groups * group_rows;
    const size_t pb = col_ws_bytes(Mc, C, 2);
    MYOLO_NEED_WS(align256(pb) + 2 * C * sizeof(double));
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + align256(pb));
    hipStream_t s = (hipStream_t)stream;
    OpBnBwdSparse op{dy_compact, x, idx, scale, shift, mean, var, C, act, group_rows};
    run_colreduce(op, Mc, C, part, tot, s, FinBnBwd{dgamma, dbeta});
    const long long nq = (long long)M * C / 4;
    hipLaunchKernelGGL(bn_bwd_dx_sparse_kernel, dim3((unsigned)(M / group_rows)), dim3(256), 0, s, dy_compact, x, inv, scale, shift, mean,
                       var, tot, dx, nq, C, act, group_rows, 1.0f / (float)M);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_bn_bwd_rowsparse_coeffs(const float* dy_compact, const float* x, const int32_t* idx, const float* mean, const float* var,
                                  const float* scale, const float* shift, float* dgamma, float* dbeta, float* ka, float* kb, int64_t M,
                                  int C, int n_groups, int group_rows, int act, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(dy_compact && x && idx && mean && var && scale && shift && dgamma && dbeta && ka && kb, "bn_bwd_rowsparse_coeffs: null pointer");
    MYOLO_REQUIRE(M > 0 && (C & 3) == 0 && n_groups > 0 && group_rows > 0 && M % group_rows == 0, "bn_bwd_rowsparse_coeffs: bad sizes");
    const long long Mc = (long long)n_groups * group_rows;
    const size_t pb = col_ws_bytes(Mc, C, 2);
    MYOLO_NEED_WS(align256(pb) + 2 * C * sizeof(double));
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + align256(pb));
    OpBnBwdSparse op{dy_compact, x, idx, scale, shift, mean, var, C, act, group_rows};
    run_colreduce(op, Mc, C, part, tot, (hipStream_t)stream, FinBnBwdCoef{dgamma, dbeta, ka, kb, scale, mean, var, 1.0f / (float)M});
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_conv3x3s2_c3_fwd(const float* x, const float* w, float* y, int N, int H, int W, int Cout, void* stream)
{
    MYOLO_REQUIRE(x && w && y && N > 0 && (H & 1) == 0 && (W & 1) == 0 && (Cout & 3) == 0, "conv3x3s2_c3_fwd: bad arguments");
    const long long total = (long long)N * (H / 2) * (W / 2) * (Cout / 4);
    if (conv1_fwd_rows_ok(H, W, Cout) && (long long)N * conv1_fwd_rows_chunks(H) < (1ll << 31))
        conv1_fwd_rows_launch(x, w, y, N, H, W, Cout, nullptr, (hipStream_t)stream);
    else
        hipLaunchKernelGGL(conv1_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 27 * Cout * sizeof(float), (hipStream_t)stream,
                           x, w, y, N, H, W, Cout, (double*)nullptr);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* inference: act(conv(x) * scale + shift) -- conv_block (model.py:42-52) with its BatchNormalization in inference mode folded into the conv's
 * store (scale / shift from myolo_bn_frozen_coeffs[_batched]); bit-identical to myolo_conv3x3s2_c3_fwd + myolo_bn_apply_act */
int myolo_conv3x3s2_c3_affine_act_fwd(const float* x, const float* w, const float* scale, const float* shift, int act, float* y, int N, int H, int W,
                                      int Cout, void* stream)
{
    MYOLO_REQUIRE(x && w && y && scale && shift && N > 0 && (H & 1) == 0 && (W & 1) == 0 && (Cout & 3) == 0, "conv3x3s2_c3_affine_act_fwd: bad arguments");
    if (conv1_fwd_rows_ok(H, W, Cout) && (long long)N * conv1_fwd_rows_chunks(H) < (1ll << 31)) {
        conv1_fwd_rows_launch(x, w, y, N, H, W, Cout, nullptr, (hipStream_t)stream, scale, shift, act);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    const int rc = myolo_conv3x3s2_c3_fwd(x, w, y, N, H, W, Cout, stream);          // shapes the row kernel does not take: the two launches (in place)
    if (rc != MYOLO_OK) return rc;
    return myolo_bn_apply_act(y, scale, shift, y, (int64_t)N * (H / 2) * (W / 2), Cout, act, stream);
}

/* conv_block of the backbone in training mode (model.py:42-52): the conv and the batch statistics of its output (what myolo_bn_stats
 * gives) in two launches -- the conv leaves per-workgroup partial sums, the finish turns them into mean / var / scale / shift / moving
 * averages.  ws: myolo_conv3x3s2_c3_bnstats_ws_bytes. */
// rows of statistics partials the fused conv1 forward may write: one per workgroup (1024 for the grid-stride kernel)
static size_t conv1_stat_rows(int N, int H, int W, int Cout)
{
    const size_t rows = conv1_fwd_rows_ok(H, W, Cout) ? (size_t)N * conv1_fwd_rows_chunks(H) : 0;
    return rows > 1024 ? rows : 1024;
}

size_t myolo_conv3x3s2_c3_bnstats_ws_bytes(int N, int H, int W, int Cout)
{
    const long long M = (long long)N * (H / 2) * (W / 2);
    const size_t fused = align256(conv1_stat_rows(N, H, W, Cout) * 2 * Cout * sizeof(double)) + 2 * Cout * sizeof(double);
    const size_t plain = align256(col_ws_bytes(M, Cout, 2)) + 2 * Cout * sizeof(double);
    return fused > plain ? fused : plain;
}

int myolo_conv3x3s2_c3_bnstats_fwd(const float* x, const float* w, float* y, const float* gamma, const float* beta, float* mean, float* var,
                                   float* scale, float* shift, float* moving_mean, float* moving_var, int N, int H, int W, int Cout, int phases,
                                   void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && w && y && gamma && beta && mean && var && scale && shift && N > 0 && (H & 1) == 0 && (W & 1) == 0 && (Cout & 3) == 0 && (phases & 3) != 0,
                  "conv3x3s2_c3_bnstats_fwd: bad arguments");
    MYOLO_NEED_WS(myolo_conv3x3s2_c3_bnstats_ws_bytes(N, H, W, Cout));
    hipStream_t s = (hipStream_t)stream;
    const long long M = (long long)N * (H / 2) * (W / 2);
    const long long total = M * (Cout / 4);
    const int cq = Cout / 4;
    if (cq <= 256 && (256 % cq) == 0 && !g_myolo_opt.no_trunk_fusion) {
        int blocks = ew_blocks(total);
        if (blocks > 1024) blocks = 1024;
        double* part = (double*)ws;
        double* tot = (double*)((char*)ws + align256(conv1_stat_rows(N, H, W, Cout) * 2 * Cout * sizeof(double)));
        const bool rows = conv1_fwd_rows_ok(H, W, Cout) && (long long)N * conv1_fwd_rows_chunks(H) < (1ll << 31);
        if (rows) blocks = N * conv1_fwd_rows_chunks(H);
        if ((phases & 1) && rows) conv1_fwd_rows_launch(x, w, y, N, H, W, Cout, part, s);
        else if (phases & 1) hipLaunchKernelGGL(conv1_fwd_kernel, dim3(blocks), dim3(256), 27 * Cout * sizeof(float), s, x, w, y, N, H, W, Cout, part);
        if (phases & 2)
            hipLaunchKernelGGL((colreduce_finish<FinBnStats>), dim3((Cout + 3) / 4), dim3(256), 0, s, part, tot, blocks, 2 * Cout, Cout,
                               FinBnStats{gamma, beta, mean, var, scale, shift, moving_mean, moving_var, (double)M, g_myolo_opt.bn_fused_tf_variance});
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    if (phases & 1) hipLaunchKernelGGL(conv1_fwd_kernel, dim3(ew_blocks(total)), dim3(256), 27 * Cout * sizeof(float), s, x, w, y, N, H, W, Cout, (double*)nullptr);
    MYOLO_CHECK_LAUNCH();
    if (!(phases & 2)) return MYOLO_OK;
    return myolo_bn_stats_launch(y, gamma, beta, mean, var, scale, shift, moving_mean, moving_var, M, Cout, ws, ws_bytes, s);
}

static int conv3x3s2_c3_bwd_weight_colreduce(const float* x, const float* dy, float* dw, int N, int H, int W, int Cout, void* ws,
                                             size_t ws_bytes, void* stream);

// conv1's weight gradient (3x3 / s2, 3 input channels, model.py:42-52), LDS-staged.  The column-reduction form (OpConv1Dw) fetched the 27 patch
// values of an output pixel with 27 scalar loads in EACH of the Co/4 lanes that share the pixel: 190 us for 70 MB, the last kernel of the step's
// backward chain.  Here a workgroup owns a chunk of output rows of one image: the three input rows of an output row are put in LDS once (left
// zero column included), a thread = (pixel lane, channel quad) walks the row's pixels with one 16-byte load of dy and 27 LDS reads each; the
// 27 x 4 sums per thread are combined over the pixel lanes in LDS and leave as one row of partials per workgroup, summed by
// colreduce_finish<FinD2F> in workgroup order (deterministic).
#define C1W_ROWS 8
__global__ __launch_bounds__(256) void conv1_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy, double* __restrict__ part,
                                                          int H, int W, int Co, int chunks)
{
    extern __shared__ __attribute__((aligned(16))) float c1w_lds[];        // [3][(W + 1) * 3] input rows | reduction scratch [4 waves][27][Co]
    const int Ho = H / 2, Wo = W / 2;
    const int cq = Co / 4, pl = 256 / cq;                  // cq a power of two <= 64 (checked by the launcher)
    const int tid = threadIdx.x, cl_i = tid % cq, pl_i = tid / cq;
    const int n = blockIdx.x / chunks, ch = blockIdx.x - n * chunks;
    const int oy0 = ch * C1W_ROWS, oy1 = min(oy0 + C1W_ROWS, Ho);
    const int rowf = (W + 1) * 3, nst = 3 * rowf;
    // staging: element e = k * rowf + q, q = (ix + 1) * 3 + ci; a thread's elements are the same for every output row (only iy moves):
    // their offsets are formed once, the next row's values are fetched into registers while this row is accumulated
    constexpr int SMAX = 12;                               // 3 * (W + 1) * 3 / 256 <= 12 for W <= 340
    int soff[SMAX], sk[SMAX];
    float sv[SMAX];
#pragma unroll
    for (int t = 0; t < SMAX; ++t) {
        const int e = tid + t * 256;
        const int k = e / rowf, q = e - k * rowf, ix = q / 3 - 1;
        sk[t] = e < nst ? k : -100000;                     // (never a valid input row)
        soff[t] = (e < nst && ix >= 0) ? ix * 3 + (q - (ix + 1) * 3) : -1;
    }
    auto fetch = [&](int oy) {
#pragma unroll
        for (int t = 0; t < SMAX; ++t) {
            const int iy = 2 * oy + sk[t] - 1;
            sv[t] = (soff[t] >= 0 && iy >= 0 && iy < H) ? x[((long long)n * H + iy) * W * 3 + soff[t]] : 0.f;
        }
    };
    float4 acc[27];
#pragma unroll
    for (int v = 0; v < 27; ++v) acc[v] = f4zero();
    fetch(oy0);
    for (int oy = oy0; oy < oy1; ++oy) {
        __syncthreads();                                   // the previous row's readers are done
#pragma unroll
        for (int t = 0; t < SMAX; ++t)
            if (tid + t * 256 < nst) c1w_lds[tid + t * 256] = sv[t];
        __syncthreads();
        if (oy + 1 < oy1) fetch(oy + 1);
        const float* grow = dy + (((long long)n * Ho + oy) * Wo) * Co + cl_i * 4;
        for (int ox = pl_i; ox < Wo; ox += pl) {
            const float4 g = ld4g(grow + (long long)ox * Co);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const float* r = c1w_lds + ky * rowf + 2 * ox * 3;      // (ix + 1) * 3 with ix = 2 ox + kx - 1
#pragma unroll
                for (int j = 0; j < 9; ++j) {                           // j = kx * 3 + ci
                    const float a = r[j];
                    float4& A = acc[ky * 9 + j];
                    A.x = fmaf(a, g.x, A.x); A.y = fmaf(a, g.y, A.y); A.z = fmaf(a, g.z, A.z); A.w = fmaf(a, g.w, A.w);
                }
            }
        }
    }
    // combine the pixel lanes: inside a wave by shuffles (the lanes of one channel quad are cq apart), the four waves through LDS, fixed order
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll                                  // (a runtime index into acc[] would move the 27 accumulators to scratch)
    for (int v = 0; v < 27; ++v) {
        for (int m = cq; m < 64; m <<= 1) {
            acc[v].x += __shfl_xor(acc[v].x, m, 64); acc[v].y += __shfl_xor(acc[v].y, m, 64);
            acc[v].z += __shfl_xor(acc[v].z, m, 64); acc[v].w += __shfl_xor(acc[v].w, m, 64);
        }
    }
    __syncthreads();
    float4* red = reinterpret_cast<float4*>(c1w_lds);       // [wave][27][cq]
    if (cq >= 64 || lane < cq) {
#pragma unroll
        for (int v = 0; v < 27; ++v) red[(wave * 27 + v) * cq + (lane % cq)] = acc[v];
    }
    __syncthreads();
    for (int e = tid; e < 27 * cq; e += 256) {
        const float4 a = red[e], b = red[27 * cq + e], c = red[2 * 27 * cq + e], d = red[3 * 27 * cq + e];
        const int v = e / cq, q4 = e - v * cq;
        double* o = part + ((long long)blockIdx.x * 27 + v) * Co + q4 * 4;
        o[0] = ((double)a.x + (double)b.x) + ((double)c.x + (double)d.x); o[1] = ((double)a.y + (double)b.y) + ((double)c.y + (double)d.y);
        o[2] = ((double)a.z + (double)b.z) + ((double)c.z + (double)d.z); o[3] = ((double)a.w + (double)b.w) + ((double)c.w + (double)d.w);
    }
}

static bool conv1_wgrad_lds_ok(int H, int W, int Cout)
{
    const int cq = Cout / 4;
    return cq >= 1 && cq <= 64 && (cq & (cq - 1)) == 0 && (H & 1) == 0 && (W & 1) == 0 && W <= 340 && !(g_myolo_opt.tune0 & 1024);
}

int myolo_conv3x3s2_c3_bwd_weight(const float* x, const float* dy, float* dw, int N, int H, int W, int Cout, void* ws,
                                  size_t ws_bytes, void* stream)
{
    if (x && dy && dw && N > 0 && (Cout & 3) == 0 && conv1_wgrad_lds_ok(H, W, Cout)) {
        const int chunks = (H / 2 + C1W_ROWS - 1) / C1W_ROWS;
        const long long wgs = (long long)N * chunks;
        const size_t pb = align256((size_t)wgs * 27 * Cout * sizeof(double));
        if (wgs < (1ll << 30) && ws && pb + 27 * Cout * sizeof(double) <= ws_bytes) {
            double* part = (double*)ws;
            double* tot = (double*)((char*)ws + pb);
            hipStream_t s = (hipStream_t)stream;
            size_t lds = (size_t)3 * (W + 1) * 3 * sizeof(float);
            if (lds < (size_t)4 * 27 * Cout * sizeof(float)) lds = (size_t)4 * 27 * Cout * sizeof(float);
            hipLaunchKernelGGL(conv1_wgrad_kernel, dim3((unsigned)wgs), dim3(256), lds, s, x, dy, part, H, W, Cout, chunks);
            const int nvc = 27 * Cout;
            hipLaunchKernelGGL((colreduce_finish<FinD2F>), dim3((nvc + 7) / 8), dim3(256), 0, s, part, tot, (int)wgs, nvc, Cout, FinD2F{dw});
            MYOLO_CHECK_LAUNCH();
            return MYOLO_OK;
        }
    }
    return conv3x3s2_c3_bwd_weight_colreduce(x, dy, dw, N, H, W, Cout, ws, ws_bytes, stream);
}

static int conv3x3s2_c3_bwd_weight_colreduce(const float* x, const float* dy, float* dw, int N, int H, int W, int Cout, void* ws,
                                             size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && dy && dw && N > 0 && (Cout & 3) == 0, "conv3x3s2_c3_bwd_weight: bad arguments");
    const long long M = (long long)N * (H / 2) * (W / 2);
    MYOLO_REQUIRE(M < (1ll << 31), "conv3x3s2_c3_bwd_weight: more than 2^31 output pixels");
    const size_t pb = col_ws_bytes(M, Cout, 27);
    MYOLO_NEED_WS(align256(pb) + 27 * Cout * sizeof(double));
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + align256(pb));
    hipStream_t s = (hipStream_t)stream;
    OpConv1Dw op{x, dy, H, W, Cout};
    run_colreduce(op, M, Cout, part, tot, s, FinD2F{dw});
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

// workgroups of the depthwise forward launch for this shape; nblk = rows of statistics partials of the fused form
static void dw_fwd_grid(int N, int H, int W, int C, int stride, int& which, dim3& grid, int& nblk)
{
    if (dw_rows_ok(H, W, C)) {
        const DwRowsGeom g = dw_rows_geom(N, H, W, C, stride);
        which = 10;
        grid = dim3((unsigned)g.tiles, 1, 1);
        nblk = g.nblk;
        return;
    }
    const int Ho = H / stride, Wo = W / stride;
    if (stride == 1) {
        const int per_row = ((Wo + 3) / 4) * (C / 4);
        // strips of 4 output rows where that still leaves >= ~1000 workgroups; the small late layers keep more, shorter strips
        const long long wg4 = (long long)((per_row + 255) / 256) * ((Ho + 3) / 4) * N;
        const int minwg = 400;
        if (Ho >= 4 && wg4 >= minwg && !g_myolo_opt.dw_rows1) { which = 0; grid = dim3((per_row + 255) / 256, (Ho + 3) / 4, N); }
        else if (Ho >= 2 && !g_myolo_opt.dw_rows1) { which = 1; grid = dim3((per_row + 255) / 256, (Ho + 1) / 2, N); }
        else { which = 2; grid = dim3((per_row + 255) / 256, Ho, N); }
    } else {
        const int per_row = ((Wo + 1) / 2) * (C / 4);
        const long long wg2 = (long long)((per_row + 255) / 256) * ((Ho + 1) / 2) * N;
        const int minwg = 400;
        if (Ho >= 2 && wg2 >= minwg && !g_myolo_opt.dw_rows1) { which = 3; grid = dim3((per_row + 255) / 256, (Ho + 1) / 2, N); }
        else { which = 4; grid = dim3((per_row + 255) / 256, Ho, N); }
    }
    nblk = (int)(grid.x * grid.y * grid.z);
}

static int dw_fwd_launch(const float* x, const float* w, float* y, int N, int H, int W, int C, int stride, DwAffine af, void* stream,
                         DwFuse fu = DwFuse{{nullptr, nullptr, MYOLO_ACT_NONE}, nullptr})
{
    MYOLO_REQUIRE(x && w && y && N > 0 && (C & 3) == 0 && (stride == 1 || stride == 2), "dwconv3x3[_affine_act]_fwd: bad arguments");
    MYOLO_REQUIRE(stride == 1 || ((H & 1) == 0 && (W & 1) == 0), "dwconv3x3[_affine_act]_fwd: stride 2 needs even H, W");
    hipStream_t s = (hipStream_t)stream;
    const int Ho = H / stride, Wo = W / stride;
    int which, nblk;
    dim3 grid;
    dw_fwd_grid(N, H, W, C, stride, which, grid, nblk);
    if (which == 10) {
        MYOLO_REQUIRE(!(af.scale && (fu.in.scale || fu.stat)), "dwconv3x3_fwd: input map / statistics and an output map in one launch are not supported");
        const DwRowsGeom g = dw_rows_geom(N, H, W, C, stride);
        MYOLO_REQUIRE(g.tiles < (1ll << 31), "dwconv3x3_fwd: too many tiles");
        if (stride == 1) {
            if (g.cqb == 32) dw_rows_launch<1, 32>(g, x, w, y, H, W, C, af, fu, s);
            else if (g.cqb == 16) dw_rows_launch<1, 16>(g, x, w, y, H, W, C, af, fu, s);
            else dw_rows_launch<1, 8>(g, x, w, y, H, W, C, af, fu, s);
        } else {
            if (g.cqb == 32) dw_rows_launch<2, 32>(g, x, w, y, H, W, C, af, fu, s);
            else if (g.cqb == 16) dw_rows_launch<2, 16>(g, x, w, y, H, W, C, af, fu, s);
            else dw_rows_launch<2, 8>(g, x, w, y, H, W, C, af, fu, s);
        }
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    switch (which) {
    case 0: hipLaunchKernelGGL((dw_fwd_kernel<1, 4, 4>), grid, dim3(256), 0, s, x, w, y, N, H, W, C, Ho, Wo, af, fu); break;
    case 1: hipLaunchKernelGGL((dw_fwd_kernel<1, 4, 2>), grid, dim3(256), 0, s, x, w, y, N, H, W, C, Ho, Wo, af, fu); break;
    case 2: hipLaunchKernelGGL((dw_fwd_kernel<1, 4, 1>), grid, dim3(256), 0, s, x, w, y, N, H, W, C, Ho, Wo, af, fu); break;
    case 3: hipLaunchKernelGGL((dw_fwd_kernel<2, 2, 2>), grid, dim3(256), 0, s, x, w, y, N, H, W, C, Ho, Wo, af, fu); break;
    default: hipLaunchKernelGGL((dw_fwd_kernel<2, 2, 1>), grid, dim3(256), 0, s, x, w, y, N, H, W, C, Ho, Wo, af, fu); break;
    }
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_dwconv3x3_fwd(const float* x, const float* w, float* y, int N, int H, int W, int C, int stride, void* stream)
{
    return dw_fwd_launch(x, w, y, N, H, W, C, stride, DwAffine{nullptr, nullptr, MYOLO_ACT_NONE}, stream);
}

/* inference: y = act(dwconv3x3(x) * scale + shift) in one launch -- the frozen BatchNorm after a depthwise conv, folded; equals
 * myolo_dwconv3x3_fwd followed by myolo_bn_apply_act bit for bit (model.py:57-66 with the BatchNormalization layers in inference mode) */
int myolo_dwconv3x3_affine_act_fwd(const float* x, const float* w, const float* scale, const float* shift, int act, float* y,
                                   int N, int H, int W, int C, int stride, void* stream)
{
    MYOLO_REQUIRE(scale && shift, "dwconv3x3_affine_act_fwd: bad arguments");
    return dw_fwd_launch(x, w, y, N, H, W, C, stride, DwAffine{scale, shift, act}, stream);
}

/* Training-mode depthwise block, first half (keras_applications _depthwise_conv_block, model.py:68-77 / 256-268, with BatchNormalization
 * on batch statistics): y = dwconv3x3(act_in(x * in_scale + in_shift)) and the BatchNorm statistics of y (what myolo_bn_stats gives) in
 * TWO launches -- the conv, which leaves per-workgroup partial sums of its own output, and the finish.  in_scale == NULL: x is used as
 * it is.  The normalised input is never written and y is not re-read for its statistics. */
size_t myolo_dwconv3x3_bnstats_ws_bytes(int N, int H, int W, int C, int stride)
{
    int which, nblk;
    dim3 grid;
    dw_fwd_grid(N, H, W, C, stride, which, grid, nblk);
    const size_t fused = align256((size_t)nblk * 2 * C * sizeof(double)) + 2 * C * sizeof(double);
    const long long M = (long long)N * (H / stride) * (W / stride);
    const size_t plain = align256(col_ws_bytes(M, C, 2)) + 2 * C * sizeof(double);
    return fused > plain ? fused : plain;
}

int myolo_dwconv3x3_bnstats_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w, float* y,
                                const float* gamma, const float* beta, float* mean, float* var, float* scale, float* shift,
                                float* moving_mean, float* moving_var, int N, int H, int W, int C, int stride, int phases,
                                void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(gamma && beta && mean && var && scale && shift && !in_scale == !in_shift && (phases & 3) != 0, "dwconv3x3_bnstats_fwd: bad arguments");
    MYOLO_REQUIRE(N > 0 && (C & 3) == 0 && (stride == 1 || stride == 2), "dwconv3x3_bnstats_fwd: bad arguments");
    MYOLO_NEED_WS(myolo_dwconv3x3_bnstats_ws_bytes(N, H, W, C, stride));
    hipStream_t s = (hipStream_t)stream;
    const long long M = (long long)N * (H / stride) * (W / stride);
    const int cq = C / 4;
    const DwAffine in{in_scale, in_shift, in_act};
    const FinBnStats fin{gamma, beta, mean, var, scale, shift, moving_mean, moving_var, (double)M, g_myolo_opt.bn_fused_tf_variance};
    if ((dw_rows_ok(H, W, C) || (cq <= 256 && (256 % cq) == 0)) && !g_myolo_opt.no_trunk_fusion) {
        int which, nblk;
        dim3 grid;
        dw_fwd_grid(N, H, W, C, stride, which, grid, nblk);
        double* part = (double*)ws;
        double* tot = (double*)((char*)ws + align256((size_t)nblk * 2 * C * sizeof(double)));
        if (phases & 1) {
            const int rc = dw_fwd_launch(x, w, y, N, H, W, C, stride, DwAffine{nullptr, nullptr, MYOLO_ACT_NONE}, stream, DwFuse{in, part});
            if (rc != MYOLO_OK) return rc;
        }
        if (phases & 2) hipLaunchKernelGGL((colreduce_finish<FinBnStats>), dim3((C + 3) / 4), dim3(256), 0, s, part, tot, nblk, 2 * C, C, fin);
    } else {           // channel counts the in-kernel reduction does not take: the conv (input still normalised on load), then the statistics pass
        if (phases & 1) {
            const int rc = dw_fwd_launch(x, w, y, N, H, W, C, stride, DwAffine{nullptr, nullptr, MYOLO_ACT_NONE}, stream, DwFuse{in, nullptr});
            if (rc != MYOLO_OK) return rc;
        }
        if (phases & 2) {
            double* part = (double*)ws;
            double* tot = (double*)((char*)ws + align256(col_ws_bytes(M, C, 2)));
            OpStats op{y, C};
            run_colreduce(op, M, C, part, tot, s, fin);
        }
    }
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* rows of [2][C] double partial sums myolo_dwconv3x3_bwd_data_bnsums leaves for these sizes; 0 = that entry point does not apply (the caller
 * then uses myolo_dwconv3x3_bwd_data + myolo_bn_act_bwd) */
int myolo_dwconv3x3_bwd_data_bnsums_rows(int N, int H, int W, int C, int stride)
{
    if (N <= 0 || (C & 3) || g_myolo_opt.dw_bwd_legacy || (g_myolo_opt.tune0 & 262144)) return 0;
    if (stride == 1) return dw_rows_ok(H, W, C) ? dw_rows_geom(N, H, W, C, 1).nblk : 0;
    if (stride != 2 || (H & 1) || (W & 1)) return 0;
    const int cq = C / 4;
    const long long total = (long long)N * (H / 2) * (W / 2) * cq;
    if (cq > 256 || (256 % cq) != 0 || total >= (1ll << 31) * 256) return 0;
    // 256-thread workgroups over the dy elements; with cq > 64 a workgroup covers 256 / cq < 4 pixels: that many rows of partials would outweigh
    // the pass they replace -- only the thin layers (conv_dw_2, conv_dw_4) qualify
    if (cq > 64) return 0;
    return (int)((total + 255) / 256);
}

static int dw_bwd_data_impl(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int stride, DwBnBwd bw, void* stream);

int myolo_dwconv3x3_bwd_data(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int stride, void* stream)
{
    return dw_bwd_data_impl(dy, w, dx, N, H, W, C, stride, DwBnBwd{nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr}, stream);
}

/* myolo_dwconv3x3_bwd_data whose output dx is the gradient reaching a training-mode BatchNorm (+ activation) with pre-BN tensor xbn [N,H,W,C] and the
 * coefficients of its forward: the data-gradient kernel also leaves that BatchNorm's backward sums as `rows` x [2][C] double partials in part
 * (rows = myolo_dwconv3x3_bwd_data_bnsums_rows(...) > 0), to be finished by myolo_bn_act_bwd_from_partials -- the separate pass over (dx, xbn) that
 * myolo_bn_act_bwd starts with disappears (model.py:51 / 68-77 backward). */
int myolo_dwconv3x3_bwd_data_bnsums(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int stride, const float* xbn,
                                    const float* scale, const float* shift, const float* mean, const float* var, int act, double* part, int rows,
                                    void* stream)
{
    MYOLO_REQUIRE(xbn && scale && shift && mean && var && part, "dwconv3x3_bwd_data_bnsums: bad arguments");
    MYOLO_REQUIRE(rows > 0 && rows == myolo_dwconv3x3_bwd_data_bnsums_rows(N, H, W, C, stride), "dwconv3x3_bwd_data_bnsums: rows does not match these sizes");
    return dw_bwd_data_impl(dy, w, dx, N, H, W, C, stride, DwBnBwd{xbn, scale, shift, mean, var, act, part}, stream);
}

/* the rest of myolo_bn_act_bwd(batch_stats = 1) when the sums already exist as nblk rows of [2][C] double partials: fixed-order finish (dgamma, dbeta),
 * then dx.  ws: 2 * C doubles. */
int myolo_bn_act_bwd_from_partials(const float* dy, const float* x, const float* mean, const float* var, const float* scale, const float* shift,
                                   float* dx, float* dgamma, float* dbeta, int64_t M, int C, int act, const double* part, int nblk, void* ws,
                                   size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(dy && x && mean && var && scale && shift && dx && dgamma && dbeta && part && nblk > 0 && M > 0 && (C & 3) == 0,
                  "bn_act_bwd_from_partials: bad arguments");
    MYOLO_NEED_WS(2 * (size_t)C * sizeof(double));
    double* tot = (double*)ws;
    hipStream_t s = (hipStream_t)stream;
    if (nblk >= 2048) hipLaunchKernelGGL((colreduce_finish<FinBnBwd, 128>), dim3((C + 3) / 4), dim3(1024), 0, s, part, tot, nblk, 2 * C, C, FinBnBwd{dgamma, dbeta});
    else hipLaunchKernelGGL((colreduce_finish<FinBnBwd>), dim3((C + 3) / 4), dim3(256), 0, s, part, tot, nblk, 2 * C, C, FinBnBwd{dgamma, dbeta});
    const long long nq = (long long)M * C / 4;
    hipLaunchKernelGGL(bn_bwd_dx_kernel, dim3(ew_blocks4(nq)), dim3(256), 0, s, dy, x, scale, shift, mean, var, tot, dx, nq, C, act, 1, 1.0f / (float)M);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

static int dw_bwd_data_impl(const float* dy, const float* w, float* dx, int N, int H, int W, int C, int stride, DwBnBwd bw, void* stream)
{
    MYOLO_REQUIRE(dy && w && dx && N > 0 && (C & 3) == 0 && (stride == 1 || stride == 2), "dwconv3x3_bwd_data: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    const int Ho = H / stride, Wo = W / stride;
    const int per_row = W * (C / 4);
    if (stride == 1 && dw_rows_ok(H, W, C) && !g_myolo_opt.dw_bwd_legacy) {
        // stride 1: dx = dy (*) w rotated by 180 degrees, SAME padding -- the forward's row-sliding kernel (the round-3 gather kernel fetched
        // 1.5-2.2x the algorithmic bytes, profiles/r4_pmc_trunk.json)
        const DwRowsGeom g = dw_rows_geom(N, H, W, C, 1);
        const DwAffine none{nullptr, nullptr, MYOLO_ACT_NONE};
        const DwFuse nof{none, nullptr, bw};
        if (g.cqb == 32) dw_rows_launch<1, 32>(g, dy, w, dx, H, W, C, none, nof, s, true);
        else if (g.cqb == 16) dw_rows_launch<1, 16>(g, dy, w, dx, H, W, C, none, nof, s, true);
        else dw_rows_launch<1, 8>(g, dy, w, dx, H, W, C, none, nof, s, true);
    } else if (stride == 1)
        hipLaunchKernelGGL((dw_bwd_data_kernel<1>), dim3((per_row + 255) / 256, H, N), dim3(256), 0, s, dy, w, dx, N, H, W, C, Ho, Wo);
    else if (!(H & 1) && !(W & 1) && !g_myolo_opt.dw_bwd_legacy && (long long)N * Ho * Wo * (C / 4) < (1ll << 40)) {
        const long long total = (long long)N * Ho * Wo * (C / 4);
        if (bw.x) hipLaunchKernelGGL((dw_bwd_data_s2_kernel<true>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dy, w, dx, Ho, Wo, C, total, bw);
        else hipLaunchKernelGGL((dw_bwd_data_s2_kernel<false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dy, w, dx, Ho, Wo, C, total, bw);
    } else
        hipLaunchKernelGGL((dw_bwd_data_kernel<2>), dim3((per_row + 255) / 256, H, N), dim3(256), 0, s, dy, w, dx, N, H, W, C, Ho, Wo);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

static int dw_bwd_weight_impl(const float* x, DwAffine in, const float* dy, float* dw, int N, int H, int W, int C, int stride, void* ws,
                              size_t ws_bytes, void* stream);

size_t myolo_dwconv3x3_bwd_weight_ws_bytes(int N, int H, int W, int C, int stride)
{
    if (N <= 0 || (C & 3) || (stride != 1 && stride != 2)) return 0;
    const long long M = (long long)N * (H / stride) * (W / stride);
    size_t need = align256(col_ws_bytes(M, C, 9)) + 9 * (size_t)C * sizeof(double);          // the generic column reduction (always accepted)
    if (dw_rows_ok(H, W, C)) {
        const DwRowsGeom g = dw_rows_geom(N, H, W, C, stride);
        const size_t rows = align256((size_t)g.nblk * 9 * C * sizeof(double)) + 9 * (size_t)C * sizeof(double);
        if (rows > need) need = rows;
    }
    const size_t tiled = align256((size_t)768 * 9 * C * sizeof(double)) + 9 * (size_t)C * sizeof(double);
    return need > tiled ? need : tiled;
}

int myolo_dwconv3x3_bwd_weight(const float* x, const float* dy, float* dw, int N, int H, int W, int C, int stride, void* ws,
                               size_t ws_bytes, void* stream)
{
    return dw_bwd_weight_impl(x, DwAffine{nullptr, nullptr, MYOLO_ACT_NONE}, dy, dw, N, H, W, C, stride, ws, ws_bytes, stream);
}

/* the same gradient when the conv's input was act_in(x * in_scale + in_shift) formed on load (myolo_dwconv3x3_bnstats_fwd): x is the
 * producing layer's pre-BN output, normalised again on load here */
int myolo_dwconv3x3_bwd_weight_affine_in(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* dy, float* dw,
                                         int N, int H, int W, int C, int stride, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(in_scale && in_shift, "dwconv3x3_bwd_weight_affine_in: bad arguments");
    return dw_bwd_weight_impl(x, DwAffine{in_scale, in_shift, in_act}, dy, dw, N, H, W, C, stride, ws, ws_bytes, stream);
}

}  // extern "C"

static int dw_bwd_weight_impl(const float* x, DwAffine in, const float* dy, float* dw, int N, int H, int W, int C, int stride, void* ws,
                              size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && dy && dw && N > 0 && (C & 3) == 0 && (stride == 1 || stride == 2), "dwconv3x3_bwd_weight: bad arguments");
    const int Ho = H / stride, Wo = W / stride;
    const long long M = (long long)N * Ho * Wo;
    hipStream_t s = (hipStream_t)stream;
    const int cq = C / 4;
    if (dw_rows_ok(H, W, C) && !g_myolo_opt.dw_wgrad_generic && !g_myolo_opt.dw_bwd_legacy) {
        const DwRowsGeom g = dw_rows_geom(N, H, W, C, stride);
        const size_t pb2 = align256((size_t)g.nblk * 9 * C * sizeof(double));
        if (pb2 + 9 * C * sizeof(double) <= ws_bytes && ws && g.tiles < (1ll << 31)) {
            double* part2 = (double*)ws;
            double* tot2 = (double*)((char*)ws + pb2);
            const unsigned xcd = (g.tiles % 8 == 0 && g.tiles >= 64) ? (unsigned)(g.tiles / 8) : 0u;
            const bool r6 = in.scale && in.act == MYOLO_ACT_RELU6;
#define DW_WG_GO(S_, CQB_) do { if (r6) hipLaunchKernelGGL((dw_rows_wgrad_kernel<S_, CQB_, true>), dim3((unsigned)g.tiles), dim3(256), 0, s, x, dy, part2, H, W, C, Ho, Wo, g.strips, g.chunks, g.rc, g.ncb, xcd, in); \
                               else hipLaunchKernelGGL((dw_rows_wgrad_kernel<S_, CQB_, false>), dim3((unsigned)g.tiles), dim3(256), 0, s, x, dy, part2, H, W, C, Ho, Wo, g.strips, g.chunks, g.rc, g.ncb, xcd, in); } while (0)
            if (stride == 1) { if (g.cqb == 32) DW_WG_GO(1, 32); else if (g.cqb == 16) DW_WG_GO(1, 16); else DW_WG_GO(1, 8); }
            else { if (g.cqb == 32) DW_WG_GO(2, 32); else if (g.cqb == 16) DW_WG_GO(2, 16); else DW_WG_GO(2, 8); }
#undef DW_WG_GO
            const int nvc = 9 * C;
            hipLaunchKernelGGL((colreduce_finish<FinD2F>), dim3((nvc + 7) / 8), dim3(256), 0, s, part2, tot2, g.nblk, nvc, C, FinD2F{dw});
            MYOLO_CHECK_LAUNCH();
            return MYOLO_OK;
        }
    }
    if (cq <= 256 && (256 % cq) == 0 && !g_myolo_opt.dw_wgrad_generic) {
        // tiled kernel: 2 rows x 4 columns (stride 1) / 2 x 2 (stride 2) of output pixels per thread
        const int TW = stride == 1 ? 4 : 2, TH = 2;
        const int wt = (Wo + TW - 1) / TW, per_row = wt * cq;
        // ~768 workgroups in all (three per CU), each walking its share of the (image, row group) units
        const int gx = (per_row + 255) / 256;
        const int units = ((Ho + TH - 1) / TH) * N;
        int gy = 768 / gx;
        if (gy < 1) gy = 1;
        if (gy > units) gy = units;
        const int zb = units;
        const dim3 grid(gx, gy, 1);
        const long long nblk = (long long)grid.x * grid.y;
        const size_t pb2 = align256((size_t)nblk * 9 * C * sizeof(double));
        if (pb2 + 9 * C * sizeof(double) <= ws_bytes && ws && nblk <= (1 << 20)) {
            double* part2 = (double*)ws;
            double* tot2 = (double*)((char*)ws + pb2);
            if (stride == 1) hipLaunchKernelGGL((dw_wgrad_kernel<1, 4, 2>), grid, dim3(256), 0, s, x, dy, part2, N, H, W, C, Ho, Wo, in, zb);
            else hipLaunchKernelGGL((dw_wgrad_kernel<2, 2, 2>), grid, dim3(256), 0, s, x, dy, part2, N, H, W, C, Ho, Wo, in, zb);
            const int nvc = 9 * C;
            hipLaunchKernelGGL((colreduce_finish<FinD2F>), dim3((nvc + 7) / 8), dim3(256), 0, s, part2, tot2, (int)nblk, nvc, C, FinD2F{dw});
            MYOLO_CHECK_LAUNCH();
            return MYOLO_OK;
        }
    }
    const size_t pb = col_ws_bytes(M, C, 9);
    MYOLO_NEED_WS(align256(pb) + 9 * C * sizeof(double));
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + align256(pb));
    OpDwDw op{x, dy, H, W, C, Ho, Wo, stride, in};
    run_colreduce(op, M, C, part, tot, s, FinD2F{dw});
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

extern "C" {

int myolo_crop_and_resize_fwd(const float* image, const float* boxes, const int32_t* box_ind, float* out, int B, int H, int W,
                              int C, int nb, int crop_h, int crop_w, void* stream)
{
    MYOLO_REQUIRE(image && boxes && box_ind && out && B > 0 && (C & 3) == 0 && nb >= 0, "crop_and_resize_fwd: bad arguments");
    if (nb == 0) return MYOLO_OK;
    MYOLO_REQUIRE(nb <= 65535, "crop_and_resize_fwd: at most 65535 boxes per call (got %d)", nb);
    hipLaunchKernelGGL(crop_fwd_kernel, dim3(crop_h, nb), dim3(256), 0, (hipStream_t)stream, image, boxes, box_ind, out,
                       H, W, C, nb, crop_h, crop_w);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_crop_and_resize_bwd_image(const float* dout, const float* boxes, const int32_t* box_ind, float* dimage, int B,
                                    int H, int W, int C, int nb, int crop_h, int crop_w, void* stream)
{
    MYOLO_REQUIRE(dout && boxes && box_ind && dimage && B > 0 && (C & 3) == 0 && nb >= 0, "crop_and_resize_bwd_image: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    (void)hipMemsetAsync(dimage, 0, (size_t)B * H * W * C * sizeof(float), s);
    if (nb == 0) return MYOLO_OK;
    const long long total = (long long)nb * crop_h * crop_w * (C / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(crop_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dout, boxes, box_ind, dimage, H, W, C, nb,
                       crop_h, crop_w);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_roialign_bwd_grouped(const float* dout, const float* boxes, float* dimage, int B, int H, int W, int C, int R,
                               int crop_h, int crop_w, void* stream)
{
    MYOLO_REQUIRE(dout && boxes && dimage && B > 0 && R > 0 && (C & 3) == 0, "roialign_bwd_grouped: bad arguments");
    const long long total = (long long)B * H * W * (C / 4);
    if (C == 256 && ((long long)H * W) % 4 == 0 && R <= 1536 && !g_myolo_opt.crop_bwd_nolds) {
        if ((H & 3) == 0 && (W & 3) == 0 && (g_myolo_opt.tune0 & 131072)) {
            // a 2 x 2 pixel quad per wave, a 4 x 4 tile per workgroup (crop_bwd_quadwave_kernel)
            const unsigned wgs = (unsigned)((long long)B * (H / 4) * (W / 4));
            const unsigned xcd = (wgs % 8 == 0 && wgs >= 64) ? wgs / 8 : 0u;
            hipLaunchKernelGGL(crop_bwd_quadwave_kernel, dim3(wgs), dim3(256), (size_t)R * 8 * sizeof(float), (hipStream_t)stream, dout, boxes, dimage, H, W, R,
                               crop_h, crop_w, xcd);
            MYOLO_CHECK_LAUNCH();
            return MYOLO_OK;
        }
        const int mode = g_myolo_opt.tune0 & 7;        // ablation (kbench): 1 = round-3 pixel order, 3 = tiles without the XCD-contiguous order, 4 = 4 x 4 tiles
        const bool t4 = (H % 4) == 0 && (W % 4) == 0 && mode == 4;      // (4 x 4 tiles, 1024 threads: measured slower than 2 x 2 -- 0.368 against 0.346 ms)
        const int quad = ((H | W) & 1) == 0 && mode != 1;
        const unsigned wgs = (unsigned)(total / (t4 ? 1024 : 256));
        const unsigned xcd = (wgs % 8 == 0 && wgs >= 64 && mode != 3 && mode != 1) ? wgs / 8 : 0u;
        if (t4)
            hipLaunchKernelGGL(crop_bwd_grouped_lds_kernel<4>, dim3(wgs), dim3(1024), (size_t)R * 8 * sizeof(float),
                               (hipStream_t)stream, dout, boxes, dimage, H, W, R, crop_h, crop_w, xcd, 1);
        else
            hipLaunchKernelGGL(crop_bwd_grouped_lds_kernel<2>, dim3(wgs), dim3(256), (size_t)R * 8 * sizeof(float),
                               (hipStream_t)stream, dout, boxes, dimage, H, W, R, crop_h, crop_w, xcd, quad);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    long long blocks = (total + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(crop_bwd_grouped_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dout, boxes, dimage, B, H,
                       W, C, R, crop_h, crop_w);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_mask_head_out_fwd(const float* x, const float* w, const float* bias, float* p, int64_t M, int Cin, int C, void* stream)
{
    MYOLO_REQUIRE(x && w && bias && p && M > 0 && (Cin & 3) == 0 && C >= 1 && C <= MASK_MAXC, "mask_head_out_fwd: bad arguments (1<=C<=8)");
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (M + 3) / 4;
    if (blocks > 16384) blocks = 16384;
#define MO_CASE(K) case K: hipLaunchKernelGGL((mask_out_fwd_kernel<K>), dim3((unsigned)blocks), dim3(256), 0, s, x, w, bias, p, M, Cin); break;
    switch (C) { MO_CASE(1) MO_CASE(2) MO_CASE(3) MO_CASE(4) MO_CASE(5) MO_CASE(6) MO_CASE(7) MO_CASE(8) }
#undef MO_CASE
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_mask_head_out_bwd(const float* x, const float* w, const float* dz, float* dx, float* dw, float* db, int64_t M, int Cin,
                            int C, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && w && dz && dx && dw && db && M > 0 && (Cin & 3) == 0 && C >= 1 && C <= MASK_MAXC, "mask_head_out_bwd: bad arguments");
    hipStream_t s = (hipStream_t)stream;
    int rc = MYOLO_EINVAL;
#define MO_CASE(K) case K: rc = mask_out_bwd_impl<K>(x, w, dz, dx, dw, db, M, Cin, ws, ws_bytes, s); break;
    switch (C) { MO_CASE(1) MO_CASE(2) MO_CASE(3) MO_CASE(4) MO_CASE(5) MO_CASE(6) MO_CASE(7) MO_CASE(8) }
#undef MO_CASE
    if (rc) return rc;
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_mask_bce(const float* target_masks, const int32_t* target_class_ids, const float* pred, float loss_weight,
                   float* loss_out, float* dz, int NR, int h, int w, int C, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(target_masks && target_class_ids && pred && loss_out && dz && NR > 0 && C > 0, "mask_bce: bad arguments");
    const int nblk = 1024;
    MYOLO_NEED_WS(256 + nblk * sizeof(double));
    int* npos = (int*)ws;
    double* part = (double*)((char*)ws + 256);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(bce_count_kernel, dim3(1), dim3(256), 0, s, target_class_ids, NR, npos);
    hipLaunchKernelGGL(bce_kernel, dim3(nblk), dim3(256), 0, s, target_masks, target_class_ids, pred, npos, loss_weight, part, dz,
                       NR, h * w, C);
    hipLaunchKernelGGL(bce_finish_kernel, dim3(1), dim3(256), 0, s, part, nblk, npos, h * w, loss_out);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1, float beta2, float eps,
                    float grad_scale, void* stream)
{
    MYOLO_REQUIRE(p && g && m && v && n > 0, "adam_step: bad arguments");
    hipLaunchKernelGGL(adam_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long long)n, lr_t, beta1,
                       beta2, eps, grad_scale);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_stream_copy(const void* src, void* dst, size_t nbytes, int variant, int blocks, void* stream)
{
    MYOLO_REQUIRE(src && dst && nbytes >= 16 && (nbytes & 15) == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && variant >= 0 && variant <= 4,
                  "stream_copy: needs 16-byte aligned buffers, a multiple of 16 bytes and variant 0..4");
    const long long n4 = (long long)(nbytes / 16);
    if (blocks <= 0) blocks = 256 * 8;           // 8 workgroups of 256 threads per CU
    const dim3 g((unsigned)blocks), b(256);
    hipStream_t s = (hipStream_t)stream;
    const f32x4n* sp = (const f32x4n*)src;
    f32x4n* dp = (f32x4n*)dst;
    switch (variant) {
    case 0: hipLaunchKernelGGL(stream_copy_kernel<0>, g, b, 0, s, sp, dp, n4); break;
    case 1: hipLaunchKernelGGL(stream_copy_kernel<1>, g, b, 0, s, sp, dp, n4); break;
    case 2: hipLaunchKernelGGL(stream_copy_kernel<2>, g, b, 0, s, sp, dp, n4); break;
    case 3: hipLaunchKernelGGL(stream_copy_kernel<3>, g, b, 0, s, sp, dp, n4); break;
    default: hipLaunchKernelGGL(stream_copy_kernel<4>, g, b, 0, s, sp, dp, n4); break;
    }
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

// flop of one launch = blocks * 4 waves * iters * 8 * (32768 | 4096); blocks <= 0: two workgroups per CU (two waves per SIMD)
int myolo_mfma_probe(int kind, int iters, int blocks, float* out, void* stream)
{
    MYOLO_REQUIRE(out && iters > 0 && kind >= 0 && kind <= 3, "mfma_probe: kind 0 (bf16 32x32x16), 1 (f32 32x32x2), 2 / 3 (bf16, two / one dependent chains), iters > 0, out = blocks*256 floats");
    if (blocks <= 0) blocks = 512;
    if (kind == 0) hipLaunchKernelGGL(mfma_probe_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 0.75f);
    else if (kind == 2) hipLaunchKernelGGL(mfma_probe_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 0.75f);
    else if (kind == 3) hipLaunchKernelGGL(mfma_probe_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 0.75f);
    else hipLaunchKernelGGL(mfma_probe_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, out, iters, 0.75f);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_add_inplace(float* a, const float* b, int64_t n, void* stream)
{
    MYOLO_REQUIRE(a && b && n > 0, "add_inplace: bad arguments");
    hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, b, (long long)n);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* y[i] = x[i] / 255 as the correctly rounded float32 of the float64 quotient -- what `image / 255.` stored into a float32 batch gives
 * (myolo_utils.py:824): a 256-entry table built once per launch in LDS from exactly that expression.  n % 4 == 0 (an image row of
 * pixels x 3 channels always is for the even image sizes of this path; the tail is done bytewise otherwise). */
int myolo_u8_to_unit_f32(const uint8_t* x, float* y, int64_t n, void* stream)
{
    MYOLO_REQUIRE(x && y && n > 0, "u8_to_unit_f32: bad arguments");
    const long long quads = (long long)((n + 3) / 4);
    hipLaunchKernelGGL(u8_unit_kernel, dim3(ew_blocks(quads)), dim3(256), 0, (hipStream_t)stream, x, y, (long long)n);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_fill(float* a, float value, int64_t n, void* stream)
{
    MYOLO_REQUIRE(a && n > 0, "fill: bad arguments");
    hipLaunchKernelGGL(fill_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, a, value, (long long)n);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

}  // extern "C"
