// fp32 MFMA (v_mfma_f32_32x32x2_f32) implicit-GEMM kernels for gfx950 -- the MFMA-bound part of
// the Mask-YOLO hot path: pointwise 1x1 convs, 3x3 'same' convs (feature_map, mask head),
// 2x2/s2 transposed conv, and their data / weight gradients.
//
// Two kernels, each with three A-operand addressing modes (no im2col buffer is ever built):
//   gemm_nn : C[M,N]  = A(m,k) * B[k,n]        (forward, and dX with pre-transformed weights)
//   gemm_tn : C[Ka,N] = sum_m A(m,ka) * B[m,n] (weight gradients; split over m, then reduced)
// A modes : PLAIN  A(m,k)        = A[m*lda + k]
//           CONV3  k=(tap,c)     = X[pixel(m) + (ty-1,tx-1)][c]   zero outside the image
//           DECONV k=(ky,kx,c)   = Y[2*pixel(m) + (ky,kx)][c]     (gather from the 2Hx2W grid)
//
// Tiling: 128x128 block tile, BK=16, 256 threads = 4 waves (2x2), each wave 64x64 = 2x2 MFMA
// 32x32 tiles (64 accumulator VGPRs).  The K order inside a BK chunk is permuted so that MFMA
// step j takes k=j from lane-half 0 and k=8+j from lane-half 1 (the sum over k is order-free as
// long as A and B agree).  LDS is double-buffered, one barrier per K step; global loads for
// step t+1 are issued before the MFMAs of step t and written to LDS after them.
// A tile is stored k-major ([k][m], row length 130) so fragment reads are conflict-free b32.
#include "myolo_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BM 128
#define BN 128
#define BK 16
#define LDAS (BM + 2)

enum { AM_PLAIN = 0, AM_CONV3 = 1, AM_DECONV = 2 };
enum { EP_PLAIN = 0, EP_DECONV = 1, EP_DECONV_MASK = 2 };

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const float* scale;   // optional per-column affine applied after the bias (folded frozen BatchNorm)
    const float* shift;
    long long M;      // rows of the pixel space
    int N;            // output columns
    int K;            // NN: reduction length;  TN: number of output rows (Ka)
    long long lda, ldb, ldc;
    int H, W;         // pixel grid of the row space (gather modes)
    int Cc;           // channels per tap of the A operand (gather modes)
    int Co;           // EP_DECONV: output channels per tap
    int act;
    int nt;                  // streaming (non-temporal) stores for outputs much larger than the L2
    int ksplits;             // NN fast path: >1 = split the K tiles over blockIdx.y, raw partials to `part`
    float* part;             // [ksplits][M][N] fp32 partial sums (workspace)
    long long m_per_split;   // TN only
    long long sA, sB, sC;    // fast kernels: batch strides in elements (grid z = batch index; 0 = not batched)
    int batch;               // TN split partial layout [split][batch][K*N]
    const float* w2;         // EP_DECONV_MASK: the 1x1 mask conv's kernel [Co][ncls]
    int ncls;                // EP_DECONV_MASK: classes (<= 4); partial logits go to `part` [slab][4*M][ncls]
    // training-mode BatchNorm fusion of the pointwise convs (AM_PLAIN; model.py:68-77 / 256-268 on batch statistics):
    const float* a_scale;    // NN: A := act(A * a_scale[k] + a_shift[k]) on load (the producing layer's BatchNorm + activation, never written);
    const float* a_shift;    // TN: the same per column ka of A (the weight gradient's x operand)
    int a_act;
    double* stat;            // NN, EP_PLAIN, ksplits == 1: per row-tile partial sums of the OUTPUT columns, [M tiles][2][N] doubles (sum, sum of squares)
};

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// per-column affine of the epilogue (a folded frozen BatchNorm): 1 / 0 when none is given
__device__ __forceinline__ void col_affine(const GemmArgs& p, int col, float& cs, float& ct)
{
    cs = p.scale ? p.scale[col] : 1.f;
    ct = p.scale ? p.shift[col] : 0.f;
}
__device__ __forceinline__ float gemm_act(float v, int act)
{
    if (act == MYOLO_ACT_RELU) return fmaxf(v, 0.f);
    if (act == MYOLO_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}

// ------------------------------------------------------------------------------------------
template <int AMODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nn(GemmArgs p)
{
    __shared__ float As[2][BK][LDAS];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.N + BN - 1) / BN;
    const long long bid = blockIdx.x;
    const int tn = (int)(bid % ntn);
    const long long m0 = (bid / ntn) * BM;
    const int n0 = tn * BN;

    // ---- per-thread A rows (two rows, one float4 of k each) ----
    const int ar = tid >> 2, akq = (tid & 3) * 4;
    long long am[2];
    bool avalid[2];
    int ay[2], ax[2];
    long long abase[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        long long m = m0 + ar + 64 * i;
        avalid[i] = m < p.M;
        if (!avalid[i]) m = 0;
        am[i] = m;
        if (AMODE != AM_PLAIN) {
            const long long hw = (long long)p.H * p.W;
            const long long n_img = m / hw;
            const int rem = (int)(m - n_img * hw);
            ay[i] = rem / p.W;
            ax[i] = rem - ay[i] * p.W;
            if (AMODE == AM_DECONV) abase[i] = n_img * 4 * hw + (long long)ay[i] * 4 * p.W + 2 * ax[i];
        }
    }
    // ---- per-thread B elements ----
    const int bk = tid >> 5, bn4 = (tid & 31) * 4;
    const bool bvec = (p.N & 3) == 0 && (p.ldb & 3) == 0;
    const bool avec = (p.K & 3) == 0 && (p.lda & 3) == 0;

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int nk = (p.K + BK - 1) / BK;
    float4 ra[2], rb[2];

    auto gload = [&](int kt) {
        const int k0 = kt * BK;
        // A
        if (AMODE == AM_PLAIN) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int k = k0 + akq;
                const float* ap = p.A + am[i] * p.lda + k;
                if (avalid[i] && avec && k + 3 < p.K) {
                    ra[i] = ld4(ap);
                } else {
                    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (avalid[i]) {
                        if (k + 0 < p.K) v.x = ap[0];
                        if (k + 1 < p.K) v.y = ap[1];
                        if (k + 2 < p.K) v.z = ap[2];
                        if (k + 3 < p.K) v.w = ap[3];
                    }
                    ra[i] = v;
                }
            }
        } else if (AMODE == AM_CONV3) {
            const int tap = k0 / p.Cc, c0 = k0 - tap * p.Cc;
            const int ty = tap / 3, tx = tap - ty * 3;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int yy = ay[i] + ty - 1, xx = ax[i] + tx - 1;
                const bool v = avalid[i] && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W;
                const long long off = (am[i] + (long long)(ty - 1) * p.W + (tx - 1)) * p.Cc + c0 + akq;
                ra[i] = v ? ld4(p.A + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int tap = k0 / p.Cc, c0 = k0 - tap * p.Cc;
            const int ky = tap >> 1, kx = tap & 1;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const long long off = (abase[i] + (long long)ky * 2 * p.W + kx) * p.Cc + c0 + akq;
                ra[i] = avalid[i] ? ld4(p.A + off) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        // B
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int k = k0 + bk + 8 * i;
            const int n = n0 + bn4;
            const float* bp = p.B + (long long)k * p.ldb + n;
            if (k < p.K && bvec && n + 3 < p.N) {
                rb[i] = ld4(bp);
            } else {
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k < p.K) {
                    if (n + 0 < p.N) v.x = bp[0];
                    if (n + 1 < p.N) v.y = bp[1];
                    if (n + 2 < p.N) v.z = bp[2];
                    if (n + 3 < p.N) v.w = bp[3];
                }
                rb[i] = v;
            }
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            As[buf][akq + 0][ar + 64 * i] = ra[i].x;
            As[buf][akq + 1][ar + 64 * i] = ra[i].y;
            As[buf][akq + 2][ar + 64 * i] = ra[i].z;
            As[buf][akq + 3][ar + 64 * i] = ra[i].w;
            *reinterpret_cast<float4*>(&Bs[buf][bk + 8 * i][bn4]) = rb[i];
        }
    };

    gload(0);
    sstore(0);
    __syncthreads();

    const int half = lane >> 5, l31 = lane & 31;
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) gload(kt + 1);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = half * 8 + j;
            const float a0 = As[cur][kk][wm * 64 + l31];
            const float a1 = As[cur][kk][wm * 64 + 32 + l31];
            const float b0 = Bs[cur][kk][wn * 64 + l31];
            const float b1 = Bs[cur][kk][wn * 64 + 32 + l31];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue ----
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= p.M) continue;
            long long rowoff;
            if (EPI == EP_PLAIN) {
                rowoff = row * p.ldc;
            } else {
                const long long hw = (long long)p.H * p.W;
                const long long n_img = row / hw;
                const int rem = (int)(row - n_img * hw);
                const int y = rem / p.W, x = rem - y * p.W;
                rowoff = n_img * 4 * hw + (long long)y * 4 * p.W + 2 * x;   // pixel index on the 2Hx2W grid
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int col = n0 + wn * 64 + u * 32 + l31;
                if (col >= p.N) continue;
                float v = acc[t][u][r];
                if (EPI == EP_PLAIN) {
                    if (p.bias) v += p.bias[col];
                    if (p.scale) { float cs, ct; col_affine(p, col, cs, ct); v = fmaf(v, cs, ct); }
                    v = gemm_act(v, p.act);
                    p.C[rowoff + col] = v;
                } else {
                    const int tap = col / p.Co, co = col - tap * p.Co;
                    if (p.bias) v += p.bias[co];
                    if (p.act == MYOLO_ACT_RELU) v = fmaxf(v, 0.f);
                    p.C[(rowoff + (long long)(tap >> 1) * 2 * p.W + (tap & 1)) * p.Co + co] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// C_part[split][Ka][N] = sum over this split's rows m of A(m,ka) * B[m,n]
template <int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_tn(GemmArgs p)
{
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.N + BN - 1) / BN;
    const int tn = blockIdx.x % ntn;
    const int tka = blockIdx.x / ntn;
    const int ka0 = tka * BM, n0 = tn * BN;
    const long long ms = (long long)blockIdx.y * p.m_per_split;
    long long me = ms + p.m_per_split;
    if (me > p.M) me = p.M;

    const int lr = tid >> 5, c4 = (tid & 31) * 4;
    const int ka = ka0 + c4;
    int tap = 0, c0 = ka;
    if (AMODE != AM_PLAIN) { tap = ka / p.Cc; c0 = ka - tap * p.Cc; }
    const bool avec = (p.lda & 3) == 0 && (p.K & 3) == 0;
    const bool bvec = (p.ldb & 3) == 0 && (p.N & 3) == 0;
    const long long hw = (long long)p.H * p.W;

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    float4 ra[2], rb[2];
    auto gload = [&](long long mbase) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long m = mbase + lr + 8 * i;
            const bool mv = m < me;
            // ---- A ----
            float4 va = make_float4(0.f, 0.f, 0.f, 0.f);
            if (mv && ka < p.K) {
                const float* ap = nullptr;
                if (AMODE == AM_PLAIN) {
                    ap = p.A + m * p.lda + ka;
                } else {
                    const long long n_img = m / hw;
                    const int rem = (int)(m - n_img * hw);
                    const int y = rem / p.W, x = rem - y * p.W;
                    if (AMODE == AM_CONV3) {
                        const int ty = tap / 3, tx = tap - ty * 3;
                        const int yy = y + ty - 1, xx = x + tx - 1;
                        if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W)
                            ap = p.A + (m + (long long)(ty - 1) * p.W + (tx - 1)) * p.Cc + c0;
                    } else {
                        const int ky = tap >> 1, kx = tap & 1;
                        ap = p.A + (n_img * 4 * hw + (long long)(2 * y + ky) * 2 * p.W + 2 * x + kx) * p.Cc + c0;
                    }
                }
                if (ap) {
                    if (avec && ka + 3 < p.K) {
                        va = ld4(ap);
                    } else {
                        va.x = ap[0];
                        if (ka + 1 < p.K) va.y = ap[1];
                        if (ka + 2 < p.K) va.z = ap[2];
                        if (ka + 3 < p.K) va.w = ap[3];
                    }
                }
            }
            ra[i] = va;
            // ---- B ----
            float4 vb = make_float4(0.f, 0.f, 0.f, 0.f);
            const int n = n0 + c4;
            if (mv && n < p.N) {
                const float* bp = p.B + m * p.ldb + n;
                if (bvec && n + 3 < p.N) {
                    vb = ld4(bp);
                } else {
                    vb.x = bp[0];
                    if (n + 1 < p.N) vb.y = bp[1];
                    if (n + 2 < p.N) vb.z = bp[2];
                    if (n + 3 < p.N) vb.w = bp[3];
                }
            }
            rb[i] = vb;
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<float4*>(&As[buf][lr + 8 * i][c4]) = ra[i];
            *reinterpret_cast<float4*>(&Bs[buf][lr + 8 * i][c4]) = rb[i];
        }
    };

    const long long nsteps = (me > ms) ? (me - ms + BK - 1) / BK : 0;
    const int half = lane >> 5, l31 = lane & 31;
    if (nsteps > 0) {
        gload(ms);
        sstore(0);
        __syncthreads();
        int cur = 0;
        for (long long st = 0; st < nsteps; ++st) {
            if (st + 1 < nsteps) gload(ms + (st + 1) * BK);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kk = half * 8 + j;
                const float a0 = As[cur][kk][wm * 64 + l31];
                const float a1 = As[cur][kk][wm * 64 + 32 + l31];
                const float b0 = Bs[cur][kk][wn * 64 + l31];
                const float b1 = Bs[cur][kk][wn * 64 + 32 + l31];
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (st + 1 < nsteps) sstore(cur ^ 1);
            __syncthreads();
            cur ^= 1;
        }
    }
    float* Cp = p.C + (long long)blockIdx.y * p.K * p.N;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = ka0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= p.K) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int col = n0 + wn * 64 + u * 32 + l31;
                if (col < p.N) Cp[(long long)row * p.N + col] = acc[t][u][r];
            }
        }
}

// ==========================================================================================
// Fast paths (N % 4 == 0, 16-byte aligned leading dimensions, channels per tap % 16 == 0).
// Same tiling as the generic kernels above, but the main loop is ONE basic block:
//   * operands come through raw buffer descriptors whose base is advanced per workgroup, so a 32-bit
//     byte offset always suffices and an invalid row / zero-padding tap is just an offset beyond
//     num_records -- the hardware returns 0, no exec-mask branch;
//   * the (tap, channel) position of the k tile is carried in scalar counters (no division), row
//     offsets advance incrementally;
//   * MFMA fragments are prefetched one step ahead, and the LDS image of tile t+1 is written in the
//     middle of tile t's MFMA sequence,
// so address arithmetic, global loads, ds_reads and ds_writes all issue in the shadow of the 64-cycle
// fp32 MFMAs instead of serialising between them (generic loop: 69 % MFMA-busy; this loop: 85 %;
// with the global loads removed the same loop reaches 93 %, see profiles/).
// ==========================================================================================
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define OOB_OFF 0x7fffff00u

__device__ __forceinline__ float4 bufld4(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const float* base, long long nbytes)
{
    if (nbytes < 0) nbytes = 0;
    if (nbytes > 0x7ffffe00ll) nbytes = 0x7ffffe00ll;      // stays below OOB_OFF
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(unsigned)nbytes, 0x00020000);
}

#define MFMA_STEP()                                                                             \
    {                                                                                           \
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb0, acc[0][0], 0, 0, 0);         \
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0, fb1, acc[0][1], 0, 0, 0);         \
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb0, acc[1][0], 0, 0, 0);         \
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1, fb1, acc[1][1], 0, 0, 0);         \
    }

// WNT = MFMA 32x32 tiles per wave along N: 2 -> 128x128 block tile (4 waves/SIMD), 4 -> 128x256 (2 waves/SIMD,
// A fetched once for 256 output channels, 25 % fewer loads and LDS reads per MFMA).
template <int AMODE, int EPI, int WNT = 2>
__global__ __launch_bounds__(256, 2) void gemm_nn_fast(GemmArgs p)
{
    __shared__ float As[2][BK][LDAS];
    constexpr int BN_ = 64 * WNT;          // block tile width
    constexpr int BLANES = BN_ / 4;        // lanes covering one B row with float4
    constexpr int BROWS = 256 / BLANES;    // B rows loaded per pass
    constexpr int NB = BK / BROWS;         // passes (float4 per thread)
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN_];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.N + BN_ - 1) / BN_;
    p.A += (long long)blockIdx.z * p.sA;       // batched launch (Winograd: one GEMM per transform point)
    p.B += (long long)blockIdx.z * p.sB;
    p.C += (long long)blockIdx.z * p.sC;
    // XCD-aware order: workgroup b runs on XCD b % 8 (observed; used for speed only).  Give each XCD a
    // contiguous run of tiles so the N-tiles that share an A tile are consecutive on ONE L2.
    long long bid;
    {
        const long long nwg = gridDim.x, orig = blockIdx.x;
        const long long q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tn = (int)(bid % ntn);
    const long long m0 = (bid / ntn) * BM;
    const int n0 = tn * BN_;
    const long long hw = (long long)p.H * p.W;

    // ---- A descriptor: base advanced to this workgroup's first reachable row ----
    long long base_row, end_row;
    long long row_elems;
    if (AMODE == AM_PLAIN) {
        base_row = m0; end_row = (m0 + BM < p.M) ? m0 + BM : p.M; row_elems = p.lda;
    } else if (AMODE == AM_CONV3) {
        base_row = m0 - (p.W + 1); if (base_row < 0) base_row = 0;
        end_row = m0 + BM + p.W + 1; if (end_row > p.M) end_row = p.M;
        row_elems = p.Cc;
    } else {
        base_row = 4 * m0 - 2 * p.W; if (base_row < 0) base_row = 0;
        end_row = 4 * (m0 + BM) + 2 * p.W + 2; if (end_row > 4 * p.M) end_row = 4 * p.M;
        row_elems = p.Cc;
    }
    const __amdgpu_buffer_rsrc_t ra_desc = make_rsrc(p.A + base_row * row_elems, (end_row - base_row) * row_elems * 4);
    const __amdgpu_buffer_rsrc_t rb_desc = make_rsrc(p.B, (long long)p.K * p.ldb * 4);

    const int ar = tid >> 2, akq = (tid & 3) * 4;
    bool avalid[2];
    int ay[2], ax[2];
    unsigned arow[2];          // byte offset of this row's element akq relative to the descriptor base
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long m = m0 + ar + 64 * i;
        avalid[i] = m < p.M;
        const long long mm = avalid[i] ? m : m0;
        ay[i] = 0; ax[i] = 0;
        if (AMODE == AM_PLAIN) {
            arow[i] = (unsigned)((mm - base_row) * row_elems + akq) * 4u;
        } else {
            const long long n_img = mm / hw;
            const int rem = (int)(mm - n_img * hw);
            ay[i] = rem / p.W;
            ax[i] = rem - ay[i] * p.W;
            if (AMODE == AM_CONV3) arow[i] = (unsigned)((mm - base_row) * row_elems + akq) * 4u;
            else arow[i] = (unsigned)((4 * mm - 2 * ax[i] - base_row) * row_elems + akq) * 4u;
        }
    }
    const int bk = tid / BLANES, bn4 = (tid % BLANES) * 4;
    const bool bcol_ok = (n0 + bn4) < p.N;
    const unsigned brow_stride = (unsigned)p.ldb * 4u;
    unsigned boff0 = bcol_ok ? ((unsigned)bk * (unsigned)p.ldb + (unsigned)(n0 + bn4)) * 4u : OOB_OFF;   // advanced to kt_begin below

    f32x16 acc[2][WNT];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < WNT; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int nk_total = p.K / BK;
    // split-K (under-filled grids): this workgroup owns k tiles [kt_begin, kt_begin + nk)
    const int ksp = p.ksplits > 1 ? p.ksplits : 1;
    const int kt_per = (nk_total + ksp - 1) / ksp;
    const int kt_begin = (int)blockIdx.y * kt_per;
    const int nk = max(0, min(kt_per, nk_total - kt_begin));
    int tap = 0, c0 = 0;       // scalar position of the k tile inside (tap, channel)
    // CONV3 K order: 32-channel group outermost, then the 9 taps, then the two 16-channel halves, so the
    // 18 k tiles that touch one 128-byte line of X run back to back (the 9 taps re-read the same pixels
    // shifted by <= W+1 rows): the re-reads hit L2 instead of the fabric.  The sum over K is order-free.
    const bool grouped = (AMODE == AM_CONV3) && (p.Cc % 32) == 0;
    int grp = 0, hh = 0;
    if (kt_begin > 0) {          // position of k tile kt_begin in the loop order used below
        if (grouped) { grp = kt_begin / 18; const int rem = kt_begin - grp * 18; tap = rem >> 1; hh = rem & 1; c0 = grp * 32 + hh * 16; }
        else if (AMODE == AM_PLAIN) { c0 = kt_begin * BK; }
        else { const int per_tap = p.Cc / BK; tap = kt_begin / per_tap; c0 = (kt_begin - tap * per_tap) * BK; }
    }
    float4 ra[2], rb[NB];
    float4 rsc = make_float4(1.f, 1.f, 1.f, 1.f), rsh = make_float4(0.f, 0.f, 0.f, 0.f);     // AM_PLAIN prologue: this k tile's per-k affine

    auto gload = [&]() {
        if (AMODE == AM_PLAIN) {
#pragma unroll
            for (int i = 0; i < 2; ++i) ra[i] = bufld4(ra_desc, avalid[i] ? arow[i] + (unsigned)c0 * 4u : OOB_OFF);
            if (p.a_scale) { rsc = ld4(p.a_scale + c0 + akq); rsh = ld4(p.a_shift + c0 + akq); }
        } else if (AMODE == AM_CONV3) {
            const int ty = (tap * 11) >> 5, tx = tap - ty * 3;          // tap/3 for tap < 9
            const int shift = ((ty - 1) * p.W + (tx - 1)) * p.Cc + c0;   // elements
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bool v = avalid[i] && (unsigned)(ay[i] + ty - 1) < (unsigned)p.H && (unsigned)(ax[i] + tx - 1) < (unsigned)p.W;
                ra[i] = bufld4(ra_desc, v ? arow[i] + (unsigned)(shift * 4) : OOB_OFF);
            }
        } else {
            const int ky = tap >> 1, kx = tap & 1;
            const int shift = (ky * 2 * p.W + kx) * p.Cc + c0;
#pragma unroll
            for (int i = 0; i < 2; ++i) ra[i] = bufld4(ra_desc, avalid[i] ? arow[i] + (unsigned)(shift * 4) : OOB_OFF);
        }
        if (AMODE == AM_CONV3) {      // B row of this k tile = tap*Cc + c0 (+bk)
            const unsigned krow = (unsigned)(tap * p.Cc + c0 + bk);
            const unsigned bo = bcol_ok ? (krow * (unsigned)p.ldb + (unsigned)(n0 + bn4)) * 4u : OOB_OFF;
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = bufld4(rb_desc, bcol_ok ? bo + (unsigned)(i * BROWS) * brow_stride : OOB_OFF);
        } else {                                                          // k rows always < K (K % BK == 0)
#pragma unroll
            for (int i = 0; i < NB; ++i) rb[i] = bufld4(rb_desc, bcol_ok ? boff0 + (unsigned)(i * BROWS) * brow_stride : OOB_OFF);
            boff0 = bcol_ok ? boff0 + (unsigned)BK * brow_stride : OOB_OFF;
        }
        if (grouped) {
            hh ^= 1;
            if (hh == 0) { ++tap; if (tap == 9) { tap = 0; ++grp; } }
            c0 = grp * 32 + hh * 16;
        } else {
            c0 += BK;
            if (AMODE != AM_PLAIN && c0 == p.Cc) { c0 = 0; ++tap; }
        }
    };
    auto sstore = [&](int buf) {
        if (AMODE == AM_PLAIN && p.a_scale) {          // (rows beyond M become act(shift): they are neither stored nor counted)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ra[i].x = gemm_act(fmaf(ra[i].x, rsc.x, rsh.x), p.a_act); ra[i].y = gemm_act(fmaf(ra[i].y, rsc.y, rsh.y), p.a_act);
                ra[i].z = gemm_act(fmaf(ra[i].z, rsc.z, rsh.z), p.a_act); ra[i].w = gemm_act(fmaf(ra[i].w, rsc.w, rsh.w), p.a_act);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            As[buf][akq + 0][ar + 64 * i] = ra[i].x;
            As[buf][akq + 1][ar + 64 * i] = ra[i].y;
            As[buf][akq + 2][ar + 64 * i] = ra[i].z;
            As[buf][akq + 3][ar + 64 * i] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) *reinterpret_cast<float4*>(&Bs[buf][bk + BROWS * i][bn4]) = rb[i];
    };

    if (bcol_ok && AMODE != AM_CONV3) boff0 += (unsigned)kt_begin * (unsigned)BK * brow_stride;
    const int half = lane >> 5, l31 = lane & 31;
    const int arow_l = wm * 64 + l31, bcol_l = wn * 32 * WNT + l31;
    if (nk > 0) {
        gload();
        sstore(0);
    }
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload();
        float fa[2], fb[WNT];
        fa[0] = As[cur][half * 8][arow_l]; fa[1] = As[cur][half * 8][arow_l + 32];
#pragma unroll
        for (int u = 0; u < WNT; ++u) fb[u] = Bs[cur][half * 8][bcol_l + 32 * u];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float na[2] = {0.f, 0.f}, nb[WNT] = {};
            if (j < 7) {
                const int kk = half * 8 + j + 1;
                na[0] = As[cur][kk][arow_l]; na[1] = As[cur][kk][arow_l + 32];
#pragma unroll
                for (int u = 0; u < WNT; ++u) nb[u] = Bs[cur][kk][bcol_l + 32 * u];
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < WNT; ++u)
                    acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], fb[u], acc[t][u], 0, 0, 0);
            if (j == 3 && more) sstore(cur ^ 1);   // tile t+1 lands in the other buffer mid-sequence
            fa[0] = na[0]; fa[1] = na[1];
#pragma unroll
            for (int u = 0; u < WNT; ++u) fb[u] = nb[u];
        }
        __syncthreads();
        cur ^= 1;
    }

    if (p.ksplits > 1) {          // raw partial sums; bias / affine / activation / scatter happen in splitk_epilogue
        float* Pp = p.part + (long long)blockIdx.y * p.M * p.N;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= p.M) continue;
#pragma unroll
                for (int u = 0; u < WNT; ++u) {
                    const int col = n0 + wn * 32 * WNT + u * 32 + l31;
                    if (col < p.N) Pp[row * p.N + col] = acc[t][u][r];
                }
            }
        return;
    }
    if constexpr (EPI == EP_DECONV_MASK && WNT == 2) {
        // myolo_mask_deconv + ReLU + the 1x1 myolo_mask conv (model.py:711-714) without ever writing the
        // [N,2H,2W,Co] tensor: this tile holds 128 of the Co channels of ONE tap (Co % 128 == 0) for 128 input pixels.
        // Per class, each lane forms relu(acc + bias) * w2 for its 32 row slots and 2 columns, then a 5-step
        // reduce-scatter butterfly over the 32 lanes of its half leaves lane l the sum of row slot l over the wave's
        // 64 columns.  The (Co/128)*2 column slabs are summed in fixed order by deconv_mask_finish (deterministic).
        float cbm[WNT];
        int cow[WNT];
        int tap = 0;
#pragma unroll
        for (int u = 0; u < WNT; ++u) {
            const int col = n0 + wn * 32 * WNT + u * 32 + l31;
            tap = col / p.Co;
            cow[u] = col - tap * p.Co;
            cbm[u] = p.bias ? p.bias[cow[u]] : 0.f;
        }
        // After the butterfly lane l31 owns row slot r = l31 & 15 of the 32-row block t = l31 >> 4.
        const int rr = l31 & 15;
        const long long row = m0 + wm * 64 + (l31 >> 4) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
        float* dst = nullptr;
        if (row < p.M) {
            const long long n_img = row / hw;
            const int rem = (int)(row - n_img * hw);
            const int y = rem / p.W, x = rem - y * p.W;
            const long long pix = n_img * 4 * hw + (long long)(2 * y + (tap >> 1)) * 2 * p.W + 2 * x + (tap & 1);
            const int slab = ((n0 - tap * p.Co) / BN_) * 2 + wn;
            dst = p.part + ((long long)slab * 4 * p.M + pix) * p.ncls;
        }
#pragma unroll 1
        for (int c = 0; c < p.ncls; ++c) {          // one class and one 32-row block at a time: live set = acc + 16 values
            const float w0 = p.w2[cow[0] * p.ncls + c], w1 = p.w2[cow[1] * p.ncls + c];
            float outv = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float cur[16];
#pragma unroll
                for (int j = 0; j < 16; ++j)
                    cur[j] = fmaf(fmaxf(acc[t][1][j] + cbm[1], 0.f), w1, fmaxf(acc[t][0][j] + cbm[0], 0.f) * w0);
#pragma unroll
                for (int st = 0; st < 4; ++st) {     // reduce-scatter over the 16 lanes sharing l31 >> 4 ...
                    const int m = 1 << st;
                    const bool bit = (l31 >> st) & 1;
#pragma unroll
                    for (int i = 0; i < (8 >> st); ++i) {
                        const float a = cur[2 * i], b = cur[2 * i + 1];
                        const float keep = bit ? b : a, send = bit ? a : b;
                        cur[i] = keep + __shfl_xor(send, m, 64);
                    }
                }
                const float tot = cur[0] + __shfl_xor(cur[0], 16, 64);      // ... then add the other 16 lanes' columns
                if ((l31 >> 4) == t) outv = tot;
            }
            if (dst) dst[c] = outv;
        }
        return;
    }
    // ---- epilogue: per-column parameters are loaded once, then 64 row-contiguous 128-byte stores per wave ----
    float cb[WNT], cs[WNT], ct[WNT];
    int ccol[WNT], ctap[WNT];
    bool cok[WNT];
#pragma unroll
    for (int u = 0; u < WNT; ++u) {
        const int col = n0 + wn * 32 * WNT + u * 32 + l31;
        cok[u] = col < p.N;
        const int colc = cok[u] ? col : 0;
        ctap[u] = 0;
        ccol[u] = colc;
        if (EPI == EP_DECONV) { ctap[u] = colc / p.Co; ccol[u] = colc - ctap[u] * p.Co; }
        cb[u] = p.bias ? p.bias[ccol[u]] : 0.f;
        col_affine(p, colc, cs[u], ct[u]);
    }
    const int act = p.act;
    if constexpr (AMODE == AM_PLAIN && EPI == EP_PLAIN) {
        if (p.stat) {
            // column sums of what this tile writes (bias included, before any epilogue affine): lane -> its 32 row slots, then the
            // other half-wave (same columns, other rows), then the two waves that share the columns through LDS (free after the k
            // loop); one row of partials per row tile, summed later in a fixed order by colreduce_finish: bit-reproducible
            float s1[WNT], s2[WNT];
#pragma unroll
            for (int u = 0; u < WNT; ++u) { s1[u] = 0.f; s2[u] = 0.f; }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row >= p.M) continue;
#pragma unroll
                    for (int u = 0; u < WNT; ++u) { const float v = acc[t][u][r] + cb[u]; s1[u] += v; s2[u] = fmaf(v, v, s2[u]); }
                }
            float* sred = &Bs[0][0][0];               // [2 (wm)][2 (sum, sumsq)][BN_]
#pragma unroll
            for (int u = 0; u < WNT; ++u) {
                s1[u] += __shfl_xor(s1[u], 32, 64);
                s2[u] += __shfl_xor(s2[u], 32, 64);
                if (half == 0) {
                    sred[(wm * 2 + 0) * BN_ + wn * 32 * WNT + u * 32 + l31] = s1[u];
                    sred[(wm * 2 + 1) * BN_ + wn * 32 * WNT + u * 32 + l31] = s2[u];
                }
            }
            __syncthreads();
            if (tid < 2 * BN_) {
                const int v = tid / BN_, cc = tid - v * BN_;
                if (n0 + cc < p.N)
                    p.stat[((m0 / BM) * 2 + v) * p.N + n0 + cc] = (double)sred[(0 * 2 + v) * BN_ + cc] + (double)sred[(1 * 2 + v) * BN_ + cc];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= p.M) continue;
            long long rowoff;
            if (EPI == EP_PLAIN) {
                rowoff = row * p.ldc;
            } else {
                const long long n_img = row / hw;
                const int rem = (int)(row - n_img * hw);
                const int y = rem / p.W, x = rem - y * p.W;
                rowoff = n_img * 4 * hw + (long long)y * 4 * p.W + 2 * x;
            }
#pragma unroll
            for (int u = 0; u < WNT; ++u) {
                if (!cok[u]) continue;
                float v = gemm_act(fmaf(acc[t][u][r] + cb[u], cs[u], ct[u]), act);
                float* dst = (EPI == EP_PLAIN) ? p.C + rowoff + ccol[u]
                                               : p.C + (rowoff + (long long)(ctap[u] >> 1) * 2 * p.W + (ctap[u] & 1)) * p.Co + ccol[u];
                if (p.nt) __builtin_nontemporal_store(v, dst);
                else *dst = v;
            }
        }
    }
}

// weight-gradient fast path: C_part[split][Ka][N] = sum_m A(m,ka) B[m,n]
template <int AMODE>
__global__ __launch_bounds__(256, 2) void gemm_tn_fast(GemmArgs p)
{
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.N + BN - 1) / BN;
    p.A += (long long)blockIdx.z * p.sA;
    p.B += (long long)blockIdx.z * p.sB;
    const int tn = blockIdx.x % ntn;
    const int tka = blockIdx.x / ntn;
    const int ka0 = tka * BM, n0 = tn * BN;
    const long long ms = (long long)blockIdx.y * p.m_per_split;
    long long me = ms + p.m_per_split;
    if (me > p.M) me = p.M;
    const long long hw = (long long)p.H * p.W;

    const int lr = tid >> 5, c4 = (tid & 31) * 4;
    const int ka = ka0 + c4;
    const bool ka_ok = ka < p.K;
    int tap = 0, c0 = ka;
    if (AMODE != AM_PLAIN) { tap = ka / p.Cc; c0 = ka - tap * p.Cc; }
    const int ty = tap / 3, tx = tap - ty * 3;      // CONV3
    const int ky = tap >> 1, kx = tap & 1;          // DECONV

    // descriptors: A rows reachable from [ms, me); B rows [ms, me)
    long long a_base, a_end, a_row_elems;
    if (AMODE == AM_PLAIN) { a_base = ms; a_end = me; a_row_elems = p.lda; }
    else if (AMODE == AM_CONV3) {
        a_base = ms - (p.W + 1); if (a_base < 0) a_base = 0;
        a_end = me + p.W + 1; if (a_end > p.M) a_end = p.M;
        a_row_elems = p.Cc;
    } else {
        a_base = 4 * ms - 2 * p.W; if (a_base < 0) a_base = 0;
        a_end = 4 * me + 2 * p.W + 2; if (a_end > 4 * p.M) a_end = 4 * p.M;
        a_row_elems = p.Cc;
    }
    const __amdgpu_buffer_rsrc_t ra_desc = make_rsrc(p.A + a_base * a_row_elems, (a_end - a_base) * a_row_elems * 4);
    const __amdgpu_buffer_rsrc_t rb_desc = make_rsrc(p.B + ms * p.ldb, (me - ms) * p.ldb * 4);
    const bool bcol_ok = (n0 + c4) < p.N;

    // this thread's two rows; (y, x) and the linear byte offsets advance by BK rows per step (32-bit, branch-free)
    long long mrow[2];
    int ry[2], rx[2];
    unsigned alin[2], blin[2];
    const unsigned a_step = (unsigned)(BK * a_row_elems) * 4u * (AMODE == AM_DECONV ? 4u : 1u);
    const unsigned b_step = (unsigned)(BK * p.ldb) * 4u;
    const unsigned two_cc = (unsigned)(2 * a_row_elems) * 4u;
    long long tapshift = 0;
    if (AMODE == AM_CONV3) tapshift = (long long)(ty - 1) * p.W + (tx - 1);
    if (AMODE == AM_DECONV) tapshift = (long long)ky * 2 * p.W + kx;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        mrow[i] = ms + lr + 8 * i;
        ry[i] = 0; rx[i] = 0;
        if (AMODE != AM_PLAIN) {
            const long long mm = mrow[i] < p.M ? mrow[i] : 0;
            const long long n_img = mm / hw;
            const int rem = (int)(mm - n_img * hw);
            ry[i] = rem / p.W;
            rx[i] = rem - ry[i] * p.W;
        }
        // (for CONV3 the value may wrap when the tap is out of the image; such rows are masked below)
        if (AMODE == AM_PLAIN) alin[i] = (unsigned)((mrow[i] - a_base) * a_row_elems + ka) * 4u;
        else if (AMODE == AM_CONV3) alin[i] = (unsigned)((mrow[i] + tapshift - a_base) * a_row_elems + c0) * 4u;
        else alin[i] = (unsigned)((4 * mrow[i] + tapshift - a_base) * a_row_elems + c0) * 4u;   // minus 2*x*Cc per step
        blin[i] = (unsigned)((mrow[i] - ms) * p.ldb + n0 + c4) * 4u;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    float4 ra[2], rb[2];
    bool rav[2] = {false, false};
    float4 tsc = make_float4(1.f, 1.f, 1.f, 1.f), tsh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (AMODE == AM_PLAIN && p.a_scale && ka_ok) { tsc = ld4(p.a_scale + ka); tsh = ld4(p.a_shift + ka); }
    auto gload = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bool mv = mrow[i] < me;
            rav[i] = mv && ka_ok;
            unsigned aoff = OOB_OFF;
            if (AMODE == AM_PLAIN) {
                if (mv && ka_ok) aoff = alin[i];
            } else if (AMODE == AM_CONV3) {
                const bool v = mv && ka_ok && (unsigned)(ry[i] + ty - 1) < (unsigned)p.H && (unsigned)(rx[i] + tx - 1) < (unsigned)p.W;
                if (v) aoff = alin[i];
            } else {
                if (mv && ka_ok) aoff = alin[i] - (unsigned)rx[i] * two_cc;
            }
            ra[i] = bufld4(ra_desc, aoff);
            rb[i] = bufld4(rb_desc, (mv && bcol_ok) ? blin[i] : OOB_OFF);
            mrow[i] += BK;
            alin[i] += a_step;
            blin[i] += b_step;
            if (AMODE != AM_PLAIN) {      // branch-free wrap (launcher guarantees W >= 8, H >= 2)
                int nx = rx[i] + BK;
                const int w2 = (nx >= 2 * p.W) ? 2 : ((nx >= p.W) ? 1 : 0);
                nx -= w2 * p.W;
                int ny = ry[i] + w2;
                ny = (ny >= p.H) ? ny - p.H : ny;
                rx[i] = nx; ry[i] = ny;
            }
        }
    };
    auto sstore = [&](int buf) {
        if (AMODE == AM_PLAIN && p.a_scale) {          // x := act(x * scale + shift) for the rows that exist (padding rows stay 0)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (rav[i]) {
                    ra[i].x = gemm_act(fmaf(ra[i].x, tsc.x, tsh.x), p.a_act); ra[i].y = gemm_act(fmaf(ra[i].y, tsc.y, tsh.y), p.a_act);
                    ra[i].z = gemm_act(fmaf(ra[i].z, tsc.z, tsh.z), p.a_act); ra[i].w = gemm_act(fmaf(ra[i].w, tsc.w, tsh.w), p.a_act);
                }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            *reinterpret_cast<float4*>(&As[buf][lr + 8 * i][c4]) = ra[i];
            *reinterpret_cast<float4*>(&Bs[buf][lr + 8 * i][c4]) = rb[i];
        }
    };

    const long long nsteps = (me > ms) ? (me - ms + BK - 1) / BK : 0;
    const int half = lane >> 5, l31 = lane & 31;
    const int arow_l = wm * 64 + l31, bcol_l = wn * 64 + l31;
    if (nsteps > 0) {
        gload();
        sstore(0);
        __syncthreads();
        int cur = 0;
        for (long long st = 0; st < nsteps; ++st) {
            const bool more = st + 1 < nsteps;
            if (more) gload();
            float fa0 = As[cur][half * 8][arow_l], fa1 = As[cur][half * 8][arow_l + 32];
            float fb0 = Bs[cur][half * 8][bcol_l], fb1 = Bs[cur][half * 8][bcol_l + 32];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float na0 = 0.f, na1 = 0.f, nb0 = 0.f, nb1 = 0.f;
                if (j < 7) {
                    const int kk = half * 8 + j + 1;
                    na0 = As[cur][kk][arow_l]; na1 = As[cur][kk][arow_l + 32];
                    nb0 = Bs[cur][kk][bcol_l]; nb1 = Bs[cur][kk][bcol_l + 32];
                }
                MFMA_STEP()
                if (j == 3 && more) sstore(cur ^ 1);
                fa0 = na0; fa1 = na1; fb0 = nb0; fb1 = nb1;
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    const int nb = p.batch > 1 ? p.batch : 1;
    float* Cp = p.C + ((long long)blockIdx.y * nb + blockIdx.z) * p.K * p.N;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = ka0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= p.K) continue;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int col = n0 + wn * 64 + u * 32 + l31;
                if (col < p.N) Cp[(long long)row * p.N + col] = acc[t][u][r];
            }
        }
}

// ------------------------------------------------------------------------------------------
// Weight gradient of a THIN pointwise conv (Cin <= 64, Cout <= 128: conv_pw_1..3, 154 MB of activations for a 32x64 result):
//   dW[Cin][Cout] = sum_m x[m][ci] * dy[m][co].
// The 128x128-tile kernel above wastes 3/4 (or 15/16) of its tile on such a shape and runs at 0.5-0.8 TB/s.  Here one wave
// owns the WHOLE Cin x Cout result as KT x NT MFMA 32x32 tiles and streams rows: the MFMA "k" index is the row m, so lane
// (i = l&31, h = l>>5) needs x[m + h][32*kt + i] and dy[m + h][32*nt + i] -- contiguous 128-byte segments straight from global
// memory, no LDS, no transposition.  Each wave takes every (gridDim.x*4)-th pair of rows; the four waves of a workgroup are summed
// through LDS, the workgroups' partials by splitk_reduce (fixed order: bit-reproducible).
// ------------------------------------------------------------------------------------------
template <int KT, int NT>
__global__ __launch_bounds__(256) void pw_wgrad_thin(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ part,
                                                     long long M, int Cin, int Cout, const float* __restrict__ a_scale,
                                                     const float* __restrict__ a_shift, int a_act)
{
    __shared__ float red[4][32 * 32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    f32x16 acc[KT][NT];
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    float xs[KT], xt[KT];                       // x := act(x * scale + shift) on load (the producing BatchNorm, see GemmArgs::a_scale)
#pragma unroll
    for (int a = 0; a < KT; ++a) { xs[a] = a_scale ? a_scale[32 * a + l31] : 1.f; xt[a] = a_scale ? a_shift[32 * a + l31] : 0.f; }
    const long long nwaves = (long long)gridDim.x * 4;
    const long long w0 = (long long)blockIdx.x * 4 + wave;
    // rows in chunks of 2*RS per wave-iteration (RS MFMA steps of 2 rows): RS independent loads per operand tile in flight
    constexpr int RS = (KT + NT <= 3) ? 8 : 4;
    for (long long m0 = w0 * (2 * RS); m0 < M; m0 += nwaves * (2 * RS)) {
        float xa[RS][KT], yb[RS][NT];
#pragma unroll
        for (int st = 0; st < RS; ++st) {
            const long long m = m0 + 2 * st + half;
            const bool ok = m < M;
#pragma unroll
            for (int a = 0; a < KT; ++a) {
                float v = ok ? x[m * Cin + 32 * a + l31] : 0.f;
                if (a_scale && ok) v = gemm_act(fmaf(v, xs[a], xt[a]), a_act);
                xa[st][a] = v;
            }
#pragma unroll
            for (int b = 0; b < NT; ++b) yb[st][b] = ok ? dy[m * Cout + 32 * b + l31] : 0.f;
        }
#pragma unroll
        for (int st = 0; st < RS; ++st)
#pragma unroll
            for (int a = 0; a < KT; ++a)
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[st][a], yb[st][b], acc[a][b], 0, 0, 0);
    }
    // sum the four waves tile by tile through LDS, then one coalesced store of the workgroup's partial
    float* dst = part + (long long)blockIdx.x * Cin * Cout;
#pragma unroll
    for (int a = 0; a < KT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave][((r & 3) + 8 * (r >> 2) + 4 * half) * 32 + l31] = acc[a][b][r];
            __syncthreads();
            for (int e = threadIdx.x; e < 1024; e += 256) {
                const float v = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
                dst[(long long)(32 * a + (e >> 5)) * Cout + 32 * b + (e & 31)] = v;
            }
        }
}

// out[i] = sum_s part[s][i] for SMALL n (a few thousand elements) and many splits: splitk_reduce would leave one short serial chain
// per element.  Workgroup = 64 consecutive elements x 4 split lanes, 8 loads in flight per thread, fixed summation order.
__global__ __launch_bounds__(256) void partial_sum_small(const float* __restrict__ part, float* __restrict__ out, int n, int splits)
{
    __shared__ float red[4][64];
    const int e = blockIdx.x * 64 + (threadIdx.x & 63), sl = threadIdx.x >> 6;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (e < n) {
        int k = sl;
        for (; k + 28 < splits; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) s[u] += part[(long long)(k + 4 * u) * n + e];
        }
        for (; k < splits; k += 4) s[0] += part[(long long)k * n + e];
    }
    red[sl][threadIdx.x & 63] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    __syncthreads();
    if (sl == 0 && e < n) out[e] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// out = sum over splits of part[split]; n4 = n/4 float4 elements.  Loads of 4 splits are issued together.
__global__ __launch_bounds__(256) void splitk_reduce(const float* __restrict__ part, float* __restrict__ out, long long n, int splits)
{
    const long long n4 = (n & 3) ? 0 : (n >> 2);      // vector path only when every split slab stays 16-byte aligned
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float4* p4 = reinterpret_cast<const float4*>(part);
    for (; i < n4; i += stride) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        int k = 0;
        for (; k + 4 <= splits; k += 4) {
            const float4 a = p4[(long long)(k + 0) * n4 + i], b = p4[(long long)(k + 1) * n4 + i];
            const float4 c = p4[(long long)(k + 2) * n4 + i], d = p4[(long long)(k + 3) * n4 + i];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
            s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
            s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
        }
        for (; k < splits; ++k) {
            const float4 a = p4[(long long)k * n4 + i];
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        reinterpret_cast<float4*>(out)[i] = s;
    }
    // tail (n % 4 elements)
    for (long long j = (n4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        float t = 0.f;
        for (int k = 0; k < splits; ++k) t += part[(long long)k * n + j];
        out[j] = t;
    }
}

// split-K epilogue of the NN kernels: out = act((sum_s part[s] + bias) * scale + shift), PLAIN or deconv scatter
template <int EPI>
__global__ __launch_bounds__(256) void splitk_epilogue(GemmArgs p)
{
    const long long total = p.M * (long long)(p.N >> 2);
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const int nq = p.N >> 2;
    const long long slab = p.M * (long long)p.N;
    const long long hw = (long long)p.H * p.W;
    for (; i < total; i += stride) {
        const long long row = i / nq;
        const int col = (int)(i - row * nq) * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = 0; k < p.ksplits; ++k) {
            const float4 a = *reinterpret_cast<const float4*>(p.part + k * slab + row * p.N + col);
            s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        }
        float v[4] = {s.x, s.y, s.z, s.w};
        int tap = 0, cc = col;
        if (EPI == EP_DECONV) { tap = col / p.Co; cc = col - tap * p.Co; }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (p.bias) v[e] += p.bias[cc + e];
            if (p.scale) { float cs, ct; col_affine(p, col + e, cs, ct); v[e] = fmaf(v[e], cs, ct); }
            v[e] = gemm_act(v[e], p.act);
        }
        float* dst;
        if (EPI == EP_PLAIN) {
            dst = p.C + row * p.ldc + col;
        } else {
            const long long n_img = row / hw;
            const int rem = (int)(row - n_img * hw);
            const int y = rem / p.W, x = rem - y * p.W;
            dst = p.C + (n_img * 4 * hw + (long long)y * 4 * p.W + 2 * x + (long long)(tap >> 1) * 2 * p.W + (tap & 1)) * p.Co + cc;
        }
        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// out[z][c][r] = in[zmap(z)][r][c]; reverse!=0 maps z -> nz-1-z (3x3 tap rotation by 180 degrees)
__global__ void transpose_batched(const float* __restrict__ in, float* __restrict__ out, int R, int C, int nz, int reverse)
{
    __shared__ float tile[32][33];
    const int z = blockIdx.z;
    const int zi = reverse ? (nz - 1 - z) : z;
    const float* ip = in + (long long)zi * R * C;
    float* op = out + (long long)z * R * C;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int r = r0 + i, c = c0 + threadIdx.x;
        tile[i][threadIdx.x] = (r < R && c < C) ? ip[(long long)r * C + c] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.y; i < 32; i += blockDim.y) {
        const int c = c0 + i, r = r0 + threadIdx.x;
        if (r < R && c < C) op[(long long)c * R + r] = tile[threadIdx.x][i];
    }
}

// p[pix][c] = sigmoid(b2[c] + sum over the column slabs of the fused deconv epilogue), slabs in fixed order
__global__ __launch_bounds__(256) void deconv_mask_finish(const float* __restrict__ part, const float* __restrict__ b2,
                                                          float* __restrict__ out, long long npix, int ncls, int nslabs)
{
    const long long total = npix * ncls;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float sacc = b2[(int)(i % ncls)];
        for (int k = 0; k < nslabs; ++k) sacc += part[(long long)k * total + i];
        out[i] = 1.f / (1.f + expf(-sacc));
    }
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static int launch_transpose(const float* in, float* out, int R, int C, int nz, int reverse, hipStream_t s)
{
    dim3 grid((C + 31) / 32, (R + 31) / 32, nz), block(32, 8);
    hipLaunchKernelGGL(transpose_batched, grid, block, 0, s, in, out, R, C, nz, reverse);
    return 0;
}

// pointwise layers on the bf16x6 kernels (128 x 256 / 256 x 256 tiles) only from this many rows up: the 7x7 layers (1568 rows) have too
// few tiles for them and keep the split-K fp32 kernels (measured: 52 -> 72 us and 75 -> 123 us the other way round)
static inline long long pw_x6_min_rows() { return g_myolo_opt.pw_x6_min_rows > 0 ? g_myolo_opt.pw_x6_min_rows : 4096; }

template <int AMODE, int EPI>
static int launch_nn(const GemmArgs& a, hipStream_t s, void* sk_ws = nullptr, size_t sk_ws_bytes = 0, long long sk_max_tiles = 512,
                     int* path = nullptr)      // *path: 0 generic kernel, 1 fast kernel, >= 2 fast kernel with that many K splits
{
    if (path) *path = 0;
    const long long tiles = cdiv64(a.M, BM) * ((a.N + BN - 1) / BN);
    if (tiles <= 0) return MYOLO_OK;
    const bool aligned = (a.N & 3) == 0 && (a.ldb & 3) == 0 && ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0;
    const bool kfast = (AMODE == AM_PLAIN) ? ((a.K % BK) == 0 && (a.lda & 3) == 0) : ((a.Cc % BK) == 0);
    GemmArgs& am = const_cast<GemmArgs&>(a);
    am.nt = ((long long)a.M * a.N * 4 > (64ll << 20)) && !g_myolo_opt.no_nt;
    if (aligned && kfast && !g_myolo_opt.gemm_generic) {
        // under-filled grid (fewer tiles than the 1024 resident-workgroup slots) and a long K loop: split K so the
        // whole chip works on it; a lone 128x128 tile with K = 2304 takes ~185 us however few tiles there are
        const int nk = a.K / BK;
        int splits = 1;
        if (sk_ws && tiles < sk_max_tiles && nk >= 16 && (a.N & 3) == 0 && (EPI == EP_PLAIN ? (a.ldc & 3) == 0 : (a.Co & 3) == 0) &&
            !g_myolo_opt.no_splitk) {
            splits = (int)(1024 / tiles);
            if (splits > nk / 8) splits = nk / 8;
            const size_t per = (size_t)a.M * a.N * sizeof(float);
            if ((size_t)splits * per > sk_ws_bytes) splits = (int)(sk_ws_bytes / per);
            if (splits < 2) splits = 1;
        }
        if (path) *path = splits > 1 ? splits : 1;
        if (g_myolo_opt.gemm_w256 && (a.N % 256) == 0 && tiles >= 1024 && !a.stat) {
            const long long tiles256 = cdiv64(a.M, BM) * (a.N / 256);
            hipLaunchKernelGGL((gemm_nn_fast<AMODE, EPI, 4>), dim3((unsigned)tiles256), dim3(256), 0, s, a);
            return MYOLO_OK;
        }
        if (splits > 1) {
            am.ksplits = splits;
            am.part = (float*)sk_ws;
            am.stat = nullptr;                 // (column statistics come out of the one-pass epilogue only; the caller runs a statistics pass)
            hipLaunchKernelGGL((gemm_nn_fast<AMODE, EPI>), dim3((unsigned)tiles, splits), dim3(256), 0, s, a);
            const long long total = a.M * (long long)(a.N / 4);
            int blocks = (int)((total + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL((splitk_epilogue<EPI>), dim3(blocks), dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL((gemm_nn_fast<AMODE, EPI>), dim3((unsigned)tiles), dim3(256), 0, s, a);
        }
    } else
        hipLaunchKernelGGL((gemm_nn<AMODE, EPI>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    return MYOLO_OK;
}

static int choose_splits(long long M, int Ka, int N)
{
    // 256 CUs x 4 resident workgroups (33 KB LDS, <=128 VGPRs): pick the split count that fills those 1024
    // slots as evenly as possible -- with e.g. 792 workgroups a quarter of the CUs hold 4 and set the time.
    const long long tiles = (long long)((Ka + BM - 1) / BM) * ((N + BN - 1) / BN);
    long long splits = 1024 / tiles;
    const long long max_by_rows = cdiv64(M, 8 * BK);     // at least 8 K-steps per split
    if (splits > max_by_rows) splits = max_by_rows;
    if (splits < 1) splits = 1;
    if (splits > 1024) splits = 1024;
    return (int)splits;
}

static size_t tn_ws_bytes(long long M, int Ka, int N)
{
    const int splits = choose_splits(M, Ka, N);
    return splits > 1 ? (size_t)splits * Ka * N * sizeof(float) : 0;
}

template <int AMODE>
static int launch_tn(GemmArgs a, float* out, void* ws, size_t ws_bytes, hipStream_t s, const char* who)
{
    const int splits = choose_splits(a.M, a.K, a.N);
    const size_t need = splits > 1 ? (size_t)splits * a.K * a.N * sizeof(float) : 0;
    if (need > ws_bytes || (need && !ws)) {
        myolo_set_error("%s: workspace too small (%zu needed, %zu given)", who, need, ws_bytes);
        return MYOLO_EWORKSPACE;
    }
    long long mps = cdiv64(a.M, splits);
    mps = cdiv64(mps, BK) * BK;
    a.m_per_split = mps;
    a.C = splits > 1 ? (float*)ws : out;
    const int tiles = ((a.K + BM - 1) / BM) * ((a.N + BN - 1) / BN);
    const bool fast = (a.N & 3) == 0 && (a.ldb & 3) == 0 && (a.K & 3) == 0 && ((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0 &&
                      ((AMODE == AM_PLAIN) ? (a.lda & 3) == 0 : ((a.Cc & 3) == 0 && a.W >= 8 && a.H >= 2)) &&
                      !g_myolo_opt.gemm_generic;
    if (fast)
        hipLaunchKernelGGL((gemm_tn_fast<AMODE>), dim3(tiles, splits), dim3(256), 0, s, a);
    else
        hipLaunchKernelGGL((gemm_tn<AMODE>), dim3(tiles, splits), dim3(256), 0, s, a);
    if (splits > 1) {
        const long long n = (long long)a.K * a.N;      // n % 4 == 0 whenever N % 4 == 0; partial slabs are 16-byte aligned then
        int blocks = (int)((n / 4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(splitk_reduce, dim3(blocks), dim3(256), 0, s, (const float*)ws, out, n, splits);
    }
    return MYOLO_OK;
}

// ---- batched plain GEMMs for wino_kernels.hip (one GEMM per Winograd transform point, grid z = point) ----
// C[z] (M x N) = A[z] (M x K) * B[z] (K x N); all row-major and dense.
int myolo_gemm_nn_batched(const float* A, const float* B, float* C, long long M, int K, int N, int batch, hipStream_t s)
{
    if ((K % BK) || (N & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15)) {
        myolo_set_error("gemm_nn_batched: needs K %% %d == 0, N %% 4 == 0 and 16-byte aligned operands", BK);
        return MYOLO_EINVAL;
    }
    GemmArgs a = {};
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldb = N; a.ldc = N; a.act = MYOLO_ACT_NONE;
    a.sA = M * (long long)K; a.sB = (long long)K * N; a.sC = M * (long long)N; a.batch = batch;
    a.nt = g_myolo_opt.wino_nt ? 1 : 0;          // the product is read back at once by the output transform
    const long long tiles = cdiv64(M, BM) * ((N + BN - 1) / BN);
    if (tiles <= 0 || batch <= 0) return MYOLO_OK;
    if (g_myolo_opt.wino_w256 && (N % 256) == 0) {      // tuning knob: 128x256 tiles (A read once)
        const long long tiles256 = cdiv64(M, BM) * (N / 256);
        hipLaunchKernelGGL((gemm_nn_fast<AM_PLAIN, EP_PLAIN, 4>), dim3((unsigned)tiles256, 1, batch), dim3(256), 0, s, a);
        return MYOLO_OK;
    }
    hipLaunchKernelGGL((gemm_nn_fast<AM_PLAIN, EP_PLAIN>), dim3((unsigned)tiles, 1, batch), dim3(256), 0, s, a);
    return MYOLO_OK;
}

size_t myolo_gemm_tn_batched_ws_bytes(long long M, int Ka, int N, int batch)
{
    const long long tiles = (long long)((Ka + BM - 1) / BM) * ((N + BN - 1) / BN) * batch;
    long long splits = 1024 / tiles;
    const long long max_by_rows = cdiv64(M, 8 * BK);
    if (splits > max_by_rows) splits = max_by_rows;
    if (splits < 1) splits = 1;
    return splits > 1 ? (size_t)splits * batch * Ka * N * sizeof(float) : 0;
}

// C[z] (Ka x N) = A[z]^T (M x Ka)^T * B[z] (M x N): the sum over the M rows is split so that tiles x batch x splits fills the chip
int myolo_gemm_tn_batched(const float* A, const float* B, float* C, long long M, int Ka, int N, int batch, void* ws, size_t ws_bytes,
                          hipStream_t s)
{
    if ((N & 3) || (Ka & 3) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) {
        myolo_set_error("gemm_tn_batched: needs Ka %% 4 == 0, N %% 4 == 0 and 16-byte aligned operands");
        return MYOLO_EINVAL;
    }
    const size_t need = myolo_gemm_tn_batched_ws_bytes(M, Ka, N, batch);
    if (need > ws_bytes || (need && !ws)) {
        myolo_set_error("gemm_tn_batched: workspace too small (%zu needed, %zu given)", need, ws_bytes);
        return MYOLO_EWORKSPACE;
    }
    const long long per = (long long)batch * Ka * N;
    const int splits = need ? (int)(need / (per * sizeof(float))) : 1;
    GemmArgs a = {};
    a.A = A; a.B = B; a.M = M; a.N = N; a.K = Ka; a.lda = Ka; a.ldb = N;
    a.sA = M * (long long)Ka; a.sB = M * (long long)N; a.batch = batch;
    long long mps = cdiv64(M, splits);
    a.m_per_split = cdiv64(mps, BK) * BK;
    a.C = splits > 1 ? (float*)ws : C;
    const int tiles = ((Ka + BM - 1) / BM) * ((N + BN - 1) / BN);
    hipLaunchKernelGGL((gemm_tn_fast<AM_PLAIN>), dim3(tiles, splits, batch), dim3(256), 0, s, a);
    if (splits > 1) {
        int blocks = (int)((per / 4 + 255) / 256);
        if (blocks > 2048) blocks = 2048;
        if (blocks < 1) blocks = 1;
        hipLaunchKernelGGL(splitk_reduce, dim3(blocks), dim3(256), 0, s, (const float*)ws, C, per, splits);
    }
    return MYOLO_OK;
}

extern "C" {

size_t myolo_workspace_bytes(int64_t rows, int cin, int cout)
{
    // largest user: weight-gradient split-K partials of a 3x3 conv (Ka = 9*cin) or deconv (Ka = 4*cout)
    size_t a = tn_ws_bytes(rows, 9 * cin, cout);
    size_t b = tn_ws_bytes(rows, 4 * cout, cin);
    size_t c = (size_t)9 * cin * cout * sizeof(float);          // transformed weights
    size_t m = a > b ? a : b;
    return align256(m > c ? m : c) + align256(c) + (size_t)(1 << 20);
}

}  // extern "C"

// ---- pointwise conv with a handful of output columns (conv_23: 1024 -> N_BOX*(5+classes) = 35 or 40; model.py:271) ----
// Neither MFMA kernel takes an N that is not a multiple of 4, and the generic one walks K = 1024 serially in a few workgroups (138 us for
// 676 x 1024 x 35).  Here a workgroup owns four rows, a lane is an output column, each of the four waves takes a quarter of K and the
// partial sums meet in LDS: x is read once (16-byte broadcast loads), w once per workgroup from L2.
template <int NW>
__global__ __launch_bounds__(64 * NW) void pw_skinny_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                               float* __restrict__ y, long long M, int K, int N)
{
    // NW waves split K (NW = 16 from K = 512 up: the loop is a chain of load latencies -- 64 steps per wave at K = 1024 with four waves took 32 us for
    // 0.05 GFLOP, on the critical path between the trunk and the YOLO loss / decode; sixteen waves walk 16 steps each)
    __shared__ float red[NW][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.x * 4;
    const int kq = K / NW, k0 = wave * kq;
    const int col = lane < N ? lane : 0;
    const float* xr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) xr[r] = x + (r0 + r < M ? r0 + r : M - 1) * (long long)K + k0;
    const float* wp = w + (long long)k0 * N + col;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int k = 0; k < kq; k += 4) {        // eight iterations' loads in flight: the loop is pure load latency
        const float w0 = wp[(long long)k * N], w1 = wp[(long long)(k + 1) * N], w2 = wp[(long long)(k + 2) * N], w3 = wp[(long long)(k + 3) * N];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float4 xv = ld4(xr[r] + k);
            acc[r] = fmaf(xv.x, w0, acc[r]);
            acc[r] = fmaf(xv.y, w1, acc[r]);
            acc[r] = fmaf(xv.z, w2, acc[r]);
            acc[r] = fmaf(xv.w, w3, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wave][r][lane] = acc[r];
    __syncthreads();
    // wave r finishes row r: the NW partial sums in pairs of neighbours, then in wave order
    const long long row = r0 + wave;
    if (wave < 4 && row < M && lane < N) {
        float v = 0.f;
        if (NW == 4) v = (red[0][wave][lane] + red[1][wave][lane]) + (red[2][wave][lane] + red[3][wave][lane]);
        else {
#pragma unroll
            for (int q = 0; q < NW; q += 2) v += red[q][wave][lane] + red[q + 1][wave][lane];
        }
        if (bias) v += bias[lane];
        y[row * N + lane] = v;
    }
}

// Data gradient of the THIN pointwise layers (conv_pw_1..4: 32-128 input channels, up to 256 output channels, 25 088 - 401 408 rows):
//     dx [M][N = Cin] = dy [M][K = Cout] * w^T,  w [N][K] -- already the [n][k] operand.
// These launches move 40-150 MB and a handful of MFLOP per row tile; through gemm_nn_fast (128-row LDS tiles built for big K) they ran at
// 1.0-1.5 TB/s at the very end of the step's backward chain.  Here a wave owns 32 rows and ALL N columns and feeds v_mfma_f32_32x32x2f32
// straight from registers: lane (row l31, half h) loads 16 bytes of its row per step, k = 8 j + 4 h .. + 3, and the matching 16 bytes of w's row
// n = l31 (+ 32 u) -- the reduction index may be walked in any order as long as both operands agree, so no transposition and no LDS.  Fixed
// order, one accumulator per output: deterministic.  fp32 operands on the fp32 matrix pipe (the same arithmetic as gemm_nn_fast).
template <int NU>
__global__ __launch_bounds__(256) void pw_bwd_data_thin_kernel(const float* __restrict__ dy, const float* __restrict__ w, float* __restrict__ dx,
                                                               long long M, int K, int N)
{
    const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
    const long long nblk = (M + 31) / 32;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwave = (long long)gridDim.x * 4;
    const int nj = K / 8;
    const float* wb = w + (long long)l31 * K + 4 * half;
    for (long long blk = wave0; blk < nblk; blk += nwave) {
        const long long row = blk * 32 + l31;
        const float* ap = dy + (row < M ? row : M - 1) * K + 4 * half;       // (rows past the end repeat the last one; never stored)
        f32x16 acc[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
#pragma unroll 2
        for (int j = 0; j < nj; ++j) {
            const float4 a = *reinterpret_cast<const float4*>(ap + 8 * j);
            float4 b[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) b[u] = *reinterpret_cast<const float4*>(wb + (long long)u * 32 * K + 8 * j);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[u].x, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[u].y, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[u].z, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[u].w, acc[u], 0, 0, 0);
            }
        }
        // acc[u][r]: row (r & 3) + 8 (r >> 2) + 4 half of the block, column 32 u + l31: 128-byte row segments per store
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long orow = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (orow < M) {
#pragma unroll
                for (int u = 0; u < NU; ++u) dx[orow * N + 32 * u + l31] = acc[u][r];
            }
        }
    }
}

// Forward of the THIN pointwise layers in training (conv_pw_1..3: 32 / 64 input, 64 / 128 output channels, 100 352 - 401 408 rows), the mirror of the
// kernel above:  y [M][N] = act(x * in_scale + in_shift) [M][K] * w [K][N], column sums of y and y^2 for the BatchNorm that follows.
// A wave owns 32 rows and all N columns; lane (row l31, half h) loads 16 bytes of its row per step (k = 8 j + 4 h .. + 3), normalises them in
// registers, and reads the matching 16 bytes of w^T's row n from LDS (w, 8-32 KB, transposed into LDS once per workgroup, rows padded to K + 4
// floats: conflict-free 16-byte reads).  The statistics: 16 rows per lane in fp32, then doubles -- over the wave's row blocks, the two half-waves,
// the four waves (LDS); one row of partials per workgroup, summed in a fixed order by the finish kernel: bit-reproducible.
template <int NU, int NJ>
__global__ __launch_bounds__(256) void pw_fwd_thin_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                          double* __restrict__ stat, long long M, const float* __restrict__ a_scale,
                                                          const float* __restrict__ a_shift, int a_act, const float* __restrict__ o_scale = nullptr,
                                                          const float* __restrict__ o_shift = nullptr, int o_act = MYOLO_ACT_NONE)
{
    constexpr int K = 8 * NJ, N = 32 * NU, LDW = K + 4;
    extern __shared__ __align__(16) float pwt_lds[];             // [N][K + 4] floats, afterwards [4][2][N] doubles
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    for (int e = tid; e < K * N; e += 256) {
        const int k = e / N, n = e - k * N;
        pwt_lds[n * LDW + k] = w[e];
    }
    float4 sc[NJ], sh[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        sc[j] = a_scale ? *reinterpret_cast<const float4*>(a_scale + 8 * j + 4 * half) : make_float4(1.f, 1.f, 1.f, 1.f);
        sh[j] = a_scale ? *reinterpret_cast<const float4*>(a_shift + 8 * j + 4 * half) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int act = a_scale ? a_act : MYOLO_ACT_NONE;
    // inference (o_scale): the frozen BatchNorm + activation behind the conv on the way out (splitk_epilogue's / bn_apply_kernel's expressions); no statistics then
    float osc[NU], osh[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) { osc[u] = o_scale ? o_scale[32 * u + l31] : 1.f; osh[u] = o_scale ? o_shift[32 * u + l31] : 0.f; }
    __syncthreads();
    const long long nblk = (M + 31) / 32;
    const long long wave0 = (long long)blockIdx.x * 4 + wave, nwave = (long long)gridDim.x * 4;
    const float* wb = pwt_lds + l31 * LDW + 4 * half;
    double d1[NU], d2[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) { d1[u] = 0.0; d2[u] = 0.0; }
    for (long long blk = wave0; blk < nblk; blk += nwave) {
        const long long row = blk * 32 + l31;
        const float* ap = x + (row < M ? row : M - 1) * K + 4 * half;        // (rows past the end repeat the last one; never stored, never summed)
        float4 a[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) a[j] = *reinterpret_cast<const float4*>(ap + 8 * j);
        f32x16 acc[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            float4 v = a[j];
            v.x = gemm_act(fmaf(v.x, sc[j].x, sh[j].x), act); v.y = gemm_act(fmaf(v.y, sc[j].y, sh[j].y), act);
            v.z = gemm_act(fmaf(v.z, sc[j].z, sh[j].z), act); v.w = gemm_act(fmaf(v.w, sc[j].w, sh[j].w), act);
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const float4 b = *reinterpret_cast<const float4*>(wb + u * 32 * LDW + 8 * j);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x, b.x, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.y, b.y, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.z, b.z, acc[u], 0, 0, 0);
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.w, b.w, acc[u], 0, 0, 0);
            }
        }
        // acc[u][r]: row (r & 3) + 8 (r >> 2) + 4 half of the block, column 32 u + l31
        float s1[NU], s2[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) { s1[u] = 0.f; s2[u] = 0.f; }
        const bool full = blk * 32 + 32 <= M;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long orow = blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (full || orow < M) {
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    float t = acc[u][r];
                    if (o_scale) t = gemm_act(fmaf(t, osc[u], osh[u]), o_act);
                    y[orow * N + 32 * u + l31] = t;
                    s1[u] += t; s2[u] = fmaf(t, t, s2[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) { d1[u] += (double)s1[u]; d2[u] += (double)s2[u]; }
    }
    if (!stat) return;
    __syncthreads();                                         // every wave is done with w^T in LDS
    double* red = reinterpret_cast<double*>(pwt_lds);          // [4 waves][2][N]
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        d1[u] += __shfl_xor(d1[u], 32, 64);
        d2[u] += __shfl_xor(d2[u], 32, 64);
        if (half == 0) { red[(wave * 2 + 0) * N + 32 * u + l31] = d1[u]; red[(wave * 2 + 1) * N + 32 * u + l31] = d2[u]; }
    }
    __syncthreads();
    for (int e = tid; e < 2 * N; e += 256)
        stat[(long long)blockIdx.x * 2 * N + e] = (red[0 * 2 * N + e] + red[1 * 2 * N + e]) + (red[2 * 2 * N + e] + red[3 * 2 * N + e]);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Pointwise layers with FEW rows and a long K (the 7x7 layers of the training step, M = 1568; every layer of the YOLO head of an inference
// forward, M = 676 / 2704 at Rice-416 batch 4): y [M][N] = act((act_in(x * a_scale + a_shift) [M][K] * w [K][N] + bias) * scale + shift).
// gemm_nn_fast ran them as split-K launches of 128 x 128 tiles -- raw partials to HBM, a second launch (splitk_epilogue) to sum them, and
// 352 workgroups on 256 CUs: 28 + 9 us for 1.4 GFLOP (M = 2704, 512 -> 512) where the fp32 matrix pipe needs 9.
// Here a workgroup owns 32 T rows x 128 columns and its four waves split K among themselves: a wave multiplies the whole tile over a quarter
// of K straight from registers (no LDS in the loop, no barrier), the four partial tiles meet in LDS once, are summed in wave order
// (deterministic) and leave through the epilogue as 16-byte stores.  One launch, no partials in HBM, at most 256 workgroups of equal work (one round).
// Operand layout without a transposition: lane (l31, half) loads 16 bytes of its ROW of x per step (k = 8 j + 4 half + e, e = 0..3, the
// thin kernels' order: the reduction index may be walked in any order as long as both operands agree) and, for each e, 16 bytes of w's row
// k: columns n0 + 4 l31 .. + 3.  Component c of that float4 is the B operand of column block c, i.e. block c holds the columns
// n0 + 4 l + c -- a permutation of the tile's columns that the epilogue undoes for free (a lane ends up with four CONSECUTIVE columns).
// fp32 operands on the fp32 matrix pipe: the arithmetic of gemm_nn_fast, another summation order.
template <int T>
__global__ __launch_bounds__(256) void pw_smallm_kernel(GemmArgs p)
{
    __shared__ __attribute__((aligned(16))) float part[4 * 4 * T * 16 * 64];       // [wave][column block][t][r][lane]: 64 T KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
    const int ntn = p.N >> 7;
    const int tn = (int)(blockIdx.x % (unsigned)ntn);                              // the column tiles of one row tile are neighbours: x stays in L2
    const long long m0 = (long long)(blockIdx.x / (unsigned)ntn) * (32 * T);
    const int n0 = tn << 7;
    const int kq = p.K >> 2, k0 = wave * kq + 4 * half, nj = kq >> 3;
    const float* ap[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const long long row = m0 + 32 * t + l31;
        ap[t] = p.A + (row < p.M ? row : p.M - 1) * p.lda + k0;                    // (rows past the end repeat the last one; never stored)
    }
    const float* bp = p.B + (long long)k0 * p.ldb + n0 + 4 * l31;
    const bool aff = p.a_scale != nullptr;
    const float* scp = aff ? p.a_scale + k0 : p.A;                                 // (never dereferenced beyond valid floats when unused: same offsets as x's row)
    const float* shp = aff ? p.a_shift + k0 : p.A;
    const int a_act = aff ? p.a_act : MYOLO_ACT_NONE;

    f32x16 acc[T][4];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][c][r] = 0.f;

    // three register stages: step j's operands are requested two steps ahead (a step is 16 T MFMAs = 1024 T cycles, an L2 hit ~2000)
    struct Stage { float4 a[T], b[4], sc, sh; };
    Stage s0, s1, s2;
    auto load = [&](Stage& st, int j) {
        const int jj = j < nj ? j : nj - 1;                                        // past the end: the last step again (no load inside a branch)
#pragma unroll
        for (int t = 0; t < T; ++t) st.a[t] = ld4(ap[t] + 8 * jj);
#pragma unroll
        for (int e = 0; e < 4; ++e) st.b[e] = ld4(bp + (long long)(8 * jj + e) * p.ldb);
        st.sc = ld4(scp + 8 * jj); st.sh = ld4(shp + 8 * jj);
    };
    auto step = [&](const Stage& st) {
        float av[T][4];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            av[t][0] = st.a[t].x; av[t][1] = st.a[t].y; av[t][2] = st.a[t].z; av[t][3] = st.a[t].w;
        }
        if (aff) {
            const float sc[4] = {st.sc.x, st.sc.y, st.sc.z, st.sc.w}, sh[4] = {st.sh.x, st.sh.y, st.sh.z, st.sh.w};
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) av[t][e] = gemm_act(fmaf(av[t][e], sc[e], sh[e]), a_act);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float bv[4] = {st.b[e].x, st.b[e].y, st.b[e].z, st.b[e].w};
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[t][c] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t][e], bv[c], acc[t][c], 0, 0, 0);
        }
    };
    load(s0, 0); load(s1, 1); load(s2, 2);
    for (int j = 0; j < nj; j += 3) {
        step(s0); load(s0, j + 3);
        if (j + 1 < nj) step(s1);
        load(s1, j + 4);
        if (j + 2 < nj) step(s2);
        load(s2, j + 5);
    }

    // the four waves' partial tiles meet in LDS ...
    float* mine = part + wave * (4 * T * 1024);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) mine[((c * T + t) * 16 + r) * 64 + lane] = acc[t][c][r];
    __syncthreads();
    // ... and wave w finishes the rows r = 4 w .. 4 w + 3 of every 32-row block: the partials summed in wave order, then splitk_epilogue's expressions
    float bs[4] = {0.f, 0.f, 0.f, 0.f}, cs[4], ct[4];
    const int col = n0 + 4 * l31;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (p.bias) bs[e] = p.bias[col + e];
        col_affine(p, col + e, cs[e], ct[e]);
    }
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int r = 4 * wave + rr;
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float sacc = 0.f;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) sacc += part[wv * (4 * T * 1024) + ((c * T + t) * 16 + r) * 64 + lane];
                v[c] = sacc;
            }
            const long long row = m0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row < p.M) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (p.bias) v[e] += bs[e];
                    if (p.scale) v[e] = fmaf(v[e], cs[e], ct[e]);
                    v[e] = gemm_act(v[e], p.act);
                }
                *reinterpret_cast<float4*>(p.C + row * p.ldc + col) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
}

// 32 T = rows of a workgroup's tile for this shape, 0 = not this kernel (too many rows: the big-tile kernels fill the chip; short K: nothing to split)
static int pw_smallm_T(long long M, int K, int N)
{
    if (g_myolo_opt.pw_no_smallm || (N & 127) != 0 || (K & 31) != 0 || K < 256 || M <= 0) return 0;
    const long long w1 = cdiv64(M, 32) * (N >> 7), w2 = cdiv64(M, 64) * (N >> 7);
    if (w1 <= 256) return 1;                // (measured both ways per shape, tools/experiments/pw_smallm.py: 32-row tiles win exactly when they fit one round)
    return w2 <= 256 ? 2 : 0;               // (257-384 tiles of 64 rows = two rounds: conv_pw_5 of an inference forward, 10816 x 256 -> 256, ran 36.6 us here against 30.1 on gemm_nn_fast)
}
static bool pw_smallm_ok(const GemmArgs& a)
{
    return pw_smallm_T(a.M, a.K, a.N) != 0 && (((uintptr_t)a.A | (uintptr_t)a.B | (uintptr_t)a.C) & 15) == 0 && (a.lda & 3) == 0 && (a.ldb & 3) == 0 &&
           (a.ldc & 3) == 0 && !g_myolo_opt.gemm_generic;
}
static void pw_smallm_launch(const GemmArgs& a, hipStream_t s)
{
    const int T = pw_smallm_T(a.M, a.K, a.N);
    const unsigned wgs = (unsigned)(cdiv64(a.M, 32 * T) * (a.N >> 7));
    if (T == 1) hipLaunchKernelGGL(pw_smallm_kernel<1>, dim3(wgs), dim3(256), 0, s, a);
    else hipLaunchKernelGGL(pw_smallm_kernel<2>, dim3(wgs), dim3(256), 0, s, a);
}

static bool pw_fwd_thin_ok(long long M, int Cin, int Cout)
{
    // 128 output channels (conv_pw_3, 100 352 rows = one row block per wave): 43 us here against 36 us on gemm_nn_fast -- the 32 KB of w^T every
    // workgroup stages are not amortised; tune0 & 8192 lets the test reach that instantiation
    return (Cin == 32 || Cin == 64) && (Cout == 64 || (Cout == 128 && (g_myolo_opt.tune0 & 8192))) && M >= 8192 && !(g_myolo_opt.tune0 & 4096) &&
           !g_myolo_opt.no_trunk_fusion;
}

// workgroups of pw_fwd_thin_kernel = rows of statistics partials it writes (never more than the 128-row tiles the scratch is sized for)
static int pw_fwd_thin_wgs(long long M)
{
    long long wgs = ((M + 31) / 32 + 3) / 4;
    if (wgs > 2048) wgs = 2048;
    return (int)wgs;
}

static void pw_fwd_thin_launch(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w, float* y, double* stat,
                               long long M, int Cin, int Cout, hipStream_t s, const float* o_scale = nullptr, const float* o_shift = nullptr,
                               int o_act = MYOLO_ACT_NONE)
{
    const unsigned wgs = (unsigned)pw_fwd_thin_wgs(M);
    size_t lds = (size_t)Cout * (Cin + 4) * sizeof(float);
    if (lds < (size_t)8 * Cout * sizeof(double)) lds = (size_t)8 * Cout * sizeof(double);
#define PWT(NU_, NJ_) hipLaunchKernelGGL((pw_fwd_thin_kernel<NU_, NJ_>), dim3(wgs), dim3(256), lds, s, x, w, y, stat, M, in_scale, in_shift, in_act, o_scale, o_shift, o_act)
    if (Cin == 32 && Cout == 64) PWT(2, 4);
    else if (Cin == 32) PWT(4, 4);
    else if (Cout == 64) PWT(2, 8);
    else PWT(4, 8);
#undef PWT
}

static bool pw_bwd_data_thin_ok(long long M, int Cin, int Cout)
{
    // (Cin = 128, i.e. conv_pw_4 with 25 088 rows: 784 waves of 32 steps x 16 MFMAs each do not fill the chip -- gemm_nn_fast keeps that layer)
    return (Cin == 32 || Cin == 64) && Cout <= 256 && (Cout % 8) == 0 && M >= 8192 && !(g_myolo_opt.tune0 & 2048);
}


extern "C" {

int myolo_pwconv1x1_fwd(const float* x, const float* w, const float* bias, float* y,
                        int64_t M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && w && y && M > 0 && Cin > 0 && Cout > 0, "pwconv1x1_fwd: bad arguments");
    GemmArgs a = {};
    a.A = x; a.B = w; a.C = y; a.bias = bias; a.M = M; a.N = Cout; a.K = Cin;
    a.lda = Cin; a.ldb = Cout; a.ldc = Cout; a.act = MYOLO_ACT_NONE;
    if (Cout <= 64 && (Cout & 3) != 0 && (Cin & 15) == 0 && ((uintptr_t)x & 15) == 0 && !g_myolo_opt.gemm_generic) {
        if (Cin >= 512 && (Cin & 63) == 0 && !g_myolo_opt.pw_skinny_nw4)
            hipLaunchKernelGGL(pw_skinny_fwd_kernel<16>, dim3((unsigned)cdiv64(M, 4)), dim3(1024), 0, (hipStream_t)stream, x, w, bias, y, (long long)M, Cin, Cout);
        else
            hipLaunchKernelGGL(pw_skinny_fwd_kernel<4>, dim3((unsigned)cdiv64(M, 4)), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, (long long)M, Cin, Cout);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    if (!bias && pw_fwd_thin_ok(M, Cin, Cout) && (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0 && !g_myolo_opt.gemm_generic) {
        // the thin layers (conv_pw_1 / 2) on the register-fed kernel of the training forward, without statistics: the same bits as
        // myolo_pwconv1x1_bnstats_fwd's y, and what myolo_pwconv1x1_affine_act_fwd applies its affine to
        pw_fwd_thin_launch(x, nullptr, nullptr, MYOLO_ACT_NONE, w, y, nullptr, M, Cin, Cout, (hipStream_t)stream);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    if (pw_smallm_ok(a)) {                  // few rows, long K: one launch, the waves of a workgroup split K (pw_smallm_kernel)
        pw_smallm_launch(a, (hipStream_t)stream);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    // ws (optional): split-K partials for the 14x14 / 7x7 layers, whose few output tiles and long K loop (a serial chain of
    // load -> LDS -> MFMA steps) would leave most of the chip idle
    // (measured, tools/pw_layers.py: 7x7 layers 83 -> 52 us and 77 -> 30 us; the 14x14 layers' 196 tiles are better left alone)
    launch_nn<AM_PLAIN, EP_PLAIN>(a, (hipStream_t)stream, ws, ws_bytes, 128);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* inference: y = act((x @ w) * scale + shift) in one launch -- the frozen BatchNorm after a pointwise conv, folded (scale / shift from
 * myolo_bn_frozen_coeffs[_batched]); equals myolo_pwconv1x1_fwd followed by myolo_bn_apply_act bit for bit (model.py:68-76 with the
 * BatchNormalization layers in inference mode) */
int myolo_pwconv1x1_affine_act_fwd(const float* x, const float* w, const float* scale, const float* shift, int act, float* y,
                                   int64_t M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && w && y && scale && shift && M > 0 && Cin > 0 && Cout > 0, "pwconv1x1_affine_act_fwd: bad arguments");
    GemmArgs a = {};
    a.A = x; a.B = w; a.C = y; a.M = M; a.N = Cout; a.K = Cin;
    a.lda = Cin; a.ldb = Cout; a.ldc = Cout; a.act = act;
    a.scale = scale; a.shift = shift;
    if (pw_fwd_thin_ok(M, Cin, Cout) && (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0 && !g_myolo_opt.gemm_generic) {
        // conv_pw_1 / 2 at inference sizes: 33.9 / 19.9 us on gemm_nn_fast (Rice-416, batch 4); the affine + activation on the accumulators
        pw_fwd_thin_launch(x, nullptr, nullptr, MYOLO_ACT_NONE, w, y, nullptr, M, Cin, Cout, (hipStream_t)stream, scale, shift, act);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    if (pw_smallm_ok(a)) {                  // the YOLO head of an inference forward (M = 676 / 2704 at Rice-416 batch 4): no split-K pair of launches
        pw_smallm_launch(a, (hipStream_t)stream);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    launch_nn<AM_PLAIN, EP_PLAIN>(a, (hipStream_t)stream, ws, ws_bytes, 128);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_pwconv1x1_bwd_data(const float* dy, const float* w, float* dx,
                             int64_t M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(dy && w && dx && M > 0, "pwconv1x1_bwd_data: bad arguments");
    if (pw_bwd_data_thin_ok(M, Cin, Cout) && (((uintptr_t)dy | (uintptr_t)w | (uintptr_t)dx) & 15) == 0) {
        long long wgs = ((M + 31) / 32 + 3) / 4;
        if (wgs > 4096) wgs = 4096;                       // (16 workgroups per CU: the waves walk the row blocks)
        hipStream_t st = (hipStream_t)stream;
        if (Cin == 32) hipLaunchKernelGGL(pw_bwd_data_thin_kernel<1>, dim3((unsigned)wgs), dim3(256), 0, st, dy, w, dx, (long long)M, Cout, Cin);
        else hipLaunchKernelGGL(pw_bwd_data_thin_kernel<2>, dim3((unsigned)wgs), dim3(256), 0, st, dy, w, dx, (long long)M, Cout, Cin);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    MYOLO_NEED_WS((size_t)Cin * Cout * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    if (g_myolo_opt.wino_x6 && !g_myolo_opt.pw_no_x6 && (Cin % 256) == 0 && (Cout % 16) == 0 && M >= pw_x6_min_rows() &&
        ws_bytes >= myolo_matmul_f32_ws_bytes(Cout, Cin, 1, MYOLO_PRODUCTS_BF16X6) && (((uintptr_t)dy | (uintptr_t)w | (uintptr_t)dx) & 15) == 0) {
        // FP32_MATMUL = "bf16x6": dx [M][Cin] = dy [M][Cout] * w^T, and w [Cin][Cout] IS the transposed operand [N][K] the NT kernel wants:
        // no transpose launch, six exact bf16 piece products per fp32 product on the bf16 matrix pipe (csrc/wino_mm.hip)
        return myolo_matmul_f32_impl(dy, w, dx, M, Cout, Cin, 1, MYOLO_PRODUCTS_BF16X6, ws, ws_bytes, stream, true);
    }
    // w^T [Cout][Cin]: into ws, or already prepared for this step (prepared-weights registry, csrc/myolo_common.h)
    const float* wt = (const float*)myolo_wprep_resolve(w, WP_TRANSPOSE, Cin, Cout, 10, (size_t)Cin * Cout * sizeof(float), ws, s,
                                                        [=](void* d, hipStream_t st) { launch_transpose(w, (float*)d, Cin, Cout, 1, 0, st); });
    GemmArgs a = {};
    a.A = dy; a.B = wt; a.C = dx; a.M = M; a.N = Cin; a.K = Cout;
    a.lda = Cout; a.ldb = Cin; a.ldc = Cin;
    const size_t wbytes = align256((size_t)Cin * Cout * sizeof(float));
    if (pw_smallm_ok(a)) {                  // the 7x7 layers' data gradients (M = 1568)
        pw_smallm_launch(a, s);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    launch_nn<AM_PLAIN, EP_PLAIN>(a, s, (char*)ws + wbytes, ws_bytes > wbytes ? ws_bytes - wbytes : 0);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

static int pw_bwd_weight_impl(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* dy, float* dw,
                              int64_t M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream);

int myolo_pwconv1x1_bwd_weight(const float* x, const float* dy, float* dw,
                               int64_t M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    return pw_bwd_weight_impl(x, nullptr, nullptr, MYOLO_ACT_NONE, dy, dw, M, Cin, Cout, ws, ws_bytes, stream);
}

/* the same gradient when the conv's input was act_in(x * in_scale + in_shift) formed on load (myolo_pwconv1x1_bnstats_fwd): x is the
 * producing layer's pre-BN output, normalised again on load here.  Needs Cin % 4 == 0, Cout % 4 == 0 (the fast kernels). */
int myolo_pwconv1x1_bwd_weight_affine_in(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* dy, float* dw,
                                         int64_t M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(in_scale && in_shift && (Cin & 3) == 0 && (Cout & 3) == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0 &&
                  !g_myolo_opt.gemm_generic, "pwconv1x1_bwd_weight_affine_in: needs in_scale / in_shift, Cin %% 4 == 0, Cout %% 4 == 0, 16-byte aligned operands");
    return pw_bwd_weight_impl(x, in_scale, in_shift, in_act, dy, dw, M, Cin, Cout, ws, ws_bytes, stream);
}

}  // extern "C"

static int pw_bwd_weight_impl(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* dy, float* dw,
                              int64_t M, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && dy && dw && M > 0, "pwconv1x1_bwd_weight: bad arguments");
    if ((Cin == 32 || Cin == 64) && (Cout == 64 || Cout == 128) && M >= 16384 && !g_myolo_opt.gemm_generic) {
        // thin layer: one wave holds the whole result (pw_wgrad_thin)
        const int nblk = M >= 300000 ? 512 : 256;     // measured: tools/pw_layers.py (tune0 is a bit mask of A/B switches: it must not size anything here)
        const size_t need = (size_t)nblk * Cin * Cout * sizeof(float);
        if (ws && need <= ws_bytes) {
            hipStream_t s = (hipStream_t)stream;
            float* part = (float*)ws;
#define THIN(KT_, NT_) hipLaunchKernelGGL((pw_wgrad_thin<KT_, NT_>), dim3(nblk), dim3(256), 0, s, x, dy, part, (long long)M, Cin, Cout, in_scale, in_shift, in_act)
            if (Cin == 32 && Cout == 64) THIN(1, 2);
            else if (Cin == 32 && Cout == 128) THIN(1, 4);
            else if (Cin == 64 && Cout == 64) THIN(2, 2);
            else THIN(2, 4);
#undef THIN
            const int n = Cin * Cout;
            hipLaunchKernelGGL(partial_sum_small, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, (const float*)part, dw, n, nblk);
            MYOLO_CHECK_LAUNCH();
            return MYOLO_OK;
        }
    }
    if (g_myolo_opt.wino_x6 && !g_myolo_opt.pw_no_x6 && myolo_gemm_tn_x6_ok(Cin, Cout) && M >= pw_x6_min_rows() && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw) & 15) == 0) {
        // FP32_MATMUL = "bf16x6": dw [Cin][Cout] = x^T dy on wino_tn_x6_kernel (256 x 256 tiles, M split over workgroups, fixed-order reduce)
        const long long rows[1] = {M}, off[1] = {0};
        const int nq[1] = {1};
        if (myolo_gemm_tn_x6_ws_bytes(1, rows, nq, Cin, Cout) <= ws_bytes && ws) {
            const int rc = myolo_gemm_tn_x6_runs(x, dy, dw, 1, rows, off, off, nq, Cin, Cout, ws, ws_bytes, (hipStream_t)stream, in_scale, in_shift, in_act);
            if (rc) return rc;
            MYOLO_CHECK_LAUNCH();
            return MYOLO_OK;
        }
    }
    GemmArgs a = {};
    a.A = x; a.B = dy; a.M = M; a.N = Cout; a.K = Cin; a.lda = Cin; a.ldb = Cout;
    a.a_scale = in_scale; a.a_shift = in_shift; a.a_act = in_act;
    int rc = launch_tn<AM_PLAIN>(a, dw, ws, ws_bytes, (hipStream_t)stream, "pwconv1x1_bwd_weight");
    if (rc) return rc;
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

extern "C" {

/* Training-mode depthwise block, second half (keras_applications _depthwise_conv_block, model.py:68-77 / 256-268, BatchNormalization on
 * batch statistics): y = act_in(x * in_scale + in_shift) @ w and the BatchNorm statistics of y (what myolo_bn_stats gives: mean, var,
 * folded scale / shift, moving averages) -- the producing layer's BatchNorm + ReLU6 applied while the GEMM loads its A operand, the
 * column sums of the output taken in the GEMM's epilogue; two launches (GEMM, finish).  in_scale == NULL: x as it is.
 * Needs Cin % 16 == 0, Cout % 4 == 0, 16-byte aligned operands (myolo_pwconv1x1_bnstats_ok). */
int myolo_pwconv1x1_bnstats_ok(int Cin, int Cout) { return (Cin % BK) == 0 && (Cout & 3) == 0 && !g_myolo_opt.gemm_generic ? 1 : 0; }

// scratch of the split-K form the launcher may pick for the small deep layers (fewer than 128 output tiles): 8 partial outputs
static size_t pw_split_bytes(int64_t M, int Cout)
{
    const long long tiles = cdiv64(M, BM) * ((Cout + BN - 1) / BN);
    return tiles < 128 ? align256((size_t)8 * M * Cout * sizeof(float)) : 0;
}

size_t myolo_pwconv1x1_bnstats_ws_bytes(int64_t M, int Cin, int Cout)
{
    const size_t tiles = (size_t)cdiv64(M, BM);
    const size_t fused = align256(tiles * 2 * Cout * sizeof(double)) + align256(2 * Cout * sizeof(double)) +
                         ((Cin >= 128 && (Cout % 256) == 0) ? myolo_pw_x6_split_bytes(Cin, Cout) : 0);
    // split-K path / ablation: the partial outputs, then a statistics pass over y (<= 1024 slabs of 2*Cout doubles)
    const size_t split = pw_split_bytes(M, Cout) + align256((size_t)1024 * 2 * Cout * sizeof(double)) + align256(2 * Cout * sizeof(double));
    return fused > split ? fused : split;
}

}  // extern "C"

// statistics pass over a [M][C] tensor + finish (csrc/mem_kernels.hip)
int myolo_bn_stats_launch(const float* x, const float* gamma, const float* beta, float* mean, float* var, float* scale, float* shift,
                          float* moving_mean, float* moving_var, long long M, int C, void* ws, size_t ws_bytes, hipStream_t s);

extern "C" {

int myolo_pwconv1x1_bnstats_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w, float* y,
                                const float* gamma, const float* beta, float* mean, float* var, float* scale, float* shift,
                                float* moving_mean, float* moving_var, int64_t M, int Cin, int Cout, int phases, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && w && y && gamma && beta && mean && var && scale && shift && M > 0 && !in_scale == !in_shift && (phases & 3) != 0, "pwconv1x1_bnstats_fwd: bad arguments");
    MYOLO_REQUIRE(myolo_pwconv1x1_bnstats_ok(Cin, Cout) && ((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0,
                  "pwconv1x1_bnstats_fwd: needs Cin %% 16 == 0, Cout %% 4 == 0 and 16-byte aligned operands");
    MYOLO_NEED_WS(myolo_pwconv1x1_bnstats_ws_bytes(M, Cin, Cout));
    hipStream_t s = (hipStream_t)stream;
    GemmArgs a = {};
    a.A = x; a.B = w; a.C = y; a.M = M; a.N = Cout; a.K = Cin;
    a.lda = Cin; a.ldb = Cout; a.ldc = Cout; a.act = MYOLO_ACT_NONE;
    a.a_scale = in_scale; a.a_shift = in_shift; a.a_act = in_act;
    const int tiles = (int)cdiv64(M, BM);
    const size_t pbytes = align256((size_t)tiles * 2 * Cout * sizeof(double));
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + pbytes);
    if (myolo_pw_x6_ok(Cin, Cout) && M >= pw_x6_min_rows() && !g_myolo_opt.no_trunk_fusion) {
        // FP32_MATMUL = "bf16x6": the layers with >= 256 input and a multiple of 256 output channels on the bf16 matrix pipe (six exact piece
        // products per fp32 product, csrc/wino_mm.hip), same on-load BatchNorm and epilogue column sums; three launches (split, GEMM, finish)
        void* split = (char*)ws + pbytes + align256(2 * Cout * sizeof(double));
        if (phases & 1) myolo_pw_x6_fwd(x, in_scale, in_shift, in_act, w, y, part, M, Cin, Cout, split, s);
        MYOLO_CHECK_LAUNCH();
        if (phases & 2) myolo_bn_stats_from_partials(part, tot, tiles, Cout, (double)M, gamma, beta, mean, var, scale, shift, moving_mean, moving_var, s);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    if (pw_fwd_thin_ok(M, Cin, Cout)) {
        // conv_pw_1..3: the register-fed kernel (pw_fwd_thin_kernel); its partial rows are one per workgroup
        const int wgs = pw_fwd_thin_wgs(M);
        if (phases & 1) pw_fwd_thin_launch(x, in_scale, in_shift, in_act, w, y, part, M, Cin, Cout, s);
        MYOLO_CHECK_LAUNCH();
        if (phases & 2) myolo_bn_stats_from_partials(part, tot, wgs, Cout, (double)M, gamma, beta, mean, var, scale, shift, moving_mean, moving_var, s);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    if (pw_smallm_ok(a)) {
        // the 7x7 layers (M = 1568): the small-M kernel with the producing BatchNorm on its loads, then a statistics pass over y (a few MB) as
        // behind the split-K pair it replaces
        if (phases & 1) pw_smallm_launch(a, s);
        MYOLO_CHECK_LAUNCH();
        if (!(phases & 2)) return MYOLO_OK;
        return myolo_bn_stats_launch(y, gamma, beta, mean, var, scale, shift, moving_mean, moving_var, M, Cout, ws, ws_bytes, s);
    }
    a.stat = g_myolo_opt.no_trunk_fusion ? nullptr : part;
    // the split-K scratch shares ws with the partials: when the launcher picks split-K it drops a.stat (nothing is written there)
    const size_t skb = pw_split_bytes(M, Cout);
    // the launcher's own rule (launch_nn): split-K only for < 128 output tiles, K >= 256, and at least two splits of >= 8 k tiles
    const long long otiles = cdiv64(M, BM) * ((Cout + BN - 1) / BN);
    const int nkt = Cin / BK;
    const bool one_pass = !(skb != 0 && !g_myolo_opt.no_splitk && nkt >= 16 && otiles < 128 && (1024 / otiles >= 2) && nkt / 8 >= 2);
    if (phases & 1) {
        int path = 0;
        launch_nn<AM_PLAIN, EP_PLAIN>(a, s, skb ? ws : nullptr, skb, 128, &path);
        MYOLO_CHECK_LAUNCH();
        if ((path == 1) != one_pass && path != 0) { myolo_set_error("pwconv1x1_bnstats_fwd: launcher path %d disagrees with the planned one", path); return MYOLO_EINVAL; }
    }
    if (!(phases & 2)) return MYOLO_OK;
    if (one_pass && !g_myolo_opt.no_trunk_fusion) {
        myolo_bn_stats_from_partials(part, tot, tiles, Cout, (double)M, gamma, beta, mean, var, scale, shift, moving_mean, moving_var, s);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    // split-K (the 7x7 layers: a few MB) or the ablation switch: a statistics pass over y
    return myolo_bn_stats_launch(y, gamma, beta, mean, var, scale, shift, moving_mean, moving_var, M, Cout, (char*)ws + skb, ws_bytes - skb, s);
}

int myolo_conv3x3_fwd(const float* x, const float* w, const float* bias, float* y,
                      int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0, "conv3x3_fwd: bad arguments");
    MYOLO_REQUIRE(Cin % BK == 0, "conv3x3_fwd: Cin must be a multiple of %d (got %d)", BK, Cin);
    GemmArgs a = {};
    a.A = x; a.B = w; a.C = y; a.bias = bias; a.M = (long long)N * H * W; a.N = Cout; a.K = 9 * Cin;
    a.ldb = Cout; a.ldc = Cout; a.H = H; a.W = W; a.Cc = Cin;
    launch_nn<AM_CONV3, EP_PLAIN>(a, (hipStream_t)stream, ws, ws_bytes);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_conv3x3_affine_act_fwd(const float* x, const float* w, const float* bias, const float* scale, const float* shift,
                                 float* y, int N, int H, int W, int Cin, int Cout, int act, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && w && y && scale && shift && N > 0 && H > 0 && W > 0, "conv3x3_affine_act_fwd: bad arguments");
    MYOLO_REQUIRE(Cin % BK == 0, "conv3x3_affine_act_fwd: Cin must be a multiple of %d (got %d)", BK, Cin);
    MYOLO_REQUIRE(act == MYOLO_ACT_NONE || act == MYOLO_ACT_RELU, "conv3x3_affine_act_fwd: act must be NONE or RELU");
    GemmArgs a = {};
    a.A = x; a.B = w; a.C = y; a.bias = bias; a.scale = scale; a.shift = shift; a.act = act;
    a.M = (long long)N * H * W; a.N = Cout; a.K = 9 * Cin;
    a.ldb = Cout; a.ldc = Cout; a.H = H; a.W = W; a.Cc = Cin;
    launch_nn<AM_CONV3, EP_PLAIN>(a, (hipStream_t)stream, ws, ws_bytes);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_conv3x3_bwd_data(const float* dy, const float* w, float* dx,
                           int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(dy && w && dx && N > 0, "conv3x3_bwd_data: bad arguments");
    MYOLO_REQUIRE(Cout % BK == 0, "conv3x3_bwd_data: Cout must be a multiple of %d (got %d)", BK, Cout);
    MYOLO_NEED_WS((size_t)9 * Cin * Cout * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    // ws[tap][co][ci] = w[8-tap][ci][co]  : dx = conv3x3_same(dy, rot180(w)^T)
    const float* wt = (const float*)myolo_wprep_resolve(w, WP_TRANSPOSE, Cin, Cout, 91, (size_t)9 * Cin * Cout * sizeof(float), ws, s,
                                                        [=](void* d, hipStream_t st) { launch_transpose(w, (float*)d, Cin, Cout, 9, 1, st); });
    GemmArgs a = {};
    a.A = dy; a.B = wt; a.C = dx; a.M = (long long)N * H * W; a.N = Cin; a.K = 9 * Cout;
    a.ldb = Cin; a.ldc = Cin; a.H = H; a.W = W; a.Cc = Cout;
    const size_t wbytes = align256((size_t)9 * Cin * Cout * sizeof(float));
    launch_nn<AM_CONV3, EP_PLAIN>(a, s, (char*)ws + wbytes, ws_bytes > wbytes ? ws_bytes - wbytes : 0);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_conv3x3_bwd_weight(const float* x, const float* dy, float* dw,
                             int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && dy && dw && N > 0, "conv3x3_bwd_weight: bad arguments");
    MYOLO_REQUIRE((Cin & 3) == 0, "conv3x3_bwd_weight: Cin must be a multiple of 4 (got %d)", Cin);
    GemmArgs a = {};
    a.A = x; a.B = dy; a.M = (long long)N * H * W; a.N = Cout; a.K = 9 * Cin; a.ldb = Cout;
    a.H = H; a.W = W; a.Cc = Cin;
    int rc = launch_tn<AM_CONV3>(a, dw, ws, ws_bytes, (hipStream_t)stream, "conv3x3_bwd_weight");
    if (rc) return rc;
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_deconv2x2s2_fwd(const float* x, const float* w, const float* bias, float* y,
                          int N, int H, int W, int Cin, int Cout, int act, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && w && y && N > 0, "deconv2x2s2_fwd: bad arguments");
    MYOLO_REQUIRE(Cin > 0 && Cout > 0, "deconv2x2s2_fwd: bad channels");
    MYOLO_NEED_WS((size_t)4 * Cin * Cout * sizeof(float));
    hipStream_t s = (hipStream_t)stream;
    if (myolo_deconv_x6_ok(Cin, Cout, 0) && (long long)N * H * W >= 4096 && ws_bytes >= (size_t)4 * Cin * Cout * 6 &&
        (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0) {
        myolo_deconv_x6_fwd(x, w, bias, y, (long long)N * H * W, H, W, Cin, Cout, act, ws, s);      // FP32_MATMUL = "bf16x6" (csrc/wino_mm.hip)
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    // ws[ci][(ky,kx,co)]
    const float* wt = (const float*)myolo_wprep_resolve(w, WP_TRANSPOSE, 4 * Cout, Cin, 10, (size_t)4 * Cin * Cout * sizeof(float), ws, s,
                                                        [=](void* d, hipStream_t st) { launch_transpose(w, (float*)d, 4 * Cout, Cin, 1, 0, st); });
    GemmArgs a = {};
    a.A = x; a.B = wt; a.C = y; a.bias = bias; a.M = (long long)N * H * W; a.N = 4 * Cout; a.K = Cin;
    a.lda = Cin; a.ldb = 4 * Cout; a.H = H; a.W = W; a.Co = Cout; a.act = act;
    launch_nn<AM_PLAIN, EP_DECONV>(a, s);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

}  // extern "C"

void myolo_launch_deconv_mask_finish(const float* part, const float* b2, float* out, long long npix, int ncls, int nslabs, hipStream_t s)
{
    long long blocks = (npix * ncls + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(deconv_mask_finish, dim3((unsigned)blocks), dim3(256), 0, s, part, b2, out, npix, ncls, nslabs);
}

extern "C" {

size_t myolo_deconv2x2s2_mask_ws_bytes(int N, int H, int W, int Cin, int Cout, int ncls)
{
    // (transposed filters, or their three-bf16-piece split: 6 bytes per value) + the partial logits of the column slabs
    return align256((size_t)4 * Cin * Cout * 6) +
           (size_t)(Cout / BN) * 2 * 4 * N * H * W * ncls * sizeof(float);
}

static int deconv2x2s2_mask_fwd_impl(const float* x, const float* w, const float* bias, const float* w2, const float* b2, float* p_out,
                                     int N, int H, int W, int Cin, int Cout, int ncls, void* ws, size_t ws_bytes, void* stream,
                                     const int32_t* keep_inv, float* keep_d, int keep_cap);

int myolo_deconv2x2s2_mask_fwd(const float* x, const float* w, const float* bias, const float* w2, const float* b2, float* p_out,
                               int N, int H, int W, int Cin, int Cout, int ncls, void* ws, size_t ws_bytes, void* stream)
{
    return deconv2x2s2_mask_fwd_impl(x, w, bias, w2, b2, p_out, N, H, W, Cin, Cout, ncls, ws, ws_bytes, stream, nullptr, nullptr, 0);
}

int myolo_deconv2x2s2_mask_fwd_keep(const float* x, const float* w, const float* bias, const float* w2, const float* b2, float* p_out,
                                    int N, int H, int W, int Cin, int Cout, int ncls, const int32_t* keep_inv, float* keep_d, int keep_cap,
                                    void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(keep_inv && keep_d && keep_cap > 0, "deconv2x2s2_mask_fwd_keep: keep_inv / keep_d / keep_cap missing");
    MYOLO_REQUIRE(myolo_deconv_mask_mm_ok(Cin, Cout), "deconv2x2s2_mask_fwd_keep: needs Cin %% 16 == 0 and Cout %% 256 == 0 (got %d, %d)", Cin, Cout);
    return deconv2x2s2_mask_fwd_impl(x, w, bias, w2, b2, p_out, N, H, W, Cin, Cout, ncls, ws, ws_bytes, stream, keep_inv, keep_d, keep_cap);
}

static int deconv2x2s2_mask_fwd_impl(const float* x, const float* w, const float* bias, const float* w2, const float* b2, float* p_out,
                                     int N, int H, int W, int Cin, int Cout, int ncls, void* ws, size_t ws_bytes, void* stream,
                                     const int32_t* keep_inv, float* keep_d, int keep_cap)
{
    MYOLO_REQUIRE(x && w && bias && w2 && b2 && p_out && N > 0 && H > 0 && W > 0, "deconv2x2s2_mask_fwd: bad arguments");
    MYOLO_REQUIRE(Cout % BN == 0 && Cin % BK == 0 && (Cin & 3) == 0 && ncls >= 1 && ncls <= 4,
                  "deconv2x2s2_mask_fwd: needs Cout %% %d == 0, Cin %% %d == 0, 1 <= classes <= 4 (got %d, %d, %d)", BN, BK, Cout, Cin, ncls);
    const size_t wb = align256((size_t)4 * Cin * Cout * 6);
    MYOLO_NEED_WS(myolo_deconv2x2s2_mask_ws_bytes(N, H, W, Cin, Cout, ncls));
    hipStream_t s = (hipStream_t)stream;
    MYOLO_REQUIRE(((uintptr_t)x & 15) == 0, "deconv2x2s2_mask_fwd: x must be 16-byte aligned");
    if (myolo_deconv_mask_mm_ok(Cin, Cout)) {          // csrc/wino_mm.hip: 128x256 tiles, b128 fragments; bf16x6 with option "wino_x6"
        float* part = (float*)((char*)ws + wb);
        int finished = 0;
        const int rc = myolo_deconv_mask_mm(x, w, bias, w2, part, ws, (long long)N * H * W, H, W, Cin, Cout, ncls, s, keep_inv, keep_d, keep_cap, b2, p_out, &finished);
        if (rc != MYOLO_OK) return rc;
        if (!finished) myolo_launch_deconv_mask_finish(part, b2, p_out, 4ll * N * H * W, ncls, Cout / 128, s);      // a wave covers 128 channels there
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    launch_transpose(w, (float*)ws, 4 * Cout, Cin, 1, 0, s);     // ws[ci][(ky,kx,co)]
    GemmArgs a = {};
    a.A = x; a.B = (const float*)ws; a.bias = bias; a.M = (long long)N * H * W; a.N = 4 * Cout; a.K = Cin;
    a.lda = Cin; a.ldb = 4 * Cout; a.H = H; a.W = W; a.Co = Cout; a.act = MYOLO_ACT_RELU;
    a.w2 = w2; a.ncls = ncls; a.part = (float*)((char*)ws + wb);
    MYOLO_REQUIRE(((uintptr_t)x & 15) == 0, "deconv2x2s2_mask_fwd: x must be 16-byte aligned");
    const long long tiles = cdiv64(a.M, BM) * (a.N / BN);
    hipLaunchKernelGGL((gemm_nn_fast<AM_PLAIN, EP_DECONV_MASK>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    myolo_launch_deconv_mask_finish(a.part, b2, p_out, 4 * a.M, ncls, (Cout / BN) * 2, s);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_deconv2x2s2_bwd_data(const float* dy, const float* w, float* dx,
                               int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(dy && w && dx && N > 0, "deconv2x2s2_bwd_data: bad arguments");
    MYOLO_REQUIRE(Cout % BK == 0, "deconv2x2s2_bwd_data: Cout must be a multiple of %d", BK);
    if (myolo_deconv_x6_ok(Cin, Cout, 1) && (long long)N * H * W >= 4096 && ws && ws_bytes >= (size_t)4 * Cin * Cout * 6 &&
        (((uintptr_t)dy | (uintptr_t)w | (uintptr_t)dx) & 15) == 0) {
        myolo_deconv_x6_bwd_data(dy, w, dx, (long long)N * H * W, H, W, Cin, Cout, ws, (hipStream_t)stream);
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    GemmArgs a = {};
    a.A = dy; a.B = w; a.C = dx; a.M = (long long)N * H * W; a.N = Cin; a.K = 4 * Cout;
    a.ldb = Cin; a.ldc = Cin; a.H = H; a.W = W; a.Cc = Cout;
    launch_nn<AM_DECONV, EP_PLAIN>(a, (hipStream_t)stream, ws, ws_bytes);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_deconv2x2s2_bwd_weight(const float* x, const float* dy, float* dw,
                                 int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && dy && dw && N > 0, "deconv2x2s2_bwd_weight: bad arguments");
    MYOLO_REQUIRE((Cout & 3) == 0, "deconv2x2s2_bwd_weight: Cout must be a multiple of 4");
    if (g_myolo_opt.wino_x6 && !g_myolo_opt.deconv_no_x6 && (Cin % 256) == 0 && (Cout % 64) == 0 && ((4 * Cout) % 256) == 0 && (long long)N * H * W >= 4096 && ws &&
        myolo_deconv_x6_bwd_weight_ws_bytes((long long)N * H * W, Cin, Cout) <= ws_bytes && (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw) & 15) == 0) {
        // FP32_MATMUL = "bf16x6": wino_tn_x6_kernel with the four taps of dy gathered into the A operand (csrc/wino_mm.hip)
        const int rc = myolo_deconv_x6_bwd_weight(x, dy, dw, (long long)N * H * W, H, W, Cin, Cout, ws, ws_bytes, (hipStream_t)stream);
        if (rc) return rc;
        MYOLO_CHECK_LAUNCH();
        return MYOLO_OK;
    }
    GemmArgs a = {};
    a.A = dy; a.B = x; a.M = (long long)N * H * W; a.N = Cin; a.K = 4 * Cout; a.ldb = Cin;
    a.H = H; a.W = W; a.Cc = Cout;
    int rc = launch_tn<AM_DECONV>(a, dw, ws, ws_bytes, (hipStream_t)stream, "deconv2x2s2_bwd_weight");
    if (rc) return rc;
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

}  // extern "C"
