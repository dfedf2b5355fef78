// Winograd with F(6x6,3x3) tiles for the 14x14 maps of the mask head (myolo_mask_conv2-4, model.py:693-709), fp32.
//
// 14 = 6 + 4 + 4: per direction one F(6,3) tile (8 interpolation points {0, 1, -1, 2, -2, 1/2, -1/2, inf}) and two F(4,3) tiles,
// whose 6 points {0, 1, -1, 2, -2, inf} are a SUBSET S6 of those 8.  The Cook-Toom filter transform of F(4,3) differs from the
// F(6,3) one on S6 only by a per-point factor rho = (x^2 - 1/4) = (-1/4, 3/4, 3/4, 15/4, 15/4, 1), which is folded into the data
// transform of the F(4,3) tiles -- so ONE set of 64 transformed filters (G8 g G8^T) serves all nine tiles of an image, an F(4,3)
// direction simply has no row in the planes of the two points it does not use (the construction of wino_kernels.hip's
// F(2,3)-in-F(4,3) tiling, one level up).  An image costs (8 + 6 + 6)^2 = 400 point-tiles instead of 484 (F(4,3)/F(2,3) tiling) or
// 576 (uniform F(4,3)): 17.4 % fewer multiplications and 17.4 % smaller V / M planes than wino_kernels.hip.  Exact in exact
// arithmetic (tools/wino63_numerics.py: 7e-14 in fp64); in fp32 its error against a float64 convolution is 1.2-1.5x the F(4,3)
// tiling's (max 9e-6, rms 6e-7 of the output maximum at K = 256; only one tile in nine is F(6,3) in both directions).
//
// Planes: 64 planes of [rows][C], ordered by who uses them (i = vertical point, j = horizontal point, C2 = {1/2, -1/2}):
//   g0  i, j in S6      36 planes, 9 rows per image (every tile)                row = img*9 + ty*3 + tx
//   g1  i in S6, j in C2 12 planes, 3 rows per image (tiles of tile column 0)   row = img*3 + ty
//   g2  i in C2, j in S6 12 planes, 3 rows per image (tiles of tile row 0)      row = img*3 + tx
//   g3  i, j in C2        4 planes, 1 row  per image (tile (0,0))               row = img
// -> three runs of equal-height planes (36, 24, 4) for the one-launch batched GEMM of csrc/wino_mm.hip.
//
// The layer boundary (output transform + bias + folded-BN affine + ReLU -> 14x14x64 activation tile in LDS -> input transform) is
// one kernel: workgroup = (image, 64-channel slice), nine waves = the nine tiles (so a wave's tile class is uniform and every
// transform is straight-line code), lane = channel: every global access is 256 contiguous bytes.
#include "myolo_common.h"

#define W63_HW 14
#define W63_CS 64            // channels per workgroup
#define W63_TILES 9

__host__ __device__ constexpr int w63_s6(int p) { return p == 7 ? 5 : p; }        // index of point p inside S6 (p != 5, 6)
// plane of transform point (i, j), 0 <= i, j < 8 in the order {0, 1, -1, 2, -2, 1/2, -1/2, inf}
__host__ __device__ constexpr int w63_q(int i, int j)
{
    const bool ci = (i == 5 || i == 6), cj = (j == 5 || j == 6);
    return !ci ? (!cj ? w63_s6(i) * 6 + w63_s6(j) : 36 + w63_s6(i) * 2 + (j - 5)) : (!cj ? 48 + (i - 5) * 6 + w63_s6(j) : 60 + (i - 5) * 2 + (j - 5));
}
__host__ __device__ constexpr int w63_grp(int i, int j) { return ((i == 5 || i == 6) ? 2 : 0) + ((j == 5 || j == 6) ? 1 : 0); }
__host__ __device__ constexpr int w63_qfirst(int g) { return g == 0 ? 0 : g == 1 ? 36 : g == 2 ? 48 : 60; }

// ---- one-dimensional transforms.  CLS = 6: F(6,3) (8 points, patch of 8, 6 outputs); CLS = 4: F(4,3) on S6 with rho folded in ----
// T = float, or f2 = two independent columns (rows) at once: the same expression trees element by element (v_pk_* on gfx950)
typedef float f2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T w63_zero() { return (T)(0.f); }
// one rounding: a * b + c (v_fma_f32 / v_pk_fma_f32)
__device__ __forceinline__ float w63_fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ f2 w63_fma(float a, f2 b, f2 c) { const f2 av = {a, a}; return __builtin_elementwise_fma(av, b, c); }
// B^T (8x8 / 6x6 with rho): the only transform with multipliers that are not powers of two (4.25, 1.25, 2.5, 5, 5.25, 0.75, 3.75).  Where an expression
// sums TWO products the compiler may fuse either one, and which one it picks depends on the code around the call (seen: the scalar and the packed
// instantiation of `0.25 d2 - 1.25 d4` differed) -- so contraction is off here and every fused multiply-add is written out: the inexact product is the
// fused one, the power-of-two product (exact) is formed first.  Legacy and packed kernels then agree bit for bit by construction.
template <int CLS, typename T = float>
__device__ __forceinline__ void w63_bt(const T d[8], T t[8])
{
#pragma clang fp contract(off)
    if (CLS == 6) {
        const T e0 = w63_fma(-4.25f, d[4], d[2] + d[6]), o0 = w63_fma(-4.25f, d[3], d[1] + d[5]);
        const T e1 = w63_fma(-1.25f, d[4], 0.25f * d[2]) + d[6], o1 = w63_fma(2.f, d[5], w63_fma(-2.5f, d[3], 0.5f * d[1]));
        const T e2 = w63_fma(-5.f, d[4], 4.f * d[2]) + d[6], o2 = w63_fma(0.5f, d[5], w63_fma(-2.5f, d[3], 2.f * d[1]));
        t[0] = w63_fma(5.25f, d[2] - d[4], d[6] - d[0]);
        t[1] = e0 + o0;
        t[2] = e0 - o0;
        t[3] = e1 + o1;
        t[4] = e1 - o1;
        t[5] = e2 + o2;
        t[6] = e2 - o2;
        t[7] = w63_fma(5.25f, d[3] - d[5], d[7] - d[1]);
    } else {            // rho .* (B6^T d), patch d[0..5]
        t[0] = w63_fma(-0.25f, d[4], w63_fma(1.25f, d[2], -d[0]));
        t[1] = 0.75f * w63_fma(-4.f, d[1] + d[2], d[3] + d[4]);                 // (products by powers of two are exact: fused or not, the same bits)
        t[2] = 0.75f * (w63_fma(4.f, d[1] - d[2], -d[3]) + d[4]);
        t[3] = 3.75f * (w63_fma(2.f, d[3] - d[1], -d[2]) + d[4]);
        t[4] = 3.75f * (w63_fma(2.f, d[1] - d[3], -d[2]) + d[4]);
        t[5] = w63_zero<T>();
        t[6] = w63_zero<T>();
        t[7] = w63_fma(-5.f, d[3], 4.f * d[1]) + d[5];
    }
}
template <int CLS, typename T = float>
__device__ __forceinline__ void w63_at(const T m[8], T y[6])
{
    const T s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    if (CLS == 6) {
        const T s56 = m[5] + m[6], d56 = m[5] - m[6];
        y[0] = m[0] + s12 + s34 + s56;
        y[1] = d12 + 2.f * d34 + 0.5f * d56;
        y[2] = s12 + 4.f * s34 + 0.25f * s56;
        y[3] = d12 + 8.f * d34 + 0.125f * d56;
        y[4] = s12 + 16.f * s34 + 0.0625f * s56;
        y[5] = d12 + 32.f * d34 + 0.03125f * d56 + m[7];
    } else {
        y[0] = m[0] + s12 + s34;
        y[1] = d12 + 2.f * d34;
        y[2] = s12 + 4.f * s34;
        y[3] = d12 + 8.f * d34 + m[7];
        y[4] = w63_zero<T>();
        y[5] = w63_zero<T>();
    }
}
// G8 (8x3) on a 3-vector
__device__ __forceinline__ void w63_g(const float g[3], float u[8])
{
    u[0] = -g[0];
    u[1] = -(2.f / 9.f) * (g[0] + g[1] + g[2]);
    u[2] = -(2.f / 9.f) * (g[0] - g[1] + g[2]);
    u[3] = g[0] * (1.f / 90.f) + g[1] * (1.f / 45.f) + g[2] * (2.f / 45.f);
    u[4] = g[0] * (1.f / 90.f) - g[1] * (1.f / 45.f) + g[2] * (2.f / 45.f);
    u[5] = g[0] * (32.f / 45.f) + g[1] * (16.f / 45.f) + g[2] * (8.f / 45.f);
    u[6] = g[0] * (32.f / 45.f) - g[1] * (16.f / 45.f) + g[2] * (8.f / 45.f);
    u[7] = g[2];
}

// A (8x6 / 6x4): the adjoint of w63_at -- Q = A dY A^T of the weight gradient
template <int CLS, typename T = float>
__device__ __forceinline__ void w63_a(const T d[6], T q[8])
{
    if (CLS == 6) {
        const T e = d[0] + d[2] + d[4], o = d[1] + d[3] + d[5];
        const T e2 = d[0] + 4.f * d[2] + 16.f * d[4], o2 = 2.f * d[1] + 8.f * d[3] + 32.f * d[5];
        const T eh = d[0] + 0.25f * d[2] + 0.0625f * d[4], oh = 0.5f * d[1] + 0.125f * d[3] + 0.03125f * d[5];
        q[0] = d[0];
        q[1] = e + o;  q[2] = e - o;
        q[3] = e2 + o2; q[4] = e2 - o2;
        q[5] = eh + oh; q[6] = eh - oh;
        q[7] = d[5];
    } else {
        const T e = d[0] + d[2], o = d[1] + d[3], e2 = d[0] + 4.f * d[2], o2 = 2.f * d[1] + 8.f * d[3];
        q[0] = d[0];
        q[1] = e + o;  q[2] = e - o;
        q[3] = e2 + o2; q[4] = e2 - o2;
        q[5] = w63_zero<T>(); q[6] = w63_zero<T>();
        q[7] = d[3];
    }
}
// G8^T (3x8) on an 8-vector
__device__ __forceinline__ void w63_gt(const float d[8], float w[3])
{
    w[0] = -d[0] - (2.f / 9.f) * (d[1] + d[2]) + (1.f / 90.f) * (d[3] + d[4]) + (32.f / 45.f) * (d[5] + d[6]);
    w[1] = (2.f / 9.f) * (d[2] - d[1]) + (1.f / 45.f) * (d[3] - d[4]) + (16.f / 45.f) * (d[5] - d[6]);
    w[2] = -(2.f / 9.f) * (d[1] + d[2]) + (2.f / 45.f) * (d[3] + d[4]) + (8.f / 45.f) * (d[5] + d[6]) + d[7];
}

struct W63Planes {
    long long base[4];       // element offset of (this tile's row, this lane's channel) in the first plane of each group
    long long stride[4];     // plane stride of the group (elements)
};
__device__ __forceinline__ W63Planes w63_planes(long long NR, long long img, int ty, int tx, int C, int c)
{
    W63Planes p;
    const long long r0 = 9 * NR, r1 = 3 * NR;
    p.stride[0] = r0 * C; p.stride[1] = r1 * C; p.stride[2] = r1 * C; p.stride[3] = NR * C;
    p.base[0] = (img * 9 + ty * 3 + tx) * C + c;
    p.base[1] = 36 * r0 * C + (img * 3 + ty) * C + c;
    p.base[2] = (36 * r0 + 12 * r1) * C + (img * 3 + tx) * C + c;
    p.base[3] = (36 * r0 + 24 * r1) * C + img * C + c;
    return p;
}
#define W63_ADDR(pl, i, j) ((pl).base[w63_grp(i, j)] + (long long)(w63_q(i, j) - w63_qfirst(w63_grp(i, j))) * (pl).stride[w63_grp(i, j)])
template <int CLS>
__device__ __forceinline__ constexpr bool w63_used(int p) { return CLS == 6 || (p != 5 && p != 6); }

struct W63Args {
    const float* src;        // FROM_M: M planes;  FROM_ACT: activation [NR,14,14,C]
    float* Vn;               // TO_V: the next conv's V planes
    float* y;                // activation out [NR,14,14,C] or NULL
    const int32_t* flags;    // y is written where flags[img] != 0 (NULL: everywhere)
    const float* bias;       // FROM_M: added before the affine (NULL: 0)
    const float* scale;      // per-channel affine (NULL: identity)
    const float* shift;
    long long NR;
    int C, act;
    // FROM_LAZY (data gradient behind a training-mode BatchNorm with a row-sparse upstream gradient, see wino_kernels.hip LazyBn):
    // src = the BN's input x, the operand is  scale * (dyc * [act passes](scale*x + shift)) + (ka + kb*x),  dyc compact per image
    const float* dyc;        // [n_pos images][14*14][C]
    const int32_t* inv;      // [NR] compact slot of an image or -1
    const float* ka;
    const float* kb;
    // FROM_CROP (ROIAlign fused into the input transform, tf.image.crop_and_resize model.py:385-387): src = feature map [B,FH,FW,C]
    const float* boxes;      // [NR][4] y1, x1, y2, x2 (normalised)
    const int32_t* bind;     // [NR] image of each box
    int FH, FW;
    // FROM_M with BatchNorm statistics: per-image partial sums [NR][2*C] (sum | sum of squares) of the values written to y
    double* stats;
    float* Qn;               // TO_VQ: the adjoint-output-transformed planes (TO_Q alone writes them to Vn)
    int order;               // workgroup order, see the kernel (set by w63_launch from option "w63_order")
    // FROM_M: the value BEFORE the affine + activation (A^T m A + bias = the conv's pre-BatchNorm output) written where flags[img] != 0 (NULL:
    // everywhere); with ypre set, `flags` governs ypre and y (if any) is written everywhere.  The exact-sparsity backward reads the frozen BatchNorms'
    // backward off THIS tensor for the positive ROIs: no (a - beta) / gamma reconstruction from the post-activation value
    float* ypre;
    // keep_cap > 0: `flags` holds SLOTS (myolo_positive_index: slot >= 0 or -1) and the flagged tensor (ypre, or y when there is no ypre) is written in
    // COMPACT order -- image img at row block flags[img] if 0 <= flags[img] < keep_cap -- so that the sparse backward needs no gather
    int keep_cap;
};

// per-image control words (flags, slots, box indices, boxes) are written by EARLIER kernels and never by these: read through the constant address
// space so that a wave-uniform index becomes an s_load -- a vector load here would put an s_waitcnt vmcnt(0) in front of every unit of the persistent
// kernel, i.e. wait for all of the prefetched rows and all of the previous unit's stores
__device__ __forceinline__ int w63_cload(const int32_t* p, long long i) { return ((const __attribute__((address_space(4))) int32_t*)p)[i]; }
__device__ __forceinline__ float w63_cload(const float* p, long long i) { return ((const __attribute__((address_space(4))) float*)p)[i]; }
// where the flagged output of image img goes: its own row block, the compact one of its slot, or -1 (not kept)
__device__ __forceinline__ long long w63_keep_dst(const W63Args& a, long long img)
{
    if (!a.flags) return img;
    const int f = w63_cload(a.flags, img);
    if (a.keep_cap > 0) return (f >= 0 && f < a.keep_cap) ? (long long)f : -1;
    return f != 0 ? img : -1;
}

__device__ __forceinline__ float w63_act(float v, int act)
{
    if (act == MYOLO_ACT_RELU) return fmaxf(v, 0.f);
    if (act == MYOLO_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}

// output transform of one tile (class CY x CX) from its M values + bias/affine/activation -> LDS tile (and y)
template <int CY, int CX, bool TO_LDS>
__device__ __forceinline__ void w63_front_m(const W63Args& a, const W63Planes& pl, float* act_lds, int oy, int ox, int lane, long long img, int c,
                                            bool wr, float& s1, float& s2)
{
    constexpr int MY = CY, MX = CX;                  // outputs per direction
    float tmp[6][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (!w63_used<CX>(j)) continue;
        float m[8], r[6];
#pragma unroll
        for (int i = 0; i < 8; ++i) m[i] = w63_used<CY>(i) ? a.src[W63_ADDR(pl, i, j)] : 0.f;
        w63_at<CY>(m, r);
#pragma unroll
        for (int i = 0; i < MY; ++i) tmp[i][j] = r[i];
    }
    const float b = a.bias ? a.bias[c] : 0.f;
    const float sc = a.scale ? a.scale[c] : 1.f, sh = a.scale ? a.shift[c] : 0.f;
    const long long yd = a.ypre ? img : w63_keep_dst(a, img);          // (wr: yd >= 0)
    float* ybase = wr ? a.y + (yd * W63_HW * W63_HW) * a.C + c : nullptr;
    const long long pd = a.ypre ? w63_keep_dst(a, img) : -1;
    const bool wrp = pd >= 0;
    float* pbase = wrp ? a.ypre + (pd * W63_HW * W63_HW) * a.C + c : nullptr;
#pragma unroll
    for (int i = 0; i < MY; ++i) {
        float r[6];
        w63_at<CX>(tmp[i], r);
#pragma unroll
        for (int j = 0; j < MX; ++j) {
            const float pre = r[j] + b;
            const float v = w63_act(fmaf(pre, sc, sh), a.act);
            const int pix = (oy + i) * W63_HW + ox + j;
            if (TO_LDS) act_lds[pix * W63_CS + lane] = v;
            if (wr) ybase[(long long)pix * a.C] = v;
            if (wrp) pbase[(long long)pix * a.C] = pre;
            s1 += v;
            s2 = fmaf(v, v, s2);
        }
    }
}

// input transform of one tile (class CY x CX) from the LDS activation tile -> V planes
template <int CY, int CX>
__device__ __forceinline__ void w63_back_v(const W63Args& a, const W63Planes& pl, const float* act_lds, int py0, int px0, int lane)
{
    constexpr int NY = CY + 2, NX = CX + 2;          // patch size per direction
    float tmp[8][8];
#pragma unroll
    for (int j = 0; j < NX; ++j) {
        float d[8], r[8];
        const int xx = px0 + j;
        const bool xin = (unsigned)xx < (unsigned)W63_HW;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int yy = py0 + i;
            d[i] = (i < NY && xin && (unsigned)yy < (unsigned)W63_HW) ? act_lds[(yy * W63_HW + xx) * W63_CS + lane] : 0.f;
        }
        w63_bt<CY>(d, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) tmp[i][j] = r[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (!w63_used<CY>(i)) continue;
        float d[8], r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = j < NX ? tmp[i][j] : 0.f;
        w63_bt<CX>(d, r);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (w63_used<CX>(j)) a.Vn[W63_ADDR(pl, i, j)] = r[j];
    }
}

// Q = A dY A^T of one tile (class CY x CX) from the LDS tile (holding dY) -> Q planes
template <int CY, int CX>
__device__ __forceinline__ void w63_back_q(float* __restrict__ dst, const W63Planes& pl, const float* act_lds, int oy, int ox, int lane)
{
    constexpr int MY = CY, MX = CX;
    float tmp[8][6];
#pragma unroll
    for (int j = 0; j < MX; ++j) {
        float d[6], r[8];
#pragma unroll
        for (int i = 0; i < 6; ++i) d[i] = i < MY ? act_lds[((oy + i) * W63_HW + ox + j) * W63_CS + lane] : 0.f;
        w63_a<CY>(d, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) tmp[i][j] = r[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if (!w63_used<CY>(i)) continue;
        float d[6], r[8];
#pragma unroll
        for (int j = 0; j < 6; ++j) d[j] = j < MX ? tmp[i][j] : 0.f;
        w63_a<CX>(d, r);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (w63_used<CX>(j)) dst[W63_ADDR(pl, i, j)] = r[j];
    }
}

struct W63CropAxis { int lo, hi; float w; bool ok; };
// same float expressions as crop_fwd_kernel / wino_in_crop_kernel (csrc/wino_kernels.hip)
__device__ __forceinline__ W63CropAxis w63_crop_axis(float b0, float b1, int size, int crop, int idx)
{
    W63CropAxis a;
    a.ok = true;
    float in;
    if (crop > 1) {
        const float scale = (b1 - b0) * (float)(size - 1) / (float)(crop - 1);
        in = b0 * (float)(size - 1) + (float)idx * scale;
    } else {
        in = 0.5f * (b0 + b1) * (float)(size - 1);
    }
    if (in < 0.f || in > (float)(size - 1)) a.ok = false;      // extrapolation value 0
    a.lo = (int)floorf(in); a.hi = (int)ceilf(in); a.w = in - (float)a.lo;
    if (!a.ok) { a.lo = 0; a.hi = 0; a.w = 0.f; }
    return a;
}

enum { W63_FROM_M = 0, W63_FROM_ACT = 1, W63_FROM_LAZY = 2, W63_FROM_CROP = 3 };
enum { W63_TO_V = 0, W63_TO_NONE = 1, W63_TO_Q = 2, W63_TO_VQ = 3 };      // TO_VQ: both transforms of ONE tile fill (V -> a.Vn, Q -> a.Qn)

// one (image, 64-channel slice) unit of the layer boundary
template <int FRONT, int BACK>
__device__ __forceinline__ void w63_unit_body(const W63Args& a, float* act_lds, long long unit)
{
    const int ns = a.C / W63_CS;
    const long long img = a.order ? unit / ns : unit % a.NR;
    const int slice = (int)(a.order ? unit - img * ns : unit / a.NR);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = slice * W63_CS + lane;
    const int ty = wave / 3, tx = wave - ty * 3;
    const W63Planes pl = w63_planes(a.NR, img, ty, tx, a.C, c);
    const long long ydst = a.ypre ? img : w63_keep_dst(a, img);
    const bool wr = a.y && ydst >= 0;
    if (FRONT == W63_FROM_M) {
        const int oy = ty == 0 ? 0 : 2 + 4 * ty, ox = tx == 0 ? 0 : 2 + 4 * tx;           // output origin: 0, 6, 10
        float s1 = 0.f, s2 = 0.f;
        constexpr bool L = BACK != W63_TO_NONE;      // without a back half nobody reads the LDS tile: it is neither filled nor allocated
        if (ty == 0) { if (tx == 0) w63_front_m<6, 6, L>(a, pl, act_lds, oy, ox, lane, img, c, wr, s1, s2); else w63_front_m<6, 4, L>(a, pl, act_lds, oy, ox, lane, img, c, wr, s1, s2); }
        else         { if (tx == 0) w63_front_m<4, 6, L>(a, pl, act_lds, oy, ox, lane, img, c, wr, s1, s2); else w63_front_m<4, 4, L>(a, pl, act_lds, oy, ox, lane, img, c, wr, s1, s2); }
        if (BACK == W63_TO_NONE && a.stats) {   // BatchNorm statistics of this image's 196 x 64 values: the nine tiles' partial sums, added in tile order
            float* red = act_lds;               // [2][W63_TILES][W63_CS] (w63_lds_bytes)
            red[wave * W63_CS + lane] = s1; red[(W63_TILES + wave) * W63_CS + lane] = s2;
            __syncthreads();
            if (wave == 0) {
                double t1 = 0, t2 = 0;
#pragma unroll
                for (int k = 0; k < W63_TILES; ++k) { t1 += (double)red[k * W63_CS + lane]; t2 += (double)red[(W63_TILES + k) * W63_CS + lane]; }
                double* dst = a.stats + img * 2 * a.C;
                dst[c] = t1;
                dst[a.C + c] = t2;
            }
        }
    } else if (FRONT == W63_FROM_CROP) {
        // lane = (pixel lane >> 4 of a group of four, channel quad lane & 15): every corner read is 16 bytes per lane, 4 x 256 B per wave
        const float* bxp = a.boxes + img * 4;
        const float by1 = bxp[0], bx1 = bxp[1], by2 = bxp[2], bx2 = bxp[3];
        const int cq = (lane & 15) * 4;
        const float* fb = a.src + (long long)a.bind[img] * a.FH * a.FW * a.C + slice * W63_CS + cq;
        constexpr int NP = 3;                                                          // pixels (x 4 corners) in flight per lane
        constexpr int STEP = 4 * W63_TILES;                                            // pixels per pass of the workgroup
        for (int p0 = wave * 4 + (lane >> 4); p0 < W63_HW * W63_HW; p0 += NP * STEP) {
            float4 tl[NP], tr[NP], bl[NP], br[NP];
            float wx[NP], wy[NP];
            bool ok[NP];
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int pix = min(p0 + k * STEP, W63_HW * W63_HW - 1);
                const int py = pix / W63_HW, px = pix - py * W63_HW;
                const W63CropAxis ay = w63_crop_axis(by1, by2, a.FH, W63_HW, py), ax = w63_crop_axis(bx1, bx2, a.FW, W63_HW, px);
                ok[k] = ay.ok && ax.ok; wx[k] = ax.w; wy[k] = ay.w;
                tl[k] = *reinterpret_cast<const float4*>(fb + ((long long)ay.lo * a.FW + ax.lo) * a.C);
                tr[k] = *reinterpret_cast<const float4*>(fb + ((long long)ay.lo * a.FW + ax.hi) * a.C);
                bl[k] = *reinterpret_cast<const float4*>(fb + ((long long)ay.hi * a.FW + ax.lo) * a.C);
                br[k] = *reinterpret_cast<const float4*>(fb + ((long long)ay.hi * a.FW + ax.hi) * a.C);
            }
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int pix = p0 + k * STEP;
                if (pix >= W63_HW * W63_HW) continue;
                float4 o;
                float top, bot;
                top = tl[k].x + (tr[k].x - tl[k].x) * wx[k]; bot = bl[k].x + (br[k].x - bl[k].x) * wx[k]; o.x = top + (bot - top) * wy[k];
                top = tl[k].y + (tr[k].y - tl[k].y) * wx[k]; bot = bl[k].y + (br[k].y - bl[k].y) * wx[k]; o.y = top + (bot - top) * wy[k];
                top = tl[k].z + (tr[k].z - tl[k].z) * wx[k]; bot = bl[k].z + (br[k].z - bl[k].z) * wx[k]; o.z = top + (bot - top) * wy[k];
                top = tl[k].w + (tr[k].w - tl[k].w) * wx[k]; bot = bl[k].w + (br[k].w - bl[k].w) * wx[k]; o.w = top + (bot - top) * wy[k];
                if (!ok[k]) o = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(&act_lds[pix * W63_CS + cq]) = o;
            }
        }
    } else {
        const float sc = a.scale ? a.scale[c] : 1.f, sh = a.scale ? a.shift[c] : 0.f;
        const float* xb = a.src + (img * W63_HW * W63_HW) * a.C + c;
        float* yb = wr ? a.y + (ydst * W63_HW * W63_HW) * a.C + c : nullptr;
        float ka = 0.f, kb = 0.f;
        const float* gb = nullptr;
        if (FRONT == W63_FROM_LAZY) {
            ka = a.ka[c]; kb = a.kb[c];
            const int slot = a.inv[img];
            if (slot >= 0) gb = a.dyc + ((long long)slot * W63_HW * W63_HW) * a.C + c;
        }
        for (int p0 = wave; p0 < W63_HW * W63_HW; p0 += 4 * W63_TILES) {             // 4 loads in flight per lane
            float v[4], gq[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pix = p0 + k * W63_TILES;
                v[k] = pix < W63_HW * W63_HW ? xb[(long long)pix * a.C] : 0.f;
                gq[k] = (FRONT == W63_FROM_LAZY && gb && pix < W63_HW * W63_HW) ? gb[(long long)pix * a.C] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int pix = p0 + k * W63_TILES;
                if (pix >= W63_HW * W63_HW) continue;
                float t;
                if (FRONT == W63_FROM_LAZY) {
                    const float z = fmaf(v[k], sc, sh);
                    const float pass = a.act == MYOLO_ACT_RELU ? (z > 0.f ? 1.f : 0.f) : a.act == MYOLO_ACT_RELU6 ? ((z > 0.f && z < 6.f) ? 1.f : 0.f) : 1.f;
                    t = fmaf(sc, gq[k] * pass, fmaf(kb, v[k], ka));
                } else {
                    t = w63_act(fmaf(v[k], sc, sh), a.act);
                }
                act_lds[pix * W63_CS + lane] = t;
                if (wr) yb[(long long)pix * a.C] = t;
            }
        }
    }
    if (BACK == W63_TO_NONE) return;
    __syncthreads();
    if (BACK == W63_TO_Q || BACK == W63_TO_VQ) {
        float* qd = BACK == W63_TO_Q ? a.Vn : a.Qn;
        const int oy = ty == 0 ? 0 : 2 + 4 * ty, ox = tx == 0 ? 0 : 2 + 4 * tx;
        if (ty == 0) { if (tx == 0) w63_back_q<6, 6>(qd, pl, act_lds, oy, ox, lane); else w63_back_q<6, 4>(qd, pl, act_lds, oy, ox, lane); }
        else         { if (tx == 0) w63_back_q<4, 6>(qd, pl, act_lds, oy, ox, lane); else w63_back_q<4, 4>(qd, pl, act_lds, oy, ox, lane); }
        if (BACK == W63_TO_Q) return;
    }
    const int py0 = ty == 0 ? -1 : 1 + 4 * ty, px0 = tx == 0 ? -1 : 1 + 4 * tx;           // patch origin: -1, 5, 9
    if (ty == 0) { if (tx == 0) w63_back_v<6, 6>(a, pl, act_lds, py0, px0, lane); else w63_back_v<6, 4>(a, pl, act_lds, py0, px0, lane); }
    else         { if (tx == 0) w63_back_v<4, 6>(a, pl, act_lds, py0, px0, lane); else w63_back_v<4, 4>(a, pl, act_lds, py0, px0, lane); }
}
// ---- round 3-5 form of the kernel (scalar transforms, one address computation per access): kept behind option "w63_legacy" as the
// reference of tests/test_gpu_ops.py::test_wino63_boundary_packed_equals_legacy -- the packed form below gives the same bits ----
template <int FRONT, int BACK>
__global__ __launch_bounds__(W63_TILES * 64) void wino63_boundary_legacy_kernel(W63Args a)
{
    extern __shared__ __attribute__((aligned(16))) float act_lds[];         // [14][14][64]
    // workgroup -> (image, 64-channel slice).  a.order = 1: the slices of one image are consecutive workgroups
    w63_unit_body<FRONT, BACK>(a, act_lds, blockIdx.x);
}

// =====================================================================================================================================
// The kernel the step runs (round 6): PERSISTENT workgroups, one per CU, that walk the units (image, 64-channel slice) and have the NEXT
// unit's planes on their way while the current unit is transformed.
//
// What the round-5 kernel did wrong, measured (tools/experiments/census, tools/experiments/w63/trace_*.py; profiles/r6_notes.md):
//   * a CU of this part admits workgroups as if every SIMD had to take ceil(waves / 4) of their waves: the nine-wave workgroups ran ONE per
//     CU at 81..96 VGPRs (the occupancy calculator says two), so each CU alternated between "nine waves wait for their 400 loads" and "nine
//     waves transform" -- the memory pipe of a CU idle ~40 % of the time; with the transforms removed the same access pattern moves 6 TB/s;
//   * eight-wave workgroups do run two per CU -- in lockstep (both start when both slots free up, both load, both transform): no gain.
// So the overlap is made explicit: after a wave's first pass has consumed its M values, it requests the same rows of the workgroup's next
// unit into the same registers; they arrive during the second pass, the barrier, the input transform and the stores.  The LDS tile is
// double-buffered (one barrier per unit).  Also new: every 1-D transform runs on TWO columns (first pass) or TWO rows (second pass) at once
// as f2 values (v_pk_add_f32 / v_pk_fma_f32, half the arithmetic instructions), plane addresses are wave-uniform (SGPR base + lane offset),
// per-channel constants are loaded once per workgroup (its slice never changes: the grid is a multiple of C / 64), the activation fronts
// request all of a wave's 22 pixels at once.  Same expression trees per element as the legacy form.
#ifdef W63_TRACE
__device__ unsigned long long w63_trace_buf[8192 * 9 * 8];
#define W63_T(k) do { asm volatile("" ::: "memory"); if ((threadIdx.x & 63) == 0 && blockIdx.x < 256 && w63_it >= 8 && w63_it < 12) w63_trace_buf[((blockIdx.x * 4 + (w63_it - 8)) * 9 + (threadIdx.x >> 6)) * 8 + (k)] = clock64(); asm volatile("" ::: "memory"); } while (0)
#else
#define W63_T(k)
#endif
// Plane addressing of the persistent kernel: ONE buffer resource per plane set (32-bit byte offsets: the set must stay below 4 GiB, w63_launch
// falls back to the legacy kernel otherwise), wave-uniform offset of (plane, this tile's row, this slice's first channel) in an SGPR, the lane's 4 bytes
// in the VGPR offset: no per-access address arithmetic on the vector ALU and one scalar register per access instead of two.
struct W63U {
    unsigned base[4];        // byte offset of (this tile's row, this slice's first channel) in the first plane of each group
    unsigned sb;             // NR * C * 4: plane strides are 9, 3, 3, 1 times this
};
__device__ __forceinline__ W63U w63p_planes(unsigned NR, unsigned img, int ty, int tx, unsigned C, unsigned c0)
{
    W63U p;
    p.sb = NR * C * 4u;
    p.base[0] = ((img * 9 + ty * 3 + tx) * C + c0) * 4u;
    p.base[1] = 36u * 9u * p.sb + ((img * 3 + ty) * C + c0) * 4u;
    p.base[2] = (36u * 9u + 12u * 3u) * p.sb + ((img * 3 + tx) * C + c0) * 4u;
    p.base[3] = (36u * 9u + 24u * 3u) * p.sb + (img * C + c0) * 4u;
    return p;
}
__host__ __device__ constexpr unsigned w63_gmult(int g) { return g == 0 ? 9u : g == 3 ? 1u : 3u; }
#define W63P_OFF(pl, i, j) ((pl).base[w63_grp(i, j)] + (unsigned)(w63_q(i, j) - w63_qfirst(w63_grp(i, j))) * w63_gmult(w63_grp(i, j)) * (pl).sb)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t w63_rsrc(const void* base, unsigned nbytes)
{
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)nbytes, 0x00020000);
}
#ifndef W63_LD_AUX
#define W63_LD_AUX 0
#endif
#ifndef W63_ST_AUX
#define W63_ST_AUX 0
#endif
__device__ __forceinline__ float w63_ld(__amdgpu_buffer_rsrc_t r, int voff, unsigned soff)
{
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, (int)soff, W63_LD_AUX));
}
__device__ __forceinline__ void w63_st(float v, __amdgpu_buffer_rsrc_t r, int voff, unsigned soff)
{
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, (int)soff, W63_ST_AUX);
}
// the k-th used point of a direction of class CLS (CLS 6: 0..7; CLS 4: 0, 1, 2, 3, 4, 7) and the number of used points
template <int CLS> __device__ __forceinline__ constexpr int w63_pt(int k) { return CLS == 6 ? k : (k == 5 ? 7 : k); }
template <int CLS> __device__ __forceinline__ constexpr int w63_npt() { return CLS == 6 ? 8 : 6; }
// slot of point p in an array indexed [pair][half] over the used points
template <int CLS> __device__ __forceinline__ constexpr int w63_slot(int p) { return CLS == 6 ? p : (p == 7 ? 5 : p); }

template <int ACT>
__device__ __forceinline__ float w63p_act(float v)
{
    if (ACT == MYOLO_ACT_RELU) return fmaxf(v, 0.f);
    if (ACT == MYOLO_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}


#define W63_NPW ((W63_HW * W63_HW + W63_TILES - 1) / W63_TILES)      // pixels per wave of an activation front: 22

// request the M values of one tile: pf[(pair of used columns * 8 + row) * 2 + half]
template <int CY, int CX>
__device__ __forceinline__ void w63q_issue_m(float (&pf)[64], __amdgpu_buffer_rsrc_t src, const W63U& pl, int lane4)
{
#pragma unroll
    for (int jp = 0; jp < w63_npt<CX>() / 2; ++jp) {
        const int j0 = w63_pt<CX>(2 * jp), j1 = w63_pt<CX>(2 * jp + 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (!w63_used<CY>(i)) continue;
            pf[(jp * 8 + i) * 2 + 0] = w63_ld(src, lane4, W63P_OFF(pl, i, j0));
            pf[(jp * 8 + i) * 2 + 1] = w63_ld(src, lane4, W63P_OFF(pl, i, j1));
        }
    }
}
// first pass of the output transform: A^T m, two columns at a time
template <int CY, int CX>
__device__ __forceinline__ void w63q_col_m(const float (&pf)[64], f2 (&tmp)[6][4])
{
#pragma unroll
    for (int jp = 0; jp < w63_npt<CX>() / 2; ++jp) {
        f2 m[8], r[6];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (w63_used<CY>(i)) { m[i].x = pf[(jp * 8 + i) * 2 + 0]; m[i].y = pf[(jp * 8 + i) * 2 + 1]; }
            else m[i] = w63_zero<f2>();
        }
        w63_at<CY, f2>(m, r);
#pragma unroll
        for (int i = 0; i < CY; ++i) tmp[i][jp] = r[i];
    }
}
struct W63Keep { __amdgpu_buffer_rsrc_t y, p; bool wy, wp; };      // wave-uniform: the slice's y / pre-BatchNorm block of the unit (row stride C), written or not
// second pass, two rows at a time + bias / affine / activation -> LDS tile (and y, ypre, statistics)
template <int CY, int CX, int BACK, int ACT>
__device__ __forceinline__ void w63q_row_m(const f2 (&tmp)[6][4], float b, float sc, float sh, const W63Keep& kp, unsigned C, float* act_lds, int oy, int ox,
                                           int lane, float& s1, float& s2)
{
    const int lane4 = lane * 4;
    constexpr int MY = CY, MX = CX;
    const f2 b2 = {b, b}, sc2 = {sc, sc}, sh2 = {sh, sh};
#pragma unroll
    for (int ip = 0; ip < MY / 2; ++ip) {
        f2 m[8], r[6];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (w63_used<CX>(j)) {
                const int sl = w63_slot<CX>(j);
                m[j].x = tmp[2 * ip][sl >> 1][sl & 1]; m[j].y = tmp[2 * ip + 1][sl >> 1][sl & 1];
            } else m[j] = w63_zero<f2>();
        }
        w63_at<CX, f2>(m, r);
        const int pix0 = (oy + 2 * ip) * W63_HW + ox;
        float vy[6];
#pragma unroll
        for (int j = 0; j < MX; ++j) {                // short live ranges: every value leaves as soon as it exists
            const f2 pre = r[j] + b2;
            f2 v = __builtin_elementwise_fma(pre, sc2, sh2);
            v.x = w63p_act<ACT>(v.x); v.y = w63p_act<ACT>(v.y);
            if (BACK != W63_TO_NONE) {
                act_lds[(pix0 + j) * W63_CS + lane] = v.x;
                act_lds[(pix0 + W63_HW + j) * W63_CS + lane] = v.y;
            }
            if (kp.wy) { w63_st(v.x, kp.y, lane4, (unsigned)(pix0 + j) * C * 4u); w63_st(v.y, kp.y, lane4, (unsigned)(pix0 + W63_HW + j) * C * 4u); }
            if (kp.wp) { w63_st(pre.x, kp.p, lane4, (unsigned)(pix0 + j) * C * 4u); w63_st(pre.y, kp.p, lane4, (unsigned)(pix0 + W63_HW + j) * C * 4u); }
            if (BACK == W63_TO_NONE) { s1 += v.x; s2 = fmaf(v.x, v.x, s2); vy[j] = v.y; }      // statistics in the order of the scalar form: row by row
        }
        if (BACK == W63_TO_NONE) {
#pragma unroll
            for (int j = 0; j < MX; ++j) { s1 += vy[j]; s2 = fmaf(vy[j], vy[j], s2); }
        }
    }
}

// input transform of one tile (class CY x CX) from the LDS activation tile -> V planes.  edge_y / edge_x: the patch's last row / column lies
// outside the map (tile row / column 2); its first one does for tile row / column 0 (CLS 6), known at compile time
template <int CY, int CX>
__device__ __forceinline__ void w63p_back_v(__amdgpu_buffer_rsrc_t Vn, const W63U& pl, const float* act_lds, int py0, int px0, bool edge_y, bool edge_x, int lane)
{
    const int lane4 = lane * 4;
    constexpr int NY = CY + 2, NX = CX + 2;          // patch size per direction
    f2 tmp[8][4];                                     // [vertical point][pair of patch columns]
#pragma unroll
    for (int jp = 0; jp < NX / 2; ++jp) {
        f2 d[8], r[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            d[i] = w63_zero<f2>();
            if (i >= NY || (CY == 6 && i == 0)) continue;                     // row -1 of tile row 0
            const int yy = min(py0 + i, W63_HW - 1);
            const bool yout = (CY == 4 && i == NY - 1) ? edge_y : false;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = 2 * jp + h;
                if (CX == 6 && j == 0) continue;                              // column -1 of tile column 0
                const int xx = min(px0 + j, W63_HW - 1);
                const bool out = yout || ((CX == 4 && j == NX - 1) ? edge_x : false);
                const float t = act_lds[(yy * W63_HW + xx) * W63_CS + lane];
                d[i][h] = out ? 0.f : t;
            }
        }
        w63_bt<CY, f2>(d, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) tmp[i][jp] = r[i];
    }
#pragma unroll
    for (int ipp = 0; ipp < w63_npt<CY>() / 2; ++ipp) {
        const int i0 = w63_pt<CY>(2 * ipp), i1 = w63_pt<CY>(2 * ipp + 1);
        f2 d[8], r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < NX) { d[j].x = tmp[i0][j >> 1][j & 1]; d[j].y = tmp[i1][j >> 1][j & 1]; }
            else d[j] = w63_zero<f2>();
        }
        w63_bt<CX, f2>(d, r);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (w63_used<CX>(j)) { w63_st(r[j].x, Vn, lane4, W63P_OFF(pl, i0, j)); w63_st(r[j].y, Vn, lane4, W63P_OFF(pl, i1, j)); }
    }
}

// Q = A dY A^T of one tile (class CY x CX) from the LDS tile (holding dY) -> Q planes
template <int CY, int CX>
__device__ __forceinline__ void w63p_back_q(__amdgpu_buffer_rsrc_t dst, const W63U& pl, const float* act_lds, int oy, int ox, int lane)
{
    const int lane4 = lane * 4;
    constexpr int MY = CY, MX = CX;
    f2 tmp[8][3];                                     // [vertical point][pair of tile columns]
#pragma unroll
    for (int jp = 0; jp < MX / 2; ++jp) {
        f2 d[6], r[8];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i < MY) { d[i].x = act_lds[((oy + i) * W63_HW + ox + 2 * jp) * W63_CS + lane]; d[i].y = act_lds[((oy + i) * W63_HW + ox + 2 * jp + 1) * W63_CS + lane]; }
            else d[i] = w63_zero<f2>();
        }
        w63_a<CY, f2>(d, r);
#pragma unroll
        for (int i = 0; i < 8; ++i) tmp[i][jp] = r[i];
    }
#pragma unroll
    for (int ipp = 0; ipp < w63_npt<CY>() / 2; ++ipp) {
        const int i0 = w63_pt<CY>(2 * ipp), i1 = w63_pt<CY>(2 * ipp + 1);
        f2 d[6], r[8];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (j < MX) { d[j].x = tmp[i0][j >> 1][j & 1]; d[j].y = tmp[i1][j >> 1][j & 1]; }
            else d[j] = w63_zero<f2>();
        }
        w63_a<CX, f2>(d, r);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (w63_used<CX>(j)) { w63_st(r[j].x, dst, lane4, W63P_OFF(pl, i0, j)); w63_st(r[j].y, dst, lane4, W63P_OFF(pl, i1, j)); }
    }
}


// the persistent loop of one wave = one tile position (ty, tx) of class CY x CX
template <int CY, int CX, int FRONT, int BACK, int ACT>
__device__ __forceinline__ void w63q_loop(const W63Args& a, float* lds, const int ty, const int tx, const int wave, const int lane)
{
    const unsigned C = (unsigned)a.C, NR = (unsigned)a.NR;
    const unsigned ns = C / W63_CS, nunits = NR * ns, stride = gridDim.x;
    unsigned unit = blockIdx.x;
    const unsigned slice = unit % ns;                         // the same for every unit of this workgroup: gridDim.x % ns == 0 (w63_launch)
    const unsigned c0 = slice * W63_CS;
    const int c = (int)c0 + lane, lane4 = lane * 4;
    const int oy = ty == 0 ? 0 : 2 + 4 * ty, ox = tx == 0 ? 0 : 2 + 4 * tx;               // output origin: 0, 6, 10
    const int py0 = ty == 0 ? -1 : 1 + 4 * ty, px0 = tx == 0 ? -1 : 1 + 4 * tx;           // patch origin: -1, 5, 9
    const bool ey = ty == 2, ex = tx == 2;
    const int tile = ty * 3 + tx;
    const unsigned set_bytes = 400u * NR * C * 4u;            // < 4 GiB (w63_launch)
    const unsigned img_bytes = (unsigned)(W63_HW * W63_HW) * C * 4u;
    // per-channel constants of the slice
    const float cb = (FRONT == W63_FROM_M && a.bias) ? a.bias[c] : 0.f;
    const float sc = a.scale ? a.scale[c] : 1.f, sh = a.scale ? a.shift[c] : 0.f;
    const float ka = FRONT == W63_FROM_LAZY ? a.ka[c] : 0.f, kb = FRONT == W63_FROM_LAZY ? a.kb[c] : 0.f;
    const __amdgpu_buffer_rsrc_t rsrc = w63_rsrc(a.src, FRONT == W63_FROM_M ? set_bytes : 0u);      // FROM_M: the M planes
    const __amdgpu_buffer_rsrc_t rv = w63_rsrc(a.Vn, set_bytes), rq = w63_rsrc(BACK == W63_TO_VQ ? a.Qn : a.Vn, set_bytes);
    constexpr int LDS_TILE = BACK == W63_TO_NONE ? 2 * W63_TILES * W63_CS : W63_HW * W63_HW * W63_CS;   // floats per buffer
    float pf[64];                                             // FROM_M: the tile's M values; FROM_ACT / FROM_LAZY: [0, 22) x, [32, 54) compact dy
    int slot = -1;                                            // FROM_LAZY: compact slot of the image whose values are in pf
    // the unit's activation rows of this wave: pixel wave + 9 k of image img (clamped: the 22nd of waves 7, 8 re-reads pixel 195, never used)
#define W63Q_ISSUE_ACT(IMG)                                                                                                          \
    do {                                                                                                                             \
        const __amdgpu_buffer_rsrc_t rx = w63_rsrc(a.src + ((long long)(IMG) * W63_HW * W63_HW) * a.C + c0, img_bytes);              \
        _Pragma("unroll") for (int k = 0; k < W63_NPW; ++k)                                                                          \
            pf[k] = w63_ld(rx, lane4, (unsigned)min(wave + k * W63_TILES, W63_HW * W63_HW - 1) * C * 4u);                             \
        if (FRONT == W63_FROM_LAZY) {                                                                                                \
            slot = w63_cload(a.inv, (IMG));                                                                                          \
            if (slot >= 0) {                                                                                                         \
                const __amdgpu_buffer_rsrc_t rg = w63_rsrc(a.dyc + ((long long)slot * W63_HW * W63_HW) * a.C + c0, img_bytes);       \
                _Pragma("unroll") for (int k = 0; k < W63_NPW; ++k)                                                                  \
                    pf[32 + k] = w63_ld(rg, lane4, (unsigned)min(wave + k * W63_TILES, W63_HW * W63_HW - 1) * C * 4u);                \
            }                                                                                                                        \
        }                                                                                                                            \
    } while (0)
    if (unit < nunits) {
        const unsigned img = unit / ns;
        if (FRONT == W63_FROM_M) w63q_issue_m<CY, CX>(pf, rsrc, w63p_planes(NR, img, ty, tx, C, c0), lane4);
        else if (FRONT == W63_FROM_ACT || FRONT == W63_FROM_LAZY) W63Q_ISSUE_ACT(img);
    }
    int buf = 0;
#ifdef W63_TRACE
    int w63_it = -1;
#endif
    for (; unit < nunits; unit += stride, buf ^= 1) {
#ifdef W63_TRACE
        ++w63_it;
#endif
        W63_T(0);
        const unsigned img = unit / ns;
        const unsigned nxt = unit + stride;
        const bool more = nxt < nunits;
        const unsigned nimg = nxt / ns;
        float* act = lds + buf * LDS_TILE;
        const W63U pl = w63p_planes(NR, img, ty, tx, C, c0);
        if (FRONT == W63_FROM_M) {
            W63Keep kp;
            const long long yd = a.ypre ? (long long)img : w63_keep_dst(a, img);
            kp.wy = a.y && yd >= 0;
            kp.y = w63_rsrc(a.y + (yd * W63_HW * W63_HW) * a.C + c0, kp.wy ? img_bytes : 0u);
            const long long pd = a.ypre ? w63_keep_dst(a, img) : -1;
            kp.wp = pd >= 0;
            kp.p = w63_rsrc(a.ypre + (pd * W63_HW * W63_HW) * a.C + c0, kp.wp ? img_bytes : 0u);
            f2 tmp[6][4];                             // [output row][pair of used columns]
            w63q_col_m<CY, CX>(pf, tmp);
            W63_T(1);
            if (more) w63q_issue_m<CY, CX>(pf, rsrc, w63p_planes(NR, nimg, ty, tx, C, c0), lane4);      // the next unit's rows into the registers just consumed
            float s1 = 0.f, s2 = 0.f;
            w63q_row_m<CY, CX, BACK, ACT>(tmp, cb, sc, sh, kp, C, act, oy, ox, lane, s1, s2);
            if (BACK == W63_TO_NONE && a.stats) {     // BatchNorm statistics of this image's 196 x 64 values: the nine tiles' partial sums, added in tile order
                act[tile * W63_CS + lane] = s1; act[(W63_TILES + tile) * W63_CS + lane] = s2;
                __syncthreads();
                if (wave == 0) {
                    double t1 = 0, t2 = 0;
#pragma unroll
                    for (int k = 0; k < W63_TILES; ++k) { t1 += (double)act[k * W63_CS + lane]; t2 += (double)act[(W63_TILES + k) * W63_CS + lane]; }
                    double* dst = a.stats + (long long)img * 2 * a.C;
                    dst[c] = t1;
                    dst[a.C + c] = t2;
                }
            }
        } else if (FRONT == W63_FROM_CROP) {
            // lane = (pixel lane >> 4 of a group of four, channel quad lane & 15): every corner read is 16 bytes per lane, 4 x 256 B per wave
            const float by1 = w63_cload(a.boxes, (long long)img * 4), bx1 = w63_cload(a.boxes, (long long)img * 4 + 1);
            const float by2 = w63_cload(a.boxes, (long long)img * 4 + 2), bx2 = w63_cload(a.boxes, (long long)img * 4 + 3);
            const int cq = (lane & 15) * 4;
            const float* fb = a.src + (long long)w63_cload(a.bind, img) * a.FH * a.FW * a.C + c0 + cq;
            constexpr int NP = 3;                                                          // pixels (x 4 corners) in flight per lane
            constexpr int STEP = 4 * W63_TILES;                                            // pixels per pass of the workgroup
            for (int p0 = wave * 4 + (lane >> 4); p0 < W63_HW * W63_HW; p0 += NP * STEP) {
                float4 tl[NP], tr[NP], bl[NP], br[NP];
                float wx[NP], wy[NP];
                bool ok[NP];
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int pix = min(p0 + k * STEP, W63_HW * W63_HW - 1);
                    const int py = pix / W63_HW, px = pix - py * W63_HW;
                    const W63CropAxis ay = w63_crop_axis(by1, by2, a.FH, W63_HW, py), ax = w63_crop_axis(bx1, bx2, a.FW, W63_HW, px);
                    ok[k] = ay.ok && ax.ok; wx[k] = ax.w; wy[k] = ay.w;
                    tl[k] = *reinterpret_cast<const float4*>(fb + ((long long)ay.lo * a.FW + ax.lo) * a.C);
                    tr[k] = *reinterpret_cast<const float4*>(fb + ((long long)ay.lo * a.FW + ax.hi) * a.C);
                    bl[k] = *reinterpret_cast<const float4*>(fb + ((long long)ay.hi * a.FW + ax.lo) * a.C);
                    br[k] = *reinterpret_cast<const float4*>(fb + ((long long)ay.hi * a.FW + ax.hi) * a.C);
                }
#pragma unroll
                for (int k = 0; k < NP; ++k) {
                    const int pix = p0 + k * STEP;
                    if (pix >= W63_HW * W63_HW) continue;
                    float4 o;
                    float top, bot;
                    top = tl[k].x + (tr[k].x - tl[k].x) * wx[k]; bot = bl[k].x + (br[k].x - bl[k].x) * wx[k]; o.x = top + (bot - top) * wy[k];
                    top = tl[k].y + (tr[k].y - tl[k].y) * wx[k]; bot = bl[k].y + (br[k].y - bl[k].y) * wx[k]; o.y = top + (bot - top) * wy[k];
                    top = tl[k].z + (tr[k].z - tl[k].z) * wx[k]; bot = bl[k].z + (br[k].z - bl[k].z) * wx[k]; o.z = top + (bot - top) * wy[k];
                    top = tl[k].w + (tr[k].w - tl[k].w) * wx[k]; bot = bl[k].w + (br[k].w - bl[k].w) * wx[k]; o.w = top + (bot - top) * wy[k];
                    if (!ok[k]) o = make_float4(0.f, 0.f, 0.f, 0.f);
                    *reinterpret_cast<float4*>(&act[pix * W63_CS + cq]) = o;
                }
            }
        } else {
            // FROM_ACT / FROM_LAZY: a wave owns the pixels wave, wave + 9, ...
            const long long ydst = w63_keep_dst(a, img);
            const bool wy = a.y && ydst >= 0;
            const __amdgpu_buffer_rsrc_t ry = w63_rsrc(a.y + (ydst * W63_HW * W63_HW) * a.C + c0, wy ? img_bytes : 0u);
            float v[W63_NPW];
            if (FRONT == W63_FROM_LAZY) {
#pragma unroll
                for (int k = 0; k < W63_NPW; ++k) {
                    const float z = fmaf(pf[k], sc, sh);
                    const float pass = a.act == MYOLO_ACT_RELU ? (z > 0.f ? 1.f : 0.f) : a.act == MYOLO_ACT_RELU6 ? ((z > 0.f && z < 6.f) ? 1.f : 0.f) : 1.f;
                    const float g = slot >= 0 ? pf[32 + k] : 0.f;
                    v[k] = fmaf(sc, g * pass, fmaf(kb, pf[k], ka));
                }
            } else {
#pragma unroll
                for (int k = 0; k < W63_NPW; ++k) v[k] = w63p_act<ACT>(fmaf(pf[k], sc, sh));
            }
            if (more) W63Q_ISSUE_ACT(nimg);
#pragma unroll
            for (int k = 0; k < W63_NPW; ++k) {
                const int pix = wave + k * W63_TILES;
                if (pix < W63_HW * W63_HW) act[pix * W63_CS + lane] = v[k];
            }
            if (wy) {
#pragma unroll
                for (int k = 0; k < W63_NPW; ++k) {
                    const int pix = wave + k * W63_TILES;
                    if (pix < W63_HW * W63_HW) w63_st(v[k], ry, lane4, (unsigned)pix * C * 4u);
                }
            }
        }
        W63_T(2);
        if (BACK == W63_TO_NONE) continue;
        __syncthreads();                              // the one barrier of a unit: the tile is complete; the other buffer is free again after the NEXT one
        if (BACK == W63_TO_Q || BACK == W63_TO_VQ) w63p_back_q<CY, CX>(rq, pl, act, oy, ox, lane);
        W63_T(3);
        if (BACK == W63_TO_V || BACK == W63_TO_VQ) w63p_back_v<CY, CX>(rv, pl, act, py0, px0, ey, ex, lane);
        W63_T(4);
    }
#undef W63Q_ISSUE_ACT
}

template <int FRONT, int BACK, int ACT>
__global__ __launch_bounds__(W63_TILES * 64) void wino63_boundary_kernel(W63Args a)
{
    extern __shared__ __attribute__((aligned(16))) float act_lds[];         // 2 x [14][14][64]; TO_NONE: 2 x [2][9][64] for the statistics
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ty = wave / 3, tx = wave - ty * 3;
    if (ty == 0) { if (tx == 0) w63q_loop<6, 6, FRONT, BACK, ACT>(a, act_lds, ty, tx, wave, lane); else w63q_loop<6, 4, FRONT, BACK, ACT>(a, act_lds, ty, tx, wave, lane); }
    else         { if (tx == 0) w63q_loop<4, 6, FRONT, BACK, ACT>(a, act_lds, ty, tx, wave, lane); else w63q_loop<4, 4, FRONT, BACK, ACT>(a, act_lds, ty, tx, wave, lane); }
}

// (A persistent form -- a few workgroups per CU walking the units in a loop, the unit body an out-of-line call because the inlined loop made the
// compiler keep the 64 plane offsets live across iterations: 168 VGPRs + scratch -- was measured in round 3: 1.25 against 0.81 ms.  One workgroup
// serialises load -> transform -> store; the one-unit-per-workgroup launch overlaps them across workgroups.  Removed; profiles/r3_notes.md.)

// dU [64][Ci][Co] -> dw [3,3,Ci,Co] = G8^T dU G8
__global__ __launch_bounds__(256) void wino63_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Ci, int Co)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Ci * Co) return;
    const long long plane = (long long)Ci * Co;
    float tmp[3][8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float col[8], r[3];
#pragma unroll
        for (int i = 0; i < 8; ++i) col[i] = dU[w63_q(i, j) * plane + idx];
        w63_gt(col, r);
#pragma unroll
        for (int k = 0; k < 3; ++k) tmp[k][j] = r[k];
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float r[3];
        w63_gt(tmp[ky], r);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) dw[(ky * 3 + kx) * plane + idx] = r[kx];
    }
}

// w [3,3,Ci,Co] -> 64 planes U[q] = (G8 g G8^T)[i][j] in w63_q order; layout per plane: 0 = [Ci][Co], 1 = [Co][Ci] (transposed for
// wino_mm_kernel), 2 = the split-bf16 operand order of wino_mm_x6_kernel (see wino_kernels.hip: wino_w_kernel)
__global__ __launch_bounds__(256) void wino63_w_kernel(const float* __restrict__ w, float* __restrict__ U, int Ci, int Co, int layout, int flip)
{
    // flip = 1: the 180-degree rotated filter with (ci, co) exchanged (data gradient): the multiply's K = Co, N = Ci then
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Ci * Co) return;
    // consecutive threads walk the contiguous axis of the destination: the multiply's k for layouts 1 and 2
    const bool ci_fast = layout != 0 && !flip;
    const int ci = ci_fast ? idx % Ci : idx / Co, co = ci_fast ? idx / Ci : idx % Co;
    const int K = flip ? Co : Ci, N = flip ? Ci : Co, k = flip ? co : ci, n = flip ? ci : co;
    float g[3][3], tmp[8][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[((flip ? (2 - ky) * 3 + (2 - kx) : ky * 3 + kx) * Ci + ci) * (long long)Co + co];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const float col[3] = {g[0][kx], g[1][kx], g[2][kx]};
        float u[8];
        w63_g(col, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) tmp[i][kx] = u[i];
    }
    const long long plane = (long long)Ci * Co;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float u[8];
        w63_g(tmp[i], u);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = w63_q(i, j);
            if (layout == 0) U[q * plane + (long long)k * N + n] = u[j];
            else if (layout == 1) U[q * plane + (long long)n * K + k] = u[j];
            else {
                const int nkc = K >> 4;
                __bf16* rec = reinterpret_cast<__bf16*>(U) + ((((long long)q * nkc + (k >> 4)) * (N >> 5) + (n >> 5)) * 6 + ((k >> 3) & 1)) * 256 + (n & 31) * 8 + (k & 7);
                const __bf16 p1 = (__bf16)u[j];
                const float r1 = u[j] - (float)p1;
                const __bf16 p2 = (__bf16)r1;
                rec[0] = p1; rec[512] = p2; rec[1024] = (__bf16)(r1 - (float)p2);
            }
        }
    }
}

static int w63_layout(int K, int N) { return myolo_gemm_nt_batched_x6(K, N) ? 2 : 1; }

static int w63_num_cus()
{
    static int n = 0;
    if (!n) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess || prop.multiProcessorCount <= 0) return 256;
        n = prop.multiProcessorCount;
    }
    return n;
}
template <int FRONT, int BACK, int ACT>
static void w63_launch_act(const W63Args& b, unsigned grid, size_t lds, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wino63_boundary_kernel<FRONT, BACK, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((wino63_boundary_kernel<FRONT, BACK, ACT>), dim3(grid), dim3(W63_TILES * 64), lds, s, b);
}
template <int FRONT, int BACK>
static int w63_launch(const W63Args& a, hipStream_t s)
{
    W63Args b = a;
    b.order = g_myolo_opt.w63_order ? 0 : 1;
    const long long units = a.NR * (a.C / W63_CS);
    if (units <= 0 || units >= (1LL << 31)) return MYOLO_EINVAL;
    // The ROIAlign front gathers from the (cache-resident) feature map in five dependent passes: nothing to prefetch, and what hides its latency is a
    // SECOND workgroup on the CU -- the one-unit kernel at 80 VGPRs / 50 KB of LDS fits two (0.45 ms against 0.53-0.55 for the persistent form): it stays.
    // Plane sets of 4 GiB and more (32-bit buffer offsets) also take the one-unit kernel.
    const bool one_unit = g_myolo_opt.w63_legacy || FRONT == W63_FROM_CROP || (unsigned long long)400 * a.NR * a.C * 4ull >= (1ull << 32);
    if (one_unit) {
        static bool attr_set = false;
        const size_t lds = (size_t)W63_HW * W63_HW * W63_CS * sizeof(float);
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void*)wino63_boundary_legacy_kernel<FRONT, BACK>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        hipLaunchKernelGGL((wino63_boundary_legacy_kernel<FRONT, BACK>), dim3((unsigned)units), dim3(W63_TILES * 64), lds, s, b);
        return MYOLO_OK;
    }
    // one persistent workgroup per CU (option "w63_wgs": workgroups per CU, default 1), a multiple of the slices per image so that a workgroup's slice
    // never changes; two LDS buffers (without a back half nobody reads the tile: only the statistics' partial sums pass through LDS)
    const int ns = a.C / W63_CS;
    const int per_cu = g_myolo_opt.w63_wgs > 0 ? g_myolo_opt.w63_wgs : 1;
    long long grid = (long long)w63_num_cus() * per_cu;
    grid -= grid % ns;
    if (grid < ns) grid = ns;
    if (grid > units) grid = units;                                        // units % ns == 0
    const size_t lds = 2 * sizeof(float) * (BACK == W63_TO_NONE ? (size_t)2 * W63_TILES * W63_CS : (size_t)W63_HW * W63_HW * W63_CS);
    // the activation is a compile-time parameter where the kernel applies it element by element (FROM_LAZY reads it as a mask: run time)
    if (FRONT == W63_FROM_LAZY || FRONT == W63_FROM_CROP || a.act == MYOLO_ACT_NONE) w63_launch_act<FRONT, BACK, MYOLO_ACT_NONE>(b, (unsigned)grid, lds, s);
    else if (a.act == MYOLO_ACT_RELU) w63_launch_act<FRONT, BACK, MYOLO_ACT_RELU>(b, (unsigned)grid, lds, s);
    else w63_launch_act<FRONT, BACK, MYOLO_ACT_RELU6>(b, (unsigned)grid, lds, s);
    return MYOLO_OK;
}

extern "C" {

/* whether the F(6,3)/F(4,3) tiling is available for a conv: 14x14 maps, Cin and Cout handled by the one-launch multiply */
int myolo_wino63_ok(int H, int W, int Cin, int Cout)
{
    return H == W63_HW && W == W63_HW && (Cin % W63_CS) == 0 && (Cout % W63_CS) == 0 && myolo_gemm_nt_batched_ok(Cin, Cout) ? 1 : 0;
}
/* elements of the 64 V (or M) planes of an [N,14,14,C] tensor: 400 point-tiles per image */
size_t myolo_wino63_plane_elems(int N, int C) { return (size_t)400 * (size_t)N * (size_t)C; }
/* floats to allocate for the transformed filters (64 planes, 6 bytes per value in the split-bf16 layout) */
size_t myolo_wino63_u_elems(int Cin, int Cout) { return align256((size_t)64 * Cin * Cout * 6) / sizeof(float); }

int myolo_wino63_weight_transform(const float* w, float* U, int Cin, int Cout, void* stream)
{
    MYOLO_REQUIRE(w && U && myolo_wino63_ok(W63_HW, W63_HW, Cin, Cout), "wino63_weight_transform: needs Cin, Cout multiples of 64 with Cin %% 16 == 0, Cout %% 256 == 0 (got %d, %d)", Cin, Cout);
    hipLaunchKernelGGL(wino63_w_kernel, dim3((Cin * Cout + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, U, Cin, Cout, w63_layout(Cin, Cout), 0);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* M[q] = V[q] * U[q] for the 64 points: one launch over the three runs of equal-height planes (csrc/wino_mm.hip) */
// the transformed filters of a layer: into `scratch`, or the copy prepared for this step (prepared-weights registry, csrc/myolo_common.h).
// flip = 1: rotated, (ci, co)-exchanged filters of the data gradient (Cin / Cout as the FORWARD conv has them, as wino63_w_kernel takes them)
static const float* w63_filters(const float* w, float* scratch, int Cin, int Cout, int flip, hipStream_t s)
{
    const int layout = flip ? w63_layout(Cout, Cin) : w63_layout(Cin, Cout);
    return (const float*)myolo_wprep_resolve(w, WP_WINO63_U, Cin, Cout, flip * 16 + layout, myolo_wino63_u_elems(Cin, Cout) * sizeof(float), scratch, s,
                                             [=](void* d, hipStream_t st) {
        hipLaunchKernelGGL(wino63_w_kernel, dim3((Cin * Cout + 255) / 256), dim3(256), 0, st, w, (float*)d, Cin, Cout, layout, flip);
    });
}

/* weight_transform + multiply in one call: U_scratch (myolo_wino63_u_elems floats) is written only when the filters are not already prepared */
int myolo_wino63_multiply_w(const float* V, const float* w, float* U_scratch, float* M, int N, int Cin, int Cout, void* stream)
{
    MYOLO_REQUIRE(V && w && U_scratch && M && N > 0 && myolo_wino63_ok(W63_HW, W63_HW, Cin, Cout), "wino63_multiply_w: bad arguments");
    const float* U = w63_filters(w, U_scratch, Cin, Cout, 0, (hipStream_t)stream);
    return myolo_wino63_multiply(V, U, M, N, Cin, Cout, stream);
}

int myolo_wino63_multiply(const float* V, const float* U, float* M, int N, int Cin, int Cout, void* stream)
{
    MYOLO_REQUIRE(V && U && M && N > 0 && myolo_wino63_ok(W63_HW, W63_HW, Cin, Cout), "wino63_multiply: bad arguments");
    const long long NR = N;
    const long long rows[3] = {9 * NR, 3 * NR, NR};
    const int nq[3] = {36, 24, 4};
    const long long prow[3] = {0, 36 * 9 * NR, 36 * 9 * NR + 24 * 3 * NR};       // first row of each run, counted over all planes
    const long long pq[3] = {0, 36, 60};
    long long ao[3], bo[3], co[3];
    for (int k = 0; k < 3; ++k) { ao[k] = prow[k] * Cin; bo[k] = pq[k] * (long long)Cin * Cout; co[k] = prow[k] * Cout; }
    const int rc = myolo_gemm_nt_batched_runs(V, U, M, 3, rows, ao, bo, co, nq, Cin, Cout, (hipStream_t)stream);
    if (rc != MYOLO_OK) return rc;
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* x [N,14,14,C] -> act(x * scale + shift) (scale NULL: identity) -> V; the activation is also written to y where flags allow */
int myolo_wino63_input_transform(const float* x, const float* scale, const float* shift, int act, float* y, const int32_t* flags, float* V,
                                 int N, int C, void* stream)
{
    MYOLO_REQUIRE(x && V && N > 0 && (C % W63_CS) == 0 && !scale == !shift, "wino63_input_transform: bad arguments (C %% 64 == 0)");
    W63Args a{x, V, y, flags, nullptr, scale, shift, N, C, act};
    w63_launch<W63_FROM_ACT, W63_TO_V>(a, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* layer boundary: M_i -> act((A^T m A + bias) * scale + shift) -> V_{i+1}; y (NULL: never) written where flags[img] != 0 (NULL: always) */
int myolo_wino63_output_input_transform(const float* M, const float* bias, const float* scale, const float* shift, float* y,
                                        const int32_t* flags, float* Vn, int N, int C, int act, void* stream)
{
    MYOLO_REQUIRE(M && Vn && N > 0 && (C % W63_CS) == 0 && !scale == !shift, "wino63_output_input_transform: bad arguments (C %% 64 == 0)");
    W63Args a{M, Vn, y, flags, bias, scale, shift, N, C, act};
    w63_launch<W63_FROM_M, W63_TO_V>(a, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* the two calls above with the conv's PRE-BatchNorm output (A^T m A + bias) kept for the flagged ROIs: ypre written where flags[img] != 0 (NULL:
 * everywhere); y (output_transform_keep_pre: the activation every ROI's consumer reads; may be NULL) is written everywhere */
int myolo_wino63_output_input_transform_keep_pre(const float* M, const float* bias, const float* scale, const float* shift, float* ypre,
                                                 const int32_t* flags, float* Vn, int N, int C, int act, void* stream)
{
    MYOLO_REQUIRE(M && Vn && ypre && N > 0 && (C % W63_CS) == 0 && !scale == !shift, "wino63_output_input_transform_keep_pre: bad arguments (C %% 64 == 0)");
    W63Args a{M, Vn, nullptr, flags, bias, scale, shift, N, C, act};
    a.ypre = ypre;
    w63_launch<W63_FROM_M, W63_TO_V>(a, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_wino63_output_transform_keep_pre(const float* M, const float* bias, const float* scale, const float* shift, float* y, float* ypre,
                                           const int32_t* flags, int N, int C, int act, void* stream)
{
    MYOLO_REQUIRE(M && ypre && N > 0 && (C % W63_CS) == 0 && !scale == !shift, "wino63_output_transform_keep_pre: bad arguments (C %% 64 == 0)");
    W63Args a{M, nullptr, y, flags, bias, scale, shift, N, C, act};
    a.ypre = ypre;
    w63_launch<W63_FROM_M, W63_TO_NONE>(a, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* ---- the three calls above with the kept tensor written in COMPACT order: slots [N] = compact slot of an image or -1 (myolo_positive_index), rows of
 * image n at block slots[n] when 0 <= slots[n] < cap.  input_transform_slots: y_compact = the activation act(x * scale + shift) of the kept images;
 * the *_keep_pre_slots pair: ypre_compact = the conv's pre-BatchNorm output of the kept images (y of output_transform stays dense). ---- */
int myolo_wino63_input_transform_slots(const float* x, const float* scale, const float* shift, int act, float* y_compact, const int32_t* slots, int cap,
                                       float* V, int N, int C, void* stream)
{
    MYOLO_REQUIRE(x && V && y_compact && slots && cap > 0 && N > 0 && (C % W63_CS) == 0 && !scale == !shift, "wino63_input_transform_slots: bad arguments (C %% 64 == 0)");
    W63Args a{x, V, y_compact, slots, nullptr, scale, shift, N, C, act};
    a.keep_cap = cap;
    w63_launch<W63_FROM_ACT, W63_TO_V>(a, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}
int myolo_wino63_output_input_transform_keep_pre_slots(const float* M, const float* bias, const float* scale, const float* shift, float* ypre_compact,
                                                       const int32_t* slots, int cap, float* Vn, int N, int C, int act, void* stream)
{
    MYOLO_REQUIRE(M && Vn && ypre_compact && slots && cap > 0 && N > 0 && (C % W63_CS) == 0 && !scale == !shift,
                  "wino63_output_input_transform_keep_pre_slots: bad arguments (C %% 64 == 0)");
    W63Args a{M, Vn, nullptr, slots, bias, scale, shift, N, C, act};
    a.ypre = ypre_compact; a.keep_cap = cap;
    w63_launch<W63_FROM_M, W63_TO_V>(a, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}
int myolo_wino63_output_transform_keep_pre_slots(const float* M, const float* bias, const float* scale, const float* shift, float* y, float* ypre_compact,
                                                 const int32_t* slots, int cap, int N, int C, int act, void* stream)
{
    MYOLO_REQUIRE(M && ypre_compact && slots && cap > 0 && N > 0 && (C % W63_CS) == 0 && !scale == !shift,
                  "wino63_output_transform_keep_pre_slots: bad arguments (C %% 64 == 0)");
    W63Args a{M, nullptr, y, slots, bias, scale, shift, N, C, act};
    a.ypre = ypre_compact; a.keep_cap = cap;
    w63_launch<W63_FROM_M, W63_TO_NONE>(a, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_wino63_output_transform(const float* M, const float* bias, const float* scale, const float* shift, float* y, int N, int C, int act,
                                  void* stream)
{
    MYOLO_REQUIRE(M && y && N > 0 && (C % W63_CS) == 0 && !scale == !shift, "wino63_output_transform: bad arguments (C %% 64 == 0)");
    W63Args a{M, nullptr, y, nullptr, bias, scale, shift, N, C, act};
    w63_launch<W63_FROM_M, W63_TO_NONE>(a, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* ROIAlign fused into the input transform (myolo_wino_input_transform_roialign on this tiling): V straight from the feature map
 * [B,FH,FW,C] for nb boxes (y1,x1,y2,x2) / box_ind, crop 14x14; the crops are never written */
int myolo_wino63_input_transform_roialign(const float* feature, const float* boxes, const int32_t* box_ind, float* V, int B, int FH, int FW,
                                          int C, int nb, void* stream)
{
    MYOLO_REQUIRE(feature && boxes && box_ind && V && B > 0 && FH > 0 && FW > 0 && nb > 0 && (C % W63_CS) == 0,
                  "wino63_input_transform_roialign: bad arguments (C %% 64 == 0)");
    W63Args a{};
    a.src = feature; a.Vn = V; a.NR = nb; a.C = C; a.act = MYOLO_ACT_NONE; a.boxes = boxes; a.bind = box_ind; a.FH = FH; a.FW = FW;
    w63_launch<W63_FROM_CROP, W63_TO_V>(a, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

size_t myolo_wino63_output_transform_bn_ws_bytes(int N, int C) { return align256((size_t)N * 2 * C * sizeof(double)) + 2 * C * sizeof(double); }

/* conv + bias -> y, and the training-mode BatchNorm statistics of y in the same pass (myolo_wino_output_transform_bn_stats on this
 * tiling; model.py:690): mean / var / folded scale, shift and the moving averages, exactly as myolo_bn_stats */
int myolo_wino63_output_transform_bn_stats(const float* M, const float* bias, float* y, int N, int C, const float* gamma, const float* beta,
                                           float* mean, float* var, float* scale, float* shift, float* moving_mean, float* moving_var,
                                           void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(M && y && gamma && beta && mean && var && scale && shift && N > 0 && (C % W63_CS) == 0, "wino63_output_transform_bn_stats: bad arguments");
    MYOLO_NEED_WS(myolo_wino63_output_transform_bn_ws_bytes(N, C));
    hipStream_t s = (hipStream_t)stream;
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + align256((size_t)N * 2 * C * sizeof(double)));
    W63Args a{};
    a.src = M; a.y = y; a.bias = bias; a.NR = N; a.C = C; a.act = MYOLO_ACT_NONE; a.stats = part;
    w63_launch<W63_FROM_M, W63_TO_NONE>(a, s);
    myolo_bn_stats_from_partials(part, tot, N, C, (double)N * W63_HW * W63_HW, gamma, beta, mean, var, scale, shift, moving_mean, moving_var, s);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

#ifdef W63_TRACE
int myolo_w63_trace_read(void* dst, size_t bytes) { return hipMemcpyFromSymbol(dst, HIP_SYMBOL(w63_trace_buf), bytes) == hipSuccess ? 0 : -1; }
#endif
static const long long* w63_run_rows(long long NR, long long rows[3]) { rows[0] = 9 * NR; rows[1] = 3 * NR; rows[2] = NR; return rows; }

}  // extern "C"

// dU[q] = V[q]^T Q[q] per run of equal-height planes, then dw = G8^T dU G8
static int w63_tn_and_dw(const float* V, const float* Q, float* dU, float* dw, int N, int Cin, int Cout, void* part, size_t part_bytes, hipStream_t s)
{
    long long rows[3];
    w63_run_rows(N, rows);
    const int nq[3] = {36, 24, 4};
    const long long prow[3] = {0, 36 * rows[0], 36 * rows[0] + 24 * rows[1]};
    const long long pq[3] = {0, 36, 60};
    if (myolo_gemm_tn_x6_ok(Cin, Cout)) {          // FP32_MATMUL = "bf16x6": all 64 planes in one launch of wino_tn_x6_kernel (csrc/wino_mm.hip)
        const long long ao[3] = {prow[0] * Cin, prow[1] * Cin, prow[2] * Cin}, bo[3] = {prow[0] * Cout, prow[1] * Cout, prow[2] * Cout};
        const int rc = myolo_gemm_tn_x6_runs(V, Q, dU, 3, rows, ao, bo, nq, Cin, Cout, part, part_bytes, s);
        if (rc != MYOLO_OK) return rc;
        hipLaunchKernelGGL(wino63_dw_kernel, dim3((Cin * Cout + 255) / 256), dim3(256), 0, s, dU, dw, Cin, Cout);
        return MYOLO_OK;
    }
    for (int k = 0; k < 3; ++k) {
        const int rc = myolo_gemm_tn_batched(V + prow[k] * Cin, Q + prow[k] * Cout, dU + pq[k] * (long long)Cin * Cout, rows[k], Cin, Cout, nq[k],
                                             part, part_bytes, s);
        if (rc != MYOLO_OK) return rc;
    }
    hipLaunchKernelGGL(wino63_dw_kernel, dim3((Cin * Cout + 255) / 256), dim3(256), 0, s, dU, dw, Cin, Cout);
    return MYOLO_OK;
}

extern "C" {

size_t myolo_wino63_bwd_weight_ws_bytes(int N, int Cin, int Cout)
{
    long long rows[3];
    w63_run_rows(N, rows);
    const int nq[3] = {36, 24, 4};
    size_t pb = 0;
    for (int k = 0; k < 3; ++k) { const size_t b = myolo_gemm_tn_batched_ws_bytes(rows[k], Cin, Cout, nq[k]); if (b > pb) pb = b; }
    if (Cin % 256 == 0 && Cout % 256 == 0) { const size_t b = myolo_gemm_tn_x6_ws_bytes(3, rows, nq, Cin, Cout); if (b > pb) pb = b; }
    return align256((size_t)64 * Cin * Cout * sizeof(float)) + align256(myolo_wino63_plane_elems(N, Cout) * sizeof(float)) + align256(pb);
}

/* Weight gradient of the same conv from the V planes kept by the forward (v_saved: myolo_wino63_input_transform[_roialign]) and the
 * lazily formed gradient of the conv's output (see myolo_wino63_bwd_data_lazybn): dU[q] = V[q]^T (A dY A^T)[q], dw = G8^T dU G8 */
int myolo_wino63_bwd_weight_lazybn(const float* v_saved, const float* y_pre, const float* dy_compact, const int32_t* inv, const float* scale,
                                   const float* shift, const float* ka, const float* kb, int act, float* dw, int N, int Cin, int Cout,
                                   void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(v_saved && y_pre && inv && scale && shift && ka && kb && dw && N > 0, "wino63_bwd_weight_lazybn: bad arguments");
    MYOLO_REQUIRE((Cin % W63_CS) == 0 && (Cout % W63_CS) == 0, "wino63_bwd_weight_lazybn: channels must be multiples of 64 (got %d, %d)", Cin, Cout);
    MYOLO_NEED_WS(myolo_wino63_bwd_weight_ws_bytes(N, Cin, Cout));
    hipStream_t s = (hipStream_t)stream;
    float* dU = (float*)ws;
    float* Q = (float*)((char*)ws + align256((size_t)64 * Cin * Cout * sizeof(float)));
    void* part = (char*)Q + align256(myolo_wino63_plane_elems(N, Cout) * sizeof(float));
    const size_t part_bytes = ws_bytes - (size_t)((char*)part - (char*)ws);
    W63Args a{};
    a.src = y_pre; a.Vn = Q; a.scale = scale; a.shift = shift; a.NR = N; a.C = Cout; a.act = act; a.dyc = dy_compact; a.inv = inv; a.ka = ka; a.kb = kb;
    w63_launch<W63_FROM_LAZY, W63_TO_Q>(a, s);
    const int rc = w63_tn_and_dw(v_saved, Q, dU, dw, N, Cin, Cout, part, part_bytes, s);
    if (rc != MYOLO_OK) return rc;
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

size_t myolo_wino63_bwd_data_ws_bytes(int N, int Cin, int Cout)
{
    return align256(myolo_wino63_u_elems(Cin, Cout) * sizeof(float)) + align256(myolo_wino63_plane_elems(N, Cout) * sizeof(float)) +
           align256(myolo_wino63_plane_elems(N, Cin) * sizeof(float));
}

/* Data gradient of a 3x3 / s1 / SAME conv on 14x14 maps behind a training-mode BatchNorm + activation with a row-sparse upstream
 * gradient -- myolo_conv3x3_wino_bwd_data_lazybn (bn1 / conv1 of the mask head, model.py:687-690) on the F(6,3)/F(4,3) tiling:
 * y_pre is the BN's input (= the conv's output), dy_compact / inv / ka / kb as produced by myolo_bn_bwd_rowsparse_coeffs; the dense
 * gradient is formed while the transform loads y_pre and never written.  Needs myolo_wino63_ok(14, 14, Cout, Cin). */
int myolo_wino63_bwd_data_lazybn(const float* y_pre, const float* dy_compact, const int32_t* inv, const float* scale, const float* shift,
                                 const float* ka, const float* kb, int act, const float* w, float* dx, int N, int Cin, int Cout, void* ws,
                                 size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(y_pre && inv && scale && shift && ka && kb && w && dx && N > 0, "wino63_bwd_data_lazybn: bad arguments");
    MYOLO_REQUIRE(myolo_wino63_ok(W63_HW, W63_HW, Cout, Cin), "wino63_bwd_data_lazybn: unsupported channel counts (%d -> %d)", Cin, Cout);
    MYOLO_NEED_WS(myolo_wino63_bwd_data_ws_bytes(N, Cin, Cout));
    hipStream_t s = (hipStream_t)stream;
    float* V = (float*)((char*)ws + align256(myolo_wino63_u_elems(Cin, Cout) * sizeof(float)));
    float* Mp = (float*)((char*)V + align256(myolo_wino63_plane_elems(N, Cout) * sizeof(float)));
    const float* U = w63_filters(w, (float*)ws, Cin, Cout, 1, s);
    W63Args a{y_pre, V, nullptr, nullptr, nullptr, scale, shift, N, Cout, act, dy_compact, inv, ka, kb};
    w63_launch<W63_FROM_LAZY, W63_TO_V>(a, s);
    const int rc = myolo_wino63_multiply(V, U, Mp, N, Cout, Cin, stream);
    if (rc != MYOLO_OK) return rc;
    W63Args b{Mp, nullptr, dx, nullptr, nullptr, nullptr, nullptr, N, Cin, MYOLO_ACT_NONE};
    w63_launch<W63_FROM_M, W63_TO_NONE>(b, s);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* The two transforms of conv1's backward from ONE pass over y_pre: the lazily formed gradient of the conv's output (see
 * myolo_wino63_bwd_data_lazybn) fills the 14x14 tile in LDS once, then goes out as V (input transform: operand of the data gradient) and
 * as Q (adjoint output transform: operand of the weight gradient).  V, Q: myolo_wino63_plane_elems(N, C) floats each. */
int myolo_wino63_lazybn_transforms(const float* y_pre, const float* dy_compact, const int32_t* inv, const float* scale, const float* shift,
                                   const float* ka, const float* kb, int act, float* V, float* Q, int N, int C, void* stream)
{
    MYOLO_REQUIRE(y_pre && inv && scale && shift && ka && kb && V && Q && N > 0 && (C % W63_CS) == 0, "wino63_lazybn_transforms: bad arguments");
    W63Args a{y_pre, V, nullptr, nullptr, nullptr, scale, shift, N, C, act, dy_compact, inv, ka, kb};
    a.Qn = Q;
    w63_launch<W63_FROM_LAZY, W63_TO_VQ>(a, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

size_t myolo_wino63_bwd_data_from_v_ws_bytes(int N, int Cin, int Cout)
{
    return align256(myolo_wino63_u_elems(Cin, Cout) * sizeof(float)) + align256(myolo_wino63_plane_elems(N, Cin) * sizeof(float));
}

/* the rest of the data gradient: dx = output transform of (V x rotated, (ci,co)-exchanged filters); V from myolo_wino63_lazybn_transforms */
int myolo_wino63_bwd_data_from_v(const float* V, const float* w, float* dx, int N, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(V && w && dx && N > 0, "wino63_bwd_data_from_v: bad arguments");
    MYOLO_REQUIRE(myolo_wino63_ok(W63_HW, W63_HW, Cout, Cin), "wino63_bwd_data_from_v: unsupported channel counts (%d -> %d)", Cin, Cout);
    MYOLO_NEED_WS(myolo_wino63_bwd_data_from_v_ws_bytes(N, Cin, Cout));
    hipStream_t s = (hipStream_t)stream;
    float* Mp = (float*)((char*)ws + align256(myolo_wino63_u_elems(Cin, Cout) * sizeof(float)));
    const float* U = w63_filters(w, (float*)ws, Cin, Cout, 1, s);
    const int rc = myolo_wino63_multiply(V, U, Mp, N, Cout, Cin, stream);
    if (rc != MYOLO_OK) return rc;
    W63Args b{Mp, nullptr, dx, nullptr, nullptr, nullptr, nullptr, N, Cin, MYOLO_ACT_NONE};
    w63_launch<W63_FROM_M, W63_TO_NONE>(b, s);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

size_t myolo_wino63_bwd_weight_from_q_ws_bytes(int N, int Cin, int Cout)
{
    long long rows[3];
    w63_run_rows(N, rows);
    const int nq[3] = {36, 24, 4};
    size_t pb = 0;
    for (int k = 0; k < 3; ++k) { const size_t b = myolo_gemm_tn_batched_ws_bytes(rows[k], Cin, Cout, nq[k]); if (b > pb) pb = b; }
    if (Cin % 256 == 0 && Cout % 256 == 0) { const size_t b = myolo_gemm_tn_x6_ws_bytes(3, rows, nq, Cin, Cout); if (b > pb) pb = b; }
    return align256((size_t)64 * Cin * Cout * sizeof(float)) + align256(pb);
}

/* the rest of the weight gradient: dU[q] = V_saved[q]^T Q[q], dw = G8^T dU G8; Q from myolo_wino63_lazybn_transforms */
int myolo_wino63_bwd_weight_from_q(const float* v_saved, const float* Q, float* dw, int N, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(v_saved && Q && dw && N > 0 && (Cin % W63_CS) == 0 && (Cout % W63_CS) == 0, "wino63_bwd_weight_from_q: bad arguments");
    MYOLO_NEED_WS(myolo_wino63_bwd_weight_from_q_ws_bytes(N, Cin, Cout));
    float* dU = (float*)ws;
    void* part = (char*)ws + align256((size_t)64 * Cin * Cout * sizeof(float));
    const size_t part_bytes = ws_bytes - (size_t)((char*)part - (char*)ws);
    const int rc = w63_tn_and_dw(v_saved, Q, dU, dw, N, Cin, Cout, part, part_bytes, (hipStream_t)stream);
    if (rc != MYOLO_OK) return rc;
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* ---- the three conv operators on this tiling with the signatures of myolo_conv3x3_wino_{fwd,bwd_data,bwd_weight} (14x14 maps only) ---- */
size_t myolo_conv3x3_wino63_ws_bytes(int N, int Cin, int Cout, int which)
{
    const size_t ub = align256(myolo_wino63_u_elems(Cin, Cout) * sizeof(float));
    const size_t vi = align256(myolo_wino63_plane_elems(N, Cin) * sizeof(float)), vo = align256(myolo_wino63_plane_elems(N, Cout) * sizeof(float));
    if (which == 0 || which == 1) return ub + vi + vo;                              // U, V, M
    return myolo_wino63_bwd_weight_ws_bytes(N, Cin, Cout) + vi;                     // dU, Q, partials (+ V when it was not kept)
}

/* y = act((conv3x3_same(x, w) + bias) * scale + shift); v_keep (optional) receives the transformed input for the weight gradient.
 * Needs myolo_wino63_ok(14, 14, Cin, Cout). */
int myolo_conv3x3_wino63_fwd(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y, int N,
                             int Cin, int Cout, int act, float* v_keep, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && w && y && N > 0 && !scale == !shift, "conv3x3_wino63_fwd: bad arguments");
    MYOLO_REQUIRE(myolo_wino63_ok(W63_HW, W63_HW, Cin, Cout), "conv3x3_wino63_fwd: unsupported channel counts (%d -> %d)", Cin, Cout);
    MYOLO_NEED_WS(myolo_conv3x3_wino63_ws_bytes(N, Cin, Cout, 0));
    float* V = v_keep ? v_keep : (float*)((char*)ws + align256(myolo_wino63_u_elems(Cin, Cout) * sizeof(float)));
    float* Mp = (float*)((char*)ws + align256(myolo_wino63_u_elems(Cin, Cout) * sizeof(float)) + align256(myolo_wino63_plane_elems(N, Cin) * sizeof(float)));
    MYOLO_REQUIRE(w && myolo_wino63_ok(W63_HW, W63_HW, Cin, Cout), "conv3x3_wino63_fwd: unsupported channel counts (%d -> %d)", Cin, Cout);
    const float* U = w63_filters(w, (float*)ws, Cin, Cout, 0, (hipStream_t)stream);
    int rc = MYOLO_OK;
    if (rc == MYOLO_OK) rc = myolo_wino63_input_transform(x, nullptr, nullptr, MYOLO_ACT_NONE, nullptr, nullptr, V, N, Cin, stream);
    if (rc == MYOLO_OK) rc = myolo_wino63_multiply(V, U, Mp, N, Cin, Cout, stream);
    if (rc == MYOLO_OK) rc = myolo_wino63_output_transform(Mp, bias, scale, shift, y, N, Cout, act, stream);
    return rc;
}

/* dx = conv3x3_same(dy, rot180(w)^T): needs myolo_wino63_ok(14, 14, Cout, Cin) */
int myolo_conv3x3_wino63_bwd_data(const float* dy, const float* w, float* dx, int N, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(dy && w && dx && N > 0, "conv3x3_wino63_bwd_data: bad arguments");
    MYOLO_REQUIRE(myolo_wino63_ok(W63_HW, W63_HW, Cout, Cin), "conv3x3_wino63_bwd_data: unsupported channel counts (%d -> %d)", Cin, Cout);
    MYOLO_NEED_WS(myolo_conv3x3_wino63_ws_bytes(N, Cin, Cout, 1));
    hipStream_t s = (hipStream_t)stream;
    float* V = (float*)((char*)ws + align256(myolo_wino63_u_elems(Cin, Cout) * sizeof(float)));
    float* Mp = (float*)((char*)V + align256(myolo_wino63_plane_elems(N, Cout) * sizeof(float)));
    const float* U = w63_filters(w, (float*)ws, Cin, Cout, 1, s);
    int rc = myolo_wino63_input_transform(dy, nullptr, nullptr, MYOLO_ACT_NONE, nullptr, nullptr, V, N, Cout, stream);
    if (rc == MYOLO_OK) rc = myolo_wino63_multiply(V, U, Mp, N, Cout, Cin, stream);
    if (rc == MYOLO_OK) rc = myolo_wino63_output_transform(Mp, nullptr, nullptr, nullptr, dx, N, Cin, MYOLO_ACT_NONE, stream);
    return rc;
}

/* dw = sum over pixels of x (x) dy; the forward's transformed input may be passed instead of x (v_saved).  Channels multiples of 64. */
int myolo_conv3x3_wino63_bwd_weight(const float* x, const float* v_saved, const float* dy, float* dw, int N, int Cin, int Cout, void* ws,
                                    size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE((x || v_saved) && dy && dw && N > 0, "conv3x3_wino63_bwd_weight: bad arguments");
    MYOLO_REQUIRE((Cin % W63_CS) == 0 && (Cout % W63_CS) == 0, "conv3x3_wino63_bwd_weight: channels must be multiples of 64 (got %d, %d)", Cin, Cout);
    MYOLO_NEED_WS(myolo_conv3x3_wino63_ws_bytes(N, Cin, Cout, 2));
    hipStream_t s = (hipStream_t)stream;
    float* dU = (float*)ws;
    float* Q = (float*)((char*)ws + align256((size_t)64 * Cin * Cout * sizeof(float)));
    char* after_q = (char*)Q + align256(myolo_wino63_plane_elems(N, Cout) * sizeof(float));
    float* V = (float*)after_q;
    void* part = v_saved ? (void*)after_q : (void*)(after_q + align256(myolo_wino63_plane_elems(N, Cin) * sizeof(float)));
    const size_t part_bytes = ws_bytes - (size_t)((char*)part - (char*)ws);
    if (!v_saved) {
        const int rc = myolo_wino63_input_transform(x, nullptr, nullptr, MYOLO_ACT_NONE, nullptr, nullptr, V, N, Cin, stream);
        if (rc != MYOLO_OK) return rc;
    }
    W63Args a{};
    a.src = dy; a.Vn = Q; a.NR = N; a.C = Cout; a.act = MYOLO_ACT_NONE;
    w63_launch<W63_FROM_ACT, W63_TO_Q>(a, s);
    const int rc = w63_tn_and_dw(v_saved ? v_saved : V, Q, dU, dw, N, Cin, Cout, part, part_bytes, s);
    if (rc != MYOLO_OK) return rc;
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

}  // extern "C"
