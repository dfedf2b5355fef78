// Winograd F(4x4, 3x3) for the dense 3x3 / stride-1 / SAME convolutions of the mask head (myolo_mask_conv1-4,
// model.py:687-709) in fp32: 36 multiplications per 4x4 output tile and channel pair instead of 144, i.e. 4x fewer
// (3.06x at 14x14, where the 4x4 tiling covers 16x16).  Same arithmetic type as the direct kernel (fp32 operands,
// fp32 MFMA accumulation); only the association order of the sums differs.
//
//   forward      Y  = A^T [ sum_ci U(ci,co) .* V(ci) ] A        V = B^T d B  (6x6 input patch d, stride 4, pad 1)
//                                                               U = G g G^T  (3x3 filter g)
//   data grad    the same algorithm on dY with the filter rotated by 180 degrees and (ci,co) exchanged
//   weight grad  dU(ci,co) = sum_tiles V(ci) .* (A dY A^T)(co),  dW = G^T dU G
//
// Mixed tiling: where the last tile row / column of an image holds at most 2 valid outputs (14 = 4+4+4+2), that direction uses
// F(2,3) for those tiles.  Its interpolation points {0, 1, -1, inf} are a subset S = {0,1,2,5} of F(4,3)'s {0, 1, -1, 2, -2, inf}
// and the transformed filters differ only by a per-point factor rho = (4, -3, -3, 1), which is folded into the data transform
// of the reduced tiles -- so ONE set of 36 transformed filters serves every tile, a reduced tile simply has no row in the planes
// of the points it does not use, and at 14x14 an image costs 484 point-tiles instead of 576 (16 % fewer multiplications,
// 16 % smaller V / M planes), with F(2,3)'s better conditioning on those tiles.
//
// Data layout: 36 planes V[q] / M[q] of [rows_q][C]; q orders the points (i,j) by group: g0 = both indices in S (16 points,
// rows = all N*TH*TW tiles), g1 = j outside S (8 points, tiles of the reduced column absent), g2 = i outside S (8 points), g3 =
// neither (4 points).  Planes of one group are contiguous with the same row count, so the per-point products are one batched
// launch of the plain fp32 MFMA GEMM per group (gemm_kernels.hip, grid z = point).  Without a reduced direction (H, W multiples
// of 4 or remainder 3) every plane has T rows and the layout is [36][T][C] in q order.  The transforms are HBM-bound streaming
// kernels: one lane = 4 channels of one tile, every access a contiguous 16 B per lane.
#include "myolo_common.h"

__device__ __forceinline__ float4 f4(float a) { return make_float4(a, a, a, a); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float s, float4 a) { return make_float4(s * a.x, s * a.y, s * a.z, s * a.w); }
__device__ __forceinline__ float4 ldg4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void stg4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// B^T (6x6) on a 6-vector
template <typename T>
__device__ __forceinline__ void bt6(const T d[6], T t[6])
{
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = d[3] + d[4] - 4.f * (d[1] + d[2]);
    t[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    t[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    t[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
// A^T (4x6) on a 6-vector
template <typename T>
__device__ __forceinline__ void at6(const T m[6], T y[4])
{
    const T s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}
// A (6x4) on a 4-vector
template <typename T>
__device__ __forceinline__ void a4(const T d[4], T q[6])
{
    const T e = d[0] + d[2], o = d[1] + d[3], e4 = d[0] + 4.f * d[2], o4 = 2.f * d[1] + 8.f * d[3];
    q[0] = d[0];
    q[1] = e + o;
    q[2] = e - o;
    q[3] = e4 + o4;
    q[4] = e4 - o4;
    q[5] = d[3];
}
// G (6x3) on a 3-vector
__device__ __forceinline__ void g3(const float g[3], float u[6])
{
    u[0] = g[0] * 0.25f;
    u[1] = -(g[0] + g[1] + g[2]) * (1.f / 6.f);
    u[2] = -(g[0] - g[1] + g[2]) * (1.f / 6.f);
    u[3] = g[0] * (1.f / 24.f) + g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
    u[4] = g[0] * (1.f / 24.f) - g[1] * (1.f / 12.f) + g[2] * (1.f / 6.f);
    u[5] = g[2];
}
// G^T (3x6) on a 6-vector
__device__ __forceinline__ void gt6(const float d[6], float w[3])
{
    w[0] = d[0] * 0.25f - (d[1] + d[2]) * (1.f / 6.f) + (d[3] + d[4]) * (1.f / 24.f);
    w[1] = (d[2] - d[1]) * (1.f / 6.f) + (d[3] - d[4]) * (1.f / 12.f);
    w[2] = -(d[1] + d[2]) * (1.f / 6.f) + (d[3] + d[4]) * (1.f / 6.f) + d[5];
}

// ---- F(2,3) in one direction, embedded in the F(4,3) point set S = {0,1,2,5} (rho folded into the data transform) ----
// rho * B2^T on the first 4 entries of the patch (entries 4, 5 lie outside the image): t[3], t[4] are not produced
template <typename T>
__device__ __forceinline__ void bt6x(const T d[6], T t[6], bool red)
{
    if (red) {
        t[0] = 4.f * (d[0] - d[2]);
        t[1] = -3.f * (d[1] + d[2]);
        t[2] = 3.f * (d[1] - d[2]);
        t[3] = d[0] - d[0];
        t[4] = t[3];
        t[5] = d[1] - d[3];
    } else {
        bt6(d, t);
    }
}
// A2^T (2x4) on (m0, m1, m2, m5): y[2], y[3] are not produced
template <typename T>
__device__ __forceinline__ void at6x(const T m[6], T y[4], bool red)
{
    if (red) {
        y[0] = m[0] + m[1] + m[2];
        y[1] = m[1] - m[2] - m[5];
        y[2] = m[0] - m[0];
        y[3] = y[2];
    } else {
        at6(m, y);
    }
}
// A2 (4x2) on (d0, d1) -> (q0, q1, q2, q5)
template <typename T>
__device__ __forceinline__ void a4x(const T d[4], T q[6], bool red)
{
    if (red) {
        q[0] = d[0];
        q[1] = d[0] + d[1];
        q[2] = d[0] - d[1];
        q[3] = d[0] - d[0];
        q[4] = q[3];
        q[5] = q[3] - d[1];
    } else {
        a4(d, q);
    }
}

__host__ __device__ constexpr bool in_s(int i) { return i < 3 || i == 5; }
// plane index of transform point (i, j): group-major (see the header comment)
__host__ __device__ constexpr int q_of(int i, int j)
{
    const int si = i == 5 ? 3 : i, sj = j == 5 ? 3 : j;          // index inside S
    const int ni = i - 3, nj = j - 3;                             // index inside the complement {3, 4}
    return in_s(i) ? (in_s(j) ? si * 4 + sj : 16 + si * 2 + nj) : (in_s(j) ? 24 + ni * 4 + sj : 32 + ni * 2 + nj);
}
__host__ __device__ constexpr int grp_of(int i, int j) { return (in_s(i) ? 0 : 2) + (in_s(j) ? 0 : 1); }
__host__ __device__ constexpr int qfirst_of(int g) { return g == 0 ? 0 : g == 1 ? 16 : g == 2 ? 24 : 32; }

struct TileGeom {
    int H, W, TH, TW;
    int redv, redh;        // 1: the last tile row / column uses F(2,3) (it holds <= 2 valid outputs)
    long long T;           // tiles in total
    long long R[4];        // rows of a plane of group g
    long long gstart[4];   // first row of group g's first plane (rows are counted across all planes)
    long long rows;        // rows of all 36 planes together
};

// where one lane's tile lives in the planes: element offset of its row in the FIRST plane of each group (+ channel), and the
// plane stride of the group.  Element (i, j) of the tile is at  base[g] + (q_of(i,j) - qfirst_of(g)) * stride[g].
struct TileRows {
    long long base[4], stride[4];
    bool rv, rh;           // this tile is reduced vertically / horizontally
};
__device__ __forceinline__ TileRows tile_rows(const TileGeom& g, long long img, int ty, int tx, int C, int c)
{
    TileRows r;
    r.rv = g.redv && ty == g.TH - 1;
    r.rh = g.redh && tx == g.TW - 1;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int nv = g.TH - ((k & 2) ? g.redv : 0), nh = g.TW - ((k & 1) ? g.redh : 0);
        r.base[k] = (g.gstart[k] + img * (long long)(nv * nh) + (long long)ty * nh + tx) * C + c;      // unused when the tile is absent
        r.stride[k] = g.R[k] * (long long)C;
    }
    return r;
}
#define WINO_ACTIVE(tr, i, j) ((in_s(i) || !(tr).rv) && (in_s(j) || !(tr).rh))
#define WINO_ADDR(tr, i, j) ((tr).base[grp_of(i, j)] + (long long)(q_of(i, j) - qfirst_of(grp_of(i, j))) * (tr).stride[grp_of(i, j)])

__device__ __forceinline__ float4 affine_act4(float4 v, float4 sc, float4 sh, int act)
{
    v = make_float4(fmaf(v.x, sc.x, sh.x), fmaf(v.y, sc.y, sh.y), fmaf(v.z, sc.z, sh.z), fmaf(v.w, sc.w, sh.w));
    if (act == MYOLO_ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
    else if (act == MYOLO_ACT_RELU6)
        v = make_float4(fminf(fmaxf(v.x, 0.f), 6.f), fminf(fmaxf(v.y, 0.f), 6.f), fminf(fmaxf(v.z, 0.f), 6.f), fminf(fmaxf(v.w, 0.f), 6.f));
    return v;
}

// "Lazy" operand: the gradient reaching a conv through a training-mode BatchNorm + activation whose upstream gradient is
// row-sparse (non-zero only in the images listed by inv[img] >= 0, stored compactly in dyc).  Instead of materialising
//     dx = scale*dz + (ka + kb*x),   dz = dyc * [act passes](scale*x + shift)      (bn_bwd_dx_sparse_kernel)
// the transforms form it per element while loading the pre-BN tensor x: one read of x replaces a write and two reads of dx.
struct LazyBn {
    const float* dyc;        // [n_pos images][H*W][C] compact upstream gradient (NULL: plain operand)
    const int32_t* inv;      // [N] compact slot of an image or -1
    const float* scale;
    const float* shift;
    const float* ka;
    const float* kb;
    int act;
};
struct LazyBnCh { float4 sc, sh, ka, kb; };
__device__ __forceinline__ float lazy1(float x, float g, float sc, float sh, float ka, float kb, int act)
{
    const float t = fmaf(x, sc, sh);
    const float pass = act == MYOLO_ACT_RELU ? (t > 0.f ? 1.f : 0.f) : act == MYOLO_ACT_RELU6 ? ((t > 0.f && t < 6.f) ? 1.f : 0.f) : 1.f;
    return fmaf(sc, g * pass, fmaf(kb, x, ka));
}
__device__ __forceinline__ float4 lazy4(float4 x, float4 g, const LazyBnCh& k, int act)
{
    return make_float4(lazy1(x.x, g.x, k.sc.x, k.sh.x, k.ka.x, k.kb.x, act), lazy1(x.y, g.y, k.sc.y, k.sh.y, k.ka.y, k.kb.y, act),
                       lazy1(x.z, g.z, k.sc.z, k.sh.z, k.ka.z, k.kb.z, act), lazy1(x.w, g.w, k.sc.w, k.sh.w, k.ka.w, k.kb.w, act));
}

// X [N,H,W,C] -> V [36][T][C].  Optional per-channel affine + activation applied to every in-bounds pixel as it is loaded
// (BatchNorm apply + ReLU of the producing layer: the normalised activation is never written); the zero padding stays zero.
template <bool LAZY>
__global__ __launch_bounds__(256) void wino_in_kernel(const float* __restrict__ x, float* __restrict__ V, TileGeom g, int C,
                                                      const float* __restrict__ scale, const float* __restrict__ shift, int act, LazyBn lz)
{
    const int c4n = C >> 2;
    const long long total = g.T * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long t = idx / c4n;
        const int c = (int)(idx - t * c4n) * 4;
        const long long img = t / (g.TH * g.TW);
        const int rem = (int)(t - img * (g.TH * g.TW));
        const int ty = rem / g.TW, tx = rem - ty * g.TW;
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        const TileRows tr = tile_rows(g, img, ty, tx, C, c);
        const float* base = x + img * (long long)g.H * g.W * C + c;
        const float4 sc = scale ? ldg4(scale + c) : f4(1.f);
        const float4 sh = scale ? ldg4(shift + c) : f4(0.f);
        LazyBnCh lk;
        const float* gbase = nullptr;
        if (LAZY) {
            lk.sc = ldg4(lz.scale + c); lk.sh = ldg4(lz.shift + c); lk.ka = ldg4(lz.ka + c); lk.kb = ldg4(lz.kb + c);
            const int slot = lz.inv[img];
            if (slot >= 0) gbase = lz.dyc + (long long)slot * g.H * g.W * C + c;
        }
        float4 tmp[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 d[6], r[6];
            const int xx = x0 + j;
            const bool xin = (unsigned)xx < (unsigned)g.W;
            float4 gq[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) {           // all loads of the column first (memory-level parallelism), arithmetic after
                const int yy = y0 + i;
                const bool in = xin && (unsigned)yy < (unsigned)g.H;
                d[i] = in ? ldg4(base + ((long long)yy * g.W + xx) * C) : f4(0.f);
                if (LAZY) gq[i] = (in && gbase) ? ldg4(gbase + ((long long)yy * g.W + xx) * C) : f4(0.f);
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int yy = y0 + i;
                const bool in = xin && (unsigned)yy < (unsigned)g.H;
                if (LAZY) { if (in) d[i] = lazy4(d[i], gq[i], lk, lz.act); }
                else if (scale && in) d[i] = affine_act4(d[i], sc, sh, act);
            }
            bt6x(d, r, tr.rv);
#pragma unroll
            for (int i = 0; i < 6; ++i) tmp[i][j] = r[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float4 r[6];
            bt6x(tmp[i], r, tr.rh);
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (WINO_ACTIVE(tr, i, j)) stg4(V + WINO_ADDR(tr, i, j), r[j]);
        }
    }
}

// ROIAlign (tf.image.crop_and_resize, model.py:385-387) fused into the input transform of the conv that consumes it: the
// [boxes, crop, crop, C] tensor is never written.  One lane = 4 channels of one 4x4 output tile of one box; it forms the 6
// row and 6 column sample coordinates of its patch once (same float expressions as crop_fwd_kernel), bilinearly samples the
// 36 patch positions from the feature map (L2-resident: a few MB per image) and transforms them.
struct CropAxis { int lo, hi; float w; bool ok; };
__device__ __forceinline__ CropAxis crop_axis(float b0, float b1, int size, int crop, int idx)
{
    CropAxis a;
    a.ok = (unsigned)idx < (unsigned)crop;          // outside the crop = the conv's zero padding
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

__global__ __launch_bounds__(256) void wino_in_crop_kernel(const float* __restrict__ feat, const float* __restrict__ boxes,
                                                           const int32_t* __restrict__ bind, float* __restrict__ V, TileGeom g, int C,
                                                           int FH, int FW)
{
    const int c4n = C >> 2;
    const long long total = g.T * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long t = idx / c4n;
        const int c = (int)(idx - t * c4n) * 4;
        const long long roi = t / (g.TH * g.TW);
        const int rem = (int)(t - roi * (g.TH * g.TW));
        const int ty = rem / g.TW, tx = rem - ty * g.TW;
        const TileRows tr = tile_rows(g, roi, ty, tx, C, c);
        const float4 bx = ldg4(boxes + roi * 4);               // y1, x1, y2, x2
        const float* base = feat + (long long)bind[roi] * FH * FW * C + c;
        CropAxis ay[6], ax[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            ay[i] = crop_axis(bx.x, bx.z, FH, g.H, 4 * ty - 1 + i);
            ax[i] = crop_axis(bx.y, bx.w, FW, g.W, 4 * tx - 1 + i);
        }
        float4 tmp[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 d[6], r[6];
#pragma unroll
            for (int h3 = 0; h3 < 6; h3 += 3) {     // 3 rows x 4 corners in flight at a time (register budget)
                float4 tl[3], tr[3], bl[3], br[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {       // padding / extrapolated positions read pixel (0,0) and are zeroed below
                    const int i = h3 + k;
                    tl[k] = ldg4(base + ((long long)ay[i].lo * FW + ax[j].lo) * C);
                    tr[k] = ldg4(base + ((long long)ay[i].lo * FW + ax[j].hi) * C);
                    bl[k] = ldg4(base + ((long long)ay[i].hi * FW + ax[j].lo) * C);
                    br[k] = ldg4(base + ((long long)ay[i].hi * FW + ax[j].hi) * C);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int i = h3 + k;
                    const float wx = ax[j].w, wy = ay[i].w;
                    float4 o;
                    float top, bot;
                    top = tl[k].x + (tr[k].x - tl[k].x) * wx; bot = bl[k].x + (br[k].x - bl[k].x) * wx; o.x = top + (bot - top) * wy;
                    top = tl[k].y + (tr[k].y - tl[k].y) * wx; bot = bl[k].y + (br[k].y - bl[k].y) * wx; o.y = top + (bot - top) * wy;
                    top = tl[k].z + (tr[k].z - tl[k].z) * wx; bot = bl[k].z + (br[k].z - bl[k].z) * wx; o.z = top + (bot - top) * wy;
                    top = tl[k].w + (tr[k].w - tl[k].w) * wx; bot = bl[k].w + (br[k].w - bl[k].w) * wx; o.w = top + (bot - top) * wy;
                    d[i] = (ay[i].ok && ax[j].ok) ? o : f4(0.f);
                }
            }
            bt6x(d, r, tr.rv);
#pragma unroll
            for (int i = 0; i < 6; ++i) tmp[i][j] = r[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float4 r[6];
            bt6x(tmp[i], r, tr.rh);
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (WINO_ACTIVE(tr, i, j)) stg4(V + WINO_ADDR(tr, i, j), r[j]);
        }
    }
}

// M [36][T][C] -> Y [N,H,W,C], + bias, optional per-channel affine (folded frozen BN), activation.
// stats != NULL (training-mode BatchNorm behind this conv): every workgroup also leaves the per-channel sum and sum of
// squares of the values it wrote in stats[blockIdx.x][2*C] (double); needs (gridDim.x * 256) % (C/4) == 0 so that a thread
// keeps its channels across the grid-stride loop.
__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ M, float* __restrict__ y, const float* __restrict__ bias,
                                                       const float* __restrict__ scale, const float* __restrict__ shift, TileGeom g, int C,
                                                       int act, double* __restrict__ stats)
{
    float4 s1 = f4(0.f), s2 = f4(0.f);
    const int c4n = C >> 2;
    const long long total = g.T * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long t = idx / c4n;
        const int c = (int)(idx - t * c4n) * 4;
        const long long img = t / (g.TH * g.TW);
        const int rem = (int)(t - img * (g.TH * g.TW));
        const int ty = rem / g.TW, tx = rem - ty * g.TW;
        const TileRows tr = tile_rows(g, img, ty, tx, C, c);
        float4 tmp[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 m[6], r[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = WINO_ACTIVE(tr, i, j) ? ldg4(M + WINO_ADDR(tr, i, j)) : f4(0.f);
            at6x(m, r, tr.rv);
#pragma unroll
            for (int i = 0; i < 4; ++i) tmp[i][j] = r[i];
        }
        const float4 b = bias ? ldg4(bias + c) : f4(0.f);
        const float4 sc = scale ? ldg4(scale + c) : f4(1.f);
        const float4 sh = scale ? ldg4(shift + c) : f4(0.f);
        float* obase = y + img * (long long)g.H * g.W * C + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 r[4];
            at6x(tmp[i], r, tr.rh);
            const int yy = 4 * ty + i;
            if (yy >= g.H) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = 4 * tx + j;
                if (xx >= g.W) continue;
                float4 v = r[j] + b;
                if (scale) v = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
                if (act == MYOLO_ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                else if (act == MYOLO_ACT_RELU6)
                    v = make_float4(fminf(fmaxf(v.x, 0.f), 6.f), fminf(fmaxf(v.y, 0.f), 6.f), fminf(fmaxf(v.z, 0.f), 6.f),
                                    fminf(fmaxf(v.w, 0.f), 6.f));
                stg4(obase + ((long long)yy * g.W + xx) * C, v);
                if (stats) {
                    s1 = s1 + v;
                    s2 = make_float4(fmaf(v.x, v.x, s2.x), fmaf(v.y, v.y, s2.y), fmaf(v.z, v.z, s2.z), fmaf(v.w, v.w, s2.w));
                }
            }
        }
    }
    if (stats) {       // threads t, t + c4n, t + 2*c4n, ... of the workgroup hold the same channels
        __shared__ double red[2][256][4];
        const double a1[4] = {s1.x, s1.y, s1.z, s1.w}, a2[4] = {s2.x, s2.y, s2.z, s2.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) { red[0][threadIdx.x][k] = a1[k]; red[1][threadIdx.x][k] = a2[k]; }
        __syncthreads();
        if ((int)threadIdx.x < c4n) {
            double* dst = stats + (long long)blockIdx.x * 2 * C;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                double t1 = 0, t2 = 0;
                for (int j = threadIdx.x; j < 256; j += c4n) { t1 += red[0][j][k]; t2 += red[1][j][k]; }
                dst[threadIdx.x * 4 + k] = t1;
                dst[C + threadIdx.x * 4 + k] = t2;
            }
        }
    }
}

// dY [N,H,W,C] -> Q [36][T][C] = A dY_tile A^T (weight gradient)
template <bool LAZY>
__global__ __launch_bounds__(256) void wino_dy_kernel(const float* __restrict__ dy, float* __restrict__ Q, TileGeom g, int C, LazyBn lz)
{
    const int c4n = C >> 2;
    const long long total = g.T * c4n;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long t = idx / c4n;
        const int c = (int)(idx - t * c4n) * 4;
        const long long img = t / (g.TH * g.TW);
        const int rem = (int)(t - img * (g.TH * g.TW));
        const int ty = rem / g.TW, tx = rem - ty * g.TW;
        const TileRows tr = tile_rows(g, img, ty, tx, C, c);
        const float* base = dy + img * (long long)g.H * g.W * C + c;
        LazyBnCh lk;
        const float* gbase = nullptr;
        if (LAZY) {
            lk.sc = ldg4(lz.scale + c); lk.sh = ldg4(lz.shift + c); lk.ka = ldg4(lz.ka + c); lk.kb = ldg4(lz.kb + c);
            const int slot = lz.inv[img];
            if (slot >= 0) gbase = lz.dyc + (long long)slot * g.H * g.W * C + c;
        }
        float4 tmp[6][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float4 d[4], r[6];
            const int xx = 4 * tx + j;
            float4 gq[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int yy = 4 * ty + i;
                const bool in = xx < g.W && yy < g.H;
                d[i] = in ? ldg4(base + ((long long)yy * g.W + xx) * C) : f4(0.f);
                if (LAZY) gq[i] = (in && gbase) ? ldg4(gbase + ((long long)yy * g.W + xx) * C) : f4(0.f);
            }
            if (LAZY) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (xx < g.W && 4 * ty + i < g.H) d[i] = lazy4(d[i], gq[i], lk, lz.act);
            }
            a4x(d, r, tr.rv);
#pragma unroll
            for (int i = 0; i < 6; ++i) tmp[i][j] = r[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float4 r[6];
            a4x(tmp[i], r, tr.rh);
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (WINO_ACTIVE(tr, i, j)) stg4(Q + WINO_ADDR(tr, i, j), r[j]);
        }
    }
}

// w [3,3,Ci,Co] -> U [36][Ci][Co]   (flip = 0), planes in q order (q_of(i, j))
//                  U'[36][Co][Ci] of the 180-degree rotated filter with (ci,co) exchanged (flip = 1: data gradient)
// bt = 1: each plane transposed ([N][K] for the multiply's K x N operand: wino_mm_kernel reads both operands k-contiguous)
// bt = 2: the transposed plane split into three bf16 pieces per value (round-to-nearest; exact: 3 x 8 significand bits) in the
//         MFMA operand order of wino_mm_x6_kernel: [plane][k / 16][n / 32][piece][(k / 8) % 2][n % 32][k % 8] bf16
__global__ __launch_bounds__(256) void wino_w_kernel(const float* __restrict__ w, float* __restrict__ U, int Ci, int Co, int flip, int bt)
{
    // block = a 16 x 16 patch of (ci, co); reads run along co (w's contiguous axis); planes whose contiguous axis is ci are
    // written through an LDS transpose so both sides move 64-byte segments
    __shared__ float tr[36][16][17];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int ci0 = blockIdx.y * 16, co0 = blockIdx.x * 16;
    const int ci = ci0 + ty, co = co0 + tx;
    const bool ok = ci < Ci && co < Co;
    const bool x6 = bt == 2;
    if (x6) bt = 1;
    const bool ci_major = (flip == bt);                   // plane [ci][co]; otherwise [co][ci]
    float g[3][3], tmp[6][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            g[ky][kx] = ok ? w[((flip ? (2 - ky) * 3 + (2 - kx) : ky * 3 + kx) * Ci + ci) * (long long)Co + co] : 0.f;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const float col[3] = {g[0][kx], g[1][kx], g[2][kx]};
        float u[6];
        g3(col, u);
#pragma unroll
        for (int i = 0; i < 6; ++i) tmp[i][kx] = u[i];
    }
    const long long plane = (long long)Ci * Co;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        float u[6];
        g3(tmp[i], u);
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            if (x6) {
                if (!ok) continue;
                const int k = flip ? co : ci, n = flip ? ci : co, N = flip ? Ci : Co, nkc = (flip ? Co : Ci) >> 4;
                __bf16* rec = reinterpret_cast<__bf16*>(U) + ((((long long)q_of(i, j) * nkc + (k >> 4)) * (N >> 5) + (n >> 5)) * 6 + ((k >> 3) & 1)) * 256 + (n & 31) * 8 + (k & 7);
                const __bf16 p1 = (__bf16)u[j];
                const float r1 = u[j] - (float)p1;
                const __bf16 p2 = (__bf16)r1;
                const __bf16 p3 = (__bf16)(r1 - (float)p2);
                rec[0] = p1; rec[512] = p2; rec[1024] = p3;
            }
            else if (ci_major) { if (ok) U[q_of(i, j) * plane + (long long)ci * Co + co] = u[j]; }
            else tr[q_of(i, j)][ty][tx] = u[j];
        }
    }
    if (ci_major || x6) return;
    __syncthreads();
    const int oci = ci0 + tx, oco = co0 + ty;             // thread (ty, tx) now owns (co = co0 + ty, ci = ci0 + tx)
    if (oci < Ci && oco < Co) {
#pragma unroll
        for (int q = 0; q < 36; ++q) U[q * plane + (long long)oco * Ci + oci] = tr[q][tx][ty];
    }
}

// dU [36][Ci][Co] -> dw [3,3,Ci,Co] = G^T dU G
__global__ __launch_bounds__(256) void wino_dw_kernel(const float* __restrict__ dU, float* __restrict__ dw, int Ci, int Co)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Ci * Co) return;
    const long long plane = (long long)Ci * Co;
    float tmp[3][6];
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        float col[6], r[3];
#pragma unroll
        for (int i = 0; i < 6; ++i) col[i] = dU[q_of(i, j) * plane + idx];
        gt6(col, r);
#pragma unroll
        for (int k = 0; k < 3; ++k) tmp[k][j] = r[k];
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        float r[3];
        gt6(tmp[ky], r);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) dw[(ky * 3 + kx) * plane + idx] = r[kx];
    }
}

// Layer boundary conv_i -> conv_{i+1} in one pass per image (ROI): M_i [36][T][C] -> output transform (+bias, folded-BN
// affine, activation) -> the H x W x 32-channel activation tile in LDS -> input transform -> V_{i+1} [36][T][C].
// The activation is written to y only where it is needed later (flags[img] != 0; flags == NULL: always; y == NULL: never),
// so between two Winograd convs it neither makes the round trip through HBM nor, for most images, exists at all.
// Workgroup = (image, 32-channel slice); thread = (tile, 4 channels); needs TH*TW*8 <= 256 threads and H*W*128 B of LDS.
#define WOI_CS 32
__global__ __launch_bounds__(256) void wino_out_in_kernel(const float* __restrict__ M, float* __restrict__ Vn, float* __restrict__ y,
                                                          const int32_t* __restrict__ flags, const float* __restrict__ bias,
                                                          const float* __restrict__ scale, const float* __restrict__ shift, TileGeom g,
                                                          int C, int act, int keep_pre)
{
    // keep_pre: what is written to y is the value BEFORE the affine + activation (the conv's pre-BatchNorm output), see wino63_kernels.hip W63Args::ypre
    extern __shared__ __attribute__((aligned(16))) float ysm[];       // [H][W][WOI_CS]
    const int img = blockIdx.x;
    const int c = blockIdx.y * WOI_CS + (threadIdx.x & 7) * 4;
    const int tl = threadIdx.x >> 3;                                   // tile inside the image
    const int ty = tl / g.TW, tx = tl - ty * g.TW;
    const TileRows tr = tile_rows(g, img, ty, tx, C, c);
    const bool wr = y && (!flags || flags[img] != 0);
    {
        float4 tmp[4][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 m[6], r[4];
#pragma unroll
            for (int i = 0; i < 6; ++i) m[i] = WINO_ACTIVE(tr, i, j) ? ldg4(M + WINO_ADDR(tr, i, j)) : f4(0.f);
            at6x(m, r, tr.rv);
#pragma unroll
            for (int i = 0; i < 4; ++i) tmp[i][j] = r[i];
        }
        const float4 b = bias ? ldg4(bias + c) : f4(0.f);
        const float4 sc = scale ? ldg4(scale + c) : f4(1.f);
        const float4 sh = scale ? ldg4(shift + c) : f4(0.f);
        float* obase = y + (long long)img * g.H * g.W * C + c;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float4 r[4];
            at6x(tmp[i], r, tr.rh);
            const int yy = 4 * ty + i;
            if (yy >= g.H) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int xx = 4 * tx + j;
                if (xx >= g.W) continue;
                float4 v = r[j] + b;
                const float4 pre = v;
                if (scale) v = make_float4(v.x * sc.x + sh.x, v.y * sc.y + sh.y, v.z * sc.z + sh.z, v.w * sc.w + sh.w);
                if (act == MYOLO_ACT_RELU) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
                else if (act == MYOLO_ACT_RELU6)
                    v = make_float4(fminf(fmaxf(v.x, 0.f), 6.f), fminf(fmaxf(v.y, 0.f), 6.f), fminf(fmaxf(v.z, 0.f), 6.f),
                                    fminf(fmaxf(v.w, 0.f), 6.f));
                *reinterpret_cast<float4*>(&ysm[(yy * g.W + xx) * WOI_CS + (threadIdx.x & 7) * 4]) = v;
                if (wr) stg4(obase + ((long long)yy * g.W + xx) * C, keep_pre ? pre : v);
            }
        }
    }
    __syncthreads();
    {
        const int y0 = 4 * ty - 1, x0 = 4 * tx - 1;
        float4 tmp[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float4 d[6], r[6];
            const int xx = x0 + j;
            const bool xin = (unsigned)xx < (unsigned)g.W;
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                const int yy = y0 + i;
                d[i] = (xin && (unsigned)yy < (unsigned)g.H)
                           ? *reinterpret_cast<const float4*>(&ysm[(yy * g.W + xx) * WOI_CS + (threadIdx.x & 7) * 4]) : f4(0.f);
            }
            bt6x(d, r, tr.rv);
#pragma unroll
            for (int i = 0; i < 6; ++i) tmp[i][j] = r[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float4 r[6];
            bt6x(tmp[i], r, tr.rh);
#pragma unroll
            for (int j = 0; j < 6; ++j)
                if (WINO_ACTIVE(tr, i, j)) stg4(Vn + WINO_ADDR(tr, i, j), r[j]);
        }
    }
}

static TileGeom geom(int N, int H, int W)
{
    TileGeom g;
    g.H = H; g.W = W; g.TH = (H + 3) / 4; g.TW = (W + 3) / 4;
    // mixed tiling: F(2,3) for the last tile row / column when it holds <= 2 valid outputs (and is not the only one)
    const bool mixed = !g_myolo_opt.wino_no_mixed;
    g.redv = (mixed && g.TH > 1 && H - 4 * (g.TH - 1) <= 2) ? 1 : 0;
    g.redh = (mixed && g.TW > 1 && W - 4 * (g.TW - 1) <= 2) ? 1 : 0;
    g.T = (long long)N * g.TH * g.TW;
    long long at = 0;
    for (int k = 0; k < 4; ++k) {
        const int nv = g.TH - ((k & 2) ? g.redv : 0), nh = g.TW - ((k & 1) ? g.redh : 0);
        g.R[k] = (long long)N * nv * nh;
        g.gstart[k] = at;
        at += g.R[k] * (k == 0 ? 16 : k == 3 ? 4 : 8);
    }
    g.rows = at;
    return g;
}
static unsigned ew_grid(long long total)
{
    long long b = (total + 255) / 256;
    if (b > (1 << 20)) b = 1 << 20;
    return (unsigned)(b < 1 ? 1 : b);
}
static size_t plane_bytes(const TileGeom& g, int C) { return align256((size_t)g.rows * C * sizeof(float)); }
// transformed filters: 36 planes of Ci*Co values, 6 bytes each when stored as three bf16 pieces (always sized for that)
static size_t u_bytes(int Ci, int Co) { return align256((size_t)36 * Ci * Co * 6); }
// layout of the transformed filters for the multiply of a (K, N) layer: 0 = [K][N], 1 = [N][K], 2 = split bf16 records
static int wino_u_layout(int K, int N) { return myolo_gemm_nt_batched_x6(K, N) ? 2 : (myolo_gemm_nt_batched_ok(K, N) ? 1 : 0); }

// runs of consecutive groups whose planes have the same row count: one batched GEMM launch each
struct GroupRun { int q0, nq; long long rows, row0; };
static int group_runs(const TileGeom& g, GroupRun out[4])
{
    static const int cnt[4] = {16, 8, 8, 4};
    int n = 0, q = 0;
    for (int k = 0; k < 4; ++k) {
        if (n && out[n - 1].rows == g.R[k]) out[n - 1].nq += cnt[k];
        else { out[n].q0 = q; out[n].nq = cnt[k]; out[n].rows = g.R[k]; out[n].row0 = g.gstart[k]; ++n; }
        q += cnt[k];
    }
    return n;
}

// M[q] = V[q] * U[q] for the 36 points (U planes [Cin][Cout], or transposed when myolo_gemm_nt_batched_ok(Cin, Cout): wino_w_kernel
// and this function take the same decision from the same two numbers)
static int wino_multiply_all(const float* V, const float* U, float* M, const TileGeom& g, int Cin, int Cout, hipStream_t s)
{
    GroupRun runs[4];
    const int n = group_runs(g, runs);
    if (myolo_gemm_nt_batched_ok(Cin, Cout)) {          // all runs in one launch (csrc/wino_mm.hip)
        long long rows[4], ao[4], bo[4], co[4];
        int nq[4];
        for (int k = 0; k < n; ++k) {
            rows[k] = runs[k].rows; nq[k] = runs[k].nq;
            ao[k] = runs[k].row0 * Cin; bo[k] = (long long)runs[k].q0 * Cin * Cout; co[k] = runs[k].row0 * Cout;
        }
        return myolo_gemm_nt_batched_runs(V, U, M, n, rows, ao, bo, co, nq, Cin, Cout, s);
    }
    for (int k = 0; k < n; ++k) {
        if (runs[k].rows <= 0) continue;
        const int rc = myolo_gemm_nn_batched(V + runs[k].row0 * Cin, U + (long long)runs[k].q0 * Cin * Cout, M + runs[k].row0 * Cout,
                                             runs[k].rows, Cin, Cout, runs[k].nq, s);
        if (rc != MYOLO_OK) return rc;
    }
    return MYOLO_OK;
}
static size_t wino_tn_ws_bytes(const TileGeom& g, int Cin, int Cout)
{
    GroupRun runs[4];
    const int n = group_runs(g, runs);
    size_t m = 0;
    for (int k = 0; k < n; ++k) {
        const size_t b = myolo_gemm_tn_batched_ws_bytes(runs[k].rows, Cin, Cout, runs[k].nq);
        if (b > m) m = b;
    }
    if (myolo_gemm_tn_x6_ok(Cin, Cout)) {
        long long rows[4];
        int nq[4];
        for (int k = 0; k < n; ++k) { rows[k] = runs[k].rows; nq[k] = runs[k].nq; }
        const size_t b = myolo_gemm_tn_x6_ws_bytes(n, rows, nq, Cin, Cout);
        if (b > m) m = b;
    }
    return m;
}
// dU[q] = V[q]^T Q[q]
static int wino_tn_all(const float* V, const float* Q, float* dU, const TileGeom& g, int Cin, int Cout, void* part, size_t part_bytes,
                       hipStream_t s)
{
    GroupRun runs[4];
    const int n = group_runs(g, runs);
    if (myolo_gemm_tn_x6_ok(Cin, Cout) && !(g_myolo_opt.tune0 & 32768)) {
        // FP32_MATMUL = "bf16x6": all 36 planes in one launch of wino_tn_x6_kernel, as the F(6,3) chain does (feature_map's weight gradient, 512 -> 256
        // channels on 28 x 28: 166-185 us in the step on the fp32 matrix pipe, one gemm_tn_fast launch per run of planes)
        long long rows[4], ao[4], bo[4];
        int nq[4];
        bool all = true;
        for (int k = 0; k < n; ++k) {
            rows[k] = runs[k].rows; nq[k] = runs[k].nq; ao[k] = runs[k].row0 * Cin; bo[k] = runs[k].row0 * Cout;
            if (runs[k].rows <= 0) all = false;                // (an empty run would shift the plane numbering of myolo_gemm_tn_x6_runs)
        }
        if (all && part && myolo_gemm_tn_x6_ws_bytes(n, rows, nq, Cin, Cout) <= part_bytes)
            return myolo_gemm_tn_x6_runs(V, Q, dU, n, rows, ao, bo, nq, Cin, Cout, part, part_bytes, s);
    }
    for (int k = 0; k < n; ++k) {
        const int rc = myolo_gemm_tn_batched(V + runs[k].row0 * Cin, Q + runs[k].row0 * Cout, dU + (long long)runs[k].q0 * Cin * Cout,
                                             runs[k].rows, Cin, Cout, runs[k].nq, part, part_bytes, s);
        if (rc != MYOLO_OK) return rc;
    }
    return MYOLO_OK;
}

extern "C" {

size_t myolo_conv3x3_wino_ws_bytes(int N, int H, int W, int Cin, int Cout, int which)
{
    const TileGeom g = geom(N, H, W);
    switch (which) {
        case 0: return u_bytes(Cin, Cout) + plane_bytes(g, Cin) + plane_bytes(g, Cout);                   // fwd: U, V, M
        case 1: return u_bytes(Cin, Cout) + plane_bytes(g, Cout) + plane_bytes(g, Cin);                   // bwd data: U', V(dy), M
        default: return u_bytes(Cin, Cout) + plane_bytes(g, Cin) + plane_bytes(g, Cout) +                 // bwd weight: dU, V, Q, partials
                        align256(wino_tn_ws_bytes(g, Cin, Cout));
    }
}

/* elements of the 36 V (or M) planes of an [N,H,W,C] tensor: <= 36 * N*ceil(H/4)*ceil(W/4) * C (mixed tiling drops the rows of
 * reduced tiles from the planes of the points they do not use) */
size_t myolo_wino_plane_elems(int N, int H, int W, int C) { return (size_t)geom(N, H, W).rows * (size_t)C; }
/* floats to allocate for the transformed filters U of myolo_wino_weight_transform (any layout the multiply may choose) */
size_t myolo_wino_u_elems(int Cin, int Cout) { return u_bytes(Cin, Cout) / sizeof(float); }

// transformed filters of a layer for the F(4,3) tiling: into `scratch`, or the copy prepared for this step (prepared-weights registry)
static const float* w43_filters(const float* w, float* scratch, int Cin, int Cout, int flip, hipStream_t s)
{
    const int layout = wino_u_layout(flip ? Cout : Cin, flip ? Cin : Cout);
    return (const float*)myolo_wprep_resolve(w, WP_WINO43_U, Cin, Cout, flip * 16 + layout, u_bytes(Cin, Cout), scratch, s, [=](void* d, hipStream_t st) {
        hipLaunchKernelGGL(wino_w_kernel, dim3((Cout + 15) / 16, (Cin + 15) / 16), dim3(256), 0, st, w, (float*)d, Cin, Cout, flip, layout);
    });
}

/* ---- the four stages on their own (the engine times the multiply stage for bench.py's roofline) ---- */
int myolo_wino_weight_transform(const float* w, float* U, int Cin, int Cout, int flip, void* stream)
{
    MYOLO_REQUIRE(w && U && Cin > 0 && Cout > 0, "wino_weight_transform: bad arguments");
    hipLaunchKernelGGL(wino_w_kernel, dim3((Cout + 15) / 16, (Cin + 15) / 16), dim3(256), 0, (hipStream_t)stream, w, U, Cin, Cout, flip,
                       wino_u_layout(flip ? Cout : Cin, flip ? Cin : Cout));
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_wino_input_transform(const float* x, float* V, int N, int H, int W, int C, void* stream)
{
    return myolo_wino_input_transform_affine(x, nullptr, nullptr, MYOLO_ACT_NONE, V, N, H, W, C, stream);
}

int myolo_wino_input_transform_affine(const float* x, const float* scale, const float* shift, int act, float* V, int N, int H, int W,
                                      int C, void* stream)
{
    MYOLO_REQUIRE(x && V && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "wino_input_transform: bad arguments (C %% 4 == 0)");
    MYOLO_REQUIRE(!scale == !shift, "wino_input_transform: scale and shift go together");
    const TileGeom g = geom(N, H, W);
    hipLaunchKernelGGL(wino_in_kernel<false>, dim3(ew_grid(g.T * (C / 4))), dim3(256), 0, (hipStream_t)stream, x, V, g, C, scale, shift, act, LazyBn{});
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_wino_input_transform_roialign(const float* feature, const float* boxes, const int32_t* box_ind, float* V, int B, int FH, int FW,
                                        int C, int nb, int crop_h, int crop_w, void* stream)
{
    MYOLO_REQUIRE(feature && boxes && box_ind && V && B > 0 && FH > 0 && FW > 0 && nb > 0 && crop_h > 0 && crop_w > 0 && C > 0 && (C & 3) == 0,
                  "wino_input_transform_roialign: bad arguments (C %% 4 == 0)");
    const TileGeom g = geom(nb, crop_h, crop_w);
    hipLaunchKernelGGL(wino_in_crop_kernel, dim3(ew_grid(g.T * (C / 4))), dim3(256), 0, (hipStream_t)stream, feature, boxes, box_ind, V, g,
                       C, FH, FW);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_wino_multiply(const float* V, const float* U, float* M, int N, int H, int W, int Cin, int Cout, void* stream)
{
    MYOLO_REQUIRE(V && U && M && N > 0 && H > 0 && W > 0, "wino_multiply: bad arguments");
    const TileGeom g = geom(N, H, W);
    const int rc = wino_multiply_all(V, U, M, g, Cin, Cout, (hipStream_t)stream);
    if (rc != MYOLO_OK) return rc;
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* weight_transform (flip 0) + multiply in one call: U_scratch (myolo_wino_u_elems floats) is written only when the filters are not already prepared */
int myolo_wino_multiply_w(const float* V, const float* w, float* U_scratch, float* M, int N, int H, int W, int Cin, int Cout, void* stream)
{
    MYOLO_REQUIRE(V && w && U_scratch && M && N > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "wino_multiply_w: bad arguments");
    const float* U = w43_filters(w, U_scratch, Cin, Cout, 0, (hipStream_t)stream);
    return myolo_wino_multiply(V, U, M, N, H, W, Cin, Cout, stream);
}

int myolo_wino_output_transform(const float* M, const float* bias, const float* scale, const float* shift, float* y,
                                int N, int H, int W, int C, int act, void* stream)
{
    MYOLO_REQUIRE(M && y && N > 0 && H > 0 && W > 0 && C > 0 && (C & 3) == 0, "wino_output_transform: bad arguments (C %% 4 == 0)");
    MYOLO_REQUIRE(!scale == !shift, "wino_output_transform: scale and shift go together");
    const TileGeom g = geom(N, H, W);
    hipLaunchKernelGGL(wino_out_kernel, dim3(ew_grid(g.T * (C / 4))), dim3(256), 0, (hipStream_t)stream, M, y, bias, scale, shift, g, C, act, (double*)nullptr);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

#define WOUT_STATS_BLOCKS 2048
size_t myolo_wino_output_transform_bn_ws_bytes(int C) { return align256((size_t)WOUT_STATS_BLOCKS * 2 * C * sizeof(double)) + 2 * C * sizeof(double); }

/* conv + bias -> y, and the training-mode BatchNorm statistics of y in the same pass (model.py:690: bn1 of the mask head
 * has no training= argument): mean / var / folded scale, shift and the moving averages, exactly as myolo_bn_stats. */
int myolo_wino_output_transform_bn_stats(const float* M, const float* bias, float* y, int N, int H, int W, int C, const float* gamma,
                                         const float* beta, float* mean, float* var, float* scale, float* shift, float* moving_mean,
                                         float* moving_var, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(M && y && gamma && beta && mean && var && scale && shift && N > 0 && H > 0 && W > 0, "wino_output_transform_bn_stats: bad arguments");
    MYOLO_REQUIRE(C > 0 && (C & 3) == 0 && 256 % (C / 4) == 0, "wino_output_transform_bn_stats: C/4 must divide 256 (got C=%d)", C);
    MYOLO_NEED_WS(myolo_wino_output_transform_bn_ws_bytes(C));
    const TileGeom g = geom(N, H, W);
    hipStream_t s = (hipStream_t)stream;
    unsigned blocks = ew_grid(g.T * (C / 4));
    if (blocks > WOUT_STATS_BLOCKS) blocks = WOUT_STATS_BLOCKS;
    double* part = (double*)ws;
    double* tot = (double*)((char*)ws + align256((size_t)WOUT_STATS_BLOCKS * 2 * C * sizeof(double)));
    hipLaunchKernelGGL(wino_out_kernel, dim3(blocks), dim3(256), 0, s, M, y, bias, (const float*)nullptr, (const float*)nullptr, g, C,
                       MYOLO_ACT_NONE, part);
    myolo_bn_stats_from_partials(part, tot, (int)blocks, C, (double)N * H * W, gamma, beta, mean, var, scale, shift, moving_mean,
                                 moving_var, s);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_wino_output_input_transform(const float* M, const float* bias, const float* scale, const float* shift, float* y,
                                      const int32_t* flags, float* V_next, int N, int H, int W, int C, int act, void* stream)
{
    MYOLO_REQUIRE(M && V_next && N > 0 && H > 0 && W > 0 && C > 0, "wino_output_input_transform: bad arguments");
    MYOLO_REQUIRE(!scale == !shift, "wino_output_input_transform: scale and shift go together");
    const TileGeom g = geom(N, H, W);
    MYOLO_REQUIRE((C % WOI_CS) == 0 && g.TH * g.TW * 8 <= 256 && (size_t)H * W * WOI_CS * sizeof(float) <= 65536,
                  "wino_output_input_transform: needs C %% 32 == 0 and at most 32 tiles per image (got C=%d, %dx%d)", C, H, W);
    hipLaunchKernelGGL(wino_out_in_kernel, dim3(N, C / WOI_CS), dim3(g.TH * g.TW * 8), (size_t)H * W * WOI_CS * sizeof(float),
                       (hipStream_t)stream, M, V_next, y, flags, bias, scale, shift, g, C, act, 0);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

/* the same boundary with the conv's PRE-BatchNorm output (A^T m A + bias) written to ypre where flags[img] != 0 (NULL: everywhere) instead of the
 * activation: what the exact-sparsity backward reads bn2-4's backward off (csrc/wino63_kernels.hip: myolo_wino63_output_input_transform_keep_pre) */
int myolo_wino_output_input_transform_keep_pre(const float* M, const float* bias, const float* scale, const float* shift, float* ypre,
                                               const int32_t* flags, float* V_next, int N, int H, int W, int C, int act, void* stream)
{
    MYOLO_REQUIRE(M && V_next && ypre && N > 0 && H > 0 && W > 0 && C > 0, "wino_output_input_transform_keep_pre: bad arguments");
    MYOLO_REQUIRE(!scale == !shift, "wino_output_input_transform_keep_pre: scale and shift go together");
    const TileGeom g = geom(N, H, W);
    MYOLO_REQUIRE((C % WOI_CS) == 0 && g.TH * g.TW * 8 <= 256 && (size_t)H * W * WOI_CS * sizeof(float) <= 65536,
                  "wino_output_input_transform_keep_pre: needs C %% 32 == 0 and at most 32 tiles per image (got C=%d, %dx%d)", C, H, W);
    hipLaunchKernelGGL(wino_out_in_kernel, dim3(N, C / WOI_CS), dim3(g.TH * g.TW * 8), (size_t)H * W * WOI_CS * sizeof(float),
                       (hipStream_t)stream, M, V_next, ypre, flags, bias, scale, shift, g, C, act, 1);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_conv3x3_wino_fwd(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y,
                           int N, int H, int W, int Cin, int Cout, int act, float* v_keep, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && w && y && N > 0 && H > 0 && W > 0, "conv3x3_wino_fwd: bad arguments");
    MYOLO_REQUIRE((Cin % 16) == 0 && (Cout & 3) == 0, "conv3x3_wino_fwd: needs Cin %% 16 == 0 and Cout %% 4 == 0 (got %d, %d)", Cin, Cout);
    MYOLO_REQUIRE(!scale == !shift, "conv3x3_wino_fwd: scale and shift go together");
    const TileGeom g = geom(N, H, W);
    const size_t ub = u_bytes(Cin, Cout), vb = v_keep ? 0 : plane_bytes(g, Cin), mb = plane_bytes(g, Cout);
    MYOLO_NEED_WS(ub + vb + mb);
    hipStream_t s = (hipStream_t)stream;
    float* V = v_keep ? v_keep : (float*)((char*)ws + ub);
    float* Mp = (float*)((char*)ws + ub + vb);
    const float* U = w43_filters(w, (float*)ws, Cin, Cout, 0, s);
    hipLaunchKernelGGL(wino_in_kernel<false>, dim3(ew_grid(g.T * (Cin / 4))), dim3(256), 0, s, x, V, g, Cin, (const float*)nullptr, (const float*)nullptr, 0, LazyBn{});
    const int rc = wino_multiply_all(V, U, Mp, g, Cin, Cout, s);
    if (rc != MYOLO_OK) return rc;
    hipLaunchKernelGGL(wino_out_kernel, dim3(ew_grid(g.T * (Cout / 4))), dim3(256), 0, s, Mp, y, bias, scale, shift, g, Cout, act, (double*)nullptr);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

}  // extern "C"

static int wino_bwd_data_impl(const float* dy, const LazyBn* lazy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout,
                              void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(dy && w && dx && N > 0 && H > 0 && W > 0, "conv3x3_wino_bwd_data: bad arguments");
    MYOLO_REQUIRE((Cout % 16) == 0 && (Cin & 3) == 0, "conv3x3_wino_bwd_data: needs Cout %% 16 == 0 and Cin %% 4 == 0 (got %d, %d)", Cout, Cin);
    const TileGeom g = geom(N, H, W);
    const size_t ub = u_bytes(Cin, Cout), vb = plane_bytes(g, Cout), mb = plane_bytes(g, Cin);
    MYOLO_NEED_WS(ub + vb + mb);
    hipStream_t s = (hipStream_t)stream;
    float* V = (float*)((char*)ws + ub);
    float* Mp = (float*)((char*)ws + ub + vb);
    const float* U = w43_filters(w, (float*)ws, Cin, Cout, 1, s);
    if (lazy) hipLaunchKernelGGL(wino_in_kernel<true>, dim3(ew_grid(g.T * (Cout / 4))), dim3(256), 0, s, dy, V, g, Cout, (const float*)nullptr, (const float*)nullptr, 0, *lazy);
    else hipLaunchKernelGGL(wino_in_kernel<false>, dim3(ew_grid(g.T * (Cout / 4))), dim3(256), 0, s, dy, V, g, Cout, (const float*)nullptr, (const float*)nullptr, 0, LazyBn{});
    const int rc = wino_multiply_all(V, U, Mp, g, Cout, Cin, s);
    if (rc != MYOLO_OK) return rc;
    hipLaunchKernelGGL(wino_out_kernel, dim3(ew_grid(g.T * (Cin / 4))), dim3(256), 0, s, Mp, dx, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, g, Cin, MYOLO_ACT_NONE, (double*)nullptr);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

static int wino_bwd_weight_impl(const float* x, const float* v_saved, const float* dy, const LazyBn* lazy, float* dw, int N, int H, int W,
                                int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE((x || v_saved) && dy && dw && N > 0 && H > 0 && W > 0, "conv3x3_wino_bwd_weight: bad arguments");
    MYOLO_REQUIRE((Cin & 3) == 0 && (Cout & 3) == 0, "conv3x3_wino_bwd_weight: needs Cin %% 4 == 0 and Cout %% 4 == 0 (got %d, %d)", Cin, Cout);
    const TileGeom g = geom(N, H, W);
    const size_t ub = u_bytes(Cin, Cout), vb = v_saved ? 0 : plane_bytes(g, Cin), qb = plane_bytes(g, Cout);
    const size_t pb = align256(wino_tn_ws_bytes(g, Cin, Cout));
    MYOLO_NEED_WS(ub + vb + qb + pb);
    hipStream_t s = (hipStream_t)stream;
    float* dU = (float*)ws;
    float* V = (float*)((char*)ws + ub);
    float* Q = (float*)((char*)ws + ub + vb);
    void* part = (char*)ws + ub + vb + qb;
    if (!v_saved) hipLaunchKernelGGL(wino_in_kernel<false>, dim3(ew_grid(g.T * (Cin / 4))), dim3(256), 0, s, x, V, g, Cin, (const float*)nullptr, (const float*)nullptr, 0, LazyBn{});
    if (lazy) hipLaunchKernelGGL(wino_dy_kernel<true>, dim3(ew_grid(g.T * (Cout / 4))), dim3(256), 0, s, dy, Q, g, Cout, *lazy);
    else hipLaunchKernelGGL(wino_dy_kernel<false>, dim3(ew_grid(g.T * (Cout / 4))), dim3(256), 0, s, dy, Q, g, Cout, LazyBn{});
    const int rc = wino_tn_all(v_saved ? v_saved : V, Q, dU, g, Cin, Cout, part, pb, s);
    if (rc != MYOLO_OK) return rc;
    hipLaunchKernelGGL(wino_dw_kernel, dim3((Cin * Cout + 255) / 256), dim3(256), 0, s, dU, dw, Cin, Cout);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

extern "C" {

int myolo_conv3x3_wino_bwd_data(const float* dy, const float* w, float* dx, int N, int H, int W, int Cin, int Cout, void* ws,
                                size_t ws_bytes, void* stream)
{
    return wino_bwd_data_impl(dy, nullptr, w, dx, N, H, W, Cin, Cout, ws, ws_bytes, stream);
}

int myolo_conv3x3_wino_bwd_weight(const float* x, const float* v_saved, const float* dy, float* dw, int N, int H, int W, int Cin, int Cout,
                                  void* ws, size_t ws_bytes, void* stream)
{
    return wino_bwd_weight_impl(x, v_saved, dy, nullptr, dw, N, H, W, Cin, Cout, ws, ws_bytes, stream);
}

/* The same two gradients when dy is the gradient behind a training-mode BatchNorm + activation with a row-sparse upstream
 * gradient (bn1 of the mask head, model.py:690): y_pre is that BN's input (= this conv's output), dy_compact / inv / ka / kb
 * as produced by myolo_bn_bwd_rowsparse_coeffs.  dy itself is never materialised. */
int myolo_conv3x3_wino_bwd_data_lazybn(const float* y_pre, const float* dy_compact, const int32_t* inv, const float* scale,
                                       const float* shift, const float* ka, const float* kb, int act, const float* w, float* dx, int N,
                                       int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(y_pre && inv && scale && shift && ka && kb, "conv3x3_wino_bwd_data_lazybn: bad arguments");
    const LazyBn lz{dy_compact, inv, scale, shift, ka, kb, act};
    return wino_bwd_data_impl(y_pre, &lz, w, dx, N, H, W, Cin, Cout, ws, ws_bytes, stream);
}

int myolo_conv3x3_wino_bwd_weight_lazybn(const float* v_saved, const float* y_pre, const float* dy_compact, const int32_t* inv,
                                         const float* scale, const float* shift, const float* ka, const float* kb, int act, float* dw,
                                         int N, int H, int W, int Cin, int Cout, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(v_saved && y_pre && inv && scale && shift && ka && kb, "conv3x3_wino_bwd_weight_lazybn: bad arguments");
    const LazyBn lz{dy_compact, inv, scale, shift, ka, kb, act};
    return wino_bwd_weight_impl(nullptr, v_saved, y_pre, &lz, dw, N, H, W, Cin, Cout, ws, ws_bytes, stream);
}

}  // extern "C"
