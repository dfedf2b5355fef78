// Winograd F(4x4,3x3) 3x3 / s1 / SAME convolution of 14x14 maps (the mask-head convs, model.py:687-709) as ONE kernel:
// the transformed input V and the products M never reach HBM.
//
//   workgroup  = 2 ROIs (32 tiles of 4x4 outputs) x 64 output channels, all 36 transform points, whole K = Cin loop;
//                256 threads = 4 waves, one per SIMD (the 36 * 32 * 64 fp32 accumulators fill 288 of a lane's 512 registers)
//   wave w     = points 9w .. 9w+8, both 32-channel halves: 18 MFMA 32x32x2 accumulator tiles
//   K loop     = chunks of 8 input channels:
//                  raw x chunk  [2 ROIs][196 px][8 ch]  global -> registers -> LDS (pixel stride 10 floats: conflict-free patch reads)
//                  transform    thread = (tile, channel): 36 LDS reads -> B^T d B -> 36 LDS writes into V[36][32 tiles][8 k]
//                  multiply     A fragments: one ds_read_b128 per point (k = 4h..4h+3 of tile l&31; 16-B slot XOR-swizzled by
//                               tile bit 3), B fragments straight from global memory (the filter slab is pre-arranged so that a
//                               lane's 4 k-steps are one 16-byte load), 72 MFMAs per wave and chunk
//                double-buffered V and raw images, one barrier per chunk
//   epilogue   = accumulators -> LDS per 32-channel half -> thread = (tile, channel): A^T m A + bias, folded-BN affine, ReLU
//                -> y [N,14,14,Cout] (full 128-byte row segments)
// Uniform F(4,3) tiling (16 tiles per ROI, 576 point-tiles): the mixed F(4,3)/F(2,3) tiling of wino_kernels.hip needs a row
// granularity of 16 and >= 4 ROIs per workgroup, which the accumulator budget does not allow together with 64 output channels.
//
// STATUS: correct (tests/test_gpu_ops.py::test_conv3x3_winograd_fused_kernel) but NOT on the default path: 4.14 ms per conv at
// the config-2 shape against 3.72 ms for the three-stage op of wino_kernels.hip (profiles/r2_notes.md).  Why: on gfx950 every
// VALU / LDS instruction issued on a SIMD takes ~2-3 cycles away from a running fp32 MFMA stream, whichever wave issues it
// (tools/mfma_valu_overlap.hip), so the transform's ~5 VALU + 1 LDS instructions per MFMA cost the same time inside this kernel as
// they do in their own HBM-bound kernels -- and this kernel cannot use the mixed tiling (16 % fewer products).
#include "myolo_common.h"
#include <type_traits>

// compile-time loop: f(integral_constant<int, G>) for G = 0 .. N-1 (the body indexes register arrays with G: it must be unrolled)
template <int G, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (G < N) {
        f(std::integral_constant<int, G>{});
        static_for<G + 1, N>(f);
    }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define WF_NT 32          // tiles per workgroup (2 ROIs x 16)
#define WF_KC 8           // input channels per chunk
#define WF_RS 10          // floats per pixel in the raw LDS image
#define WF_HW 14
#define WF_PX (WF_HW * WF_HW)
#define WF_V_FLOATS (36 * WF_NT * WF_KC)              // 9216 per buffer
#define WF_PW 18          // padded raw image: rows / columns -1 .. 16 of the 14x14 map (the border stays zero = SAME padding)
#define WF_R_FLOATS (2 * WF_PW * WF_PW * WF_RS)       // 6480 per buffer
#define WF_LDS_BYTES (36 * WF_NT * 32 * 4)            // 147456: the epilogue's [36][32 tiles][32 channels] image (>= main-loop use:
                                                      // 2 x 36864 (V) + 2 x 25920 (raw) + 8192 (dump slots) = 133760)

struct WFArgs {
    const float* x;        // [NR,14,14,Cin]
    const float* Uf;       // transformed filters in the fused layout (wino_w_fused_kernel)
    const float* bias;     // [Cout] or NULL
    const float* scale;    // [Cout] or NULL (folded BatchNorm)
    const float* shift;
    float* y;              // [NR,14,14,Cout]
    int NR, Cin, Cout, act;
};

__device__ __forceinline__ void wf_bt6(const float d[6], float t[6])
{
    t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
    t[1] = d[3] + d[4] - 4.f * (d[1] + d[2]);
    t[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
    t[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
    t[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
    t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}
__device__ __forceinline__ void wf_at6(const float m[6], float y[4])
{
    const float s12 = m[1] + m[2], d12 = m[1] - m[2], s34 = m[3] + m[4], d34 = m[3] - m[4];
    y[0] = m[0] + s12 + s34;
    y[1] = d12 + 2.f * d34;
    y[2] = s12 + 4.f * s34;
    y[3] = d12 + 8.f * d34 + m[5];
}

// w [3,3,Ci,Co] -> Uf[slab = co/64][chunk = ci/8][point 36][nh 2][lane 64][s 4]  with ci%8 = 4*(lane>>5) + s, co%64 = 32*nh + (lane&31):
// a wave's B fragments of one point, one 32-channel half and one chunk are 1 KB contiguous, 16 B per lane.
__global__ __launch_bounds__(256) void wino_w_fused_kernel(const float* __restrict__ w, float* __restrict__ Uf, int Ci, int Co)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Ci * Co) return;
    const int ci = idx / Co, co = idx - ci * Co;
    float g[3][3], tmp[6][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) g[ky][kx] = w[((ky * 3 + kx) * Ci + ci) * (long long)Co + co];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const float c0 = g[0][kx], c1 = g[1][kx], c2 = g[2][kx];
        tmp[0][kx] = c0 * 0.25f;
        tmp[1][kx] = -(c0 + c1 + c2) * (1.f / 6.f);
        tmp[2][kx] = -(c0 - c1 + c2) * (1.f / 6.f);
        tmp[3][kx] = c0 * (1.f / 24.f) + c1 * (1.f / 12.f) + c2 * (1.f / 6.f);
        tmp[4][kx] = c0 * (1.f / 24.f) - c1 * (1.f / 12.f) + c2 * (1.f / 6.f);
        tmp[5][kx] = c2;
    }
    const int slab = co >> 6, nh = (co >> 5) & 1, j = co & 31;
    const int chunk = ci >> 3, kk = ci & 7, h = kk >> 2, s = kk & 3;
    const int nchunk = Ci >> 3;
    const long long base = ((long long)slab * nchunk + chunk) * 36;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const float c0 = tmp[i][0], c1 = tmp[i][1], c2 = tmp[i][2];
        float u[6];
        u[0] = c0 * 0.25f;
        u[1] = -(c0 + c1 + c2) * (1.f / 6.f);
        u[2] = -(c0 - c1 + c2) * (1.f / 6.f);
        u[3] = c0 * (1.f / 24.f) + c1 * (1.f / 12.f) + c2 * (1.f / 6.f);
        u[4] = c0 * (1.f / 24.f) - c1 * (1.f / 12.f) + c2 * (1.f / 6.f);
        u[5] = c2;
#pragma unroll
        for (int jj = 0; jj < 6; ++jj)
            Uf[((((base + i * 6 + jj) * 2 + nh) * 64) + h * 32 + j) * 4 + s] = u[jj];
    }
}

__global__ __launch_bounds__(256, 1) void wino_fused_fwd_kernel(WFArgs p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* const Vs = lds;                                // [2][36][32][8]
    float* const Rs = lds + 2 * WF_V_FLOATS;              // [2][2 ROIs][18][18][10], zero border

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = lane >> 5, l31 = lane & 31;
    const int nslab = p.Cout >> 6;
    const int slab = blockIdx.x % nslab;                  // block b runs on XCD b % 8: an XCD sees one (or two) filter slabs
    const int roi0 = (blockIdx.x / nslab) * 2;
    const int nchunk = p.Cin / WF_KC;

    // ---- transform role: thread = (tile il, channel kt) ----
    const int il = tid >> 3, kt = tid & 7;
    const int troi = il >> 4, tyx = il & 15, tty = tyx >> 2, ttx = tyx & 3;
    const int vsw = ((il >> 3) & 1) << 2;                 // swizzle of the 4-float slot inside the tile's 8-float row
    // ---- raw fill role: thread owns pixels q0 = tid and q1 = tid + 256 (< 392); out-of-range ROIs / pixels read a valid dummy
    //      address (no exec-mask branches in the loop) ----
    const int q1 = tid + 256;
    const bool q1ok = q1 < 2 * WF_PX;
    const int q1c = q1ok ? q1 : tid;
    const bool r0ok = (roi0 + (tid >= WF_PX ? 1 : 0)) < p.NR;
    const bool r1ok = q1ok && (roi0 + (q1 >= WF_PX ? 1 : 0)) < p.NR;
    const long long xrow0 = r0ok ? ((long long)roi0 * WF_PX + tid) * p.Cin : 0;
    const long long xrow1 = r1ok ? ((long long)roi0 * WF_PX + q1c) * p.Cin : 0;
    // (a ROI beyond NR -- the second half of the last workgroup when NR is odd -- reads ROI 0 instead: its results are never stored)
    // padded LDS position of a pixel q = roi*196 + y*14 + x
    auto padpos = [](int q) { const int r = q / WF_PX, rem = q - r * WF_PX, yy = rem / WF_HW, xx = rem - yy * WF_HW;
                              return ((r * WF_PW + yy + 1) * WF_PW + xx + 1) * WF_RS; };
    const int rpos0 = padpos(tid), rpos1 = padpos(q1c);
    f32x16 acc[9][2];
#pragma unroll
    for (int a = 0; a < 9; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const float* ub = p.Uf + ((long long)slab * nchunk * 36 + wave * 9) * 2 * 64 * 4 + lane * 4;      // + chunk*36*512 + (pl*2+nh)*256
    float4 bq[9][2];

    auto raw_load = [&](int chunk, float4 (&r)[4]) {
        const int c0 = chunk * WF_KC;
        r[0] = *reinterpret_cast<const float4*>(p.x + xrow0 + c0);
        r[1] = *reinterpret_cast<const float4*>(p.x + xrow0 + c0 + 4);
        r[2] = *reinterpret_cast<const float4*>(p.x + xrow1 + c0);
        r[3] = *reinterpret_cast<const float4*>(p.x + xrow1 + c0 + 4);
    };
    auto raw_store = [&](int buf, const float4 (&r)[4]) {
        float* d0 = Rs + buf * WF_R_FLOATS + rpos0;
        *reinterpret_cast<float2*>(d0 + 0) = make_float2(r[0].x, r[0].y);
        *reinterpret_cast<float2*>(d0 + 2) = make_float2(r[0].z, r[0].w);
        *reinterpret_cast<float2*>(d0 + 4) = make_float2(r[1].x, r[1].y);
        *reinterpret_cast<float2*>(d0 + 6) = make_float2(r[1].z, r[1].w);
        if (q1ok) {
            float* d1 = Rs + buf * WF_R_FLOATS + rpos1;
            *reinterpret_cast<float2*>(d1 + 0) = make_float2(r[2].x, r[2].y);
            *reinterpret_cast<float2*>(d1 + 2) = make_float2(r[2].z, r[2].w);
            *reinterpret_cast<float2*>(d1 + 4) = make_float2(r[3].x, r[3].y);
            *reinterpret_cast<float2*>(d1 + 6) = make_float2(r[3].z, r[3].w);
        }
    };
    auto transform = [&](int buf) {
        // patch origin (ty0, tx0) = (4*tty - 1, 4*ttx - 1) -> padded index (4*tty, 4*ttx): 36 loads at constant offsets
        const float* rb = Rs + buf * WF_R_FLOATS + ((troi * WF_PW + 4 * tty) * WF_PW + 4 * ttx) * WF_RS + kt;
        float* vb = Vs + buf * WF_V_FLOATS + il * WF_KC + (kt ^ vsw);
        float tmp[6][6];
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            float d[6], r[6];
#pragma unroll
            for (int i = 0; i < 6; ++i) d[i] = rb[(i * WF_PW + j) * WF_RS];
            wf_bt6(d, r);
#pragma unroll
            for (int i = 0; i < 6; ++i) tmp[i][j] = r[i];
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float r[6];
            wf_bt6(tmp[i], r);
#pragma unroll
            for (int j = 0; j < 6; ++j) vb[(i * 6 + j) * (WF_NT * WF_KC)] = r[j];
        }
    };
    auto load_b = [&](int chunk) {
        const float* u = ub + (long long)chunk * 36 * 512;
#pragma unroll
        for (int pl = 0; pl < 9; ++pl)
#pragma unroll
            for (int nh = 0; nh < 2; ++nh) bq[pl][nh] = *reinterpret_cast<const float4*>(u + (pl * 2 + nh) * 256);
    };

    // ---- prologue: zero the raw images (their borders stay zero), raw 0 -> LDS, transform 0, raw 1 -> LDS, B of chunk 0 ----
    for (int e = tid; e < 2 * WF_R_FLOATS; e += 256) Rs[e] = 0.f;
    __syncthreads();
    float4 rr[4];
    raw_load(0, rr);
    raw_store(0, rr);
    if (nchunk > 1) raw_load(1, rr);
    load_b(0);
    __syncthreads();
    transform(0);
    if (nchunk > 1) raw_store(1, rr);
    __syncthreads();

    const int arow = ((wave * 9) * WF_NT + l31) * WF_KC + ((4 * half) ^ (((l31 >> 3) & 1) << 2));
    const int rbase = ((troi * WF_PW + 4 * tty) * WF_PW + 4 * ttx) * WF_RS + kt;      // transform: patch origin in the padded raw image
    const int vbase = il * WF_KC + (kt ^ vsw);                                          // transform: this thread's slot in every V plane
    // threads without a second pixel dump it into a private scratch slot (no exec-mask branch in the loop)
    const int rdump = 2 * WF_V_FLOATS + 2 * WF_R_FLOATS + tid * 8;
    for (int c = 0; c < nchunk; ++c) {
        // ONE basic block per chunk, hand-interleaved: a 32x32x2 fp32 MFMA holds the matrix pipe for 64 cycles and a wave issues in
        // order, so everything else -- the input transform of chunk c+1 (36 LDS reads, two B^T passes, 36 LDS writes), the A-fragment
        // reads, the B-fragment loads of chunk c+1 and the raw image of chunk c+2 -- is cut into 72 small pieces, one behind each
        // MFMA (sched_barrier(0) pins the memory operations; pure arithmetic and the MFMAs still float between them -- pinning those too with
        // empty volatile asm statements was measured and changes nothing, see profiles/r2_notes.md).
        // The look-ahead of the last iterations is clamped to the last chunk (it lands in buffers nobody reads) instead of branching.
        const int cur = c & 1;
        const int c1 = min(c + 1, nchunk - 1), c2 = min(c + 2, nchunk - 1);
        const float* va = Vs + cur * WF_V_FLOATS + arow;
        const float* un = ub + (long long)c1 * 36 * 512;
        const float* rbn = Rs + (cur ^ 1) * WF_R_FLOATS + rbase;
        float* vbn = Vs + (cur ^ 1) * WF_V_FLOATS + vbase;
        float* rs0 = Rs + cur * WF_R_FLOATS + rpos0;
        float* rs1 = q1ok ? Rs + cur * WF_R_FLOATS + rpos1 : lds + rdump;
        const float* xs0 = p.x + xrow0 + c2 * WF_KC;
        const float* xs1 = p.x + xrow1 + c2 * WF_KC;
        float4 af[9];
        af[0] = *reinterpret_cast<const float4*>(va);
        float d[6], tmp[6][6], ro[6];
        static_for<0, 72>([&](auto gc) {
            constexpr int g = decltype(gc)::value;
            constexpr int pl = g >> 3, nh = (g >> 2) & 1, st = g & 3;
            const float av = st == 0 ? af[pl].x : st == 1 ? af[pl].y : st == 2 ? af[pl].z : af[pl].w;
            const float bv = st == 0 ? bq[pl][nh].x : st == 1 ? bq[pl][nh].y : st == 2 ? bq[pl][nh].z : bq[pl][nh].w;
            acc[pl][nh] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[pl][nh], 0, 0, 0);
            if constexpr (st == 3) bq[pl][nh] = *reinterpret_cast<const float4*>(un + (pl * 2 + nh) * 256);       // same registers, next chunk
            if constexpr ((g & 7) == 2 && pl < 8) af[pl + 1] = *reinterpret_cast<const float4*>(va + (pl + 1) * (WF_NT * WF_KC));
            if constexpr (g == 1) { rr[0] = *reinterpret_cast<const float4*>(xs0); rr[1] = *reinterpret_cast<const float4*>(xs0 + 4); }
            if constexpr (g == 7) { rr[2] = *reinterpret_cast<const float4*>(xs1); rr[3] = *reinterpret_cast<const float4*>(xs1 + 4); }
            if constexpr (g < 36) {             // column pass j of B^T d B
                constexpr int j = g / 6, part = g % 6;
                if constexpr (part == 0) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) d[i] = rbn[(i * WF_PW + j) * WF_RS];
                } else if constexpr (part == 2) {
                    tmp[0][j] = 4.f * d[0] - 5.f * d[2] + d[4];
                    tmp[1][j] = d[3] + d[4] - 4.f * (d[1] + d[2]);
                } else if constexpr (part == 3) {
                    tmp[2][j] = 4.f * (d[1] - d[2]) - d[3] + d[4];
                    tmp[3][j] = 2.f * (d[3] - d[1]) - d[2] + d[4];
                } else if constexpr (part == 4) {
                    tmp[4][j] = 2.f * (d[1] - d[3]) - d[2] + d[4];
                    tmp[5][j] = 4.f * d[1] - 5.f * d[3] + d[5];
                }
            } else {                            // row pass i, then its 6 stores into the V planes
                constexpr int i = (g - 36) / 6, part = (g - 36) % 6;
                if constexpr (part == 0) {
                    ro[0] = 4.f * tmp[i][0] - 5.f * tmp[i][2] + tmp[i][4];
                    ro[1] = tmp[i][3] + tmp[i][4] - 4.f * (tmp[i][1] + tmp[i][2]);
                } else if constexpr (part == 1) {
                    ro[2] = 4.f * (tmp[i][1] - tmp[i][2]) - tmp[i][3] + tmp[i][4];
                    ro[3] = 2.f * (tmp[i][3] - tmp[i][1]) - tmp[i][2] + tmp[i][4];
                } else if constexpr (part == 2) {
                    ro[4] = 2.f * (tmp[i][1] - tmp[i][3]) - tmp[i][2] + tmp[i][4];
                    ro[5] = 4.f * tmp[i][1] - 5.f * tmp[i][3] + tmp[i][5];
                } else {
                    constexpr int j0 = 2 * (part - 3);
                    vbn[(i * 6 + j0) * (WF_NT * WF_KC)] = ro[j0];
                    vbn[(i * 6 + j0 + 1) * (WF_NT * WF_KC)] = ro[j0 + 1];
                }
            }
            if constexpr (g >= 38 && ((g - 38) & 3) == 0 && (g - 38) / 4 < 8) {      // raw image of chunk c+2 -> raw[c&1], one 8-byte piece per gap
                constexpr int k = (g - 38) / 4;
                // (no arithmetic on the loaded values: anything computed from them is hoisted up to the load and its wait with it)
                float* dst = (k < 4 ? rs0 : rs1) + 2 * (k & 3);
                const float4 src = rr[k >> 1];
                *reinterpret_cast<float2*>(dst) = (k & 1) ? make_float2(src.z, src.w) : make_float2(src.x, src.y);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();
    }

    // ---- epilogue: per 32-channel half, accumulators -> LDS [36][32 tiles][32 ch] -> output transform ----
    float* Ms = lds;
    const int etile = tid >> 5, ech = tid & 31;               // two tiles per thread: etile and etile + 16... (8 tiles per pass of 256 threads)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
        __syncthreads();
#pragma unroll
        for (int pl = 0; pl < 9; ++pl)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
                Ms[((wave * 9 + pl) * WF_NT + row) * 32 + l31] = acc[pl][nh][r];
            }
        __syncthreads();
        const int co = slab * 64 + nh * 32 + ech;
        const float bv = p.bias ? p.bias[co] : 0.f;
        const float sc = p.scale ? p.scale[co] : 1.f;
        const float sh = p.scale ? p.shift[co] : 0.f;
        for (int t = etile; t < WF_NT; t += 8) {
            const int roi = roi0 + (t >> 4);
            if (roi >= p.NR) continue;
            const int ty = (t & 15) >> 2, tx = t & 3;
            float tmp[4][6];
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                float m[6], r[4];
#pragma unroll
                for (int i = 0; i < 6; ++i) m[i] = Ms[((i * 6 + j) * WF_NT + t) * 32 + ech];
                wf_at6(m, r);
#pragma unroll
                for (int i = 0; i < 4; ++i) tmp[i][j] = r[i];
            }
            float* yb = p.y + (long long)roi * WF_PX * p.Cout + co;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float r[4];
                wf_at6(tmp[i], r);
                const int yy = 4 * ty + i;
                if (yy >= WF_HW) continue;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int xx = 4 * tx + j;
                    if (xx >= WF_HW) continue;
                    float v = fmaf(r[j] + bv, sc, sh);
                    if (p.act == MYOLO_ACT_RELU) v = fmaxf(v, 0.f);
                    yb[(long long)(yy * WF_HW + xx) * p.Cout] = v;
                }
            }
        }
    }
}

extern "C" {

size_t myolo_conv3x3_wino_fused_ws_bytes(int Cin, int Cout) { return align256((size_t)36 * Cin * Cout * sizeof(float)); }

/* y = act(((conv3x3_same(x, w) + bias) * scale + shift)) for 14x14 maps in one kernel (see the file header).
 * Needs H = W = 14, Cin % 8 == 0, Cout % 64 == 0; ws >= myolo_conv3x3_wino_fused_ws_bytes (the re-arranged filters). */
int myolo_conv3x3_wino_fused_fwd(const float* x, const float* w, const float* bias, const float* scale, const float* shift, float* y,
                                 int N, int H, int W, int Cin, int Cout, int act, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && w && y && N > 0, "conv3x3_wino_fused_fwd: bad arguments");
    MYOLO_REQUIRE(H == WF_HW && W == WF_HW && (Cin % WF_KC) == 0 && (Cout % 64) == 0,
                  "conv3x3_wino_fused_fwd: needs 14x14 maps, Cin %% 8 == 0, Cout %% 64 == 0 (got %dx%d, %d, %d)", H, W, Cin, Cout);
    MYOLO_REQUIRE(!scale == !shift, "conv3x3_wino_fused_fwd: scale and shift go together");
    MYOLO_REQUIRE(act == MYOLO_ACT_NONE || act == MYOLO_ACT_RELU, "conv3x3_wino_fused_fwd: act must be NONE or RELU");
    MYOLO_NEED_WS(myolo_conv3x3_wino_fused_ws_bytes(Cin, Cout));
    hipStream_t s = (hipStream_t)stream;
    float* Uf = (float*)ws;
    hipLaunchKernelGGL(wino_w_fused_kernel, dim3((Cin * Cout + 255) / 256), dim3(256), 0, s, w, Uf, Cin, Cout);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)wino_fused_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, WF_LDS_BYTES);
        attr_set = true;
    }
    WFArgs a{x, Uf, bias, scale, shift, y, N, Cin, Cout, act};
    const unsigned blocks = (unsigned)(((N + 1) / 2) * (Cout / 64));
    hipLaunchKernelGGL(wino_fused_fwd_kernel, dim3(blocks), dim3(256), WF_LDS_BYTES, s, a);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

}  // extern "C"
