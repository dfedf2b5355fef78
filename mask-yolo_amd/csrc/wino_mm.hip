// Batched fp32 GEMM of the Winograd multiply stage with the SECOND operand stored transposed:
//     C[z] (M x N) = A[z] (M x K) * Bt[z]^T,   A [M][K], Bt [N][K], C [M][N], all dense row-major, z = transform point.
// (the 36 products per mask-head 3x3 conv, model.py:687-709; Bt = the transformed filters, which our own wino_w_kernel writes in
// whichever layout the multiply wants.)
//
// Why another GEMM kernel: on gfx950 every non-MFMA instruction a SIMD issues costs the fp32 matrix pipe ~2-3 cycles, whoever
// issues it (tools/mfma_valu_overlap.hip, profiles/r2_notes.md), and gemm_nn_fast's loop issues 41 of them per 32 MFMAs
// (16 ds_read2_b32, 6 LDS writes, 4 buffer loads, ~15 VALU).  With BOTH operands k-contiguous in LDS a lane's four consecutive
// k-steps are one ds_read_b128; a 32x32x2 MFMA only needs lanes 0-31 and 32-63 to hold two DIFFERENT k of the chunk, so the
// k order is permuted: lanes 0-31 walk k = 0..7 of the 16-deep chunk, lanes 32-63 walk k = 8..15.  Tile 128 x 256 (A is read
// once for all 256 output channels), 4 waves of 64 x 128 = 8 MFMA tiles:
//     per 16-deep chunk and wave: 64 MFMAs, 12 ds_read_b128, 6 ds_write_b128, 6 buffer_load_dwordx4, one barrier.
// LDS rows are padded to 20 floats: the 16 lanes of a ds_read_b128 phase (rows l .. l+15, same 16-byte slot) cover all 64
// banks exactly once; the b128 writes (4 lanes per row) are conflict-free as well.
// Pipeline: chunk c+1 is written to the other LDS buffer at the start of the second half of chunk c (its global loads were issued
// one chunk earlier), one barrier, then the first-half fragments of chunk c+1 are read under the remaining MFMAs of chunk c;
// the second-half fragments of a chunk are read under its first-half MFMAs.  No LDS read is waited for right after it is issued.
#include "myolo_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define MM_BM 128
#define MM_BN 256
#define MM_BK 16
#define MM_LD 20                     // padded LDS row (floats)
#define MM_OOB 0x7fffff00u

// one launch covers up to 4 "runs" of planes (groups of transform points whose planes have the same number of rows, see
// group_runs() in wino_kernels.hip): separate launches would each end in a partial wave of workgroups.
struct MMRun {
    long long rows;        // M of every plane of the run
    long long a_off;       // element offset of the run's first plane in A (planes M*K apart), likewise Bt (K*N apart) and C (M*N)
    long long b_off;
    long long c_off;
    long long tile0;       // first flattened tile index of the run
    int mtiles;            // ceil(rows / 128)
    int nq;                // planes in the run
};
struct MMArgs {
    const float* A;
    const float* Bt;
    float* C;
    int K, N;
    int nruns;
    int nt;
    // MM_EP_DECONV_MASK only (csrc/gemm_kernels.hip EP_DECONV_MASK, model.py:711-714): columns = (tap, co) of a 2x2 / s2 deconv
    const float* bias;     // [Co]
    const float* w2;       // [Co][ncls]: the 1x1 mask conv
    float* part;           // [Co/128 slabs][4*M pixels][ncls] partial logits
    int H, W, Co, ncls;
    // ... and optionally the ReLU'd deconv output itself for the images (ROIs) the caller will differentiate: keep_inv[image] = slot (< keep_cap) or -1,
    // keep_d [slot][2H][2W][Co] -- what MM_EP_DECONV would write for that image (the sparse mask-head backward reads it instead of re-running the deconv)
    const int32_t* keep_inv;
    float* keep_d;
    int keep_cap;
    // MM_EP_DECONV_MASK_T with Co == 256 (a workgroup holds all channels of its pixels): the 1x1 conv's bias and the probabilities [4*M][ncls] --
    // the two waves' 128-channel slabs meet in LDS and the kernel stores the sigmoid itself (NULL: partial logits to `part` + deconv_mask_finish)
    const float* b2;
    float* out;
    // PW (pointwise conv of the trunk in training mode, see gemm_kernels.hip myolo_pwconv1x1_bnstats_fwd): A := act(A * a_scale[k] + a_shift[k])
    // on load; stat: per row-tile partial sums of the output columns [M tiles][2][N] doubles
    const float* a_scale;
    const float* a_shift;
    int a_act;
    double* stat;
    int tune;              // MM_X6_TUNE builds only (tools/experiments/x6_tune.sh): timing-only ablations of wino_mm_x6_kernel, results are wrong
    MMRun run[4];
};

__device__ __forceinline__ float4 mm_bufld4(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t mm_rsrc(const float* base, long long nbytes)
{
    if (nbytes < 0) nbytes = 0;
    if (nbytes > 0x7ffffe00ll) nbytes = 0x7ffffe00ll;
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(unsigned)nbytes, 0x00020000);
}
__device__ __forceinline__ float f4c(const float4& v, int s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

enum { MM_EP_PLAIN = 0, MM_EP_DECONV_MASK = 1, MM_EP_DECONV = 2, MM_EP_DECONV_MASK_T = 3 };   // _T (bf16x6 kernel): the tile formed transposed, see mm_deconv_mask_epilogue_t     // DECONV: 2x2 / s2 transposed-conv scatter of a (tap, co) column tile + bias + activation
enum { MM_A_PLAIN = 0, MM_A_DECONV = 1 };                               // DECONV: A row m = the four taps of output pixel block m gathered from [N,2H,2W,Cc] (K = 4 Cc)

// Epilogue of the fused deconv + ReLU + 1x1 mask conv (model.py:711-714; csrc/gemm_kernels.hip EP_DECONV_MASK for 4 column tiles per
// wave), shared by the fp32 and the bf16x6 kernel: both hold the 128 x 256 tile as 4 waves x (2 x 4) 32x32 accumulator tiles.
__device__ __forceinline__ void mm_deconv_mask_epilogue(const MMArgs& p, const f32x16 (&acc)[2][4], long long m0, long long M, int n0,
                                                        int wm, int wn, int half, int l31)
{
        // relu(deconv + bias) times the 1x1 mask conv, never writing the [N,2H,2W,Co] tensor: this tile is 128 input pixels x the
        // 256 channels starting at n0 of ONE tap (Co % 256 == 0).  Per class each lane forms its 32 row slots' products for its 4
        // columns, then a reduce-scatter butterfly over the 32 lanes of its half leaves lane l the sum of row slot l over the
        // wave's 128 columns; the Co/128 column slabs are summed in fixed order by deconv_mask_finish (deterministic).
        const int tap = n0 / p.Co;
        const int cbase = n0 - tap * p.Co + wn * 128 + l31;
        float cbm[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) cbm[u] = p.bias[cbase + 32 * u];
        const long long hw = (long long)p.H * p.W;
        if (p.keep_d) {
            // images of this row tile (uniform): anything to keep at all?
            const long long last = (m0 + MM_BM <= M ? m0 + MM_BM : M) - 1;
            bool any = false;
            for (long long g = m0 / hw; g <= last / hw; ++g) { const int sl = p.keep_inv[g]; any = any || (sl >= 0 && sl < p.keep_cap); }
            if (any) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const long long row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (row >= M) continue;
                        const long long n_img = row / hw;
                        const int sl = p.keep_inv[n_img];
                        if (sl < 0 || sl >= p.keep_cap) continue;
                        const int rem = (int)(row - n_img * hw);
                        const int y = rem / p.W, x = rem - y * p.W;
                        float* kd = p.keep_d + ((long long)sl * 4 * hw + (long long)(2 * y + (tap >> 1)) * 2 * p.W + 2 * x + (tap & 1)) * p.Co + cbase;
#pragma unroll
                        for (int u = 0; u < 4; ++u) kd[32 * u] = fmaxf(acc[t][u][r] + cbm[u], 0.f);
                    }
            }
        }
        const int rr = l31 & 15;          // after the butterfly lane l31 owns row slot rr of the 32-row block l31 >> 4
        const long long row = m0 + wm * 64 + (l31 >> 4) * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * half;
        float* dst = nullptr;
        if (row < M) {
            const long long n_img = row / hw;
            const int rem2 = (int)(row - n_img * hw);
            const int y = rem2 / p.W, x = rem2 - y * p.W;
            const long long pix = n_img * 4 * hw + (long long)(2 * y + (tap >> 1)) * 2 * p.W + 2 * x + (tap & 1);
            const int slab = (cbase - l31) >> 7;
            dst = p.part + ((long long)slab * 4 * M + pix) * p.ncls;
        }
#pragma unroll 1
        for (int c = 0; c < p.ncls; ++c) {
            float wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) wv[u] = p.w2[(cbase + 32 * u) * p.ncls + c];
            float outv = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float cur[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float v = fmaxf(acc[t][0][j] + cbm[0], 0.f) * wv[0];
                    v = fmaf(fmaxf(acc[t][1][j] + cbm[1], 0.f), wv[1], v);
                    v = fmaf(fmaxf(acc[t][2][j] + cbm[2], 0.f), wv[2], v);
                    cur[j] = fmaf(fmaxf(acc[t][3][j] + cbm[3], 0.f), wv[3], v);
                }
#pragma unroll
                for (int st = 0; st < 4; ++st) {
                    const int m = 1 << st;
                    const bool bit = (l31 >> st) & 1;
#pragma unroll
                    for (int ii = 0; ii < (8 >> st); ++ii) {
                        const float a = cur[2 * ii], b = cur[2 * ii + 1];
                        const float keep = bit ? b : a, send = bit ? a : b;
                        cur[ii] = keep + __shfl_xor(send, m, 64);
                    }
                }
                const float tot = cur[0] + __shfl_xor(cur[0], 16, 64);
                if ((l31 >> 4) == t) outv = tot;
            }
            if (dst) dst[c] = outv;
        }
}

// {bias, w2} of the tile's 256 channels into LDS: called in the kernel's PROLOGUE (its global loads ride on the first operand loads, its LDS
// writes are published by the prologue's barrier) -- built in the epilogue it cost two barriers and an exposed L2 round trip per tile
__device__ __forceinline__ void mm_deconv_mask_table(const MMArgs& p, unsigned char* lds, int n0, int tid)
{
    float* tab = reinterpret_cast<float*>(lds);
    const int co = n0 - (n0 / p.Co) * p.Co + tid;
    const float* w2r = p.w2 + (long long)co * p.ncls;
    tab[tid] = p.bias[co];
    *reinterpret_cast<float4*>(tab + 256 + 4 * tid) = make_float4(w2r[0], p.ncls > 1 ? w2r[1] : 0.f, p.ncls > 2 ? w2r[2] : 0.f, p.ncls > 3 ? w2r[3] : 0.f);
}

// The same epilogue for a tile formed TRANSPOSED (bf16x6 kernel, MM_EP_DECONV_MASK_T: mfma(B, A) instead of mfma(A, B)): a lane then holds ONE
// pixel (row m0 + wm*64 + t*32 + l31) and 16 channels of each 32 x 32 block (register 4g+e = channel u*32 + 8g + 4*half + e of the wave's 128),
// so the channel sum of the 1x1 conv is a chain of FMAs in registers and ONE shuffle between the half-waves -- the untransposed form above
// multiplies per class and runs a 31-shuffle reduce-scatter butterfly per class and row block (~2100 instructions per wave against ~900; the
// epilogue was 0.24 of the kernel's 2.46 ms).  {bias, w2} of the tile's 256 channels wait in LDS (free after the loop).  With p.out set (Co ==
// 256) the two waves' slabs are summed in LDS in deconv_mask_finish's order and the sigmoid is stored here: no partial logits, no finish launch.
__device__ __forceinline__ void mm_deconv_mask_epilogue_t(const MMArgs& p, const f32x16 (&acc)[2][4], unsigned char* lds, long long m0, long long M, int n0,
                                                          int tid, int wm, int wn, int half, int l31)
{
    const float* tab = reinterpret_cast<const float*>(lds);       // bias [256], then w2 [256][4 classes] (mm_deconv_mask_table, filled in the kernel's prologue)
    float4* red = reinterpret_cast<float4*>(lds + 5120);          // [wm][wn][t][32 pixels] partial logits (p.out only)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int tap = n0 / p.Co;
    const int cb0 = n0 - tap * p.Co;
    // (opaque copies: nothing of the epilogue's index arithmetic may be scheduled above the main loop, which runs at 247 of 256 registers --
    //  hoisted there, the pixel / keep pointers cost 256 spills)
    asm volatile("" : "+v"(tid), "+v"(l31));
    const long long hw = (long long)p.H * p.W;
    long long pix[2];
    float* kdp[2] = {nullptr, nullptr};
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const long long row = m0 + wm * 64 + t * 32 + l31;
        pix[t] = -1;
        if (row < M) {
            const long long n_img = row / hw;
            const int rem = (int)(row - n_img * hw);
            const int y = rem / p.W, x = rem - y * p.W;
            const long long inimg = (long long)(2 * y + (tap >> 1)) * 2 * p.W + 2 * x + (tap & 1);
            pix[t] = n_img * 4 * hw + inimg;
            if (p.keep_d) {
                const int sl = p.keep_inv[n_img];
                if (sl >= 0 && sl < p.keep_cap) kdp[t] = p.keep_d + ((long long)sl * 4 * hw + inimg) * p.Co + cb0 + wn * 128 + 4 * half;
            }
        }
    }
    // the ReLU'd deconv rows of the images the caller keeps (few tiles hold one: a pass of its own, so that the class sums below stay one
    // straight line -- with the conditional stores inside it the allocator spilled 256 registers)
    if (__builtin_amdgcn_ballot_w64(kdp[0] != nullptr || kdp[1] != nullptr) != 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *reinterpret_cast<const float4*>(p.bias + cb0 + wn * 128 + u * 32 + 8 * g + 4 * half);
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    if (kdp[t])
                        *reinterpret_cast<float4*>(kdp[t] + u * 32 + 8 * g) =
                            make_float4(fmaxf(acc[t][u][4 * g] + b4.x, 0.f), fmaxf(acc[t][u][4 * g + 1] + b4.y, 0.f),
                                        fmaxf(acc[t][u][4 * g + 2] + b4.z, 0.f), fmaxf(acc[t][u][4 * g + 3] + b4.w, 0.f));
            }
    }
    // two groups of four channels per scheduling region: their ten 16-byte table reads are issued together and waited for once (one channel
    // at a time -- a read, a wait, eight FMAs -- the epilogue was slower than the butterfly it replaces: 0.29 against 0.25 ms); the four
    // class FMAs of a value as two packed ones (v_pk_fma_f32 on {class 0, 1} and {class 2, 3})
    f32x2 q01[2], q23[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { q01[t] = f32x2{0.f, 0.f}; q23[t] = f32x2{0.f, 0.f}; }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int gp = 0; gp < 2; ++gp) {
            float4 bb[2], ww[2][4];
#pragma unroll
            for (int gg = 0; gg < 2; ++gg) {
                const int cl = wn * 128 + u * 32 + 8 * (2 * gp + gg) + 4 * half;
                bb[gg] = *reinterpret_cast<const float4*>(tab + cl);
#pragma unroll
                for (int e = 0; e < 4; ++e) ww[gg][e] = *reinterpret_cast<const float4*>(tab + 256 + 4 * (cl + e));
            }
#pragma unroll
            for (int gg = 0; gg < 2; ++gg)
#pragma unroll
                for (int ep = 0; ep < 2; ++ep)
#pragma unroll
                    for (int t = 0; t < 2; ++t) {
                        const int r = 4 * (2 * gp + gg) + 2 * ep;
                        const f32x2 sv = f32x2{acc[t][u][r], acc[t][u][r + 1]} + f32x2{f4c(bb[gg], 2 * ep), f4c(bb[gg], 2 * ep + 1)};     // v_pk_add_f32
#pragma unroll
                        for (int k = 0; k < 2; ++k) {
                            const float a = fmaxf(k ? sv.y : sv.x, 0.f);
                            const f32x2 aa = {a, a};
                            const float4 w4 = ww[gg][2 * ep + k];
                            q01[t] = __builtin_elementwise_fma(aa, f32x2{w4.x, w4.y}, q01[t]);
                            q23[t] = __builtin_elementwise_fma(aa, f32x2{w4.z, w4.w}, q23[t]);
                        }
                    }
            __builtin_amdgcn_sched_barrier(0);
        }
    float ps[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) { ps[t][0] = q01[t].x; ps[t][1] = q01[t].y; ps[t][2] = q23[t].x; ps[t][3] = q23[t].y; }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) ps[t][c] += __shfl_xor(ps[t][c], 32, 64);
    if (p.out) {
        if (half == 0) {
#pragma unroll
            for (int t = 0; t < 2; ++t) red[((wm * 2 + wn) * 2 + t) * 32 + l31] = make_float4(ps[t][0], ps[t][1], ps[t][2], ps[t][3]);
        }
        __syncthreads();
        const int wn_s = __builtin_amdgcn_readfirstlane(wn);
#pragma unroll
        for (int t = 0; t < 2; ++t)                                // wave (wm, wn) finishes row block t = wn
            if (t == wn_s && half == 0 && pix[t] >= 0) {
                const float4 s0 = red[((wm * 2 + 0) * 2 + t) * 32 + l31], s1 = red[((wm * 2 + 1) * 2 + t) * 32 + l31];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < p.ncls) {
                        float sacc = p.b2[c];
                        sacc += f4c(s0, c);
                        sacc += f4c(s1, c);
                        p.out[pix[t] * p.ncls + c] = 1.f / (1.f + expf(-sacc));
                    }
            }
        return;
    }
    if (half == 0) {
        const int slab = (cb0 + wn * 128) >> 7;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (pix[t] < 0) continue;
            float* dst = p.part + ((long long)slab * 4 * M + pix[t]) * p.ncls;
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < p.ncls) dst[c] = ps[t][c];
        }
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void wino_mm_kernel(MMArgs p)
{
    __shared__ __attribute__((aligned(16))) float As[2][MM_BM * MM_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][MM_BN * MM_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int ntn = p.N / MM_BN;
    // workgroup b runs on XCD b % 8: give each XCD a contiguous run of tiles (consecutive tiles = consecutive row blocks of ONE
    // plane: its 256 KB of filters stay in that XCD's L2)
    long long bid;
    {
        const long long nwg = gridDim.x, orig = blockIdx.x;
        const long long q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    int ri = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < p.nruns && bid >= p.run[k].tile0) ri = k;
    const MMRun& R = p.run[ri];
    const long long local = bid - R.tile0;
    const long long per_plane = (long long)R.mtiles * ntn;
    const int z = (int)(local / per_plane);
    const long long rem = local - z * per_plane;
    const int n0 = (int)(rem % ntn) * MM_BN;
    const long long m0 = (rem / ntn) * MM_BM;
    const long long M = R.rows;
    const float* Ap = p.A + R.a_off + (long long)z * M * p.K;
    const float* Bp = p.Bt + R.b_off + (long long)z * p.K * p.N;
    float* Cp = p.C + R.c_off + (long long)z * M * p.N;
    const long long mend = (m0 + MM_BM < M) ? m0 + MM_BM : M;
    const __amdgpu_buffer_rsrc_t ra = mm_rsrc(Ap + m0 * p.K, (mend - m0) * p.K * 4);      // rows beyond M read 0
    const __amdgpu_buffer_rsrc_t rb = mm_rsrc(Bp + (long long)n0 * p.K, (long long)MM_BN * p.K * 4);

    // loader role: thread = (row r4 [+64 i], 16-byte slot kq) of the 16-deep chunk
    const int r4 = tid >> 2, kq = (tid & 3) * 4;
    const unsigned rowb = (unsigned)p.K * 4u;
    unsigned goff = ((unsigned)r4 * (unsigned)p.K + (unsigned)kq) * 4u;      // advanced by 64 bytes per chunk
    const int soff = r4 * MM_LD + kq;
    // two sets of staging registers: chunk c+3 is requested while chunk c is multiplied (a chunk is ~2 us of MFMA work; with one set
    // -- two chunks ahead -- the loads of a tile whose operands come from HBM under the other workgroups' write traffic arrived late)
    float4 sa[2][2], sb[2][4];
    auto gload = [&](int set) {
#pragma unroll
        for (int i = 0; i < 2; ++i) sa[set][i] = mm_bufld4(ra, goff + (unsigned)(64 * i) * rowb);
#pragma unroll
        for (int i = 0; i < 4; ++i) sb[set][i] = mm_bufld4(rb, goff + (unsigned)(64 * i) * rowb);
        goff += MM_BK * 4u;
    };
    auto sstore = [&](int buf, int set) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<float4*>(&As[buf][soff + 64 * i * MM_LD]) = sa[set][i];
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<float4*>(&Bs[buf][soff + 64 * i * MM_LD]) = sb[set][i];
    };
    // MFMA role: lane (l31, half) holds k = 8*half + 4*q + s of rows / columns l31 (+32 t, +32 u)
    const int aoff = (wm * 64 + l31) * MM_LD + half * 8;
    const int boff = (wn * 128 + l31) * MM_LD + half * 8;
    float4 fa[2][2], fb[2][4];          // [q][tile]
    auto fread = [&](int buf, int q) {
#pragma unroll
        for (int t = 0; t < 2; ++t) fa[q][t] = *reinterpret_cast<const float4*>(&As[buf][aoff + t * 32 * MM_LD + 4 * q]);
#pragma unroll
        for (int u = 0; u < 4; ++u) fb[q][u] = *reinterpret_cast<const float4*>(&Bs[buf][boff + u * 32 * MM_LD + 4 * q]);
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int nk = p.K / MM_BK;
    gload(0);
    sstore(0, 0);
    if (nk > 1) gload(1);
    if (nk > 2) gload(0);
    __syncthreads();
    fread(0, 0);

#define MM_STEP(q, s)                                                                                                  \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                                                      \
        _Pragma("unroll") for (int u = 0; u < 4; ++u)                                                                  \
            acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x2f32(f4c(fa[q][t], s), f4c(fb[q][u], s), acc[t][u], 0, 0, 0);

#define MM_CHUNK(c, cur)                                                                                               \
    {                                                                                                                  \
        fread(cur, 1);                       /* second-half fragments, used 32 MFMAs from now */                       \
        MM_STEP(0, 0) MM_STEP(0, 1) MM_STEP(0, 2) MM_STEP(0, 3)                                                        \
        if ((c) + 1 < nk) sstore(cur ^ 1, cur ^ 1);   /* chunk c+1 (requested two chunks ago) -> the buffer chunk c-1 used */ \
        if ((c) + 3 < nk) gload(cur ^ 1);             /* chunk c+3 into the set just emptied */                          \
        MM_STEP(1, 0)                                                                                                  \
        __syncthreads();                                                                                               \
        if ((c) + 1 < nk) fread(cur ^ 1, 0); /* first-half fragments of chunk c+1 under the rest of this chunk */      \
        MM_STEP(1, 1) MM_STEP(1, 2) MM_STEP(1, 3)                                                                      \
    }
    int c = 0;
    for (; c + 1 < nk; c += 2) {             // two chunks per trip: the LDS buffer and the staging set of a chunk are its parity
        MM_CHUNK(c, 0)
        MM_CHUNK(c + 1, 1)
    }
    if (c < nk) MM_CHUNK(c, 0)
#undef MM_CHUNK
#undef MM_STEP

    if constexpr (EPI == MM_EP_DECONV_MASK) {
        mm_deconv_mask_epilogue(p, acc, m0, M, n0, wm, wn, half, l31);
        return;
    }
    // ---- epilogue: 128-byte row segments per 32 lanes ----
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= M) continue;
            float* dst = Cp + row * p.N + n0 + wn * 128 + l31;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (p.nt) __builtin_nontemporal_store(acc[t][u][r], dst + 32 * u);
                else dst[32 * u] = acc[t][u][r];
            }
        }
}

// =====================================================================================================================
// The same product on the bf16 matrix pipe, at fp32 accuracy ("bf16x6"): every fp32 operand is split EXACTLY into three bf16
// pieces (x = x1 + x2 + x3: 8 + 8 + 8 significand bits), and a*b is accumulated in fp32 from the six piece products that are
// not below fp32 resolution:  a1b3 + a3b1 + a2b2 + a1b2 + a2b1 + a1b1  (dropped: a2b3, a3b2, a3b3 <= 2^-25 |ab|).
// bf16 x bf16 products are exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the result carries the same
// kind of error as the fp32 MFMA path -- measured against an fp64 reference it is slightly SMALLER (96 instead of 128 fp32
// roundings per 256-deep dot product; tests/test_gpu_ops.py::test_wino_multiply_bf16x6_accuracy, tools/split_bf16_numerics.py)
// -- while six 32-cycle bf16 MFMAs replace eight 64-cycle fp32 MFMAs per 32x32x16 block: 2.67x less matrix-pipe time.
//   A (activations V): fp32 in HBM as before; split in the loader by truncation (and / sub / perm: 44 VALU per 8 elements) into
//                LDS row records [p0h0 | p0h1 | p1h0 | p1h1 | p2h0 | p2h1 | pad] x 16 bytes = 112 bytes (piece p, k half h): a
//                lane's operand (row l&31, k = 8*(l>>5) .. +7) is one ds_read_b128 and 16 consecutive rows at one slot cover all
//                64 banks.  Double-buffered (2 x 14 KB), one barrier per 16-deep chunk.
//   B (filters): split once by wino_w_kernel (round-to-nearest) and stored in MFMA operand order
//                [plane][chunk][n / 32][piece][k half][n % 32][8 bf16]: a wave's B fragment is 1 KB contiguous and goes
//                straight from L2 to registers (the 9.4 MB of filters are L2 / MALL resident), one chunk ahead of its use --
//                no LDS traffic and no barrier for two thirds of the operand bytes.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define X6_REC 112            // bytes per LDS row record

__device__ __forceinline__ unsigned x6_top(float lo, float hi) { return __builtin_amdgcn_perm(__float_as_uint(hi), __float_as_uint(lo), 0x07060302u); }
__device__ __forceinline__ float x6_rest(float x) { return x - __uint_as_float(__float_as_uint(x) & 0xffff0000u); }
__device__ __forceinline__ bf16x8 x6_ldb(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff)
{
    u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
    return __builtin_bit_cast(bf16x8, v);
}

template <int V> struct IntK { static constexpr int value = V; };
// SWAP: the product transposed (rows = B's columns): both operands have the same fragment layout (lane & 31 = row / column, lane >> 5 = k half)
template <bool SWAP> __device__ __forceinline__ f32x16 x6_mfma(bf16x8 a, bf16x8 b, f32x16 c)
{
    if constexpr (SWAP) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ float mm_act(float v, int act)
{
    if (act == MYOLO_ACT_RELU) return fmaxf(v, 0.f);
    if (act == MYOLO_ACT_RELU6) return fminf(fmaxf(v, 0.f), 6.f);
    return v;
}

// NU = 32-column blocks per wave: 4 (tile 128 x 256) or 2 (tile 128 x 128, twice the workgroups: for the products whose 128 x 256 tiles
// do not fill the chip -- the 14x14 / 7x7 pointwise layers of the trunk; each A element is then split by two workgroups instead of one)
template <int EPI, bool PW = false, int AG = MM_A_PLAIN, int NU = 4>
__global__ __launch_bounds__(256, 2) void wino_mm_x6_kernel(MMArgs p)
{
    static_assert(NU == 4 || (NU == 2 && EPI == MM_EP_PLAIN && AG == MM_A_PLAIN), "the 128-column tile exists for the plain product only");
    constexpr int BN = 64 * NU;
    __shared__ __attribute__((aligned(16))) unsigned char As[2][MM_BM * X6_REC];
    __shared__ __attribute__((aligned(16))) unsigned char etab[EPI == MM_EP_DECONV_MASK_T ? 5120 + 4096 : 16];      // {bias, w2} table + the slab exchange

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, half = lane >> 5, l31 = lane & 31;
    const int ntn = p.N / BN;
    long long bid;
    {
        const long long nwg = gridDim.x, orig = blockIdx.x;
        const long long q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    int ri = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < p.nruns && bid >= p.run[k].tile0) ri = k;
    const MMRun& R = p.run[ri];
    const long long local = bid - R.tile0;
    const long long per_plane = (long long)R.mtiles * ntn;
    const int z = (int)(local / per_plane);
    const long long rem = local - z * per_plane;
    const int n0 = (int)(rem % ntn) * BN;
    const long long m0 = (rem / ntn) * MM_BM;
    const long long M = R.rows;
    const int nk = p.K / MM_BK;
    const float* Ap = p.A + R.a_off + (long long)z * M * p.K;
    // split filters: 6 bytes per element; b_off counts ELEMENTS of the plane sequence
    const unsigned char* Bp = reinterpret_cast<const unsigned char*>(p.Bt) + (R.b_off + (long long)z * p.K * p.N) * 6;
    float* Cp = p.C + R.c_off + (long long)z * M * p.N;
    const long long mend = (m0 + MM_BM < M) ? m0 + MM_BM : M;
    // A loader: thread = (row tid>>1, k half tid&1): 8 consecutive floats
    const int arow = tid >> 1, ah = tid & 1;
    __amdgpu_buffer_rsrc_t ra;
    unsigned aoffg;                                                                        // + 64 bytes per chunk
    unsigned ag_base = 0;                                                                  // AG: this row's tap-(0,0) pixel, bytes
    if constexpr (AG == MM_A_DECONV) {
        // A row m, column k = tap*Cc + c = dy[pixel(m, tap)][c], pixel(m, tap) = 4m - 2x + ky*2W + kx on the [N,2H,2W] grid (Cc = p.Co)
        long long base_px = 4 * m0 - 2 * p.W; if (base_px < 0) base_px = 0;
        long long end_px = 4 * (m0 + MM_BM) + 2 * p.W + 2; if (end_px > 4 * M) end_px = 4 * M;
        ra = mm_rsrc(Ap + base_px * p.Co, (end_px - base_px) * p.Co * 4);
        const long long mm = m0 + arow;
        if (mm < M) {
            const long long hw = (long long)p.H * p.W;
            const int rem = (int)(mm - (mm / hw) * hw);
            const int x = rem - (rem / p.W) * p.W;
            ag_base = (unsigned)((4 * mm - 2 * x - base_px) * p.Co + ah * 8) * 4u;
        } else ag_base = MM_OOB;
        aoffg = ag_base;
    } else {
        ra = mm_rsrc(Ap + m0 * p.K, (mend - m0) * p.K * 4);                                // rows beyond M read 0
        aoffg = ((unsigned)arow * (unsigned)p.K + (unsigned)ah * 8u) * 4u;
    }
    const __amdgpu_buffer_rsrc_t rb = mm_rsrc(reinterpret_cast<const float*>(Bp), (long long)p.K * p.N * 6);
    int ag_k = 0;                                                                          // AG: first k of the next chunk to load
    const int asto = arow * X6_REC + ah * 16;                                              // + piece * 32
    float4 sa[2];
    float4 psc[2], psh[2];                     // PW: this chunk's per-k affine (k = kch + ah*8 .. +7)
    int kch = ah * 8;
    auto gload = [&]() {
        if constexpr (AG == MM_A_DECONV) {
            const int tap = ag_k / p.Co, c0 = ag_k - tap * p.Co;
            const unsigned sh = (unsigned)(((tap >> 1) * 2 * p.W + (tap & 1)) * p.Co + c0) * 4u;
            aoffg = ag_base == MM_OOB ? MM_OOB : ag_base + sh;
            ag_k += MM_BK;
        }
        sa[0] = mm_bufld4(ra, aoffg);
        sa[1] = mm_bufld4(ra, aoffg == MM_OOB ? MM_OOB : aoffg + 16u);
#ifdef MM_X6_TUNE
        if constexpr (AG == MM_A_PLAIN) { if (!(p.tune & 16)) aoffg += MM_BK * 4u; }
#else
        if constexpr (AG == MM_A_PLAIN) aoffg += MM_BK * 4u;
#endif
        if (PW && p.a_scale) {
            psc[0] = *reinterpret_cast<const float4*>(p.a_scale + kch); psc[1] = *reinterpret_cast<const float4*>(p.a_scale + kch + 4);
            psh[0] = *reinterpret_cast<const float4*>(p.a_shift + kch); psh[1] = *reinterpret_cast<const float4*>(p.a_shift + kch + 4);
            kch += MM_BK;
        }
    };
    auto affine = [&]() {                      // the producing layer's BatchNorm + activation on the staged chunk (rows beyond M are never stored / counted)
        if (PW && p.a_scale) {
            sa[0].x = mm_act(fmaf(sa[0].x, psc[0].x, psh[0].x), p.a_act); sa[0].y = mm_act(fmaf(sa[0].y, psc[0].y, psh[0].y), p.a_act);
            sa[0].z = mm_act(fmaf(sa[0].z, psc[0].z, psh[0].z), p.a_act); sa[0].w = mm_act(fmaf(sa[0].w, psc[0].w, psh[0].w), p.a_act);
            sa[1].x = mm_act(fmaf(sa[1].x, psc[1].x, psh[1].x), p.a_act); sa[1].y = mm_act(fmaf(sa[1].y, psc[1].y, psh[1].y), p.a_act);
            sa[1].z = mm_act(fmaf(sa[1].z, psc[1].z, psh[1].z), p.a_act); sa[1].w = mm_act(fmaf(sa[1].w, psc[1].w, psh[1].w), p.a_act);
        }
    };
    auto sstore = [&](int buf) {
        affine();
        const float x[8] = {sa[0].x, sa[0].y, sa[0].z, sa[0].w, sa[1].x, sa[1].y, sa[1].z, sa[1].w};
        float r1[8], r2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { r1[e] = x6_rest(x[e]); r2[e] = x6_rest(r1[e]); }
        u32x4 p1, p2, p3;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            p1[e] = x6_top(x[2 * e], x[2 * e + 1]);
            p2[e] = x6_top(r1[2 * e], r1[2 * e + 1]);
            p3[e] = x6_top(r2[2 * e], r2[2 * e + 1]);
        }
        *reinterpret_cast<u32x4*>(As[buf] + asto) = p1;
        *reinterpret_cast<u32x4*>(As[buf] + asto + 32) = p2;
        *reinterpret_cast<u32x4*>(As[buf] + asto + 64) = p3;
    };
    const int afr = (wm * 64 + l31) * X6_REC + half * 16;          // + t * 32 rows, + piece * 32 bytes
    // B fragments: tile (n0 + wn*128)/32 + u, piece pc, this lane's k half and column: 16 bytes at
    //   ((chunk * N/32 + tile) * 6 + pc * 2 + half) * 512 + l31 * 16
    const unsigned bvo = (unsigned)(((n0 + wn * 32 * NU) >> 5) * 6 + half) * 512u + (unsigned)l31 * 16u;     // + u * 3072 + pc * 1024
    const unsigned bchunk = (unsigned)(p.N >> 5) * 3072u;
    unsigned bso = 0;                                                                                    // chunk offset (scalar)
    bf16x8 bq[NU][3];

    f32x16 acc[2][NU];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    gload();
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) bq[u][pc] = x6_ldb(rb, bvo + u * 3072u + pc * 1024u, bso);
    sstore(0);
    if (nk > 1) gload();
    if constexpr (EPI == MM_EP_DECONV_MASK_T) mm_deconv_mask_table(p, etab, n0, tid);
    __syncthreads();
    bf16x8 fa[2][3];
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) fa[0][pc] = *reinterpret_cast<const bf16x8*>(As[0] + afr + pc * 32);

    if constexpr (NU == 2 && AG == MM_A_PLAIN && EPI == MM_EP_PLAIN) {
        // ---- the 128 x 128 tile (14 x 14 / 7 x 7 pointwise layers, under one workgroup per CU): 24 MFMAs per wave and chunk are a third of a global
        // load's latency, and with the operands of chunk c+2 requested during chunk c the loop ran at one load latency per chunk (1.5 us, whatever
        // the tile width; profiles/r4_notes.md section 5).  Here chunk k's A values wait in register stage k % 3 (requested THREE chunks ahead) and
        // its B fragments in stage k % 2 (two ahead).  The loop is unrolled by hand over the period of those queues and of the two LDS buffers
        // (6), so that a stage is a register NAME: a stage that moved up by a copy made hipcc wait vmcnt(0) at the end of every iteration, and
        // so did a load inside a conditional block -- the loads past the last chunk are issued anyway, out of range / at k = 0.
        float4 qa[3][2], qsc[3][2], qsh[3][2];
        bf16x8 qb[2][NU][3];
        const float* csc = (PW && p.a_scale) ? p.a_scale : p.A;        // (never dereferenced beyond the first chunk's 16 floats when unused)
        const float* csh = (PW && p.a_scale) ? p.a_shift : p.A;
        const bool has_aff = PW && p.a_scale;
        auto gstage = [&](auto st, bool valid) {
            constexpr int S = decltype(st)::value;
#ifdef MM_X6_TUNE
            if (!(p.tune & 2048)) {
#endif
            qa[S][0] = mm_bufld4(ra, valid ? aoffg : MM_OOB);
            qa[S][1] = mm_bufld4(ra, valid ? aoffg + 16u : MM_OOB);
#ifdef MM_X6_TUNE
            }
#endif
            aoffg += MM_BK * 4u;
#ifdef MM_X6_TUNE
            if (p.tune & 8192) return;
#endif
            if constexpr (PW) {
                const int kc = (valid && has_aff) ? kch : ah * 8;
                qsc[S][0] = *reinterpret_cast<const float4*>(csc + kc); qsc[S][1] = *reinterpret_cast<const float4*>(csc + kc + 4);
                qsh[S][0] = *reinterpret_cast<const float4*>(csh + kc); qsh[S][1] = *reinterpret_cast<const float4*>(csh + kc + 4);
                kch += MM_BK;
            }
        };
        // the prologue above left chunk 1 in sa / psc / psh and chunk 0's B fragments in bq
        qa[1][0] = sa[0]; qa[1][1] = sa[1];
        if constexpr (PW) { qsc[1][0] = psc[0]; qsc[1][1] = psc[1]; qsh[1][0] = psh[0]; qsh[1][1] = psh[1]; }
#pragma unroll
        for (int u = 0; u < NU; ++u)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) {
                qb[0][u][pc] = bq[u][pc];
                qb[1][u][pc] = x6_ldb(rb, bvo + u * 3072u + pc * 1024u, nk > 1 ? bchunk : 0u);
            }
        gstage(IntK<2>{}, nk > 2);
        gstage(IntK<0>{}, nk > 3);
        auto step = [&](auto ic, int c) {
            constexpr int I = decltype(ic)::value;
            constexpr int cur = I & 1, SA = (I + 1) % 3, SB = I & 1;
            const bool more = c + 1 < nk;
#ifdef MM_X6_TUNE
            if (!(p.tune & 16384))
#endif
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) fa[1][pc] = *reinterpret_cast<const bf16x8*>(As[cur] + afr + 32 * X6_REC + pc * 32);
            float x[8] = {qa[SA][0].x, qa[SA][0].y, qa[SA][0].z, qa[SA][0].w, qa[SA][1].x, qa[SA][1].y, qa[SA][1].z, qa[SA][1].w};
            if constexpr (PW) {
                if (has_aff) {
                    const float sc8[8] = {qsc[SA][0].x, qsc[SA][0].y, qsc[SA][0].z, qsc[SA][0].w, qsc[SA][1].x, qsc[SA][1].y, qsc[SA][1].z, qsc[SA][1].w};
                    const float sh8[8] = {qsh[SA][0].x, qsh[SA][0].y, qsh[SA][0].z, qsh[SA][0].w, qsh[SA][1].x, qsh[SA][1].y, qsh[SA][1].z, qsh[SA][1].w};
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = mm_act(fmaf(x[e], sc8[e], sh8[e]), p.a_act);
                }
            }
            gstage(IntK<SA>{}, c + 4 < nk);                              // chunk c+4 takes the stage chunk c+1 just left
            const unsigned b2 = c + 2 < nk ? bso + 2u * bchunk : bso;
            u32x4 p1, p2, p3;
            bf16x8 fn0, fn1, fn2;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
#ifdef MM_X6_TUNE
#define X6_GO (!(p.tune & 512))
#else
#define X6_GO true
#endif
#define X6_TILE(t)  if (X6_GO) {                                                                                         \
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t][0], qb[SB][u][2], acc[t][u], 0, 0, 0);       \
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t][2], qb[SB][u][0], acc[t][u], 0, 0, 0);       \
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t][1], qb[SB][u][1], acc[t][u], 0, 0, 0);       \
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t][0], qb[SB][u][1], acc[t][u], 0, 0, 0);       \
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t][1], qb[SB][u][0], acc[t][u], 0, 0, 0);       \
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t][0], qb[SB][u][0], acc[t][u], 0, 0, 0); }
                X6_TILE(0)
                X6_TILE(1)
#undef X6_TILE
#undef X6_GO
#ifdef MM_X6_TUNE
                if (!(p.tune & 1024))
#endif
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) qb[SB][u][pc] = x6_ldb(rb, bvo + u * 3072u + pc * 1024u, b2);      // chunk c+2 into the stage chunk c leaves
                if (u == 0) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a0 = x[2 * e], a1 = x[2 * e + 1];
                        const float b0 = x6_rest(a0), b1 = x6_rest(a1);
                        p1[e] = x6_top(a0, a1);
                        p2[e] = x6_top(b0, b1);
                        p3[e] = x6_top(x6_rest(b0), x6_rest(b1));
                    }
#ifdef MM_X6_TUNE
                    if (more && !(p.tune & 4096)) {
#else
                    if (more) {
#endif
                        *reinterpret_cast<u32x4*>(As[cur ^ 1] + asto) = p1;
                        *reinterpret_cast<u32x4*>(As[cur ^ 1] + asto + 32) = p2;
                        *reinterpret_cast<u32x4*>(As[cur ^ 1] + asto + 64) = p3;
                    }
                } else {
#ifdef MM_X6_TUNE
                    if (!(p.tune & 4096))
#endif
                    __syncthreads();
#ifdef MM_X6_TUNE
                    if (more && !(p.tune & 16384)) {
#else
                    if (more) {
#endif
                        fn0 = *reinterpret_cast<const bf16x8*>(As[cur ^ 1] + afr);
                        fn1 = *reinterpret_cast<const bf16x8*>(As[cur ^ 1] + afr + 32);
                        fn2 = *reinterpret_cast<const bf16x8*>(As[cur ^ 1] + afr + 64);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (more) { fa[0][0] = fn0; fa[0][1] = fn1; fa[0][2] = fn2; }
            bso = more ? bso + bchunk : bso;
        };
        int c0 = 0;
        for (; c0 + 6 <= nk; c0 += 6) {
            step(IntK<0>{}, c0); step(IntK<1>{}, c0 + 1); step(IntK<2>{}, c0 + 2);
            step(IntK<3>{}, c0 + 3); step(IntK<4>{}, c0 + 4); step(IntK<5>{}, c0 + 5);
        }
        if (c0 < nk) step(IntK<0>{}, c0);
        if (c0 + 1 < nk) step(IntK<1>{}, c0 + 1);
        if (c0 + 2 < nk) step(IntK<2>{}, c0 + 2);
        if (c0 + 3 < nk) step(IntK<3>{}, c0 + 3);
        if (c0 + 4 < nk) step(IntK<4>{}, c0 + 4);
    } else
    // One barrier per chunk, and no LDS read is waited for right behind its issue:
    //   u = 0, 1: row tile 1's fragments of THIS chunk are read first thing (used six MFMAs later: each column tile runs row tile 0's
    //             six terms, then row tile 1's); chunk c+1, in registers since the previous chunk, is split into the other buffer
    //             in two slices behind these column tiles' MFMAs (that buffer's last readers finished before the previous barrier);
    //   u = 2:    barrier;
    //   u = 3:    row tile 0's fragments of chunk c+1 are read into the (now dead) registers of this chunk's.
    // The B fragments of chunk c+1 replace each column tile's registers as soon as its MFMAs are issued.
    for (int c = 0; c < nk; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < nk;
#ifdef MM_X6_TUNE
        if (!(p.tune & 64))
#endif
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) fa[1][pc] = *reinterpret_cast<const bf16x8*>(As[cur] + afr + 32 * X6_REC + pc * 32);
        affine();                                                       // (PW) chunk c+1's BatchNorm + activation, with the coefficients its gload fetched
        const float x[8] = {sa[0].x, sa[0].y, sa[0].z, sa[0].w, sa[1].x, sa[1].y, sa[1].z, sa[1].w};
        u32x4 p1, p2, p3;
#ifdef MM_X6_TUNE
        if (c + 2 < nk && !(p.tune & 128)) gload();
#else
        if (c + 2 < nk) gload();                                        // chunk c+2 (x[] holds copies of chunk c+1)
#endif
#ifdef MM_X6_TUNE
        const unsigned bnext = (p.tune & 1) ? 0u : (more ? bso + bchunk : bso);
#else
        const unsigned bnext = more ? bso + bchunk : bso;               // the last chunk re-reads itself (into registers nobody uses)
#endif
        bf16x8 fn0, fn1, fn2;                                           // row tile 0 of chunk c+1
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            // smallest terms first
#define X6_TILE(t)                                                                                                 \
            if (X6_ALL) {                                                                                          \
            acc[t][u] = x6_mfma<EPI == MM_EP_DECONV_MASK_T>(fa[t][0], bq[u][2], acc[t][u]);           \
            acc[t][u] = x6_mfma<EPI == MM_EP_DECONV_MASK_T>(fa[t][2], bq[u][0], acc[t][u]);           \
            acc[t][u] = x6_mfma<EPI == MM_EP_DECONV_MASK_T>(fa[t][1], bq[u][1], acc[t][u]); }         \
            acc[t][u] = x6_mfma<EPI == MM_EP_DECONV_MASK_T>(fa[t][0], bq[u][1], acc[t][u]);           \
            acc[t][u] = x6_mfma<EPI == MM_EP_DECONV_MASK_T>(fa[t][1], bq[u][0], acc[t][u]);           \
            acc[t][u] = x6_mfma<EPI == MM_EP_DECONV_MASK_T>(fa[t][0], bq[u][0], acc[t][u]);
#ifdef MM_X6_TUNE
#define X6_ALL (!(p.tune & 256))          /* 256: only three of the six piece products (timing only: what a three-product scheme could cost at most) */
#else
#define X6_ALL true
#endif
            X6_TILE(0)
            X6_TILE(1)
#undef X6_TILE
#undef X6_ALL
#ifdef MM_X6_TUNE
            if (!(p.tune & 2))
#endif
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) bq[u][pc] = x6_ldb(rb, bvo + u * 3072u + pc * 1024u, bnext);      // same registers, next chunk
            if (u < NU / 2) {                                           // the split of chunk c+1 in NU/2 slices
#pragma unroll
                for (int e = (8 / NU) * u; e < (8 / NU) * (u + 1); ++e) {
                    const float a0 = x[2 * e], a1 = x[2 * e + 1];
#ifdef MM_X6_TUNE
                    if (p.tune & 4) { p1[e] = x6_top(a0, a1); p2[e] = p1[e]; p3[e] = p1[e]; continue; }
#endif
                    const float b0 = x6_rest(a0), b1 = x6_rest(a1);
                    p1[e] = x6_top(a0, a1);
                    p2[e] = x6_top(b0, b1);
                    p3[e] = x6_top(x6_rest(b0), x6_rest(b1));
                }
#ifdef MM_X6_TUNE
                if (u == NU / 2 - 1 && more && !(p.tune & 32)) {
#else
                if (u == NU / 2 - 1 && more) {
#endif
                    *reinterpret_cast<u32x4*>(As[cur ^ 1] + asto) = p1;
                    *reinterpret_cast<u32x4*>(As[cur ^ 1] + asto + 32) = p2;
                    *reinterpret_cast<u32x4*>(As[cur ^ 1] + asto + 64) = p3;
                }
            } else if (u == NU / 2) {
#ifdef MM_X6_TUNE
                if (!(p.tune & 32))
#endif
                __syncthreads();
#ifdef MM_X6_TUNE
                if (more && !(p.tune & 64)) {
#else
                if (more) {
#endif
                    fn0 = *reinterpret_cast<const bf16x8*>(As[cur ^ 1] + afr);
                    fn1 = *reinterpret_cast<const bf16x8*>(As[cur ^ 1] + afr + 32);
                    fn2 = *reinterpret_cast<const bf16x8*>(As[cur ^ 1] + afr + 64);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) { fa[0][0] = fn0; fa[0][1] = fn1; fa[0][2] = fn2; }
        bso = bnext;
    }

    if constexpr (EPI == MM_EP_DECONV_MASK) {
#ifdef MM_X6_TUNE
        if (p.tune & 32768) return;        // timing only: what the epilogue costs
#endif
        mm_deconv_mask_epilogue(p, acc, m0, M, n0, wm, wn, half, l31);
        return;
    }
    if constexpr (EPI == MM_EP_DECONV_MASK_T) {
#ifdef MM_X6_TUNE
        if (p.tune & 32768) return;
#endif
        mm_deconv_mask_epilogue_t(p, acc, etab, m0, M, n0, tid, wm, wn, half, l31);
        return;
    }
    if constexpr (PW) {
        if (p.stat) {
            // column sums of the tile (BatchNorm statistics of the conv's output): lane -> its 32 row slots, the other half-wave, the two
            // waves sharing the columns through LDS (free after the loop); one row of partials per row tile: fixed-order finish
            float s1[NU], s2[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) { s1[u] = 0.f; s2[u] = 0.f; }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const long long row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (row >= M) continue;
#pragma unroll
                    for (int u = 0; u < NU; ++u) { const float v = acc[t][u][r]; s1[u] += v; s2[u] = fmaf(v, v, s2[u]); }
                }
            __syncthreads();                                     // (every wave is past its last fragment read)
            float* sred = reinterpret_cast<float*>(&As[0][0]);   // [2 (wm)][2 (sum, sumsq)][256]
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                s1[u] += __shfl_xor(s1[u], 32, 64);
                s2[u] += __shfl_xor(s2[u], 32, 64);
                if (half == 0) {
                    sred[(wm * 2 + 0) * BN + wn * 32 * NU + u * 32 + l31] = s1[u];
                    sred[(wm * 2 + 1) * BN + wn * 32 * NU + u * 32 + l31] = s2[u];
                }
            }
            __syncthreads();
            for (int e = tid; e < 2 * BN; e += 256) {
                const int v = e / BN, cc = e - v * BN;
                p.stat[((m0 / MM_BM) * 2 + v) * p.N + n0 + cc] = (double)sred[(0 * 2 + v) * BN + cc] + (double)sred[(1 * 2 + v) * BN + cc];
            }
        }
    }
    if constexpr (EPI == MM_EP_DECONV) {
        // Conv2DTranspose 2x2 / s2 (model.py:711-712): this tile's 256 columns are 256 channels of ONE tap (Co % 256 == 0); row m of the
        // GEMM = input pixel (n, y, x) -> output pixel (n, 2y + ky, 2x + kx); + bias, activation
        const int tap = n0 / p.Co;
        const int cb = n0 - tap * p.Co + wn * 128 + l31;
        float bv[4];
#pragma unroll
        for (int u = 0; u < NU; ++u) bv[u] = p.bias ? p.bias[cb + 32 * u] : 0.f;
        const long long hw = (long long)p.H * p.W;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row >= M) continue;
                const long long n_img = row / hw;
                const int rem = (int)(row - n_img * hw);
                const int y = rem / p.W, x = rem - y * p.W;
                float* dst = p.C + (n_img * 4 * hw + (long long)(2 * y + (tap >> 1)) * 2 * p.W + 2 * x + (tap & 1)) * p.Co + cb;
#pragma unroll
                for (int u = 0; u < NU; ++u) dst[32 * u] = mm_act(acc[t][u][r] + bv[u], p.a_act);
            }
        return;
    }
#ifdef MM_X6_TUNE
    if ((p.tune & 8) && acc[0][0][0] != 12345.678f) return;
#endif
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= M) continue;
            float* dst = Cp + row * p.N + n0 + wn * 32 * NU + l31;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                if (p.nt) __builtin_nontemporal_store(acc[t][u][r], dst + 32 * u);
                else dst[32 * u] = acc[t][u][r];
            }
        }
}

/* whether the Winograd multiply of a (K = Cin, N = Cout) layer runs here (and the filters are therefore stored transposed) */
// fewer 128 x 256 tiles than CUs: 128 x 128 tiles (twice the workgroups) instead; option x6_no_half_tiles = 1 is the ablation
static bool x6_half_tiles(long long tiles) { return !g_myolo_opt.x6_no_half_tiles && tiles < 256; }

bool myolo_gemm_nt_batched_ok(int K, int N) { return !g_myolo_opt.wino_no_bt && K >= MM_BK && (K % MM_BK) == 0 && (N % MM_BN) == 0; }
/* ... and whether it runs as six bf16 piece products per fp32 product (option "wino_x6"; the filters are stored split then) */
bool myolo_gemm_nt_batched_x6(int K, int N) { return g_myolo_opt.wino_x6 && myolo_gemm_nt_batched_ok(K, N); }

/* For every run r < nruns and plane z < nq[r]:  C_r[z] (rows[r] x N) = A_r[z] (rows[r] x K) * Bt_r[z]^T, with
 * A_r = A + a_off[r] (planes rows[r]*K elements apart), Bt_r = Bt + b_off[r] (K*N apart), C_r = C + c_off[r] (rows[r]*N apart).
 * Needs myolo_gemm_nt_batched_ok(K, N) and 16-byte aligned operands; ONE launch. */
int myolo_gemm_nt_batched_runs(const float* A, const float* Bt, float* C, int nruns, const long long* rows, const long long* a_off,
                               const long long* b_off, const long long* c_off, const int* nq, int K, int N, hipStream_t s)
{
    if (K < MM_BK || (K % MM_BK) || (N % MM_BN) || ((uintptr_t)A & 15) || ((uintptr_t)Bt & 15) || ((uintptr_t)C & 15) || nruns < 0 || nruns > 4) {
        myolo_set_error("gemm_nt_batched_runs: needs K %% %d == 0, N %% %d == 0, <= 4 runs and 16-byte aligned operands", MM_BK, MM_BN);
        return MYOLO_EINVAL;
    }
    MMArgs a{};
    a.A = A; a.Bt = Bt; a.C = C; a.K = K; a.N = N; a.nt = g_myolo_opt.wino_nt ? 1 : 0;
    a.tune = g_myolo_opt.tune0;
    long long tiles = 0;
    for (int r = 0; r < nruns; ++r) {
        if (rows[r] <= 0 || nq[r] <= 0) continue;
        if ((a_off[r] | b_off[r] | c_off[r]) & 3) { myolo_set_error("gemm_nt_batched_runs: run offsets must be multiples of 4 elements"); return MYOLO_EINVAL; }
        MMRun& R = a.run[a.nruns++];
        R.rows = rows[r]; R.a_off = a_off[r]; R.b_off = b_off[r]; R.c_off = c_off[r]; R.nq = nq[r];
        R.mtiles = (int)((rows[r] + MM_BM - 1) / MM_BM);
        R.tile0 = tiles;
        tiles += (long long)R.mtiles * (N / MM_BN) * nq[r];
    }
    if (tiles <= 0) return MYOLO_OK;
    if (g_myolo_opt.wino_x6) hipLaunchKernelGGL(wino_mm_x6_kernel<MM_EP_PLAIN>, dim3((unsigned)tiles), dim3(256), 0, s, a);      // Bt = split filters
    else hipLaunchKernelGGL(wino_mm_kernel<MM_EP_PLAIN>, dim3((unsigned)tiles), dim3(256), 0, s, a);
    return MYOLO_OK;
}

// ---- split filters for a plain [K][N]-shaped operand (not a Winograd plane sequence) ----
// out: the operand order of wino_mm_x6_kernel, [k / 16][n / 32][piece][(k / 8) % 2][n % 32][k % 8] bf16, of B[k][n] = src[n * K + k]
// (src_kn = 1: B[k][n] = src[k * N + n], the natural [K][N] layout of a Keras 1x1 kernel)
__device__ __forceinline__ void x6_split_one(const float* __restrict__ src, __bf16* __restrict__ out, int K, int N, int src_kn, long long idx)
{
    int n, k;
    if (src_kn) { k = (int)(idx / N); n = (int)(idx - (long long)k * N); }
    else { n = (int)(idx / K); k = (int)(idx - (long long)n * K); }
    const float v = src[idx];
    const __bf16 p1 = (__bf16)v;
    const float r1 = v - (float)p1;
    const __bf16 p2 = (__bf16)r1;
    const __bf16 p3 = (__bf16)(r1 - (float)p2);
    __bf16* rec = out + ((((long long)(k >> 4) * (N >> 5) + (n >> 5)) * 6 + ((k >> 3) & 1)) * 256) + (n & 31) * 8 + (k & 7);
    rec[0] = p1; rec[512] = p2; rec[1024] = p3;
}
__global__ __launch_bounds__(256) void x6_split_nk_kernel(const float* __restrict__ src, __bf16* __restrict__ out, int K, int N, int src_kn = 0)
{
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)K * N) return;
    x6_split_one(src, out, K, N, src_kn, idx);
}
// several weight matrices in ONE launch (the prepared-weights registry re-splits ~17 of them at the start of every training step: one launch instead of 17)
#define X6_BATCH_MAX 24
struct X6SplitBatch {
    const float* src[X6_BATCH_MAX];
    __bf16* out[X6_BATCH_MAX];
    int K[X6_BATCH_MAX], N[X6_BATCH_MAX], kn[X6_BATCH_MAX];
    unsigned blk0[X6_BATCH_MAX + 1];       // first workgroup of every matrix
    int n;
};
__global__ __launch_bounds__(256) void x6_split_batched_kernel(X6SplitBatch b)
{
    int i = 0;
#pragma unroll 1
    for (int j = 1; j < b.n; ++j)
        if (blockIdx.x >= b.blk0[j]) i = j;
    const long long idx = (long long)(blockIdx.x - b.blk0[i]) * 256 + threadIdx.x;
    if (idx >= (long long)b.K[i] * b.N[i]) return;
    x6_split_one(b.src[i], b.out[i], b.K[i], b.N[i], b.kn[i], idx);
}
/* csrc/mem_kernels.hip (myolo_wprep_refresh): n <= X6_BATCH_MAX splits in one launch; the same bytes as n calls of x6_split_nk_kernel */
int myolo_x6_split_batched(int n, const void* const* src, void* const* dst, const long long* K, const long long* N, const long long* kn, hipStream_t s)
{
    for (int base = 0; base < n; base += X6_BATCH_MAX) {
        X6SplitBatch b{};
        b.n = n - base < X6_BATCH_MAX ? n - base : X6_BATCH_MAX;
        unsigned blocks = 0;
        for (int i = 0; i < b.n; ++i) {
            b.src[i] = (const float*)src[base + i]; b.out[i] = (__bf16*)dst[base + i];
            b.K[i] = (int)K[base + i]; b.N[i] = (int)N[base + i]; b.kn[i] = (int)kn[base + i];
            b.blk0[i] = blocks;
            blocks += (unsigned)(((long long)b.K[i] * b.N[i] + 255) / 256);
        }
        b.blk0[b.n] = blocks;
        if (blocks) hipLaunchKernelGGL(x6_split_batched_kernel, dim3(blocks), dim3(256), 0, s, b);
    }
    return MYOLO_OK;
}

// the three-bf16-piece split of a WEIGHT matrix: into `scratch`, or the copy prepared for this step (prepared-weights registry, csrc/myolo_common.h)
static const float* x6_split_weights(const float* w, void* scratch, int K, int N, int src_kn, hipStream_t s)
{
    const long long total = (long long)K * N;
#ifdef MM_X6_TUNE
    if (g_myolo_opt.tune0 & (1 << 20)) return (const float*)scratch;          // timing only: the split left there by an earlier call
#endif
    return (const float*)myolo_wprep_resolve(w, WP_X6_SPLIT, K, N, src_kn, (size_t)total * 6, scratch, s, [=](void* d, hipStream_t st) {
        hipLaunchKernelGGL(x6_split_nk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w, (__bf16*)d, K, N, src_kn);
    });
}

/* whether the fused deconv + ReLU + 1x1 mask conv GEMM (csrc/gemm_kernels.hip: myolo_deconv2x2s2_mask_fwd) runs on these kernels */
bool myolo_deconv_mask_mm_ok(int Cin, int Cout) { return !g_myolo_opt.wino_no_bt && (Cin % MM_BK) == 0 && Cin >= MM_BK && (Cout % MM_BN) == 0; }
size_t myolo_deconv_mask_mm_split_bytes(int Cin, int Cout) { return align256((size_t)4 * Cin * Cout * 6); }

/* x [M][Cin] fp32, w [2,2,Cout,Cin] (Keras Conv2DTranspose: for tap (ky,kx), w[tap][co][ci] = B[ci][tap*Cout+co], i.e. w IS the
 * transposed operand [N][K] the fp32 kernel wants), split = scratch of myolo_deconv_mask_mm_split_bytes (bf16x6 only),
 * part = [Cout/128][4*M][ncls] partial logits.  Option "wino_x6": six bf16 piece products per fp32 product. */
int myolo_deconv_mask_mm(const float* x, const float* w, const float* bias, const float* w2, float* part, void* split,
                         long long M, int H, int W, int Cin, int Cout, int ncls, hipStream_t s, const int32_t* keep_inv, float* keep_d, int keep_cap,
                         const float* b2, float* p_out, int* finished)
{
    if (finished) *finished = 0;              // 1: the kernel stored the probabilities itself (no deconv_mask_finish launch wanted)
    const int K = Cin, N = 4 * Cout;
    MMArgs a{};
    a.A = x; a.C = nullptr; a.K = K; a.N = N; a.nruns = 1; a.nt = 0;
    a.bias = bias; a.w2 = w2; a.part = part; a.H = H; a.W = W; a.Co = Cout; a.ncls = ncls;
    a.keep_inv = keep_d ? keep_inv : nullptr; a.keep_d = keep_inv ? keep_d : nullptr; a.keep_cap = keep_cap;
    MMRun& R = a.run[0];
    R.rows = M; R.a_off = 0; R.b_off = 0; R.c_off = 0; R.nq = 1; R.tile0 = 0;
    R.mtiles = (int)((M + MM_BM - 1) / MM_BM);
    const long long tiles = (long long)R.mtiles * (N / MM_BN);
    a.tune = g_myolo_opt.tune0;
    if (g_myolo_opt.wino_x6) {
        a.Bt = x6_split_weights(w, split, K, N, 0, s);
        if (g_myolo_opt.deconv_mask_legacy == 1)  // rounds 3-5: the untransposed tile with the butterfly epilogue (ablation / test reference)
            hipLaunchKernelGGL(wino_mm_x6_kernel<MM_EP_DECONV_MASK>, dim3((unsigned)tiles), dim3(256), 0, s, a);
        else {
            // 2 = the transposed tile, but partial logits + the finish launch (test reference of the in-kernel finish)
            if (Cout == MM_BN && b2 && p_out && finished && g_myolo_opt.deconv_mask_legacy != 2) { a.b2 = b2; a.out = p_out; *finished = 1; }
            hipLaunchKernelGGL(wino_mm_x6_kernel<MM_EP_DECONV_MASK_T>, dim3((unsigned)tiles), dim3(256), 0, s, a);
        }
    } else {
        a.Bt = w;
        hipLaunchKernelGGL(wino_mm_kernel<MM_EP_DECONV_MASK>, dim3((unsigned)tiles), dim3(256), 0, s, a);
    }
    return MYOLO_OK;
}

// =====================================================================================================================
// Plain fp32 matrix product through the kernels above, the way its products are formed chosen PER CALL (no process switch):
//     C [M][N] = A [M][K] * B,   B given as [N][K] (b_is_nk = 1: the transposed operand both kernels want) or [K][N] (b_is_nk = 0).
// products = MYOLO_PRODUCTS_NATIVE: v_mfma_f32_32x32x2_f32;  MYOLO_PRODUCTS_BF16X6: six exact bf16 piece products per fp32 product.
// Used by the 1x1 convolutions with K, N multiples of (16, 256) and by tests/test_gpu_ops.py::test_bf16x6_* (special values, K = 2304).
__global__ __launch_bounds__(256) void mm_transpose_kn_kernel(const float* __restrict__ src, float* __restrict__ dst, int K, int N)
{
    __shared__ float t[32][33];
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8)
        if (k0 + j < K && n0 + tx < N) t[j][tx] = src[(long long)(k0 + j) * N + n0 + tx];
    __syncthreads();
    for (int j = ty; j < 32; j += 8)
        if (n0 + j < N && k0 + tx < K) dst[(long long)(n0 + j) * K + k0 + tx] = t[tx][j];
}

extern "C" size_t myolo_matmul_f32_ws_bytes(int K, int N, int b_is_nk, int products)
{
    if (products == MYOLO_PRODUCTS_BF16X6) return align256((size_t)K * N * 6);
    return b_is_nk ? 0 : align256((size_t)K * N * 4);
}

extern "C" int myolo_matmul_f32(const float* A, const float* B, float* C, int64_t M, int K, int N, int b_is_nk, int products,
                                void* ws, size_t ws_bytes, void* stream)
{
    return myolo_matmul_f32_impl(A, B, C, M, K, N, b_is_nk, products, ws, ws_bytes, stream, false);
}

// b_is_weight: B is a layer's weight tensor (its bf16x6 split may come from the prepared-weights registry); false for arbitrary operands
int myolo_matmul_f32_impl(const float* A, const float* B, float* C, int64_t M, int K, int N, int b_is_nk, int products,
                          void* ws, size_t ws_bytes, void* stream, bool b_is_weight)
{
    MYOLO_REQUIRE(A && B && C && M > 0 && K >= MM_BK && (K % MM_BK) == 0 && N >= MM_BN && (N % MM_BN) == 0,
                  "matmul_f32: needs M > 0, K %% %d == 0, N %% %d == 0", MM_BK, MM_BN);
    MYOLO_REQUIRE(products == MYOLO_PRODUCTS_NATIVE || products == MYOLO_PRODUCTS_BF16X6, "matmul_f32: products must be MYOLO_PRODUCTS_NATIVE or MYOLO_PRODUCTS_BF16X6");
    MYOLO_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "matmul_f32: operands must be 16-byte aligned");
    MYOLO_NEED_WS(myolo_matmul_f32_ws_bytes(K, N, b_is_nk, products));
    hipStream_t s = (hipStream_t)stream;
    MMArgs a{};
    a.A = A; a.C = C; a.K = K; a.N = N; a.nruns = 1; a.nt = 0;
    MMRun& R = a.run[0];
    R.rows = M; R.a_off = 0; R.b_off = 0; R.c_off = 0; R.nq = 1; R.tile0 = 0;
    R.mtiles = (int)((M + MM_BM - 1) / MM_BM);
    const long long tiles = (long long)R.mtiles * (N / MM_BN);
    if (products == MYOLO_PRODUCTS_BF16X6) {
        const long long total = (long long)K * N;
        if (b_is_weight) a.Bt = x6_split_weights(B, ws, K, N, b_is_nk ? 0 : 1, s);
        else {
            hipLaunchKernelGGL(x6_split_nk_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, B, (__bf16*)ws, K, N, b_is_nk ? 0 : 1);
            a.Bt = (const float*)ws;
        }
        if (x6_half_tiles(tiles)) hipLaunchKernelGGL((wino_mm_x6_kernel<MM_EP_PLAIN, false, MM_A_PLAIN, 2>), dim3((unsigned)(2 * tiles)), dim3(256), 0, s, a);
        else hipLaunchKernelGGL(wino_mm_x6_kernel<MM_EP_PLAIN>, dim3((unsigned)tiles), dim3(256), 0, s, a);
    } else {
        if (!b_is_nk) {
            hipLaunchKernelGGL(mm_transpose_kn_kernel, dim3((N + 31) / 32, (K + 31) / 32), dim3(256), 0, s, B, (float*)ws, K, N);
            a.Bt = (const float*)ws;
        } else a.Bt = B;
        hipLaunchKernelGGL(wino_mm_kernel<MM_EP_PLAIN>, dim3((unsigned)tiles), dim3(256), 0, s, a);
    }
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

// =====================================================================================================================
// Weight-gradient product of the Winograd conv on the bf16 matrix pipe at fp32 accuracy ("bf16x6", see wino_mm_x6_kernel):
//     C[z] (Ka x N) = A[z]^T (Ka x M) * B[z] (M x N),   A [M][Ka], B [M][N] dense row-major (the V and Q planes of conv1's backward,
//     model.py:687-690: dU[q] = V[q]^T Q[q]), M = rows of the plane = the reduction index.
// Both operands have the reduction index as their SLOW axis, the MFMA wants 8 consecutive k per lane: a loader thread owns one column
// of one operand and fetches its 16 rows of the chunk with 16 dword loads (every wave-level load is a contiguous 256-byte row segment),
// splits them EXACTLY into three bf16 pieces (truncation: and / sub / perm) and writes them k-contiguous into the LDS row record of
// its column -- the transposition happens in the register -> LDS step, no strided access anywhere.  256 x 256 tile per workgroup (each
// plane element is read from HBM once: the kernel is bound by its 3.9 GB of operand traffic, not by the matrix pipe), 8 waves of
// 128 x 64 (4 x 2 MFMA tiles x 6 piece products), M split over workgroups, fp32 partial tiles summed in a fixed order by
// tn_x6_reduce_kernel (bit-reproducible).  One barrier per 16-deep chunk; chunk c+1 is split into the other LDS buffer behind the MFMAs
// of chunk c, chunk c+2 is in flight from HBM meanwhile.
#define TN_T 256               // tile edge (rows of C and columns of C per workgroup)
struct TNRun {
    long long rows;            // M of every plane of the run
    long long a_off, b_off;    // element offsets of the run's first plane in A / B (planes rows*Ka, rows*N apart)
    long long rows_per_split;  // multiple of 16
    long long unit0;           // first (plane, split) unit of the run
    int plane0;                // index of the run's first plane in C
    int nq;                    // planes in the run
    int splits;
};
struct TNArgs {
    const float* a_scale;      // A := act(A * a_scale[col] + a_shift[col]) on load (rows that exist), NULL: none
    const float* a_shift;
    int a_act;
    int gH, gW, gCo;           // AG (deconv weight gradient): A row m, column (tap, co) = dy[pixel(m, tap)][co] on the [N,2H,2W,Co] grid; Ka = 4 Co
    const float* A;
    const float* B;
    float* part;               // [(plane, split) unit][Ka][N] fp32 partial products
    float* C;                  // [planes][Ka][N]
    int Ka, N;
    int nruns;
    int tiles_k, tiles_n;      // Ka / 256, N / 256
    long long nunits;          // (plane, split) pairs x tiles
    long long unit_base;       // first unit of this launch (option "tn_wgs": the units go out in launches of at most that many workgroups)
    int tune;                  // MM_X6_TUNE builds only (tools/experiments/tn_x6_tune.sh): timing-only ablations of wino_tn_x6_kernel, results are wrong
    TNRun run[4];
};

template <bool AG = false>
__global__ __launch_bounds__(512, 1) void wino_tn_x6_kernel(TNArgs p)
{
    __shared__ __attribute__((aligned(16))) unsigned char Ls[2][2][TN_T * X6_REC];      // [buffer][A | B][column record]: 112 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3, half = lane >> 5, l31 = lane & 31;
    // unit = ((plane, split) pair, tile): tiles of one pair are consecutive units.  One workgroup per unit; with option "tn_wgs" the units go out in
    // several launches of at most that many workgroups (unit_base): with 112 KB of LDS a CU holds ONE of these workgroups, so launches of 224 leave
    // four CUs of every XCD to whatever runs beside this kernel (the trunk's backward chain: dozens of short dependent kernels that otherwise
    // starve behind its half-millisecond workgroups)
    const int ntile = p.tiles_k * p.tiles_n;
  const long long bid = p.unit_base + blockIdx.x;
  {
    const long long pair = bid / ntile;
    const int tile = (int)(bid - pair * ntile);
    const int kat = tile / p.tiles_n, nt = tile - kat * p.tiles_n;
    int ri = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < p.nruns && pair >= p.run[k].unit0) ri = k;
    const TNRun& R = p.run[ri];
    const long long local = pair - R.unit0;
    const int z = (int)(local / R.splits), sp = (int)(local - (long long)z * R.splits);
    const long long r0 = (long long)sp * R.rows_per_split;
    long long r1 = r0 + R.rows_per_split;
    if (r1 > R.rows) r1 = R.rows;
    const long long nrows = r1 > r0 ? r1 - r0 : 0;
    const int nk = (int)((nrows + MM_BK - 1) / MM_BK);

    // loader role: thread = (operand tid >> 8, column tid & 255); 16 rows of the chunk
    const int op = tid >> 8, col = tid & 255;
    const int ld = op ? p.N : p.Ka;
    const float* base = op ? p.B + R.b_off + ((long long)z * R.rows + r0) * p.N + nt * TN_T
                           : p.A + R.a_off + ((long long)z * R.rows + r0) * p.Ka + kat * TN_T;
    __amdgpu_buffer_rsrc_t rs = mm_rsrc(base, (nrows * ld - (op ? nt : kat) * TN_T) * 4);      // rows beyond the split read 0
    unsigned voff = (unsigned)col * 4u;
    const unsigned rowb = (unsigned)ld * 4u;
    unsigned soff = 0;                                   // byte offset of the chunk's first row (advances by 16 rows)
    // AG: the A operand is gathered -- row m, column (tap, co) = dy[pixel(m, tap)][co], pixel(m, tap) = 4m - 2(m mod W) + ky*2W + kx
    long long g_m = r0, g_base_px = 0;                   // first row of the next chunk; first pixel the descriptor covers
    int g_x = 0;                                         // g_m mod W
    if constexpr (AG) {
        if (op == 0) {
            g_base_px = 4 * r0 - 2 * p.gW; if (g_base_px < 0) g_base_px = 0;
            long long end_px = 4 * r1 + 2 * p.gW + 2; if (end_px > 4 * R.rows) end_px = 4 * R.rows;
            rs = mm_rsrc(p.A + g_base_px * p.gCo, (end_px - g_base_px) * p.gCo * 4);
            const int ka = kat * TN_T + col, tap = ka / p.gCo, co = ka - tap * p.gCo;
            voff = (unsigned)(((tap >> 1) * 2 * p.gW + (tap & 1)) * p.gCo + co) * 4u;
            g_x = (int)(r0 % p.gW);
        }
    }
    float st[16];
    auto gload = [&]() {
        if (AG && op == 0) {
            int xk = g_x;
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const long long m = g_m + k;
                const unsigned so = m < r1 ? (unsigned)((4 * m - 2 * xk - g_base_px) * p.gCo) * 4u : MM_OOB;
                st[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)(so == MM_OOB ? MM_OOB : voff + so), 0, 0));
                if (++xk == p.gW) xk = 0;
            }
            g_m += 16;
            g_x = xk;
            return;
        }
#ifdef MM_X6_TUNE
        if (p.tune & 1) {                                   // no operand traffic: every load out of range
#pragma unroll
            for (int k = 0; k < 16; ++k) st[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)MM_OOB, 0, 0));
            return;
        }
        if (p.tune & 16) {                                  // the chunk's first 16 rows every time: L2 hits instead of HBM traffic
#pragma unroll
            for (int k = 0; k < 16; ++k) st[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)((unsigned)k * rowb), 0));
            return;
        }
#endif
#pragma unroll
        for (int k = 0; k < 16; ++k) st[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, (int)voff, (int)(soff + (unsigned)k * rowb), 0));
        soff += 16u * rowb;
    };
    unsigned char* const wbase = &Ls[0][op][col * X6_REC];
    // the A operand may be a pre-BN tensor whose BatchNorm + activation is applied on load (pointwise weight gradient): this thread's
    // column has one scale / shift; padding rows (beyond the split) must stay 0
    const bool pro = p.a_scale != nullptr && op == 0;
    const float csc = pro ? p.a_scale[kat * TN_T + col] : 1.f, csh = pro ? p.a_shift[kat * TN_T + col] : 0.f;
    long long rows_staged = 0;                           // first row (within the split) of the chunk in the staging registers
    auto split_store = [&](int buf, int h) {             // rows 8h .. 8h+7 of the staged chunk -> the three pieces of k half h
        u32x4 p1, p2, p3;
        if (pro) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (rows_staged + 8 * h + j < nrows) st[8 * h + j] = mm_act(fmaf(st[8 * h + j], csc, csh), p.a_act);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = st[8 * h + 2 * e], a1 = st[8 * h + 2 * e + 1];
#ifdef MM_X6_TUNE
            if (p.tune & 2) { p1[e] = x6_top(a0, a1); p2[e] = p1[e]; p3[e] = p1[e]; continue; }      // no exact split
#endif
            const float b0 = x6_rest(a0), b1 = x6_rest(a1);
            p1[e] = x6_top(a0, a1);
            p2[e] = x6_top(b0, b1);
            p3[e] = x6_top(x6_rest(b0), x6_rest(b1));
        }
#ifdef MM_X6_TUNE
        if (p.tune & 8) return;                              // no LDS writes
#endif
        unsigned char* w = wbase + buf * (2 * TN_T * X6_REC) + h * 16;
        *reinterpret_cast<u32x4*>(w) = p1;
        *reinterpret_cast<u32x4*>(w + 32) = p2;
        *reinterpret_cast<u32x4*>(w + 64) = p3;
    };
    // MFMA role: A-operand rows = C rows (columns of A): wm*128 + t*32 + l31; B-operand = C columns: wn*64 + u*32 + l31
    const int afr = (wm * 128 + l31) * X6_REC + half * 16;           // + t * 32 * X6_REC, + piece * 32
    const int bfr = (wn * 64 + l31) * X6_REC + half * 16;            // + u * 32 * X6_REC

    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    if (nk > 0) {
        gload();
        split_store(0, 0);
        split_store(0, 1);
        if (nk > 1) gload();
    }
    __syncthreads();
    for (int c = 0; c < nk; ++c) {
        const int cur = c & 1;
        const bool more = c + 1 < nk;
        rows_staged = (long long)(c + 1) * MM_BK;         // the staging registers hold chunk c+1 until it is split below
        const unsigned char* la = &Ls[cur][0][0];
        const unsigned char* lb = &Ls[cur][1][0];
        bf16x8 fb[2][3], fa[3], fn[3];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) fb[u][pc] = *reinterpret_cast<const bf16x8*>(lb + bfr + u * 32 * X6_REC + pc * 32);
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) fa[pc] = *reinterpret_cast<const bf16x8*>(la + afr + pc * 32);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (t < 3) {
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) fn[pc] = *reinterpret_cast<const bf16x8*>(la + afr + (t + 1) * 32 * X6_REC + pc * 32);
            }
#ifdef MM_X6_TUNE
            if (!(p.tune & 4))                        // (4: no MFMAs -- what the rest of the loop costs by itself)
#endif
#pragma unroll
            for (int u = 0; u < 2; ++u) {             // smallest terms first
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[u][2], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[u][0], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[u][1], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[u][1], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[u][0], acc[t][u], 0, 0, 0);
                acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[u][0], acc[t][u], 0, 0, 0);
            }
            // chunk c+1 (in the staging registers since the previous chunk) goes into the other buffer in two slices behind the MFMAs
            // of row tiles 0 and 1 (that buffer's last readers passed the previous barrier); chunk c+2 is requested behind row tile 2
            if (t == 0 && more) split_store(cur ^ 1, 0);
            if (t == 1 && more) split_store(cur ^ 1, 1);
            if (t == 2 && c + 2 < nk) gload();
            if (t < 3) { fa[0] = fn[0]; fa[1] = fn[1]; fa[2] = fn[2]; }
        }
        __syncthreads();
    }
    // ---- partial tile -> part[pair][kat*256 + row][nt*256 + col]: 128-byte row segments per 32 lanes
    float* Pp = p.part + pair * (long long)p.Ka * p.N + ((long long)kat * TN_T + wm * 128) * p.N + nt * TN_T + wn * 64 + l31;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
#pragma unroll
            for (int u = 0; u < 2; ++u) Pp[(long long)row * p.N + u * 32] = acc[t][u][r];
        }
  }
}

// C[plane] = sum over the plane's splits of part[(plane, split)], fixed order; grid (element quads / 256, planes)
__global__ __launch_bounds__(256) void tn_x6_reduce_kernel(TNArgs p)
{
    const int plane = blockIdx.y;
    int ri = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
        if (k < p.nruns && plane >= p.run[k].plane0) ri = k;
    const TNRun& R = p.run[ri];
    const long long n4 = (long long)p.Ka * p.N / 4;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4* src = reinterpret_cast<const float4*>(p.part) + (R.unit0 + (long long)(plane - R.plane0) * R.splits) * n4 + i;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 4 <= R.splits; k += 4) {
        const float4 a = src[(long long)k * n4], b = src[(long long)(k + 1) * n4], c = src[(long long)(k + 2) * n4], d = src[(long long)(k + 3) * n4];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
        s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
    }
    for (; k < R.splits; ++k) {
        const float4 a = src[(long long)k * n4];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
    }
    reinterpret_cast<float4*>(p.C)[(long long)plane * n4 + i] = s;
}

/* whether dU[q] = V[q]^T Q[q] of a (Cin, Cout) layer runs on wino_tn_x6_kernel (option "wino_x6") */
bool myolo_gemm_tn_x6_ok(int Ka, int N) { return g_myolo_opt.wino_x6 && !g_myolo_opt.tn_no_x6 && Ka >= TN_T && (Ka % TN_T) == 0 && N >= TN_T && (N % TN_T) == 0; }

// split plan shared by the size query and the launcher: ~3 equal-work units per CU
static long long tn_x6_plan(int nruns, const long long* rows, const int* nq, int Ka, int N, TNArgs* a)
{
    const int ntile = (Ka / TN_T) * (N / TN_T);
    long long total_rows = 0;
    for (int r = 0; r < nruns; ++r) total_rows += rows[r] * nq[r];
    const long long target_units = 3 * (g_myolo_opt.tn_wgs > 0 ? g_myolo_opt.tn_wgs : 256);
    long long rps = ((total_rows * ntile + target_units - 1) / target_units + MM_BK - 1) / MM_BK * MM_BK;
    if (rps < 8 * MM_BK) rps = 8 * MM_BK;
    long long pairs = 0;
    for (int iter = 0; iter < 64; ++iter) {
        pairs = 0;
        for (int r = 0; r < nruns; ++r) pairs += (long long)nq[r] * ((rows[r] + rps - 1) / rps);
        if (pairs * ntile <= target_units || rps >= total_rows) break;
        rps += MM_BK;
    }
    if (a) {
        long long unit = 0;
        int plane = 0;
        a->nruns = 0;
        for (int r = 0; r < nruns; ++r) {
            if (rows[r] <= 0 || nq[r] <= 0) continue;
            TNRun& R = a->run[a->nruns++];
            R.rows = rows[r]; R.nq = nq[r]; R.plane0 = plane; R.unit0 = unit;
            R.splits = (int)((rows[r] + rps - 1) / rps);
            R.rows_per_split = ((rows[r] + R.splits - 1) / R.splits + MM_BK - 1) / MM_BK * MM_BK;       // equal slices of the plane
            unit += (long long)R.nq * R.splits;
            plane += nq[r];
        }
    }
    return pairs;
}

static long long tn_x6_grid(long long nunits) { return g_myolo_opt.tn_wgs > 0 && nunits > g_myolo_opt.tn_wgs ? g_myolo_opt.tn_wgs : nunits; }

size_t myolo_gemm_tn_x6_ws_bytes(int nruns, const long long* rows, const int* nq, int Ka, int N)
{
    return align256((size_t)tn_x6_plan(nruns, rows, nq, Ka, N, nullptr) * Ka * N * sizeof(float));
}

/* For every run r and plane z < nq[r]:  C[plane0_r + z] (Ka x N) = A_r[z]^T B_r[z], A_r = A + a_off[r] (planes rows[r]*Ka apart),
 * B_r = B + b_off[r] (rows[r]*N apart); C planes are Ka*N apart in run order.  part: myolo_gemm_tn_x6_ws_bytes.  Two launches. */
int myolo_gemm_tn_x6_runs(const float* A, const float* B, float* C, int nruns, const long long* rows, const long long* a_off, const long long* b_off,
                          const int* nq, int Ka, int N, void* part, size_t part_bytes, hipStream_t s,
                          const float* a_scale, const float* a_shift, int a_act)
{
    if (nruns < 0 || nruns > 4 || Ka < TN_T || (Ka % TN_T) || N < TN_T || (N % TN_T) || ((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)C & 15)) {
        myolo_set_error("gemm_tn_x6_runs: needs Ka %% %d == 0, N %% %d == 0, <= 4 runs, 16-byte aligned operands", TN_T, TN_T);
        return MYOLO_EINVAL;
    }
    TNArgs a{};
    a.A = A; a.B = B; a.C = C; a.part = (float*)part; a.Ka = Ka; a.N = N; a.tiles_k = Ka / TN_T; a.tiles_n = N / TN_T;
    a.a_scale = a_scale; a.a_shift = a_shift; a.a_act = a_act;
    const long long pairs = tn_x6_plan(nruns, rows, nq, Ka, N, &a);
    if ((size_t)pairs * Ka * N * sizeof(float) > part_bytes || !part) {
        myolo_set_error("gemm_tn_x6_runs: workspace too small (%zu needed, %zu given)", (size_t)pairs * Ka * N * sizeof(float), part_bytes);
        return MYOLO_EWORKSPACE;
    }
    int k = 0, planes = 0;
    for (int r = 0; r < nruns; ++r) {
        if (rows[r] <= 0 || nq[r] <= 0) continue;
        if ((a_off[r] | b_off[r]) & 3) { myolo_set_error("gemm_tn_x6_runs: run offsets must be multiples of 4 elements"); return MYOLO_EINVAL; }
        a.run[k].a_off = a_off[r]; a.run[k].b_off = b_off[r];
        planes += nq[r];
        ++k;
    }
    if (pairs <= 0) return MYOLO_OK;
    a.nunits = pairs * a.tiles_k * a.tiles_n;
#ifdef MM_X6_TUNE
    a.tune = g_myolo_opt.tune0;
#endif
    for (long long base = 0, g = tn_x6_grid(a.nunits); base < a.nunits; base += g) {
        a.unit_base = base;
        hipLaunchKernelGGL(wino_tn_x6_kernel<false>, dim3((unsigned)(a.nunits - base < g ? a.nunits - base : g)), dim3(512), 0, s, a);
    }
    const long long n4 = (long long)Ka * N / 4;
    hipLaunchKernelGGL(tn_x6_reduce_kernel, dim3((unsigned)((n4 + 255) / 256), planes), dim3(256), 0, s, a);
    return MYOLO_OK;
}


/* pointwise conv of the trunk on the bf16 matrix pipe with six exact piece products (FP32_MATMUL = "bf16x6", layers with Cout % 256 == 0):
 * y [M][N] = act_in(x * in_scale + in_shift) [M][K] * w [K][N], optional per-row-tile partial sums of y's columns (stat).
 * ws: the split filters (K*N*6 bytes).  Two launches (split, GEMM). */
// (tune0 & 65536: from 128 input channels -- conv_pw_4, 25 088 x 128 -> 256, is 34.4 -> 28.2 us stand-alone that way and the step 0.05 ms SLOWER, four runs each:
//  one more weight split on the side stream and one more bf16-MFMA kernel in the trunk; not the default)
bool myolo_pw_x6_ok(int K, int N) { return g_myolo_opt.wino_x6 && !g_myolo_opt.pw_no_x6 && K >= ((g_myolo_opt.tune0 & 65536) ? 128 : 256) && (K % MM_BK) == 0 && (N % MM_BN) == 0; }
size_t myolo_pw_x6_split_bytes(int K, int N) { return align256((size_t)K * N * 6); }
int myolo_pw_x6_fwd(const float* x, const float* in_scale, const float* in_shift, int in_act, const float* w, float* y, double* stat,
                    long long M, int K, int N, void* split, hipStream_t s)
{
    MMArgs a{};
    a.A = x; a.C = y; a.K = K; a.N = N; a.nruns = 1; a.nt = 0;
    a.a_scale = in_scale; a.a_shift = in_shift; a.a_act = in_act; a.stat = stat;
#ifdef MM_X6_TUNE
    a.tune = g_myolo_opt.tune0;
#endif
    MMRun& R = a.run[0];
    R.rows = M; R.a_off = 0; R.b_off = 0; R.c_off = 0; R.nq = 1; R.tile0 = 0;
    R.mtiles = (int)((M + MM_BM - 1) / MM_BM);
    const long long tiles = (long long)R.mtiles * (N / MM_BN);
    a.Bt = x6_split_weights(w, split, K, N, 1, s);
    if (x6_half_tiles(tiles)) hipLaunchKernelGGL((wino_mm_x6_kernel<MM_EP_PLAIN, true, MM_A_PLAIN, 2>), dim3((unsigned)(2 * tiles)), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wino_mm_x6_kernel<MM_EP_PLAIN, true>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    return MYOLO_OK;
}


/* myolo_mask_deconv (Conv2DTranspose 2x2 / s2, model.py:711-712) forward and data gradient with six exact bf16 piece products per fp32
 * product (FP32_MATMUL = "bf16x6"): the compacted mask-head backward rebuilds the deconv output of the positive ROIs and back-propagates
 * through it with these.  fwd: y [N,2H,2W,Co] = act(deconv(x [M][Cin], w [2,2,Co,Cin]) + bias); bwd_data: dx [M][Cin] from dy [N,2H,2W,Co].
 * split = scratch of 4*Cin*Co*6 bytes.  Need Cin % 16 == 0 and Co % 256 == 0 (fwd), Co % 16 == 0 and Cin % 256 == 0 (bwd_data). */
bool myolo_deconv_x6_ok(int Cin, int Co, int which)
{
    if (!g_myolo_opt.wino_x6 || g_myolo_opt.deconv_no_x6) return false;
    return which == 0 ? ((Cin % MM_BK) == 0 && Cin >= MM_BK && (Co % MM_BN) == 0) : ((Co % MM_BK) == 0 && (Cin % MM_BN) == 0);
}
int myolo_deconv_x6_fwd(const float* x, const float* w, const float* bias, float* y, long long M, int H, int W, int Cin, int Co, int act,
                        void* split, hipStream_t s)
{
    const int K = Cin, N = 4 * Co;
    MMArgs a{};
    a.A = x; a.C = y; a.K = K; a.N = N; a.nruns = 1; a.bias = bias; a.H = H; a.W = W; a.Co = Co; a.a_act = act;
    MMRun& R = a.run[0];
    R.rows = M; R.nq = 1; R.mtiles = (int)((M + MM_BM - 1) / MM_BM);
    a.Bt = x6_split_weights(w, split, K, N, 0, s);     // w = [N][K]
    hipLaunchKernelGGL((wino_mm_x6_kernel<MM_EP_DECONV, false, MM_A_PLAIN>), dim3((unsigned)((long long)R.mtiles * (N / MM_BN))), dim3(256), 0, s, a);
    return MYOLO_OK;
}
int myolo_deconv_x6_bwd_data(const float* dy, const float* w, float* dx, long long M, int H, int W, int Cin, int Co, void* split, hipStream_t s)
{
    const int K = 4 * Co, N = Cin;
    MMArgs a{};
    a.A = dy; a.C = dx; a.K = K; a.N = N; a.nruns = 1; a.H = H; a.W = W; a.Co = Co;
    MMRun& R = a.run[0];
    R.rows = M; R.nq = 1; R.mtiles = (int)((M + MM_BM - 1) / MM_BM);
    a.Bt = x6_split_weights(w, split, K, N, 1, s);     // w = [K = (tap, co)][N = ci]
    hipLaunchKernelGGL((wino_mm_x6_kernel<MM_EP_PLAIN, false, MM_A_DECONV>), dim3((unsigned)((long long)R.mtiles * (N / MM_BN))), dim3(256), 0, s, a);
    return MYOLO_OK;
}


/* weight gradient of the same transposed conv: dw [2,2,Co,Cin] = sum_m dy[pixel(m, tap)][co] * x[m][ci] on wino_tn_x6_kernel with the A operand
 * gathered (Ka = 4 Co, N = Cin; both multiples of 256; W >= 1).  part: myolo_deconv_x6_bwd_weight_ws_bytes. */
size_t myolo_deconv_x6_bwd_weight_ws_bytes(long long M, int Cin, int Co)
{
    const long long rows[1] = {M};
    const int nq[1] = {1};
    return myolo_gemm_tn_x6_ws_bytes(1, rows, nq, 4 * Co, Cin);
}
int myolo_deconv_x6_bwd_weight(const float* x, const float* dy, float* dw, long long M, int H, int W, int Cin, int Co, void* part, size_t part_bytes,
                               hipStream_t s)
{
    const int Ka = 4 * Co, N = Cin;
    TNArgs a{};
    a.A = dy; a.B = x; a.C = dw; a.part = (float*)part; a.Ka = Ka; a.N = N; a.tiles_k = Ka / TN_T; a.tiles_n = N / TN_T;
    a.gH = H; a.gW = W; a.gCo = Co;
    const long long rows[1] = {M};
    const int nq[1] = {1};
    const long long pairs = tn_x6_plan(1, rows, nq, Ka, N, &a);
    if ((size_t)pairs * Ka * N * sizeof(float) > part_bytes || !part) {
        myolo_set_error("deconv_x6_bwd_weight: workspace too small (%zu needed, %zu given)", (size_t)pairs * Ka * N * sizeof(float), part_bytes);
        return MYOLO_EWORKSPACE;
    }
    a.run[0].a_off = 0; a.run[0].b_off = 0;
    a.nunits = pairs * a.tiles_k * a.tiles_n;
    for (long long base = 0, g = tn_x6_grid(a.nunits); base < a.nunits; base += g) {
        a.unit_base = base;
        hipLaunchKernelGGL(wino_tn_x6_kernel<true>, dim3((unsigned)(a.nunits - base < g ? a.nunits - base : g)), dim3(512), 0, s, a);
    }
    const long long n4 = (long long)Ka * N / 4;
    hipLaunchKernelGGL(tn_x6_reduce_kernel, dim3((unsigned)((n4 + 255) / 256), 1), dim3(256), 0, s, a);
    return MYOLO_OK;
}
