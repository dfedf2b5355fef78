// bf16-storage / fp32-accumulate inference kernels for the mask head (BASELINE.json configs[3]:
// "Rice 416x416, 5 anchors, 28x28 mask head, bf16, inference-only").  The mask head is 99 % of the
// inference FLOPs; on gfx950 v_mfma_f32_32x32x16_bf16 runs at 16x the
// fp32 MFMA rate, so the ROIAlign output, the four 3x3 convs (BatchNorm folded into the weights -- all
// BN layers are frozen in inference, model.py:690-708) and the 2x2 transposed conv are kept in bf16.
//
//   gemm_bf16<AMODE, EPI>:  C[m, n] = act( sum_k A(m, k) * Wt[n, k] + bias[n] )
//     A bf16, gathered im2col-free (PLAIN rows or CONV3 taps with hardware zero-fill through a raw buffer
//     descriptor), Wt bf16 [N][K] (k contiguous: weights are static, so they are stored transposed once),
//     128x128x64 tiles, 4 waves x (2x2) MFMA 32x32x16, LDS rows padded to 144 B so every ds_read_b128
//     fragment read (8 bf16 of one row) is bank-conflict-free, double-buffered, one barrier per K tile.
//     EPI PLAIN stores bf16 [M, N]; EPI DECONV scatters the 2x2/s2 transposed-conv output (bf16).
#include "myolo_common.h"
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define TBM 128
#define TBN 128
#define TBK 64
#define LDSROW 72            // bf16 elements per LDS row (64 + 8 pad = 144 bytes)
#define OOB_OFF 0x7fffff00u

template <int V> struct IntC { static constexpr int value = V; };
enum { AM_PLAIN = 0, AM_CONV3 = 1 };
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
enum { EP_PLAIN = 0, EP_DECONV = 1, EP_DECONV_MASK = 2 };

struct Bf16Args {
    const uint16_t* A;       // bf16 bits
    const uint16_t* Wt;      // [N][K]
    uint16_t* C;             // bf16 out
    const float* bias;       // [N] (EP_PLAIN) or [Co] (EP_DECONV)
    long long M;
    int N, K;
    int H, W, Cc, Co;
    int act;
    const float* w2;         // EP_DECONV_MASK: 1x1 mask conv kernel [Co][ncls] (fp32)
    float* part;             // EP_DECONV_MASK: partial logits [slab][4*M][ncls]
    const float* b2;         // EP_DECONV_MASK, FIN kernels: bias of the 1x1 mask conv [ncls]
    float* out;              //   and the probabilities [4*M][ncls] (no partial logits, no finish launch)
    int ncls;
    int tune;                // BF16_TUNE builds only (tools/experiments/bf16_tune.sh): timing-only ablations, results are wrong
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void* base, long long nbytes)
{
    if (nbytes < 0) nbytes = 0;
    if (nbytes > 0x7ffffe00ll) nbytes = 0x7ffffe00ll;
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)(unsigned)nbytes, 0x00020000);
}
__device__ __forceinline__ u32x4 bufld(__amdgpu_buffer_rsrc_t r, unsigned off)
{
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
}
__device__ __forceinline__ uint16_t f2bf(float f)
{   // round to nearest even
    unsigned u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(uint16_t h) { return __uint_as_float(((unsigned)h) << 16); }

template <int AMODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16(Bf16Args p)
{
    __shared__ __attribute__((aligned(16))) uint16_t As[2][TBM][LDSROW];
    __shared__ __attribute__((aligned(16))) uint16_t Bs[2][TBN][LDSROW];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.N + TBN - 1) / TBN;
    long long bid;
    {   // XCD-aware order (speed only): each XCD gets a contiguous run of tiles
        const long long nwg = gridDim.x, orig = blockIdx.x;
        const long long q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tn = (int)(bid % ntn);
    const long long m0 = (bid / ntn) * TBM;
    const int n0 = tn * TBN;
    const long long hw = (long long)p.H * p.W;

    // A descriptor with a per-workgroup base (32-bit offsets, out-of-range => zeros)
    long long base_row, end_row, row_elems;
    if (AMODE == AM_PLAIN) { base_row = m0; end_row = (m0 + TBM < p.M) ? m0 + TBM : p.M; row_elems = p.K; }
    else {
        base_row = m0 - (p.W + 1); if (base_row < 0) base_row = 0;
        end_row = m0 + TBM + p.W + 1; if (end_row > p.M) end_row = p.M;
        row_elems = p.Cc;
    }
    const __amdgpu_buffer_rsrc_t ra = mk_rsrc(p.A + base_row * row_elems, (end_row - base_row) * row_elems * 2);
    const __amdgpu_buffer_rsrc_t rb = mk_rsrc(p.Wt, (long long)p.N * p.K * 2);

    // each thread stages half a row (32 bf16 = 64 B = 4 x 16 B) of A and of B per K tile
    const int lrow = tid >> 1, lhalf = tid & 1;
    const long long am = m0 + lrow;
    const bool avalid = am < p.M;
    int ay = 0, ax = 0;
    if (AMODE == AM_CONV3) {
        const long long mm = avalid ? am : m0;
        const long long n_img = mm / hw;
        const int rem = (int)(mm - n_img * hw);
        ay = rem / p.W; ax = rem - ay * p.W;
    }
    const unsigned arow = (unsigned)(((avalid ? am : m0) - base_row) * row_elems + lhalf * 32) * 2u;
    const int bn = n0 + lrow;
    const bool bvalid = bn < p.N;
    const unsigned brow = (unsigned)((long long)(bvalid ? bn : 0) * p.K + lhalf * 32) * 2u;

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int nk = p.K / TBK;
    int tap = 0, c0 = 0;
    u32x4 va[4], vb[4];
    auto gload = [&](int kt) {
        unsigned aoff;
        if (AMODE == AM_PLAIN) {
            aoff = avalid ? arow + (unsigned)(kt * TBK) * 2u : OOB_OFF;
        } else {
            const int ty = (tap * 11) >> 5, tx = tap - ty * 3;
            const bool v = avalid && (unsigned)(ay + ty - 1) < (unsigned)p.H && (unsigned)(ax + tx - 1) < (unsigned)p.W;
            const int shift = ((ty - 1) * p.W + (tx - 1)) * p.Cc + c0;
            aoff = v ? arow + (unsigned)(shift * 2) : OOB_OFF;
            c0 += TBK;
            if (c0 == p.Cc) { c0 = 0; ++tap; }
        }
        const unsigned boff = bvalid ? brow + (unsigned)(kt * TBK) * 2u : OOB_OFF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            va[i] = bufld(ra, aoff == OOB_OFF ? OOB_OFF : aoff + 16u * i);
            vb[i] = bufld(rb, boff == OOB_OFF ? OOB_OFF : boff + 16u * i);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<u32x4*>(&As[buf][lrow][lhalf * 32 + i * 8]) = va[i];
            *reinterpret_cast<u32x4*>(&Bs[buf][lrow][lhalf * 32 + i * 8]) = vb[i];
        }
    };

    const int half = lane >> 5, l31 = lane & 31;
    if (nk > 0) { gload(0); sstore(0); }
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) gload(kt + 1);
#pragma unroll
        for (int ks = 0; ks < TBK / 16; ++ks) {
            const int kc = ks * 16 + half * 8;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(&As[cur][wm * 64 + l31][kc]);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(&As[cur][wm * 64 + 32 + l31][kc]);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(&Bs[cur][wn * 64 + l31][kc]);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(&Bs[cur][wn * 64 + 32 + l31][kc]);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
            if (ks == 1 && more) sstore(cur ^ 1);
        }
        __syncthreads();
        cur ^= 1;
    }

    // ---- epilogue: bias + activation, bf16 stores ----
    float cb[2];
    int ccol[2], ctap[2];
    bool cok[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int col = n0 + wn * 64 + u * 32 + l31;
        cok[u] = col < p.N;
        const int colc = cok[u] ? col : 0;
        ctap[u] = 0; ccol[u] = colc;
        if (EPI == EP_DECONV) { ctap[u] = colc / p.Co; ccol[u] = colc - ctap[u] * p.Co; }
        cb[u] = p.bias ? p.bias[ccol[u]] : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long row = m0 + wm * 64 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (row >= p.M) continue;
            long long rowoff;
            if (EPI == EP_PLAIN) rowoff = row * p.N;
            else {
                const long long n_img = row / hw;
                const int rem = (int)(row - n_img * hw);
                const int y = rem / p.W, x = rem - y * p.W;
                rowoff = n_img * 4 * hw + (long long)y * 4 * p.W + 2 * x;
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (!cok[u]) continue;
                float v = acc[t][u][r] + cb[u];
                if (p.act == MYOLO_ACT_RELU) v = fmaxf(v, 0.f);
                if (EPI == EP_PLAIN) p.C[rowoff + ccol[u]] = f2bf(v);
                else p.C[(rowoff + (long long)(ctap[u] >> 1) * 2 * p.W + (ctap[u] & 1)) * p.Co + ccol[u]] = f2bf(v);
            }
        }
}

// ---- the same GEMM staged by LDS-DMA (buffer_load_dwordx4 ... lds): no staging VGPRs, no ds_write pass ----
// LDS image per operand and buffer: 128 rows x 128 B (64 bf16), unpadded (the DMA destination is wave-uniform base +
// lane x 16 B, so padding is impossible); bank conflicts are avoided by an XOR swizzle of the 16-B chunk index with
// bits 1..3 of the row, applied on the per-lane SOURCE address when filling and on the ds_read_b128 address when reading.
// One wave instruction fills 8 rows; wave w fills rows 32w..32w+31 of A and of B (8 DMA instructions per K tile).
// The MFMA operands are swapped (weights first) so a lane ends up with 4 consecutive output channels of one row:
// the epilogue packs them to 8 B, transposes the tile through LDS and stores full 256-B row segments.
#define CS_ROW 264           // bytes per row of the epilogue staging tile (256 + 8: conflict-free ds_write_b64 / ds_read_b64)
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int AMODE, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_glds(Bf16Args p)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[2][2][TBM * 128];   // [buffer][A|B]  64 KB

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int ntn = (p.N + TBN - 1) / TBN;
    long long bid;
    {
        const long long nwg = gridDim.x, orig = blockIdx.x;
        const long long q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tn = (int)(bid % ntn);
    const long long m0 = (bid / ntn) * TBM;
    const int n0 = tn * TBN;
    const long long hw = (long long)p.H * p.W;

    long long base_row, end_row, row_elems;
    if (AMODE == AM_PLAIN) { base_row = m0; end_row = (m0 + TBM < p.M) ? m0 + TBM : p.M; row_elems = p.K; }
    else {
        base_row = m0 - (p.W + 1); if (base_row < 0) base_row = 0;
        end_row = m0 + TBM + p.W + 1; if (end_row > p.M) end_row = p.M;
        row_elems = p.Cc;
    }
    const __amdgpu_buffer_rsrc_t ra = mk_rsrc(p.A + base_row * row_elems, (end_row - base_row) * row_elems * 2);
    const __amdgpu_buffer_rsrc_t rb = mk_rsrc(p.Wt, (long long)p.N * p.K * 2);

    // per-lane source offsets of the 4 A rows and 4 B rows this lane fills
    unsigned arow[4], brow[4], amask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lrow = wave * 32 + j * 8 + (lane >> 3);
        const unsigned chunk = (unsigned)((lane & 7) ^ ((lrow >> 1) & 7));
        const long long am = m0 + lrow;
        const bool av = am < p.M;
        const long long amc = av ? am : m0;
        arow[j] = (unsigned)((amc - base_row) * row_elems) * 2u + chunk * 16u;
        amask[j] = av ? 0x1ffu : 0u;
        if (AMODE == AM_CONV3 && av) {
            const int rem = (int)(am - (am / hw) * hw);
            const int y = rem / p.W, x = rem - y * p.W;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ty = t / 3, tx = t - ty * 3;
                if ((unsigned)(y + ty - 1) < (unsigned)p.H && (unsigned)(x + tx - 1) < (unsigned)p.W) mk |= 1u << t;
            }
            amask[j] = mk;
        }
        const int bn = n0 + lrow;
        brow[j] = bn < p.N ? (unsigned)((long long)bn * p.K) * 2u + chunk * 16u : OOB_OFF;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int nk = p.K / TBK;
    int tap = 0, c0 = 0;
    auto fill = [&](int kt, int buf) {
        unsigned ashift, abit;
        if (AMODE == AM_PLAIN) { ashift = (unsigned)(kt * TBK) * 2u; abit = 0; }
        else {
            const int ty = (tap * 11) >> 5, tx = tap - ty * 3;
            ashift = (unsigned)((((ty - 1) * p.W + (tx - 1)) * p.Cc + c0) * 2);
            abit = (unsigned)tap;
            c0 += TBK;
            if (c0 == p.Cc) { c0 = 0; ++tap; }
        }
        const unsigned bshift = (unsigned)(kt * TBK) * 2u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned ao = ((amask[j] >> abit) & 1u) ? arow[j] + ashift : OOB_OFF;
            const unsigned bo = brow[j] == OOB_OFF ? OOB_OFF : brow[j] + bshift;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)&lds[buf][0][(wave * 32 + j * 8) * 128], 16, (int)ao, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)&lds[buf][1][(wave * 32 + j * 8) * 128], 16, (int)bo, 0, 0, 0);
        }
    };

    const int half = lane >> 5, l31 = lane & 31;
    const unsigned rsw = (unsigned)((l31 >> 1) & 7);
    if (nk > 0) fill(0, 0);
    __syncthreads();
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) fill(kt + 1, cur ^ 1);
        const unsigned char* Ab = &lds[cur][0][(wm * 64 + l31) * 128];
        const unsigned char* Bb = &lds[cur][1][(wn * 64 + l31) * 128];
#pragma unroll
        for (int ks = 0; ks < TBK / 16; ++ks) {
            const unsigned co = (((unsigned)(ks * 2 + half)) ^ rsw) * 16u;
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(Ab + co);
            const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(Ab + 32 * 128 + co);
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bb + co);
            const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bb + 32 * 128 + co);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
        cur ^= 1;
    }

    if constexpr (EPI == EP_DECONV_MASK) {
        // deconv + ReLU + the 1x1 mask conv: with the swapped operands a lane owns two rows (t) and 32 of the wave's 64
        // columns, so the channel sum is almost entirely in registers; the two halves meet with one shuffle.  The tile's
        // 128 columns are 128 of the Co channels of ONE tap (Co % 128 == 0); column slabs are summed by the finish kernel.
        const int tap = n0 / p.Co;
        const int cbase = n0 - tap * p.Co + wn * 64;
        auto epi = [&](auto ncc) {
            constexpr int NC = decltype(ncc)::value;           // classes this instance accumulates; p.ncls <= NC (no per-element class tests)
            float ps[2][NC];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < NC; ++c) ps[t][c] = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int co = cbase + u * 32 + 8 * g + 4 * half + e;
                        const float b = p.bias ? p.bias[co] : 0.f;
                        const float v0 = fmaxf(acc[0][u][4 * g + e] + b, 0.f), v1 = fmaxf(acc[1][u][4 * g + e] + b, 0.f);
                        const float* w2r = p.w2 + co * p.ncls;
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            const float wv = (NC == 1 || c < p.ncls) ? w2r[c < p.ncls ? c : 0] : 0.f;
                            ps[0][c] = fmaf(v0, wv, ps[0][c]);
                            ps[1][c] = fmaf(v1, wv, ps[1][c]);
                        }
                    }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < NC; ++c) ps[t][c] += __shfl_xor(ps[t][c], 32, 64);
            if (half == 0) {
                const int slab = ((n0 - tap * p.Co) / TBN) * 2 + wn;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const long long row = m0 + wm * 64 + t * 32 + l31;
                    if (row >= p.M) continue;
                    const long long n_img = row / hw;
                    const int rem = (int)(row - n_img * hw);
                    const int y = rem / p.W, x = rem - y * p.W;
                    const long long pix = n_img * 4 * hw + (long long)(2 * y + (tap >> 1)) * 2 * p.W + 2 * x + (tap & 1);
                    float* dst = p.part + ((long long)slab * 4 * p.M + pix) * p.ncls;
#pragma unroll
                    for (int c = 0; c < NC; ++c) if (c < p.ncls) dst[c] = ps[t][c];
                }
            }
        };
        if (p.ncls == 1) epi(IntC<1>{});
        else if (p.ncls == 2) epi(IntC<2>{});
        else epi(IntC<4>{});
        return;
    }
    // ---- epilogue: bias + activation, pack 4 channels, transpose through LDS, row-contiguous stores ----
    unsigned char* Cs = &lds[0][0][0];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wn * 64 + u * 32 + 8 * g + 4 * half;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + nl + e;
                    float b = 0.f;
                    if (p.bias && n < p.N) b = p.bias[EPI == EP_DECONV ? n % p.Co : n];
                    v[e] = acc[t][u][4 * g + e] + b;
                    if (p.act == MYOLO_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                }
                uint2 pk;
                pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                *reinterpret_cast<uint2*>(Cs + (wm * 64 + t * 32 + l31) * CS_ROW + nl * 2) = pk;
            }
    __syncthreads();
    const bool vec_ok = (p.N & 3) == 0 && (EPI == EP_PLAIN || (p.Co & 3) == 0);
#pragma unroll 4
    for (int it = 0; it < 16; ++it) {
        const int idx = it * 256 + tid;
        const int row = idx >> 5, cq = idx & 31;
        const long long m = m0 + row;
        const int n = n0 + cq * 4;
        if (m >= p.M || n >= p.N) continue;
        const uint2 pk = *reinterpret_cast<const uint2*>(Cs + row * CS_ROW + cq * 8);
        long long pix = m;
        if (EPI == EP_DECONV) {
            const long long n_img = m / hw;
            const int rem = (int)(m - n_img * hw);
            const int y = rem / p.W, x = rem - y * p.W;
            pix = n_img * 4 * hw + (long long)y * 4 * p.W + 2 * x;
        }
        if (vec_ok) {
            long long off;
            if (EPI == EP_PLAIN) off = pix * p.N + n;
            else { const int tp = n / p.Co, co = n - tp * p.Co; off = (pix + (long long)(tp >> 1) * 2 * p.W + (tp & 1)) * p.Co + co; }
            *reinterpret_cast<uint2*>(p.C + off) = pk;
        } else {
            const uint16_t e4[4] = {(uint16_t)(pk.x & 0xffff), (uint16_t)(pk.x >> 16), (uint16_t)(pk.y & 0xffff), (uint16_t)(pk.y >> 16)};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (n + e >= p.N) break;
                long long off;
                if (EPI == EP_PLAIN) off = pix * p.N + n + e;
                else { const int tp = (n + e) / p.Co, co = (n + e) - tp * p.Co; off = (pix + (long long)(tp >> 1) * 2 * p.W + (tp & 1)) * p.Co + co; }
                p.C[off] = e4[e];
            }
        }
    }
}

#define W2_ROW 136           // bytes per row of a wave's private 128 x 64 bf16 staging block (128 + 8)

// ---- EP_PLAIN of the 256 x 256 kernels: bias + activation, each wave transposes its 128 x 64 block through its own LDS region ----
__device__ __forceinline__ void epi_plain_256(f32x16 (&acc)[4][2], const Bf16Args& p, unsigned char* lds, long long m0, int n0)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int half = lane >> 5, l31 = lane & 31;
    unsigned char* Ws = lds + wave * 128 * W2_ROW;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = u * 32 + 8 * g + 4 * half;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = n0 + wn * 64 + nl + e;
                    float b = 0.f;
                    if (p.bias && n < p.N) b = p.bias[n];
                    v[e] = acc[t][u][4 * g + e] + b;
                    if (p.act == MYOLO_ACT_RELU) v[e] = fmaxf(v[e], 0.f);
                }
                uint2 pk;
                pk.x = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
                pk.y = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
                *reinterpret_cast<uint2*>(Ws + (t * 32 + l31) * W2_ROW + nl * 2) = pk;
            }
    __syncthreads();
#pragma unroll 4
    for (int it = 0; it < 32; ++it) {                  // 128 rows x 16 chunks of 8 B per wave / 64 lanes
        const int idx = it * 64 + lane;
        const int row = idx >> 4, cq = idx & 15;
        const long long m = m0 + wm * 128 + row;
        const int n = n0 + wn * 64 + cq * 4;
        if (m >= p.M || n >= p.N) continue;
        const uint2 pk = *reinterpret_cast<const uint2*>(Ws + row * W2_ROW + cq * 8);
        *reinterpret_cast<uint2*>(p.C + m * p.N + n) = pk;
    }
}

// ---- 256 x 256 x 64 tiles, 8 waves (2 x 4, each 128 x 64 = 4 x 2 MFMA blocks), same LDS-DMA fill and swizzle ----
// One workgroup per CU (128 KB of LDS for the two buffers): per k step a wave reads 6 fragments for 8 MFMAs (0.75 per
// MFMA against 1.0 in the 128^2 kernel) and the A tile is fetched once for all 256 output channels.  Used for the big
// launches only (>= 6 rounds of 256 workgroups); the 128^2 kernel keeps the small ones and the 2-per-CU granularity.
// EPI PLAIN (each wave transposes its own 128 x 64 block through a private LDS region) and EPI DECONV_MASK.
#define T2M 256
#define T2N 256

// LOOPN (EP_DECONV_MASK): one workgroup per 256 rows walks ALL column blocks (the four taps of the transposed conv) in one software-
// pipelined loop: K is only Cin = 256 deep, so a workgroup per (row tile, tap) spends as long waiting for its first operands and in its
// epilogue as in its four k tiles; in the long loop the next tap's tiles arrive while the previous tap's epilogue runs, and three of the
// four fetches of the activation tile come from L2.  In the inference step 0.62 against 0.67 ms (stand-alone at M = 921984: 0.85 against
// 0.89 ms) once the epilogue stopped spilling (class-count specialisation, pixel indices formed once); bf16_no_loopn=1 is the ablation.
// MEP (EP_DECONV_MASK): the 1x1 mask conv on the matrix pipe.  The accumulators are C^T (a lane holds ONE pixel and 16 channels of a 32 x 32
// block), which is exactly the B-operand layout of v_mfma_f32_32x32x16_bf16 up to a permutation of k -- so max(acc, 0) is rounded to bf16
// in place (v_cvt_pk_bf16_f32 + v_pk_max_i16: two VALU instructions per pair) and multiplied by W2^T (rows = classes, bf16 like every
// other weight of this path, the same k permutation applied when its fragments are built) without leaving the registers: 16 short MFMAs
// per wave and tap against the 128 of the main loop, instead of ~650 VALU instructions and 256 LDS reads.  The deconv output and the
// 1x1 kernel are rounded to bf16 here (the VALU epilogue kept both in fp32: bf16_mask_valu=1).  Stand-alone at M = 662480 (Rice-416,
// batch 4): 0.566 -> 0.44 ms with the finish launch; without any epilogue 0.397 (profiles/r6_notes.md section 6).
// FIN (LOOPN + MEP, Co == 256: a workgroup holds ALL channels of its pixels): the four waves' 64-channel partial logits meet in LDS, in the
// finish kernel's order (bias, slab 0..3), and the sigmoid is stored by this kernel -- no partial-logit round trip through HBM (4 slabs
// x 4 M x classes floats written and read back), no finish launch; bit-identical to partials + deconv_mask_finish.
template <int AMODE, int EPI, bool LOOPN = false, bool MEP = false, bool FIN = false>
__global__ __launch_bounds__(512, 1) void gemm_bf16_256(Bf16Args p)
{
    static_assert(!FIN || (LOOPN && MEP && EPI == EP_DECONV_MASK), "FIN needs the all-taps loop and the matrix-pipe epilogue");
    __shared__ __attribute__((aligned(16))) unsigned char lds[8 * 128 * W2_ROW > 2 * 2 * T2M * 128 ? 8 * 128 * W2_ROW : 2 * 2 * T2M * 128];
    unsigned char (*buf)[2][T2M * 128] = reinterpret_cast<unsigned char (*)[2][T2M * 128]>(lds);      // [buffer][A|B][row*128 + chunk*16]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntn = (p.N + T2N - 1) / T2N;
    long long bid;
    {
        const long long nwg = gridDim.x, orig = blockIdx.x;
        const long long q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tn = LOOPN ? 0 : (int)(bid % ntn);
    const long long m0 = (LOOPN ? bid : bid / ntn) * T2M;
    const int n0 = tn * T2N;
    const long long hw = (long long)p.H * p.W;

    long long base_row, end_row, row_elems;
    if (AMODE == AM_PLAIN) { base_row = m0; end_row = (m0 + T2M < p.M) ? m0 + T2M : p.M; row_elems = p.K; }
    else {
        base_row = m0 - (p.W + 1); if (base_row < 0) base_row = 0;
        end_row = m0 + T2M + p.W + 1; if (end_row > p.M) end_row = p.M;
        row_elems = p.Cc;
    }
    const __amdgpu_buffer_rsrc_t ra = mk_rsrc(p.A + base_row * row_elems, (end_row - base_row) * row_elems * 2);
    const __amdgpu_buffer_rsrc_t rb = mk_rsrc(p.Wt, (long long)p.N * p.K * 2);

    // EP_DECONV_MASK: {deconv bias, 1x1 mask-conv weights of up to four classes} of the tile's 256 output channels, in LDS before the
    // main loop (the epilogue used to fetch them from global memory element by element, with the accumulators pinning every register)
    __shared__ float4 etab[EPI == EP_DECONV_MASK && !MEP ? (LOOPN ? 512 : T2N) : 1];
    __shared__ float ebias[EPI == EP_DECONV_MASK && MEP ? (LOOPN ? 512 : T2N) : 1];              // MEP: the deconv bias alone
    __shared__ float4 ered[FIN ? 2 * 4 * 4 * 32 : 1];                                             // FIN: [wm][wn][t][pixel lane] partial logits of a tap
    __shared__ float etab3[EPI == EP_DECONV_MASK && !MEP ? (LOOPN ? 512 : T2N) : 1];
    // MEP: W2^T fragments [group of 16 channels][half][class 0..3][8 bf16]: element i of (group q, half h) is channel
    // 16 q + 8 (i / 4) + 4 h + i % 4 -- the channel accumulator register 8 pr + i of a lane of that half holds (q = 2 u + pr); one more
    // entry of zeros at the end for the lanes whose row of W2^T is no class
    __shared__ __attribute__((aligned(16))) uint16_t w2tab[EPI == EP_DECONV_MASK && MEP ? ((LOOPN ? 512 : T2N) / 2 + 1) * 8 : 8];
    if constexpr (EPI == EP_DECONV_MASK) {
        if (tid < (LOOPN ? p.Co : T2N)) {                        // LOOPN: all Co (<= 512) channels, indexed by channel
            const int co = LOOPN ? tid : n0 % p.Co + tid;
            const float* w2r = p.w2 + (long long)co * p.ncls;
            if constexpr (MEP) {
                ebias[tid] = p.bias ? p.bias[co] : 0.f;
                const int nent = (LOOPN ? p.Co : T2N) / 2;
                if (tid <= nent) {
                    const int q = tid >> 3, h = (tid >> 2) & 1, cls = tid & 3;
                    const int cb = LOOPN ? 0 : n0 % p.Co;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const int ch = cb + q * 16 + 8 * (i >> 2) + 4 * h + (i & 3);
                        w2tab[tid * 8 + i] = (tid < nent && cls < p.ncls) ? f2bf(p.w2[(long long)ch * p.ncls + cls]) : (uint16_t)0;
                    }
                }
            } else {
                etab[tid] = make_float4(p.bias ? p.bias[co] : 0.f, w2r[0], p.ncls > 1 ? w2r[1] : 0.f, p.ncls > 2 ? w2r[2] : 0.f);
                etab3[tid] = p.ncls > 3 ? w2r[3] : 0.f;
            }
        }
    }

    unsigned arow[4], brow[4], amask[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lrow = wave * 32 + j * 8 + (lane >> 3);
        const unsigned chunk = (unsigned)((lane & 7) ^ ((lrow >> 1) & 7));
        const long long am = m0 + lrow;
        const bool av = am < p.M;
        const long long amc = av ? am : m0;
        arow[j] = (unsigned)((amc - base_row) * row_elems) * 2u + chunk * 16u;
        amask[j] = av ? 0x1ffu : 0u;
        if (AMODE == AM_CONV3 && av) {
            const int rem = (int)(am - (am / hw) * hw);
            const int y = rem / p.W, x = rem - y * p.W;
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ty = t / 3, tx = t - ty * 3;
                if ((unsigned)(y + ty - 1) < (unsigned)p.H && (unsigned)(x + tx - 1) < (unsigned)p.W) mk |= 1u << t;
            }
            amask[j] = mk;
        }
        const int bn = n0 + lrow;
        brow[j] = bn < p.N ? (unsigned)((long long)bn * p.K) * 2u + chunk * 16u : OOB_OFF;
    }

    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int nk = p.K / TBK;
    int tap = 0, c0 = 0;
    unsigned f_ashift = 0, f_abit = 0, f_bshift = 0;
    auto fill_begin = [&](int kt) {              // operand offsets of k tile kt (CONV3: tap and channel block of the implicit im2col)
        int tnf = 0;
        if (LOOPN) { tnf = kt / nk; kt -= tnf * nk; }
        if (AMODE == AM_PLAIN) { f_ashift = (unsigned)(kt * TBK) * 2u; f_abit = 0; }
        else {
            const int ty = (tap * 11) >> 5, tx = tap - ty * 3;
            f_ashift = (unsigned)((((ty - 1) * p.W + (tx - 1)) * p.Cc + c0) * 2);
            f_abit = (unsigned)tap;
            c0 += TBK;
            if (c0 == p.Cc) { c0 = 0; ++tap; }
        }
        f_bshift = (unsigned)(kt * TBK) * 2u + (unsigned)tnf * (unsigned)(T2N * p.K * 2);
    };
    auto fill_piece = [&](int j, int b, bool live) {        // 8 rows of A and 8 rows of B per wave and piece, straight into LDS
        // (live == false, past the last k tile: out-of-range offsets, the DMA writes zeros nobody reads -- keeps the loop body branch-free,
        //  which the instruction-group scheduling below needs)
        const unsigned ao = (live && ((amask[j] >> f_abit) & 1u)) ? arow[j] + f_ashift : OOB_OFF;
        const unsigned bo = (!live || brow[j] == OOB_OFF) ? OOB_OFF : brow[j] + f_bshift;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)&buf[b][0][(wave * 32 + j * 8) * 128], 16, (int)ao, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)&buf[b][1][(wave * 32 + j * 8) * 128], 16, (int)bo, 0, 0, 0);
    };

    const int half = lane >> 5, l31 = lane & 31;
    const unsigned rsw = (unsigned)((l31 >> 1) & 7);
    if (nk > 0) {
        fill_begin(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) fill_piece(j, 0, true);
    }
    __syncthreads();
    // The loop is scheduled by hand (sched_group_barrier): a wave's MFMAs issue back to back and every LDS read / LDS-DMA piece sits in
    // the 32-cycle shadow of one of them.  Fragments are double-buffered in registers (the six ds_read_b128 of k step ks+1 ride on the
    // MFMAs of step ks, in the order the next step consumes them); the eight LDS-DMA pieces of the next k tile ride on steps 0 and 1, so
    // they have two steps to land; the last step issues two MFMAs, waits for the DMA, passes the barrier and reads the next tile's first
    // fragments under its remaining six MFMAs.  (The compiler's own schedule put the 8 DMA pieces in one block and waited on each group of
    // reads in front of its MFMAs: 979 TFLOP/s.)
    // EP_DECONV_MASK epilogue of one 256-column block (columns nb .. nb+255 = 256 channels of ONE tap): deconv + bias + ReLU and the 1x1
    // mask conv.  A lane owns one pixel per t and 32 of the wave's 64 channels, so the channel sum is almost entirely in registers; the
    // two halves meet with one shuffle.  The 64-column slabs are summed by the finish kernel.  Two pixels (t) at a time keep the
    // partial sums in 8 registers.
    // EP_DECONV_MASK: output pixel (on the 2H x 2W grid, tap (0,0)) of this lane's four rows, formed once -- the epilogue runs per column
    // block with every register taken, and the 64-bit divisions of the row -> (image, y, x) mapping used to spill there
    long long pixb[4] = {-1, -1, -1, -1};
    if constexpr (EPI == EP_DECONV_MASK) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const long long row = m0 + wm * 128 + t * 32 + (lane & 31);
            if (row < p.M) {
                const long long n_img = row / hw;
                const int rem = (int)(row - n_img * hw);
                const int y = rem / p.W, x = rem - y * p.W;
                pixb[t] = n_img * 4 * hw + (long long)(2 * y) * 2 * p.W + 2 * x;
            }
        }
    }
    // EP_DECONV_MASK: the accumulators of a column block START at the deconv bias of their channel (the table is in LDS by now), so the
    // epilogue is max(acc, 0) and the class FMAs -- one VALU instruction less per accumulator element
    auto acc_start = [&](int nb) {
        if constexpr (EPI == EP_DECONV_MASK) {
            const int tap2 = nb / p.Co;
            const int tb = LOOPN ? nb - tap2 * p.Co + wn * 64 : wn * 64;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int bi = tb + u * 32 + 8 * g + 4 * half + e;
                        float b;
                        if constexpr (MEP) b = ebias[bi]; else b = etab[bi].x;
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[t][u][4 * g + e] = b;
                    }
        }
    };
    acc_start(n0);
    auto mask_epilogue_nc = [&](auto ncc, int nb) {
        constexpr int NC = decltype(ncc)::value;                       // classes this instance accumulates (2 or 4); p.ncls <= NC
        const int tap2 = nb / p.Co;
        const int c0w = nb - tap2 * p.Co + wn * 64;                    // first channel of the wave's 64
        const int tb = LOOPN ? c0w : wn * 64;                          // its row in the {bias, w} table
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {
            float ps[2][NC];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < NC; ++c) ps[t][c] = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int cl = tb + u * 32 + 8 * g + 4 * half + e;
                        asm volatile("" : "+v"(cl));            // re-read the table in the second pass instead of holding 160 registers of it
                        const float4 bw = etab[cl];
                        float w3 = 0.f;
                        if constexpr (NC > 3) w3 = etab3[cl];
#pragma unroll
                        for (int t = 0; t < 2; ++t) {
                            const float v = fmaxf(acc[2 * tp + t][u][4 * g + e], 0.f);       // (the bias is in the accumulator since acc_start)
                            // the first two classes as ONE packed fp32 FMA (v_pk_fma_f32)
                            f32x2 pp = {ps[t][0], ps[t][1]};
                            const f32x2 ww = {bw.y, bw.z}, vv = {v, v};
                            pp = __builtin_elementwise_fma(vv, ww, pp);
                            ps[t][0] = pp.x; ps[t][1] = pp.y;
                            if constexpr (NC > 2) ps[t][2] = fmaf(v, bw.w, ps[t][2]);
                            if constexpr (NC > 3) ps[t][3] = fmaf(v, w3, ps[t][3]);
                        }
                    }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int c = 0; c < NC; ++c) ps[t][c] += __shfl_xor(ps[t][c], 32, 64);
            if (half == 0) {
                const int slab = c0w / 64;                                 // 64-column slabs: Co/64 of them per tap
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (pixb[2 * tp + t] < 0) continue;
                    const long long pix = pixb[2 * tp + t] + (long long)(tap2 >> 1) * 2 * p.W + (tap2 & 1);
                    float* dst = p.part + ((long long)slab * 4 * p.M + pix) * p.ncls;
#pragma unroll
                    for (int c = 0; c < NC; ++c) if (c < p.ncls) dst[c] = ps[t][c];
                }
            }
        }
    };
    auto mask_epilogue_mfma = [&](int nb) {
        const int tap2 = nb / p.Co;
        const int c0w = nb - tap2 * p.Co + wn * 64;                    // first channel of the wave's 64
        const int q0 = (LOOPN ? c0w : wn * 64) >> 4;
        bf16x8 wf[2][2];                                               // [u][16-channel half of the block]; rows >= 4 (no class): the zero entry
        const int zent = (LOOPN ? p.Co : T2N) / 2;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                const int ent = l31 < 4 ? ((q0 + 2 * u + pr) * 2 + half) * 4 + l31 : zent;
                wf[u][pr] = *reinterpret_cast<const bf16x8*>(&w2tab[ent * 8]);
            }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 o;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int pr = 0; pr < 2; ++pr) {
                    u32x4 rr;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {                      // (the bias is in the accumulator since acc_start)
                        const f32x2 a2 = {acc[t][u][8 * pr + 2 * j], acc[t][u][8 * pr + 2 * j + 1]};
                        s16x2 b2 = __builtin_bit_cast(s16x2, __builtin_convertvector(a2, bf16x2));
                        b2 = __builtin_elementwise_max(b2, s16x2{0, 0});         // ReLU on the bf16 bits: negative <=> sign bit <=> negative int16
                        rr[j] = __builtin_bit_cast(unsigned, b2);
                    }
                    const bf16x8 rb8 = __builtin_bit_cast(bf16x8, rr);
                    o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[u][pr], rb8, o, 0, 0, 0);
                }
            // rows 0..3 of the product = the classes, in registers 0..3 of the lower half
            if constexpr (FIN) {
                if (half == 0) ered[((wm * 4 + wn) * 4 + t) * 32 + l31] = make_float4(o[0], o[1], o[2], o[3]);
            } else if (half == 0 && pixb[t] >= 0) {
                const long long pix = pixb[t] + (long long)(tap2 >> 1) * 2 * p.W + (tap2 & 1);
                float* dst = p.part + ((long long)(c0w / 64) * 4 * p.M + pix) * p.ncls;
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c < p.ncls) dst[c] = o[c];
            }
        }
        if constexpr (FIN) {
            __syncthreads();               // (the next tap's partials are written four k-tile barriers from here)
            const int wn_s = __builtin_amdgcn_readfirstlane(wn);
#pragma unroll
            for (int t = 0; t < 4; ++t)                                // wave (wm, wn) finishes pixel group t = wn of its 128 rows
                if (t == wn_s && half == 0 && pixb[t] >= 0) {
                    const long long pix = pixb[t] + (long long)(tap2 >> 1) * 2 * p.W + (tap2 & 1);
                    float4 sl[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) sl[k] = ered[((wm * 4 + k) * 4 + t) * 32 + l31];
#pragma unroll
                    for (int c = 0; c < 4; ++c)
                        if (c < p.ncls) {
                            float sacc = p.b2[c];
#pragma unroll
                            for (int k = 0; k < 4; ++k) sacc += c == 0 ? sl[k].x : c == 1 ? sl[k].y : c == 2 ? sl[k].z : sl[k].w;
                            p.out[pix * p.ncls + c] = 1.f / (1.f + expf(-sacc));
                        }
                }
        }
    };
    auto mask_epilogue = [&](int nb) {
        if constexpr (MEP) mask_epilogue_mfma(nb);
        else if (p.ncls <= 2) mask_epilogue_nc(IntC<2>{}, nb);
        else mask_epilogue_nc(IntC<4>{}, nb);
    };
    bf16x8 fa[2][4], fb[2][2];
    auto rd = [&](int b, int ks, int slot) {
        const unsigned char* Ab = &buf[b][0][(wm * 128 + l31) * 128];
        const unsigned char* Bb = &buf[b][1][(wn * 64 + l31) * 128];
        const unsigned co = (((unsigned)(ks * 2 + half)) ^ rsw) * 16u;
        fb[slot][0] = *reinterpret_cast<const bf16x8*>(Bb + co);
        fa[slot][0] = *reinterpret_cast<const bf16x8*>(Ab + co);
        fb[slot][1] = *reinterpret_cast<const bf16x8*>(Bb + 32 * 128 + co);
#pragma unroll
        for (int t = 1; t < 4; ++t) fa[slot][t] = *reinterpret_cast<const bf16x8*>(Ab + t * 32 * 128 + co);
    };
    auto mma = [&](int sl, int t0, int t1) {
#pragma unroll
        for (int t = t0; t < t1; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[sl][u], fa[sl][t], acc[t][u], 0, 0, 0);
    };
    int cur = 0;
    if (nk > 0) rd(0, 0, 0);
    const int nk_all = LOOPN ? nk * ntn : nk;
    for (int kt = 0; kt < nk_all; ++kt) {
        const bool more = kt + 1 < nk_all;
        fill_begin(kt + 1);
#pragma unroll
        for (int ks = 0; ks < TBK / 16; ++ks) {
            const int sl = ks & 1;
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 1 < TBK / 16) {
                rd(cur, ks + 1, sl ^ 1);
                if (ks < 2) { fill_piece(2 * ks, cur ^ 1, more); fill_piece(2 * ks + 1, cur ^ 1, more); }
                mma(sl, 0, 4);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (ks < 2) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (ks < 2) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
            } else {
                mma(sl, 0, 1);
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
                rd(cur ^ 1, 0, sl ^ 1);         // (past the last k tile: stale bytes, never used)
                mma(sl, 1, 4);
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        cur ^= 1;
        if constexpr (LOOPN) {
            const int kk = kt + 1;
            if (kk % nk == 0) {                       // a column block is complete: its epilogue, then the accumulators start over
#ifdef BF16_TUNE
                if (!(p.tune & 1))
#endif
                mask_epilogue((kk / nk - 1) * T2N);
#ifdef BF16_TUNE
                if constexpr (EPI == EP_DECONV_MASK) { if (!(p.tune & 2)) acc_start((kk / nk) * T2N < p.N ? (kk / nk) * T2N : 0); }
#else
                if constexpr (EPI == EP_DECONV_MASK) acc_start((kk / nk) * T2N < p.N ? (kk / nk) * T2N : 0);
#endif
                else {
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;
                }
            }
        }
    }

    if constexpr (EPI == EP_DECONV_MASK) {
        if constexpr (!LOOPN) mask_epilogue(n0);
        return;
    }
    epi_plain_256(acc, p, lds, m0, n0);
}

// ---- 3x3 / s1 / SAME conv as an implicit GEMM whose A operand is fetched ONCE per channel block -------------------------------
// gemm_bf16_256<AM_CONV3> re-fetches the 256 activation rows of its tile from L2 for each of the nine taps; with the weight tile that
// is 64 KB of L2 -> LDS traffic per k tile, and the kernel turned out to be bound by exactly that traffic (a timing-only experiment
// that skipped the A fetch on eight of the nine taps ran 20 % faster).  Here the k loop is (channel block) x (tap): the 256 + 2(W+1)
// rows a tile can touch are put in LDS once per 64-channel block, and the nine taps read their fragments from that block at a row
// offset -- the im2col happens in the LDS address.  Pixels whose tap falls outside the image read a row of zeros kept at the end of
// LDS.  Per k tile only the weight tile (32 KB) crosses L2 -> LDS, plus 1/9 of the 40 KB activation block.
//   LDS: two activation blocks of C3_AROWS rows x 128 B (same XOR swizzle as above, keyed by the block row), two weight tiles, zero row.
#define C3_AROWS 320                     // >= 256 + 2 (W + 1): W <= 31
#define C3_LDS (2 * C3_AROWS * 128 + 2 * T2N * 128 + 128)

__global__ __launch_bounds__(512, 1) void conv3_bf16_256(Bf16Args p)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[C3_LDS];
    unsigned char* const abuf = lds;                               // [2][C3_AROWS * 128]
    unsigned char* const bbuf = lds + 2 * C3_AROWS * 128;          // [2][T2N * 128]
    const unsigned zoff = 2 * C3_AROWS * 128 + 2 * T2N * 128;      // the zero row

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int ntn = p.N / T2N;
    long long bid;
    {
        const long long nwg = gridDim.x, orig = blockIdx.x;
        const long long q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
    }
    const int tn = (int)(bid % ntn);
    const long long m0 = (bid / ntn) * T2M;
    const int n0 = tn * T2N;
    const int hw = p.H * p.W;
    const int halo = p.W + 1, arows = T2M + 2 * halo;

    long long base_row = m0 - halo; if (base_row < 0) base_row = 0;
    long long end_row = m0 + T2M + halo; if (end_row > p.M) end_row = p.M;
    const __amdgpu_buffer_rsrc_t ra = mk_rsrc(p.A + base_row * p.Cc, (end_row - base_row) * p.Cc * 2);
    const __amdgpu_buffer_rsrc_t rb = mk_rsrc(p.Wt, (long long)p.N * p.K * 2);

    if (tid < 32) *reinterpret_cast<unsigned*>(lds + zoff + tid * 4) = 0u;

    // activation block: 40 pieces of 8 rows (piece q = wave + 8 i), block row r holds input row m0 - halo + r
    unsigned aoff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int prow = (wave + 8 * i) * 8 + (lane >> 3);
        const unsigned chunk = (unsigned)((lane & 7) ^ ((prow >> 1) & 7));
        const long long g = m0 - halo + prow;
        aoff[i] = (prow < arows && g >= 0 && g < p.M) ? (unsigned)((g - base_row) * p.Cc) * 2u + chunk * 16u : OOB_OFF;
    }
    unsigned brow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int lrow = wave * 32 + j * 8 + (lane >> 3);
        const unsigned chunk = (unsigned)((lane & 7) ^ ((lrow >> 1) & 7));
        brow[j] = (unsigned)((long long)(n0 + lrow) * p.K) * 2u + chunk * 16u;
    }
    auto a_piece = [&](int i, int cb, int b, bool live) {
        const unsigned ao = (live && aoff[i] != OOB_OFF) ? aoff[i] + (unsigned)cb * 128u : OOB_OFF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_ptr_t)(abuf + b * (C3_AROWS * 128) + (wave + 8 * i) * 1024), 16, (int)ao, 0, 0, 0);
    };
    auto b_piece = [&](int j, int tap, int cb, int b, bool live) {
        const unsigned bo = live ? brow[j] + (unsigned)(tap * p.Cc + cb * TBK) * 2u : OOB_OFF;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_ptr_t)(bbuf + b * (T2N * 128) + (wave * 32 + j * 8) * 128), 16, (int)bo, 0, 0, 0);
    };

    // this lane's four output pixels (rows wm*128 + 32 t + l31 of the tile) and, per pixel, which of the nine taps stay inside the image
    const int half = lane >> 5, l31 = lane & 31;
    unsigned mask9[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const long long m = m0 + wm * 128 + t * 32 + l31;
        unsigned mk = 0;
        if (m < p.M) {
            const int rem = (int)(m % hw);
            const int y = rem / p.W, x = rem - y * p.W;
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
                const int ty = tp / 3, tx = tp - ty * 3;
                if ((unsigned)(y + ty - 1) < (unsigned)p.H && (unsigned)(x + tx - 1) < (unsigned)p.W) mk |= 1u << tp;
            }
        }
        mask9[t] = mk;
    }
    const int prow0 = wm * 128 + l31 + halo;                       // block row of pixel t = 0 at the centre tap; + 32 t, + tap shift
    const unsigned rswB = (unsigned)((l31 >> 1) & 7);
    const unsigned cB = (unsigned)(wn * 64 + l31) * 128u;

    f32x16 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][u][r] = 0.f;

    const int ncb = p.Cc / TBK;
#pragma unroll
    for (int i = 0; i < 5; ++i) a_piece(i, 0, 0, true);
#pragma unroll
    for (int j = 0; j < 4; ++j) b_piece(j, 0, 0, 0, true);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) b_piece(j, 1, 0, 1, true);         // k tile 1 (tap 1 of block 0)

    // fragment addresses of one tap: LDS byte address of this lane's row in the activation block (or the zero row) per pixel, and the
    // swizzle term of that row (32 t and the tap shift change (row >> 1) & 7 the same way for all four pixels)
    unsigned arow[4], arsw = 0;
    auto tap_addr = [&](int tap, int b) {
        const int ty = tap / 3, tx = tap - ty * 3;
        int pr = prow0 + (ty - 1) * p.W + (tx - 1);
        asm volatile("" : "+v"(pr));                  // no per-(tap, k step) address table in registers (it would be spilled)
        arsw = (unsigned)((pr >> 1) & 7) << 4;
        const unsigned base = (unsigned)(b * (C3_AROWS * 128)) + (unsigned)pr * 128u;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            unsigned mk = mask9[t];
            asm volatile("" : "+v"(mk));              // likewise for the 36 (pixel, tap) tests
            arow[t] = ((mk >> tap) & 1u) ? base + (unsigned)t * 4096u : zoff;
        }
    };
    bf16x8 fa[2][4], fb[2][2];
    auto rd = [&](int bb, int ks, int slot) {
        const unsigned c16 = (unsigned)(ks * 2 + half) << 4;
        const unsigned char* Bb = bbuf + bb * (T2N * 128) + cB + ((c16 >> 4) ^ rswB) * 16u;
        const unsigned xa = c16 ^ arsw;
        fb[slot][0] = *reinterpret_cast<const bf16x8*>(Bb);
        fa[slot][0] = *reinterpret_cast<const bf16x8*>(lds + arow[0] + xa);
        fb[slot][1] = *reinterpret_cast<const bf16x8*>(Bb + 32 * 128);
#pragma unroll
        for (int t = 1; t < 4; ++t) fa[slot][t] = *reinterpret_cast<const bf16x8*>(lds + arow[t] + xa);
    };
    auto mma = [&](int sl, int t0, int t1) {
#pragma unroll
        for (int t = t0; t < t1; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) acc[t][u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[sl][u], fa[sl][t], acc[t][u], 0, 0, 0);
    };

    // Schedule of k tile t = (cb, tap), everything in the shadow of an MFMA (see gemm_bf16_256):
    //   step 0   : fragments of step 1; for taps 0..4 one piece of the next activation block
    //   step 1, 2: fragments of steps 2, 3
    //   step 3   : next tap's row addresses; two MFMAs; wait for the weight tile of t+1; barrier (tile t's buffer is free now);
    //              first fragments of t+1 and the whole weight tile of t+2 (four pieces) under the remaining six MFMAs
    // A weight piece (L2) has four steps (of 8 MFMAs x 2 waves per SIMD) to land.  The activation piece (HBM) is the youngest VMEM
    // operation at its tile's barrier, which therefore waits with vmcnt(1) and leaves it in flight until the next tile's barrier.
    tap_addr(0, 0);
    rd(0, 0, 0);
    for (int cb = 0; cb < ncb; ++cb) {
        const int pa = cb & 1;                       // activation block buffer; weight tile buffer = (cb + tap) & 1 (nine taps: parity flips)
        const bool more_cb = cb + 1 < ncb;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int bcur = (cb + tap) & 1;
            const bool last_tap = tap == 8;
            const bool more2 = tap < 7 || more_cb;                                       // k tile t+2 exists
            const int tap1 = tap < 8 ? tap + 1 : 0;
            const int tap2 = tap < 7 ? tap + 2 : tap - 7, cb2 = tap < 7 ? cb : cb + 1;
#pragma unroll
            for (int ks = 0; ks < TBK / 16; ++ks) {
                const int sl = ks & 1;
                __builtin_amdgcn_sched_barrier(0);
                if (ks + 1 < TBK / 16) {
                    rd(bcur, ks + 1, sl ^ 1);
                    if (ks == 0 && tap < 5) a_piece(tap, cb + 1, pa ^ 1, more_cb);
                    mma(sl, 0, 4);
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (ks == 0 && tap < 5) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                } else {
                    tap_addr(tap1, last_tap ? pa ^ 1 : pa);          // the current tap's last fragments were read in step 2
                    mma(sl, 0, 1);
                    __builtin_amdgcn_sched_barrier(0);
                    // s_waitcnt vmcnt(tap < 5 ? 1 : 0) lgkmcnt(0) [gfx9 encoding: vmcnt simm[3:0], expcnt simm[6:4] = 7 (no wait), lgkmcnt simm[11:8]]
                    if (tap < 5) __builtin_amdgcn_s_waitcnt(0x0071); else __builtin_amdgcn_s_waitcnt(0x0070);
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    rd(bcur ^ 1, 0, sl ^ 1);                         // (past the last k tile: stale bytes, never used)
#pragma unroll
                    for (int j = 0; j < 4; ++j) b_piece(j, tap2, cb2, bcur, more2);
                    mma(sl, 1, 4);
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if (i >= 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __syncthreads();
    epi_plain_256(acc, p, lds, m0, n0);
}

// ROIAlign (crop_and_resize) with fp32 feature map in, bf16 out -- same coordinate arithmetic as crop_fwd_kernel
__device__ __forceinline__ bool crop_coord_b(float lo, float hi, int size, int crop, int idx, float& in)
{
    if (crop > 1) {
        const float scale = (hi - lo) * (float)(size - 1) / (float)(crop - 1);
        in = lo * (float)(size - 1) + (float)idx * scale;
    } else {
        in = 0.5f * (lo + hi) * (float)(size - 1);
    }
    return !(in < 0.f || in > (float)(size - 1));
}

__global__ __launch_bounds__(256) void crop_fwd_bf16_kernel(const float* __restrict__ img, const float* __restrict__ boxes,
                                                            const int32_t* __restrict__ bind, uint16_t* __restrict__ out,
                                                            int H, int W, int C, int ch, int cw)
{
    const int b = blockIdx.y, py = blockIdx.x;
    const int cq = C / 4;
    const float4 bx = *reinterpret_cast<const float4*>(boxes + (long long)b * 4);
    float iny;
    const bool vy = crop_coord_b(bx.x, bx.z, H, ch, py, iny);
    const int ty = (int)floorf(iny), by = (int)ceilf(iny);
    const float wy = iny - (float)ty;
    const float* base = img + (long long)bind[b] * H * W * C;
    uint16_t* orow = out + ((long long)b * ch + py) * cw * C;
    const unsigned total = (unsigned)(cw * cq);
    for (unsigned e = threadIdx.x; e < total; e += blockDim.x) {
        const int px = e / (unsigned)cq;
        const int c = (e - px * cq) * 4;
        float inx;
        const bool vx = crop_coord_b(bx.y, bx.w, W, cw, px, inx);
        float o[4] = {0.f, 0.f, 0.f, 0.f};
        if (vy && vx) {
            const int lx = (int)floorf(inx), rx = (int)ceilf(inx);
            const float wx = inx - (float)lx;
            const float4 tl = *reinterpret_cast<const float4*>(base + ((long long)ty * W + lx) * C + c);
            const float4 tr = *reinterpret_cast<const float4*>(base + ((long long)ty * W + rx) * C + c);
            const float4 bl = *reinterpret_cast<const float4*>(base + ((long long)by * W + lx) * C + c);
            const float4 br = *reinterpret_cast<const float4*>(base + ((long long)by * W + rx) * C + c);
            const float tlv[4] = {tl.x, tl.y, tl.z, tl.w}, trv[4] = {tr.x, tr.y, tr.z, tr.w};
            const float blv[4] = {bl.x, bl.y, bl.z, bl.w}, brv[4] = {br.x, br.y, br.z, br.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float top = tlv[k] + (trv[k] - tlv[k]) * wx, bot = blv[k] + (brv[k] - blv[k]) * wx;
                o[k] = top + (bot - top) * wy;
            }
        }
        uint2 pk;
        pk.x = (unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16);
        pk.y = (unsigned)f2bf(o[2]) | ((unsigned)f2bf(o[3]) << 16);
        *reinterpret_cast<uint2*>(orow + (long long)px * C + c) = pk;
    }
}

// The same crops with every feature-map column fetched ONCE per output row.  crop_fwd_bf16_kernel reads four corners per output element:
// 64 bytes from L1 / L2 for 8 bytes stored -- 2.7 GB through the vector caches for 339 MB written at the Rice-416 shape, and that traffic, not
// the HBM write, set its 0.126 ms.  The x coordinates of a box are the same for all of its rows and channels, and with boxes narrower than
// ~2 x the crop (every anchor of the Rice / Shapes configs) neighbouring sample points share columns: lx(px + 1) is lx(px) or rx(px).  Here a
// thread owns (output row, 8 channels) and walks the crop's columns; the two feature-map columns of a sample (rows ty / by, 2 x 32 bytes each)
// stay in registers, and a column is loaded only when the walk reaches one it does not hold.  The column indices are wave-uniform (they depend
// on the box alone), so the reuse tests are scalar branches.  Same expressions per element as crop_fwd_bf16_kernel: the same bits.
struct CropCol { float4 t0, t1, b0, b1; };           // rows ty / by of one feature-map column, this thread's 8 channels
__global__ __launch_bounds__(256) void crop_fwd_bf16_walk_kernel(const float* __restrict__ img, const float* __restrict__ boxes,
                                                                 const int32_t* __restrict__ bind, uint16_t* __restrict__ out,
                                                                 int H, int W, int C, int ch, int cw)
{
    const int b = blockIdx.y;
    const int co = C >> 3;
    const int item = blockIdx.x * 256 + threadIdx.x;
    const bool live = item < ch * co;
    const int py = live ? item / co : 0;
    const int c = (live ? item - py * co : 0) * 8;
    const float4 bx = *reinterpret_cast<const float4*>(boxes + (long long)b * 4);
    float iny;
    const bool vy = crop_coord_b(bx.x, bx.z, H, ch, py, iny);
    const int ty = (int)floorf(iny), by = (int)ceilf(iny);
    const float wy = iny - (float)ty;
    const float* base = img + (long long)bind[b] * H * W * C + c;
    const float* rowt = base + (long long)(vy ? ty : 0) * W * C;       // (a row outside the image is never used: its outputs are zeros)
    const float* rowb = base + (long long)(vy ? by : 0) * W * C;
    uint16_t* orow = out + ((long long)b * ch + py) * cw * C + c;
    CropCol L, R;
    L.t0 = L.t1 = L.b0 = L.b1 = R.t0 = R.t1 = R.b0 = R.b1 = make_float4(0.f, 0.f, 0.f, 0.f);
    int il = -1, ir = -1;                               // the columns L / R hold
    auto fetch = [&](CropCol& d, int x) {
        const float* pt = rowt + (long long)x * C;
        const float* pb = rowb + (long long)x * C;
        d.t0 = *reinterpret_cast<const float4*>(pt); d.t1 = *reinterpret_cast<const float4*>(pt + 4);
        d.b0 = *reinterpret_cast<const float4*>(pb); d.b1 = *reinterpret_cast<const float4*>(pb + 4);
    };
    for (int px = 0; px < cw; ++px) {
        float inx;
        const bool vx = crop_coord_b(bx.y, bx.w, W, cw, px, inx);
        uint4 pk = make_uint4(0u, 0u, 0u, 0u);
        if (__builtin_amdgcn_readfirstlane((int)vx)) {
            const int lx = __builtin_amdgcn_readfirstlane((int)floorf(inx)), rx = __builtin_amdgcn_readfirstlane((int)ceilf(inx));
            const float wx = inx - (float)lx;
            if (lx != il) { if (lx == ir) L = R; else fetch(L, lx); il = lx; }
            if (rx != ir) { if (rx == il) R = L; else fetch(R, rx); ir = rx; }
            const float tl[8] = {L.t0.x, L.t0.y, L.t0.z, L.t0.w, L.t1.x, L.t1.y, L.t1.z, L.t1.w};
            const float tr[8] = {R.t0.x, R.t0.y, R.t0.z, R.t0.w, R.t1.x, R.t1.y, R.t1.z, R.t1.w};
            const float bl[8] = {L.b0.x, L.b0.y, L.b0.z, L.b0.w, L.b1.x, L.b1.y, L.b1.z, L.b1.w};
            const float br[8] = {R.b0.x, R.b0.y, R.b0.z, R.b0.w, R.b1.x, R.b1.y, R.b1.z, R.b1.w};
            float o[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const float top = tl[k] + (tr[k] - tl[k]) * wx, bot = bl[k] + (br[k] - bl[k]) * wx;
                o[k] = vy ? top + (bot - top) * wy : 0.f;
            }
            pk.x = (unsigned)f2bf(o[0]) | ((unsigned)f2bf(o[1]) << 16);
            pk.y = (unsigned)f2bf(o[2]) | ((unsigned)f2bf(o[3]) << 16);
            pk.z = (unsigned)f2bf(o[4]) | ((unsigned)f2bf(o[5]) << 16);
            pk.w = (unsigned)f2bf(o[6]) | ((unsigned)f2bf(o[7]) << 16);
        }
        if (live) *reinterpret_cast<uint4*>(orow + (long long)px * C) = pk;
    }
}

// final 1x1 conv + bias + sigmoid, bf16 activations in, fp32 probabilities out (one wave per row)
template <int CC>
__global__ __launch_bounds__(256) void mask_out_bf16_kernel(const uint16_t* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ p, long long M, int Cin)
{
    const int lane = threadIdx.x & 63;
    const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int cq = Cin / 4;
    for (long long r = wave0; r < M; r += nwaves) {
        float acc[CC];
#pragma unroll
        for (int k = 0; k < CC; ++k) acc[k] = 0.f;
        for (int q = lane; q < cq; q += 64) {
            const uint2 pk = *reinterpret_cast<const uint2*>(x + r * Cin + q * 4);
            const float xv[4] = {bf2f((uint16_t)(pk.x & 0xffff)), bf2f((uint16_t)(pk.x >> 16)), bf2f((uint16_t)(pk.y & 0xffff)),
                                 bf2f((uint16_t)(pk.y >> 16))};
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
            s += bias[lane];
            p[r * CC + lane] = 1.f / (1.f + expf(-s));
        }
    }
}


// weight packing: fp32 [K][N] (HWIO flattened) or [N][K] -> bf16 [N][K], with the frozen BatchNorm that follows
// the conv folded in:  w'[n,k] = w[k,n] * g[n],  b'[n] = b[n] * g[n] + beta[n] - mean[n] * g[n],  g = gamma / sqrt(var + eps)
__global__ __launch_bounds__(256) void pack_weights_bf16_kernel(const float* __restrict__ w, int K, int N, int w_is_nk,
                                                                const float* __restrict__ bias, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ mean,
                                                                const float* __restrict__ var, uint16_t* __restrict__ wt,
                                                                float* __restrict__ bias_out)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * K) return;
    const int n = (int)(i / K), k = (int)(i - (long long)n * K);
    float g = 1.f;
    if (gamma) g = gamma[n] / sqrtf(var[n] + BN_EPS_F);
    const float v = w_is_nk ? w[i] : w[(long long)k * N + n];
    wt[i] = f2bf(v * g);
    if (k == 0 && bias_out) {
        const float b = bias ? bias[n] : 0.f;
        bias_out[n] = gamma ? b * g + (beta[n] - mean[n] * g) : b;
    }
}

template <int AMODE, int EPI>
static void launch_bf16(const Bf16Args& a, hipStream_t s)
{
    const long long tiles = cdiv64(a.M, TBM) * ((a.N + TBN - 1) / TBN);
    if (tiles <= 0) return;
    const bool regstage = g_myolo_opt.bf16_regstage != 0;    // ablation: register-staged variant (myolo_set_option)
    if constexpr (EPI == EP_PLAIN) {
        const bool no256 = g_myolo_opt.bf16_no256 != 0, force256 = g_myolo_opt.bf16_force256 != 0;
        const long long tiles256 = cdiv64(a.M, T2M) * ((a.N + T2N - 1) / T2N);
        if (!no256 && !regstage && (a.N % T2N) == 0 && (tiles256 >= 1536 || force256)) {
            if (AMODE == AM_CONV3 && a.W + 1 <= (C3_AROWS - T2M) / 2 && !g_myolo_opt.bf16_no_c3)
                hipLaunchKernelGGL(conv3_bf16_256, dim3((unsigned)tiles256), dim3(512), 0, s, a);
            else
                hipLaunchKernelGGL((gemm_bf16_256<AMODE, EP_PLAIN>), dim3((unsigned)tiles256), dim3(512), 0, s, a);
            return;
        }
    }
    if (regstage) hipLaunchKernelGGL((gemm_bf16<AMODE, EPI>), dim3((unsigned)tiles), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_bf16_glds<AMODE, EPI>), dim3((unsigned)tiles), dim3(256), 0, s, a);
}

extern "C" {

int myolo_conv3x3_bf16_fwd(const uint16_t* x, const uint16_t* wt, const float* bias, uint16_t* y,
                           int N, int H, int W, int Cin, int Cout, int act, void* stream)
{
    MYOLO_REQUIRE(x && wt && y && N > 0 && H > 0 && W > 0, "conv3x3_bf16_fwd: bad arguments");
    MYOLO_REQUIRE(Cin % TBK == 0, "conv3x3_bf16_fwd: Cin must be a multiple of %d (got %d)", TBK, Cin);
    Bf16Args a = {};
    a.A = x; a.Wt = wt; a.C = y; a.bias = bias; a.M = (long long)N * H * W; a.N = Cout; a.K = 9 * Cin;
    a.H = H; a.W = W; a.Cc = Cin; a.act = act;
    launch_bf16<AM_CONV3, EP_PLAIN>(a, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_deconv2x2s2_bf16_fwd(const uint16_t* x, const uint16_t* wt, const float* bias, uint16_t* y,
                               int N, int H, int W, int Cin, int Cout, int act, void* stream)
{
    MYOLO_REQUIRE(x && wt && y && N > 0 && H > 0 && W > 0, "deconv2x2s2_bf16_fwd: bad arguments");
    MYOLO_REQUIRE(Cin % TBK == 0, "deconv2x2s2_bf16_fwd: Cin must be a multiple of %d (got %d)", TBK, Cin);
    Bf16Args a = {};
    a.A = x; a.Wt = wt; a.C = y; a.bias = bias; a.M = (long long)N * H * W; a.N = 4 * Cout; a.K = Cin;
    a.H = H; a.W = W; a.Co = Cout; a.act = act;
    launch_bf16<AM_PLAIN, EP_DECONV>(a, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_deconv2x2s2_mask_bf16_fwd(const uint16_t* x, const uint16_t* wt, const float* bias, const float* w2, const float* b2, float* p_out,
                                    int N, int H, int W, int Cin, int Cout, int ncls, void* ws, size_t ws_bytes, void* stream)
{
    MYOLO_REQUIRE(x && wt && bias && w2 && b2 && p_out && N > 0 && H > 0 && W > 0, "deconv2x2s2_mask_bf16_fwd: bad arguments");
    MYOLO_REQUIRE(Cin % TBK == 0 && Cout % TBN == 0 && ncls >= 1 && ncls <= 4,
                  "deconv2x2s2_mask_bf16_fwd: needs Cin %% %d == 0, Cout %% %d == 0, 1 <= classes <= 4 (got %d, %d, %d)", TBK, TBN, Cin, Cout, ncls);
    const long long M = (long long)N * H * W;
    const int nslabs = (Cout / TBN) * 2;
    MYOLO_NEED_WS((size_t)nslabs * 4 * M * ncls * sizeof(float));
    Bf16Args a = {};
    a.A = x; a.Wt = wt; a.bias = bias; a.M = M; a.N = 4 * Cout; a.K = Cin; a.H = H; a.W = W; a.Co = Cout; a.act = MYOLO_ACT_RELU;
    a.w2 = w2; a.part = (float*)ws; a.ncls = ncls;
    a.tune = g_myolo_opt.tune0;
    const long long tiles = cdiv64(M, TBM) * (a.N / TBN);
    const long long tiles256 = cdiv64(M, T2M) * (a.N / T2N);
    const bool no256 = g_myolo_opt.bf16_no256 != 0, force256 = g_myolo_opt.bf16_force256 != 0;
    const bool valu = g_myolo_opt.bf16_mask_valu != 0;      // 1 = the VALU epilogue on the fp32 deconv output (rounds 3-5; ablation / the tighter numerics)
    if (!no256 && (Cout % T2N) == 0 && Cout <= 512 && !g_myolo_opt.bf16_no_loopn && (tiles256 >= 1536 || force256)) {
        // one workgroup per 256 rows walks all four taps in one pipelined loop (bf16_no_loopn=1: a workgroup per (row tile, tap); same
        // Cout/64 column slabs in all three kernels)
        if (valu) hipLaunchKernelGGL((gemm_bf16_256<AM_PLAIN, EP_DECONV_MASK, true, false>), dim3((unsigned)cdiv64(M, T2M)), dim3(512), 0, (hipStream_t)stream, a);
        else if (Cout == T2N && !g_myolo_opt.bf16_mask_nofin) {      // all channels of a pixel in one workgroup: sigmoid stored by the kernel itself
            a.b2 = b2; a.out = p_out;
            hipLaunchKernelGGL((gemm_bf16_256<AM_PLAIN, EP_DECONV_MASK, true, true, true>), dim3((unsigned)cdiv64(M, T2M)), dim3(512), 0, (hipStream_t)stream, a);
            MYOLO_CHECK_LAUNCH();
            return MYOLO_OK;
        }
        else hipLaunchKernelGGL((gemm_bf16_256<AM_PLAIN, EP_DECONV_MASK, true, true>), dim3((unsigned)cdiv64(M, T2M)), dim3(512), 0, (hipStream_t)stream, a);
    } else if (!no256 && (Cout % T2N) == 0 && (tiles256 >= 1536 || force256)) {
        if (valu) hipLaunchKernelGGL((gemm_bf16_256<AM_PLAIN, EP_DECONV_MASK, false, false>), dim3((unsigned)tiles256), dim3(512), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL((gemm_bf16_256<AM_PLAIN, EP_DECONV_MASK, false, true>), dim3((unsigned)tiles256), dim3(512), 0, (hipStream_t)stream, a);
    }
    else
        hipLaunchKernelGGL((gemm_bf16_glds<AM_PLAIN, EP_DECONV_MASK>), dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, a);
    myolo_launch_deconv_mask_finish(a.part, b2, p_out, 4 * M, ncls, nslabs, (hipStream_t)stream);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_crop_and_resize_bf16_fwd(const float* image, const float* boxes, const int32_t* box_ind, uint16_t* out, int B, int H, int W,
                                   int C, int nb, int crop_h, int crop_w, void* stream)
{
    MYOLO_REQUIRE(image && boxes && box_ind && out && B > 0 && (C & 3) == 0 && nb >= 0 && nb <= 65535, "crop_and_resize_bf16_fwd: bad arguments");
    if (nb == 0) return MYOLO_OK;
    if ((C & 7) == 0 && (((uintptr_t)image | (uintptr_t)out) & 15) == 0 && !g_myolo_opt.crop_bf16_legacy)
        hipLaunchKernelGGL(crop_fwd_bf16_walk_kernel, dim3((unsigned)((crop_h * (C >> 3) + 255) / 256), nb), dim3(256), 0, (hipStream_t)stream, image, boxes,
                           box_ind, out, H, W, C, crop_h, crop_w);
    else
        hipLaunchKernelGGL(crop_fwd_bf16_kernel, dim3(crop_h, nb), dim3(256), 0, (hipStream_t)stream, image, boxes, box_ind, out, H, W, C,
                           crop_h, crop_w);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_mask_head_out_bf16_fwd(const uint16_t* x, const float* w, const float* bias, float* p, int64_t M, int Cin, int C, void* stream)
{
    MYOLO_REQUIRE(x && w && bias && p && M > 0 && (Cin & 3) == 0 && C >= 1 && C <= 8, "mask_head_out_bf16_fwd: bad arguments (1<=C<=8)");
    hipStream_t s = (hipStream_t)stream;
    long long blocks = (M + 3) / 4;
    if (blocks > 16384) blocks = 16384;
#define MO_CASE(K) case K: hipLaunchKernelGGL((mask_out_bf16_kernel<K>), dim3((unsigned)blocks), dim3(256), 0, s, x, w, bias, p, M, Cin); break;
    switch (C) { MO_CASE(1) MO_CASE(2) MO_CASE(3) MO_CASE(4) MO_CASE(5) MO_CASE(6) MO_CASE(7) MO_CASE(8) }
#undef MO_CASE
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

int myolo_pack_weights_bf16(const float* w, int K, int N, int w_is_nk, const float* bias, const float* gamma, const float* beta,
                            const float* mean, const float* var, uint16_t* wt, float* bias_out, void* stream)
{
    MYOLO_REQUIRE(w && wt && K > 0 && N > 0, "pack_weights_bf16: bad arguments");
    MYOLO_REQUIRE(!gamma || (beta && mean && var && bias_out), "pack_weights_bf16: BN folding needs beta/mean/var/bias_out");
    const long long total = (long long)N * K;
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, w, K, N, w_is_nk,
                       bias, gamma, beta, mean, var, wt, bias_out);
    MYOLO_CHECK_LAUNCH();
    return MYOLO_OK;
}

}  // extern "C"
