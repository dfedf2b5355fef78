// Microbenchmark: does non-MFMA work hide behind fp32 MFMA on gfx950?   (answer, MI355X, this file's output in profiles/r2_notes.md: no)
//   /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o /tmp/ovl tools/mfma_valu_overlap.hip && /tmp/ovl
// One workgroup per CU, NW waves per SIMD.  Every wave runs a stream of independent v_mfma_f32_32x32x2_f32 (64 cycles each) with KV
// independent v_fma_f32 and KL independent ds_read_b32 behind each MFMA (sched_barrier pins the interleaving).  DEP=1: four
// consecutive MFMAs accumulate into the same tile; DEP=2: waves 0-3 issue only MFMAs, waves 4-7 only the VALU / LDS work.
// Prints ns per MFMA per SIMD: 29 ns alone; +6 ns for the first 4 VALU, ~+0.8 ns for each further one, the same with 1, 2 or 4
// waves per SIMD and with specialised waves -- the matrix pipe loses the cycles other instructions are issued in.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// one wave per SIMD (256 threads, 1 WG/CU via big LDS), loop: NT independent acc tiles round robin, K VALU fmas after each MFMA
template <int KV, int KL, int DEP, int NW>
__global__ __launch_bounds__(256 * NW, 1) void k(float* out, int iters)
{
    extern __shared__ float lds[];
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = threadIdx.x, b = 1.f;
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = threadIdx.x + i;
    float l = 0.f;
    const float* lp = lds + threadIdx.x;
    const bool mf = DEP != 2 || (threadIdx.x >> 8) == 0;      // DEP=2: waves 0-3 MFMA only, waves 4-7 VALU/LDS only
    if (DEP == 2 && !mf) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
#pragma unroll
                for (int i = 0; i < KV; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
#pragma unroll
                for (int i = 0; i < KL; ++i) l += lp[(g * KL + i) * 256];
                if (KL < 0 && (g & 3) == 0) {
#pragma unroll
                    for (int i = 0; i < -KL; ++i) { const float4 q = *reinterpret_cast<const float4*>(lds + ((g / 4 * -KL + i) * 256 + threadIdx.x) * 4); l += q.x + q.w; }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const int t = DEP == 1 ? (g >> 2) : (g & 3);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            if (DEP != 2) {
#pragma unroll
            for (int i = 0; i < KV; ++i) v[i] = fmaf(v[i], 1.0001f, 0.5f);
#pragma unroll
            for (int i = 0; i < KL; ++i) l += lp[(g * KL + i) * 256];
            if (KL < 0 && (g & 3) == 0) {
#pragma unroll
                for (int i = 0; i < -KL; ++i) { const float4 q = *reinterpret_cast<const float4*>(lds + ((g / 4 * -KL + i) * 256 + threadIdx.x) * 4); l += q.x + q.w; }
            }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = l;
    for (int i = 0; i < 12; ++i) s += v[i];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 256 * NW + threadIdx.x] = s;
}
template <int KV, int KL, int DEP, int NW>
void run(float* out)
{
    hipFuncSetAttribute((const void*)k<KV, KL, DEP, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, 100000);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    k<KV, KL, DEP, NW><<<256, 256 * NW, 100000>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KV, KL, DEP, NW><<<256, 256 * NW, 100000>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ns_per_mfma = ms * 1e6 / (iters * 16.0 * (DEP == 2 ? 1 : NW));
    printf("NW=%d KV=%2d KL=%d DEP=%d: %.3f ms, %.2f ns per MFMA (64 cyc @2.4GHz = 26.7 ns)\n", NW, KV, KL, DEP, ms, ns_per_mfma);
}
int main()
{
    float* out; hipMalloc(&out, 256 * 256 * 4 * 4);
    run<0, 0, 0, 1>(out); run<4, 0, 0, 1>(out); run<8, 0, 0, 1>(out);
    run<0, 0, 0, 2>(out); run<4, 0, 0, 2>(out); run<8, 0, 0, 2>(out); run<12, 0, 0, 2>(out); run<4, 1, 0, 2>(out);
    run<4, 0, 2, 2>(out); run<8, 0, 2, 2>(out); run<12, 0, 2, 2>(out); run<8, 1, 2, 2>(out);
    run<0, 0, 0, 4>(out); run<8, 0, 0, 4>(out);
    // LDS reads only: KL > 0 = ds_read_b32 per MFMA; KL < 0 = -KL ds_read_b128 per FOUR MFMAs (same bytes as KL b32 per MFMA)
    run<0, 1, 0, 2>(out); run<0, -1, 0, 2>(out); run<0, 2, 0, 2>(out); run<0, -2, 0, 2>(out); run<0, -3, 0, 2>(out); run<1, 0, 0, 2>(out); run<2, 0, 0, 2>(out);
    return 0;
}
