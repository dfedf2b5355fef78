// Scalar against packed (f2) instantiations of the 1-D Winograd transforms, output by output, on the GPU: how the contraction ambiguity of B^T was found
// (profiles/r6_notes.md 1d).  The transforms below are a COPY of csrc/wino63_kernels.hip as they were BEFORE the fix (implicit contraction): with today's
// w63_bt (explicit fused multiply-adds, contraction off) every row prints 0 mismatches.   hipcc --offload-arch=gfx950 -O3 -o pkcmp pkcmp.hip && ./pkcmp

#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
// T = float, or f2 = two independent columns (rows) at once: the same expression trees element by element (v_pk_* on gfx950)
typedef float f2 __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ T w63_zero() { return (T)(0.f); }
template <int CLS, typename T = float>
__device__ __forceinline__ void w63_bt(const T d[8], T t[8])
{
    if (CLS == 6) {
        const T e0 = d[2] + d[6] - 4.25f * d[4], o0 = d[1] + d[5] - 4.25f * d[3];
        const T e1 = 0.25f * d[2] - 1.25f * d[4] + d[6], o1 = 0.5f * d[1] - 2.5f * d[3] + 2.f * d[5];
        const T e2 = 4.f * d[2] - 5.f * d[4] + d[6], o2 = 2.f * d[1] - 2.5f * d[3] + 0.5f * d[5];
        t[0] = (d[6] - d[0]) + 5.25f * (d[2] - d[4]);
        t[1] = e0 + o0;
        t[2] = e0 - o0;
        t[3] = e1 + o1;
        t[4] = e1 - o1;
        t[5] = e2 + o2;
        t[6] = e2 - o2;
        t[7] = (d[7] - d[1]) + 5.25f * (d[3] - d[5]);
    } else {            // rho .* (B6^T d), patch d[0..5]
        t[0] = -d[0] + 1.25f * d[2] - 0.25f * d[4];
        t[1] = 0.75f * (d[3] + d[4] - 4.f * (d[1] + d[2]));
        t[2] = 0.75f * (4.f * (d[1] - d[2]) - d[3] + d[4]);
        t[3] = 3.75f * (2.f * (d[3] - d[1]) - d[2] + d[4]);
        t[4] = 3.75f * (2.f * (d[1] - d[3]) - d[2] + d[4]);
        t[5] = w63_zero<T>();
        t[6] = w63_zero<T>();
        t[7] = 4.f * d[1] - 5.f * d[3] + d[5];
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

template <int F, int CLS, int ZMASK>
__global__ void k(const float* s, float* os, float* ov)
{
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    float d[8]; f2 dv[8];
    for (int i = 0; i < 8; ++i) {
        float a = (ZMASK >> i) & 1 ? 0.f : s[(i * 2) * 4096 + t], b = (ZMASK >> i) & 1 ? 0.f : s[(i * 2 + 1) * 4096 + t];
        d[i] = a; dv[i].x = a; dv[i].y = b;
    }
    float r[8]; f2 rv[8];
    for (int i = 0; i < 8; ++i) { r[i] = 0; rv[i] = w63_zero<f2>(); }
    if (F == 0) { w63_bt<CLS, float>(d, r); w63_bt<CLS, f2>(dv, rv); }
    if (F == 1) { w63_at<CLS, float>(d, r); w63_at<CLS, f2>(dv, rv); }
    if (F == 2) { w63_a<CLS, float>(d, r); w63_a<CLS, f2>(dv, rv); }
    for (int i = 0; i < 8; ++i) { os[i * 4096 + t] = r[i]; ov[i * 4096 + t] = rv[i].x; }
}
template <int F, int CLS, int Z> void run(const float* ds, float* dos, float* dov, const char* name)
{
    hipLaunchKernelGGL((k<F, CLS, Z>), dim3(16), dim3(256), 0, 0, ds, dos, dov);
    std::vector<float> a(8 * 4096), b(8 * 4096);
    hipMemcpy(a.data(), dos, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), dov, b.size() * 4, hipMemcpyDeviceToHost);
    int bad[8] = {0};
    for (int i = 0; i < 8; ++i) for (int t = 0; t < 4096; ++t) if (a[i * 4096 + t] != b[i * 4096 + t]) bad[i]++;
    printf("%-22s zmask %02x: mismatches per output", name, Z); for (int i = 0; i < 8; ++i) printf(" %d", bad[i]); printf("\n");
}
int main()
{
    std::vector<float> h(16 * 4096); srand(1); for (auto& x : h) x = (rand() / (float)RAND_MAX - 0.5f) * 20.f;
    float *ds, *dos, *dov; hipMalloc(&ds, h.size() * 4); hipMalloc(&dos, 8 * 4096 * 4); hipMalloc(&dov, 8 * 4096 * 4);
    hipMemcpy(ds, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 6, 0x00>(ds, dos, dov, "bt<6>"); run<0, 6, 0x01>(ds, dos, dov, "bt<6>"); run<0, 6, 0xC0>(ds, dos, dov, "bt<6>"); run<0, 6, 0xC1>(ds, dos, dov, "bt<6>");
    run<0, 4, 0xC0>(ds, dos, dov, "bt<4>"); run<0, 4, 0xE0>(ds, dos, dov, "bt<4>"); run<0, 4, 0xC1>(ds, dos, dov, "bt<4>");
    run<1, 6, 0x00>(ds, dos, dov, "at<6>"); run<1, 4, 0x60>(ds, dos, dov, "at<4>"); run<1, 6, 0x60>(ds, dos, dov, "at<6>");
    run<2, 6, 0xC0>(ds, dos, dov, "a<6>"); run<2, 4, 0xF0>(ds, dos, dov, "a<4>"); run<2, 6, 0xF0>(ds, dos, dov, "a<6>");
    return 0;
}
