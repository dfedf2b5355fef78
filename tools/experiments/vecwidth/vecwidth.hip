// Does the bandwidth a CU can pull depend on BYTES PER LANE of its memory instructions?  Persistent workgroups (one per CU, nine waves -- the shape of
// wino63_boundary_kernel) stream rows of `planes` strided regions: every wave instruction moves 64 lanes x VEC x 4 bytes of ONE row.
//   hipcc --offload-arch=gfx950 -O3 -o vecwidth vecwidth.hip && ./vecwidth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float vf1 __attribute__((ext_vector_type(1)));
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));
template <int VEC> struct V;
template <> struct V<1> { typedef float T; };
template <> struct V<2> { typedef vf2 T; };
template <> struct V<4> { typedef vf4 T; };

// MODE 0: read only (sum), 1: write only, 2: read + write (copy)
template <int VEC, int MODE, int INFLIGHT>
__global__ __launch_bounds__(576) void k(const float* __restrict__ src, float* __restrict__ dst, long long plane_elems, int planes, long long rows_per_plane, float* sink)
{
    typedef typename V<VEC>::T T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long nw = (long long)gridDim.x * 9, w = (long long)blockIdx.x * 9 + wave;
    const int row_elems = 64 * VEC;                         // one wave instruction = one row
    T acc[INFLIGHT];
    for (int i = 0; i < INFLIGHT; ++i) acc[i] = (T)(0.f);
    float tot = 0.f;
    for (long long r = w; r < rows_per_plane; r += nw) {    // row r of every plane (64 planes: the 64 Winograd points)
        for (int p0 = 0; p0 < planes; p0 += INFLIGHT) {
#pragma unroll
            for (int i = 0; i < INFLIGHT; ++i) {
                const long long off = (long long)(p0 + i) * plane_elems + r * row_elems + lane * VEC;
                if (MODE != 1) acc[i] = *reinterpret_cast<const T*>(src + off);
            }
#pragma unroll
            for (int i = 0; i < INFLIGHT; ++i) {
                const long long off = (long long)(p0 + i) * plane_elems + r * row_elems + lane * VEC;
                if (MODE == 0) { if constexpr (VEC == 1) tot += acc[i]; else tot += acc[i][0] + acc[i][VEC - 1]; }
                else if (MODE == 1) { T v = (T)((float)lane); *reinterpret_cast<T*>(dst + off) = v; }
                else *reinterpret_cast<T*>(dst + off) = acc[i];
            }
        }
    }
    if (MODE == 0 && tot == 12345.678f) sink[0] = tot;
}
template <int VEC, int MODE, int INFLIGHT>
static void run(const float* src, float* dst, long long plane_elems, int planes, float* sink, int wgs, const char* name)
{
    const long long rows = plane_elems / (64 * VEC);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<VEC, MODE, INFLIGHT>), dim3(wgs), dim3(576), 0, 0, src, dst, plane_elems, planes, rows, sink);
    hipEventRecord(e0);
    const int it = 10;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((k<VEC, MODE, INFLIGHT>), dim3(wgs), dim3(576), 0, 0, src, dst, plane_elems, planes, rows, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
    const double bytes = (double)plane_elems * planes * 4 * (MODE == 2 ? 2 : 1);
    printf("%-10s %d B/lane, %2d in flight, %4d workgroups: %.3f ms  %.0f GB/s\n", name, VEC * 4, INFLIGHT, wgs, ms, bytes / ms / 1e6);
}
int main()
{
    const int planes = 64;
    const long long plane_elems = 4704LL * 400 / 64 * 256;           // 1.93 GB per set, as the boundary kernels' plane sets
    float *src, *dst, *sink;
    hipMalloc(&src, plane_elems * planes * 4); hipMalloc(&dst, plane_elems * planes * 4); hipMalloc(&sink, 4);
    hipMemset(src, 0, plane_elems * planes * 4); hipMemset(dst, 0, plane_elems * planes * 4);
    for (int wgs : {256, 512}) {
        run<1, 0, 16>(src, dst, plane_elems, planes, sink, wgs, "read");  run<2, 0, 16>(src, dst, plane_elems, planes, sink, wgs, "read");  run<4, 0, 16>(src, dst, plane_elems, planes, sink, wgs, "read");
        run<1, 0, 64>(src, dst, plane_elems, planes, sink, wgs, "read");  run<4, 0, 32>(src, dst, plane_elems, planes, sink, wgs, "read");
        run<1, 1, 16>(src, dst, plane_elems, planes, sink, wgs, "write"); run<2, 1, 16>(src, dst, plane_elems, planes, sink, wgs, "write"); run<4, 1, 16>(src, dst, plane_elems, planes, sink, wgs, "write");
        run<1, 2, 16>(src, dst, plane_elems, planes, sink, wgs, "copy");  run<2, 2, 16>(src, dst, plane_elems, planes, sink, wgs, "copy");  run<4, 2, 16>(src, dst, plane_elems, planes, sink, wgs, "copy");
        run<1, 2, 64>(src, dst, plane_elems, planes, sink, wgs, "copy");  run<4, 2, 32>(src, dst, plane_elems, planes, sink, wgs, "copy");
    }
    return 0;
}
