// Residency census: how many workgroups of T threads / L bytes of dynamic LDS / V VGPRs does a CU of this part hold at once?
//   hipcc --offload-arch=gfx950 -O2 -o census census.hip && ./census
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
#include <algorithm>
struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };
template <int V>
__global__ void k(Rec* r, int spin)
{
    extern __shared__ float lds[];
    if (V == 96) asm volatile("v_mov_b32 v95, 0" ::: "v95");
    if (V == 88) asm volatile("v_mov_b32 v87, 0" ::: "v87");
    if (V == 72) asm volatile("v_mov_b32 v71, 0" ::: "v71");
    if (V == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (V == 104) asm volatile("v_mov_b32 v103, 0" ::: "v103");
    if (V == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    const unsigned long long t0 = wall_clock64();
    lds[threadIdx.x] = (float)t0;
    __syncthreads();
    while (wall_clock64() - t0 < (unsigned long long)spin) { __builtin_amdgcn_s_sleep(10); }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        r[blockIdx.x] = Rec{t0, (unsigned long long)wall_clock64(), hw, xcc & 0xf};
    }
}
template <int V>
static void run(int threads, int ldsb, int grid)
{
    Rec* d; hipMalloc(&d, grid * sizeof(Rec));
    hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    int api = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&api, k<V>, threads, ldsb);
    hipLaunchKernelGGL(k<V>, dim3(grid), dim3(threads), ldsb, 0, d, 2000);   // 20 us at 100 MHz
    hipDeviceSynchronize();
    std::vector<Rec> h(grid); hipMemcpy(h.data(), d, grid * sizeof(Rec), hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<std::pair<unsigned long long, int>>> ev;
    for (auto& r : h) { unsigned id = (r.xcc << 16) | ((r.hw >> 8) & 0xff); ev[id].push_back({r.t0, 1}); ev[id].push_back({r.t1, -1}); }
    int mx = 0; double avg = 0;
    for (auto& e : ev) { std::sort(e.second.begin(), e.second.end()); int c = 0, m = 0; for (auto& p : e.second) { c += p.second; m = std::max(m, c); } mx = std::max(mx, m); avg += m; }
    printf("threads %4d vgpr %3d lds %6d: API %d, measured max resident per CU %d (mean of per-CU max %.2f over %zu CUs)\n", threads, V, ldsb, api, mx, avg / ev.size(), ev.size());
    hipFree(d);
}
int main()
{
    const int grid = 256 * 12;
    for (int ldsb : {1024, 32768, 50176, 54784, 65536}) {
        run<64>(576, ldsb, grid); run<72>(576, ldsb, grid); run<88>(576, ldsb, grid); run<96>(576, ldsb, grid); run<104>(576, ldsb, grid); run<128>(576, ldsb, grid);
    }
    for (int th : {256, 320, 384, 448, 512, 640, 768, 1024}) { run<88>(th, 1024, grid); run<64>(th, 1024, grid); }
    return 0;
}
