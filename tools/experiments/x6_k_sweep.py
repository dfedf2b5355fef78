"""Fixed cost per output tile of wino_mm_x6_kernel: the plain product C[M][256] = A[M][K] B (myolo_matmul_f32, bf16x6 products, the same
kernel) at constant FLOPs for K = 256, 512, 1024, 2048 in steady state (30 untimed launches first).  time(K) = tiles * (fixed + K * slope):
the intercept is what a tile costs beyond its K loop (start-up, first fetch, epilogue stores, workgroup turnover).
   gpurun -- python tools/experiments/x6_k_sweep.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
import torch
from myolo import _ext as X
dev = "cuda:0"
X.set_option("wino_x6", 1)
st = torch.cuda.current_stream().cuda_stream
N = 256
res = []
for K in (256, 512, 1024, 2048):
    M = 4704 * 400 * 256 // K
    A = torch.randn(M, K, device=dev) * 0.1
    B = torch.randn(N, K, device=dev) * 0.05
    C = torch.empty(M, N, device=dev)
    ws = torch.empty(max(16, X.matmul_ws_bytes(K, N, 1, X.PRODUCTS_BF16X6)), dtype=torch.uint8, device=dev)
    fn = lambda: X.call("myolo_matmul_f32", X.ptr(A), X.ptr(B), X.ptr(C), M, K, N, 1, X.PRODUCTS_BF16X6, ws.data_ptr(), ws.numel(), st)
    for _ in range(30): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 30
    tiles = (M + 127) // 128
    res.append((K, tiles, ms))
    print("K %5d  M %8d  tiles %6d: %.3f ms  %.1f fp32-equivalent TFLOP/s  -> %.3f us per tile per CU-slot (512 slots)" %
          (K, M, tiles, ms, 2.0 * M * K * N / ms / 1e9, ms * 1e3 * 512 / tiles), flush=True)
(k0, t0, m0), (k1, t1, m1) = res[0], res[2]
a0, a1 = m0 * 1e3 * 512 / t0, m1 * 1e3 * 512 / t1        # us per tile-slot
slope = (a1 - a0) / (k1 - k0)
print("per tile: slope %.4f us per unit of K, intercept %.2f us (= %.0f units of K; %.0f %% of the K = 256 tile time)" %
      (slope, a0 - slope * k0, (a0 - slope * k0) / slope, 100 * (a0 - slope * k0) / a0))
