"""Error statistics of the bf16 deconv + mask kernel's matrix-pipe epilogue (round 6, profiles/r6_notes.md section 6) against the float64 oracle with and
without the bf16 roundings of the deconv output / the 1x1 kernel, for the all-taps kernel and the one-tap-per-workgroup kernel:
    gpurun -- 'python tools/experiments/dbg_mep.py'      (uses the helpers of tests/test_gpu_bf16.py)"""
import sys, numpy as np, torch
sys.path.insert(0, "tests"); sys.path.insert(0, "mask-yolo_amd"); sys.path.insert(0, ".")
import test_gpu_bf16 as T
from test_gpu_bf16 import *
N,H,W,Cin,Cout,C = 3,14,14,256,256,2
rng = np.random.default_rng(3)
x = bf16_round(rnd(rng, N, H, W, Cin)); w = bf16_round(rnd(rng, 2, 2, Cout, Cin, scale=0.05))
b, w2, b2 = rnd(rng, Cout), rnd(rng, Cout, C, scale=0.1), rnd(rng, C)
M = N*H*W
wsb = torch.empty((Cout // 128) * 2 * 4 * M * C * 4, dtype=torch.uint8, device=DEV)
d = O.relu(O.deconv2x2s2(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64)))
ref = 1 / (1 + np.exp(-(d.reshape(-1, Cout) @ w2 + b2))).reshape(N, 2 * H, 2 * W, C)
ref16 = 1 / (1 + np.exp(-(bf16_round(d.astype(np.float32)).astype(np.float64).reshape(-1, Cout) @ bf16_round(w2).astype(np.float64) + b2))).reshape(N, 2 * H, 2 * W, C)
for loopn in (0,1):
    with X.option("bf16_force256", 1), X.option("bf16_no_loopn", 1 - loopn):
        p2 = torch.full((N, 2 * H, 2 * W, C), float("nan"), device=DEV)
        X.call("myolo_deconv2x2s2_mask_bf16_fwd", X.ptr(to_bf16_dev(x)), X.ptr(to_bf16_dev(w.reshape(4 * Cout, Cin))), X.ptr(dt(b)),
               X.ptr(dt(w2)), X.ptr(dt(b2)), X.ptr(p2), N, H, W, Cin, Cout, C, wsb.data_ptr(), wsb.numel(), X.stream())
        torch.cuda.synchronize()
    e = np.abs(p2.cpu().numpy() - ref16); e0 = np.abs(p2.cpu().numpy() - ref)
    print("loopn", loopn, "vs ref16 max %.3g q50 %.3g q90 %.3g q99 %.3g | vs ref max %.3g" % (e.max(), np.quantile(e,.5), np.quantile(e,.9), np.quantile(e,.99), e0.max()))
