"""pw_smallm_kernel against the split-K pair of gemm_nn_fast launches it replaces (library switch pw_no_smallm), per entry point and shape:
us per call (HIP events around 200 back-to-back calls, alternating).   gpurun -- 'python tools/experiments/pw_smallm.py'"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mask-yolo_amd"))
import torch
from myolo import _ext as X

dev = "cuda:0"
shapes = [("train 14x14 pw7-12", 6272, 512, 512), ("infer 26x26 pw8-12", 2704, 512, 512), ("infer 13x13 pw13", 676, 512, 1024), ("infer 13x13 pw14", 676, 1024, 1024),
          ("train 7x7 pw13", 1568, 512, 1024), ("train 7x7 pw14", 1568, 1024, 1024)]
ws = torch.empty(256 << 20, dtype=torch.uint8, device=dev)
st = torch.cuda.current_stream().cuda_stream


def time_us(fn, n=200):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


for name, M, K, N in shapes:
    x, w, y = torch.randn(M, K, device=dev), torch.randn(K, N, device=dev) * 0.1, torch.empty(M, N, device=dev)
    sc, sh = torch.rand(N, device=dev) + 0.5, torch.randn(N, device=dev)
    isc, ish = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev)
    dy, dx = torch.randn(M, N, device=dev), torch.empty(M, K, device=dev)
    calls = {
        "affine_act_fwd": lambda: X.call("myolo_pwconv1x1_affine_act_fwd", X.ptr(x), X.ptr(w), X.ptr(sc), X.ptr(sh), 2, X.ptr(y), M, K, N, ws.data_ptr(), ws.numel(), st),
        "bwd_data": lambda: X.call("myolo_pwconv1x1_bwd_data", X.ptr(dy), X.ptr(w), X.ptr(dx), M, K, N, ws.data_ptr(), ws.numel(), st),
    }
    for cn, fn in calls.items():
        res = []
        for rep in range(2):
            for off in (0, 1):
                with X.option("pw_no_smallm", off):
                    res.append((off, time_us(fn)))
        new = [t for o, t in res if o == 0]
        old = [t for o, t in res if o == 1]
        print("%-22s M=%5d K=%4d N=%4d %-15s small-M %6.1f / %6.1f us   split-K pair %6.1f / %6.1f us" % (name, M, K, N, cn, new[0], new[1], old[0], old[1]), flush=True)
