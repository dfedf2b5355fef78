"""Timing-only ablations of the 128 x 128-tile loop of wino_mm_x6_kernel<PLAIN, PW, PLAIN, 2> (the 14 x 14 pointwise layers of the trunk, one workgroup per CU):
    tools/experiments/mk_variant.sh x6tune wino_mm.hip -DMM_X6_TUNE
    gpurun -- 'MYOLO_LIB=tools/_ab/lib_x6tune.so python tools/experiments/pw_x6_tune.py'
tune0 bits read by that loop (results are wrong, durations are what is measured):
   512 no MFMAs          1024 no B fragment loads in the loop       2048 no A loads in the loop       8192 no BatchNorm-coefficient loads in the loop
  4096 no LDS stores and no barrier       16384 no LDS fragment reads       8 no stores of the result       1 << 20 (always set here) no split launch: the multiply alone"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
import torch
from myolo import _ext as X
dev = "cuda:0"
shapes = [(6272, 512, 512)] if len(sys.argv) < 4 else [tuple(int(v) for v in sys.argv[1:4])]
st = X.stream()
X.set_option("wino_x6", 1)                 # FP32_MATMUL = "bf16x6", as the engine sets it
for Mr, Cc, Co in shapes:
    nbuf = max(2, int(640e6 // (Mr * Cc * 4)) + 1)
    xs = [torch.randn(Mr, Cc, device=dev) for _ in range(nbuf)]
    sc, sh = torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1
    w, y = torch.randn(Cc, Co, device=dev) * 0.05, torch.empty(Mr, Co, device=dev)
    gam, bet = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
    outs = [torch.empty(Co, device=dev) for _ in range(6)]
    wsb = X.pw_bnstats_ws_bytes(Mr, Cc, Co)
    wsd = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    it = [0]

    def f():
        i = it[0]; it[0] += 1
        X.call("myolo_pwconv1x1_bnstats_fwd", X.ptr(xs[i % nbuf]), X.ptr(sc), X.ptr(sh), 2, X.ptr(w), X.ptr(y), X.ptr(gam), X.ptr(bet),
               *[X.ptr(o) for o in outs], Mr, Cc, Co, 1, wsd.data_ptr(), wsb, st)

    def timeit(n=40):
        for _ in range(10):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            f()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    masks = [0, 512, 1024, 2048, 8192, 2048 + 8192, 4096, 16384, 4096 + 16384, 8, 1024 + 2048 + 8192, 1024 + 2048 + 8192 + 4096 + 16384,
             512 + 1024 + 2048 + 8192 + 4096 + 16384, 512 + 1024 + 2048 + 8192 + 4096 + 16384 + 8, 0]
    for t in masks:
        f(); torch.cuda.synchronize()                      # (the split filters for the runs below, which skip the split launch: bit 20)
        with X.option("tune0", t + (1 << 20)):
            us = timeit()
        print("M=%d %d->%d tune0=%6d: %.1f us" % (Mr, Cc, Co, t, us))
