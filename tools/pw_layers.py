#!/usr/bin/env python
"""per-layer timing of the 14 pointwise (1x1) convs and 14 depthwise convs at config 2 (stand-alone launches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
import torch
from myolo import _ext as X
import bench

dev = "cuda:0"
for opt in os.environ.get("KBENCH_OPTIONS", "").split(","):
    if "=" in opt:
        X.set_option(opt.split("=")[0], int(opt.split("=")[1]))
ONLY = os.environ.get("PW_ONLY")
g = torch.Generator(device=dev).manual_seed(0)
ws = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
outs = [64, 64, 128, 256, 256, 512, 512, 512, 512, 512, 512, 512, 1024, 1024]
tot = [0.0, 0.0, 0.0]
for kind in (("fwd", "bwd_data", "bwd_weight") if not ONLY else (ONLY,)):
    tt = 0.0
    for (h, c, s), co in zip(bench.dw_layers(224, 1.0), outs):
        M = 32 * (h // s) * (h // s)
        x, w, y = torch.randn(M, c, device=dev, generator=g), torch.randn(c, co, device=dev, generator=g) * 0.05, torch.randn(M, co, device=dev, generator=g)
        dx, dw = torch.empty(M, c, device=dev), torch.empty(c, co, device=dev)
        st = X.stream()
        if kind == "fwd":
            fn = lambda: X.call("myolo_pwconv1x1_fwd", X.ptr(x), X.ptr(w), None, X.ptr(y), M, c, co, ws.data_ptr(), ws.numel(), st)
        elif kind == "bwd_data":
            fn = lambda: X.call("myolo_pwconv1x1_bwd_data", X.ptr(y), X.ptr(w), X.ptr(dx), M, c, co, ws.data_ptr(), ws.numel(), st)
        else:
            fn = lambda: X.call("myolo_pwconv1x1_bwd_weight", X.ptr(x), X.ptr(y), X.ptr(dw), M, c, co, ws.data_ptr(), ws.numel(), st)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        fl = 2.0 * M * c * co
        by = 4.0 * (M * c + M * co + c * co)
        tt += ms
        print("pw %-10s M=%7d K=%4d N=%4d: %7.1f us  %6.1f TF/s  %6.0f GB/s  (AI %.0f flop/B)" % (kind, M, c, co, ms * 1e3, fl / ms / 1e9, by / ms / 1e6, fl / by))
    print("pw %s total %.3f ms" % (kind, tt))
