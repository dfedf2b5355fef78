#!/usr/bin/env python
"""Micro-benchmark of single C-ABI kernels at the BASELINE config-2 shapes (used for tuning and for the
rocprofv3 --pmc passes whose summaries are committed under profiles/).
  python tools/kbench.py conv3x3_fwd|conv3x3_bwd_weight|conv3x3_bwd_data|deconv_fwd|roialign_fwd|roialign_bwd|dw [--iters N]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
import torch         # noqa: E402
from myolo import _ext as X   # noqa: E402


WARM = 2


def timeit(fn, iters, warm=None):
    for _ in range(WARM if warm is None else warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("which")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--warm", type=int, default=2, help="untimed launches first; an MFMA-bound kernel needs ~30 to get through the clock transient after idle")
    ap.add_argument("--rois", type=int, default=32 * 147)
    ap.add_argument("--cout", type=int, default=256)
    a = ap.parse_args()
    global WARM
    WARM = a.warm
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(0)
    rn = lambda *s: torch.randn(*s, device=dev, generator=g)   # noqa: E731
    ws = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
    NR, ps, C = a.rois, 14, 256
    M = NR * ps * ps
    st = X.stream()
    for opt in os.environ.get("KBENCH_OPTIONS", "").split(","):      # e.g. KBENCH_OPTIONS=gemm_no_glds=1,wino_no_mixed=1
        if "=" in opt:
            X.set_option(opt.split("=")[0], int(opt.split("=")[1]))
    if a.which in ("conv3x3_bf16_fwd", "deconv_bf16_fwd"):
        bf = torch.bfloat16
        x, b = rn(M, C).to(bf), rn(C)
        if a.which == "conv3x3_bf16_fwd":
            wt, y = (rn(C, 9 * C) * 0.02).to(bf), torch.empty(M, C, dtype=bf, device=dev)
            flop = 2.0 * M * 9 * C * C
            fn = lambda: X.call("myolo_conv3x3_bf16_fwd", X.ptr(x), X.ptr(wt), X.ptr(b), X.ptr(y), NR, ps, ps, C, C, 1, st)   # noqa: E731
        else:
            wt, y = (rn(4 * C, C) * 0.02).to(bf), torch.empty(4 * M, C, dtype=bf, device=dev)
            flop = 2.0 * M * C * 4 * C
            fn = lambda: X.call("myolo_deconv2x2s2_bf16_fwd", X.ptr(x), X.ptr(wt), X.ptr(b), X.ptr(y), NR, ps, ps, C, C, 1, st)   # noqa: E731
        ms = timeit(fn, a.iters)
        print("%s M=%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 2500 bf16 dense)" % (a.which, M, ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0))
    elif a.which == "deconv_mask_bf16_fwd":
        # bf16 deconv 2x2/s2 + ReLU + 1x1 mask conv + sigmoid in one GEMM launch (+ the slab-sum finish), ncls classes
        bf = torch.bfloat16
        ncls = 2                                           # RiceConfig: background + rice
        x, b = rn(M, C).to(bf), rn(C)
        wt = (rn(4 * C, C) * 0.02).to(bf)
        w2, b2, pp = rn(C, ncls) * 0.1, rn(ncls), torch.empty(4 * M, ncls, device=dev)
        wsw = torch.empty((C // 128) * 2 * 4 * M * ncls * 4, dtype=torch.uint8, device=dev)
        fn = lambda: X.call("myolo_deconv2x2s2_mask_bf16_fwd", X.ptr(x), X.ptr(wt), X.ptr(b), X.ptr(w2), X.ptr(b2), X.ptr(pp), NR, ps, ps, C, C, ncls, wsw.data_ptr(), wsw.numel(), st)   # noqa: E731
        ms = timeit(fn, a.iters)
        flop = 2.0 * M * C * 4 * C
        print("deconv_mask_bf16_fwd (ncls=%d) M=%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 2500 bf16 dense)" % (ncls, M, ms, flop / ms / 1e9, flop / ms / 1e9 / 25.0))
    elif a.which == "wino63_wgrad":
        # conv1's weight-gradient product dU = V^T Q over the 64 planes + the dU -> dw transform (myolo_wino63_bwd_weight_from_q)
        pe = X.wino63_plane_elems(NR, C)
        V, Q, dw = rn(pe), rn(pe), torch.empty(3, 3, C, C, device=dev)
        wsw = torch.empty(X.wino63_bwd_weight_from_q_ws_bytes(NR, C, C), dtype=torch.uint8, device=dev)
        fn = lambda: X.call("myolo_wino63_bwd_weight_from_q", X.ptr(V), X.ptr(Q), X.ptr(dw), NR, C, C, wsw.data_ptr(), wsw.numel(), st)   # noqa: E731
        ms = timeit(fn, a.iters)
        flop = 2.0 * 400 * NR * C * C
        print("wino63_wgrad NR=%d: %.3f ms  %.1f fp32-equivalent TFLOP/s; operand bytes %.2f GB -> %.0f GB/s" % (NR, ms, flop / ms / 1e9, 2 * pe * 4 / 1e9, 2 * pe * 4 / ms / 1e6))
    elif a.which == "wino63_boundary":
        # the layer boundary M_i -> (output transform, bias, affine, ReLU) -> LDS -> input transform -> V_{i+1}: reads and writes one plane set each
        pe = X.wino63_plane_elems(NR, C)
        Mp, Vn, b, sc, sh = rn(pe), torch.empty(pe, device=dev), rn(C), rn(C), rn(C)
        fn = lambda: X.call("myolo_wino63_output_input_transform", X.ptr(Mp), X.ptr(b), X.ptr(sc), X.ptr(sh), None, None, X.ptr(Vn), NR, C, 1, st)   # noqa: E731
        ms = timeit(fn, a.iters)
        print("wino63_boundary<M,V> NR=%d: %.3f ms  %.0f GB/s (%.2f GB read + written)" % (NR, ms, 2 * pe * 4 / ms / 1e6, 2 * pe * 4 / 1e9))
        y = torch.empty(M, C, device=dev)
        fn = lambda: X.call("myolo_wino63_output_transform", X.ptr(Mp), X.ptr(b), X.ptr(sc), X.ptr(sh), X.ptr(y), NR, C, 1, st)   # noqa: E731
        ms = timeit(fn, a.iters)
        print("wino63_boundary<M,none> NR=%d: %.3f ms  %.0f GB/s" % (NR, ms, (pe + M * C) * 4 / ms / 1e6))
        fn = lambda: X.call("myolo_wino63_input_transform", X.ptr(y), X.ptr(sc), X.ptr(sh), 1, None, None, X.ptr(Vn), NR, C, st)   # noqa: E731
        ms = timeit(fn, a.iters)
        print("wino63_boundary<act,V> NR=%d: %.3f ms  %.0f GB/s" % (NR, ms, (pe + M * C) * 4 / ms / 1e6))
    elif a.which == "wino63_lazy":
        # conv1's backward transforms: bn1's lazily formed input gradient (y_pre + the positive ROIs' compact dy) -> LDS tile -> V (data gradient operand) and
        # Q (weight gradient operand) in ONE pass: reads 0.94 GB, writes 2 x 1.93 GB
        pe = X.wino63_plane_elems(NR, C)
        y = rn(M, C)
        npos = 8
        dyc = rn(npos * ps * ps, C)
        inv = torch.full((NR,), -1, dtype=torch.int32, device=dev)
        inv[:npos] = torch.arange(npos, dtype=torch.int32, device=dev)
        sc, sh, ka, kb = rn(C), rn(C), rn(C) * 1e-3, rn(C) * 1e-3
        V, Q = torch.empty(pe, device=dev), torch.empty(pe, device=dev)
        fn = lambda: X.call("myolo_wino63_lazybn_transforms", X.ptr(y), X.ptr(dyc), X.ptr(inv), X.ptr(sc), X.ptr(sh), X.ptr(ka), X.ptr(kb), 1, X.ptr(V), X.ptr(Q), NR, C, st)   # noqa: E731
        ms = timeit(fn, a.iters)
        by = (M * C + 2 * pe) * 4.0
        print("wino63_boundary<lazy,VQ> NR=%d: %.3f ms  %.0f GB/s (%.2f GB read + written)" % (NR, ms, by / ms / 1e6, by / 1e9))
    elif a.which == "wino63_mm":
        pe = X.wino63_plane_elems(NR, C)
        V, Mp, w = rn(pe), torch.empty(pe, device=dev), rn(3, 3, C, C) * 0.02
        # KBENCH_DATA: what the operands hold -- the package power (hence the clock this kernel is granted) depends on how many bits toggle
        mode = os.environ.get("KBENCH_DATA", "randn")
        if mode == "zeros":
            V.zero_(); w.zero_()
        elif mode == "ones":
            V.fill_(1.0); w.fill_(1.0)
        elif mode == "relu":                      # half the activations exactly zero, as behind a ReLU
            V.clamp_(min=0.0)
        U = torch.empty(X.wino63_u_elems(C, C), device=dev)
        X.call("myolo_wino63_weight_transform", X.ptr(w), X.ptr(U), C, C, st)
        fn = lambda: X.call("myolo_wino63_multiply", X.ptr(V), X.ptr(U), X.ptr(Mp), NR, C, C, st)   # noqa: E731
        ms = timeit(fn, a.iters)
        flop = 2.0 * 400 * NR * C * C
        print("wino63_mm NR=%d: %.3f ms  %.1f fp32-equivalent TFLOP/s" % (NR, ms, flop / ms / 1e9))
    elif a.which == "copy":
        # HBM stream copy: the hand-written float4 kernel (myolo_stream_copy) over grid sizes, beside torch's copy_
        nbytes = 2 << 30
        src, dst = torch.empty(nbytes, dtype=torch.uint8, device=dev), torch.empty(nbytes, dtype=torch.uint8, device=dev)
        src.zero_(); dst.zero_()
        for variant, name, factor in ((0, "float4", 2), (1, "nt store", 2), (2, "nt load+store", 2), (3, "read only", 1), (4, "write only", 1)):
            for blocks in (256, 512, 1024, 2048, 4096, 8192, 16384):
                fn = lambda: X.call("myolo_stream_copy", src.data_ptr(), dst.data_ptr(), nbytes, variant, blocks, st)   # noqa: E731
                ms = timeit(fn, a.iters)
                print("copy %-14s blocks %5d: %.3f ms  %.0f GB/s" % (name, blocks, ms, factor * nbytes / ms / 1e6))
        ms = timeit(lambda: dst.copy_(src), a.iters)
        print("torch copy_: %.3f ms  %.0f GB/s" % (ms, 2 * nbytes / ms / 1e6))
    elif a.which == "mfma":
        # matrix-pipe ceiling: myolo_mfma_probe (register operands, eight independent accumulator blocks per wave) over waves per SIMD
        for kind, name, flop in ((0, "bf16 32x32x16", 32768.0), (1, "f32 32x32x2", 4096.0), (2, "bf16 2 chains", 32768.0), (3, "bf16 1 chain", 32768.0)):
            for blocks in ((256, 512, 1024) if kind < 2 else (512,)):
                it = 10000 if kind == 1 else 20000
                out = torch.zeros(blocks * 256, device=dev)
                fn = lambda: X.call("myolo_mfma_probe", kind, it, blocks, out.data_ptr(), st)   # noqa: E731
                ms = timeit(fn, 3)
                print("mfma %-14s blocks %5d: %.3f ms  %.1f TFLOP/s" % (name, blocks, ms, blocks * 4.0 * it * 8 * flop / ms / 1e9))
    elif a.which == "wino63_fwd":
        # the F(6,3)/F(4,3) tiling (csrc/wino63_kernels.hip): input transform -> one-launch multiply -> output transform
        x, w, b, y = rn(M, C), rn(3, 3, C, C) * 0.02, rn(C), torch.empty(M, C, device=dev)
        V, Mp = torch.empty(X.wino63_plane_elems(NR, C), device=dev), torch.empty(X.wino63_plane_elems(NR, C), device=dev)
        U = torch.empty(X.wino63_u_elems(C, C), device=dev)

        def fn():
            X.call("myolo_wino63_weight_transform", X.ptr(w), X.ptr(U), C, C, st)
            X.call("myolo_wino63_input_transform", X.ptr(x), None, None, 0, None, None, X.ptr(V), NR, C, st)
            X.call("myolo_wino63_multiply", X.ptr(V), X.ptr(U), X.ptr(Mp), NR, C, C, st)
            X.call("myolo_wino63_output_transform", X.ptr(Mp), X.ptr(b), None, None, X.ptr(y), NR, C, 0, st)
        ms = timeit(fn, a.iters)
        flop = 2.0 * M * 9 * C * C
        print("wino63_fwd M=%d: %.3f ms  %.1f direct-equivalent TFLOP/s (direct-conv FLOPs / time)" % (M, ms, flop / ms / 1e9))
    elif a.which.startswith("wino"):
        x, w, b, y = rn(M, C), rn(3, 3, C, C) * 0.02, rn(C), torch.empty(M, C, device=dev)
        T = NR * 16
        vk = torch.empty(36, T, C, device=dev)
        wsz = max(X.wino_ws_bytes(NR, ps, ps, C, C, k) for k in (0, 1, 2))
        wsw = torch.empty(wsz, dtype=torch.uint8, device=dev)
        flop = 2.0 * M * 9 * C * C
        if a.which == "wino_fwd":
            fn = lambda: X.call("myolo_conv3x3_wino_fwd", X.ptr(x), X.ptr(w), X.ptr(b), None, None, X.ptr(y), NR, ps, ps, C, C, 0, X.ptr(vk), wsw.data_ptr(), wsw.numel(), st)   # noqa: E731
        elif a.which == "wino_bwd_data":
            fn = lambda: X.call("myolo_conv3x3_wino_bwd_data", X.ptr(x), X.ptr(w), X.ptr(y), NR, ps, ps, C, C, wsw.data_ptr(), wsw.numel(), st)   # noqa: E731
        else:
            dw = torch.empty(3, 3, C, C, device=dev)
            X.call("myolo_conv3x3_wino_fwd", X.ptr(x), X.ptr(w), X.ptr(b), None, None, X.ptr(y), NR, ps, ps, C, C, 0, X.ptr(vk), wsw.data_ptr(), wsw.numel(), st)
            fn = lambda: X.call("myolo_conv3x3_wino_bwd_weight", None, X.ptr(vk), X.ptr(y), X.ptr(dw), NR, ps, ps, C, C, wsw.data_ptr(), wsw.numel(), st)   # noqa: E731
        ms = timeit(fn, a.iters)
        print("%s M=%d: %.3f ms  %.1f direct-equivalent TFLOP/s (direct-conv FLOPs / time)" % (a.which, M, ms, flop / ms / 1e9))
    elif a.which.startswith("conv3x3"):
        Co = a.cout
        x, w, b, y = rn(M, C), rn(3, 3, C, Co) * 0.02, rn(Co), torch.empty(M, Co, device=dev)
        flop = 2.0 * M * 9 * C * Co
        if a.which == "conv3x3_fwd":
            fn = lambda: X.call("myolo_conv3x3_fwd", X.ptr(x), X.ptr(w), X.ptr(b), X.ptr(y), NR, ps, ps, C, Co, ws.data_ptr(), ws.numel(), st)   # noqa: E731
        elif a.which == "conv3x3_bwd_data":
            fn = lambda: X.call("myolo_conv3x3_bwd_data", X.ptr(x), X.ptr(w), X.ptr(y), NR, ps, ps, C, C, ws.data_ptr(), ws.numel(), st)   # noqa: E731
        else:
            dw = torch.empty(3, 3, C, C, device=dev)
            fn = lambda: X.call("myolo_conv3x3_bwd_weight", X.ptr(x), X.ptr(y.copy_(x)), X.ptr(dw), NR, ps, ps, C, C, ws.data_ptr(), ws.numel(), st)   # noqa: E731
            y.copy_(x)
            fn = lambda: X.call("myolo_conv3x3_bwd_weight", X.ptr(x), X.ptr(y), X.ptr(dw), NR, ps, ps, C, C, ws.data_ptr(), ws.numel(), st)   # noqa: E731
        ms = timeit(fn, a.iters)
        print("%s M=%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)" % (a.which, M, ms, flop / ms / 1e9, flop / ms / 1e9 / 1.573))
    elif a.which == "deconv_mask_fwd":
        x, w, b = rn(M, C), rn(2, 2, C, C) * 0.02, rn(C)
        w2, b2, pp = rn(C, 4) * 0.1, rn(4), torch.empty(4 * M, 4, device=dev)
        wsw = torch.empty(X.deconv_mask_ws_bytes(NR, ps, ps, C, C, 4), dtype=torch.uint8, device=dev)
        fn = lambda: X.call("myolo_deconv2x2s2_mask_fwd", X.ptr(x), X.ptr(w), X.ptr(b), X.ptr(w2), X.ptr(b2), X.ptr(pp), NR, ps, ps, C, C, 4, wsw.data_ptr(), wsw.numel(), st)   # noqa: E731
        ms = timeit(fn, a.iters)
        flop = 2.0 * M * C * 4 * C
        print("deconv_mask_fwd (fused deconv+relu+1x1+sigmoid) M=%d: %.3f ms  %.1f TFLOP/s" % (M, ms, flop / ms / 1e9))
    elif a.which == "deconv_fwd":
        x, w, b, y = rn(M, C), rn(2, 2, C, C) * 0.02, rn(C), torch.empty(4 * M, C, device=dev)
        fn = lambda: X.call("myolo_deconv2x2s2_fwd", X.ptr(x), X.ptr(w), X.ptr(b), X.ptr(y), NR, ps, ps, C, C, 1, ws.data_ptr(), ws.numel(), st)   # noqa: E731
        ms = timeit(fn, a.iters)
        flop = 2.0 * M * C * 4 * C
        print("deconv_fwd M=%d: %.3f ms  %.1f TFLOP/s" % (M, ms, flop / ms / 1e9))
    elif a.which == "roialign_bf16_fwd":
        # the inference ROIAlign at the Rice-416 shape: batch 4, 52 x 52 x 256 feature map, --rois boxes (3380 = 4 x 845) of the five Rice anchors' sizes, bf16 out
        B, H = 4, 52
        NRb = a.rois if a.rois != 32 * 147 else 3380
        feat = rn(B, H, H, C)
        anch = torch.tensor([2.09, 2.48, 2.59, 3.01, 3.60, 3.64, 5.25, 4.56, 6.21, 6.25], device=dev).view(5, 2) / 13.0
        wh = anch[torch.arange(NRb, device=dev) % 5] * (0.8 + 0.4 * torch.rand(NRb, 2, device=dev, generator=g))
        ctr = torch.rand(NRb, 2, device=dev, generator=g)
        boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], 1).contiguous()
        bind = (torch.arange(NRb, device=dev, dtype=torch.int32) % B).contiguous()
        out = torch.empty(NRb * ps * ps, C, dtype=torch.bfloat16, device=dev)
        nbytes = NRb * ps * ps * C * 2 + B * H * H * C * 4
        fn = lambda: X.call("myolo_crop_and_resize_bf16_fwd", X.ptr(feat), X.ptr(boxes), X.ptr(bind), X.ptr(out), B, H, H, C, NRb, ps, ps, st)   # noqa: E731
        ms = timeit(fn, a.iters)
        print("roialign_bf16_fwd (%d boxes): %.4f ms  %.0f GB/s (%.1f%% of 8000) on %.1f MB" % (NRb, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80, nbytes / 1e6))
    elif a.which == "pw_smallm":
        # a pointwise layer of the inference YOLO head (26 x 26 x 4 rows, 512 -> 512) with the frozen BatchNorm + ReLU6 fold: pw_smallm_kernel (pw_no_smallm=1: the split-K pair)
        Mr, K, N = 4 * 26 * 26, 512, a.cout if a.cout != 256 else 512
        x, w, y = rn(Mr, K), rn(K, N) * 0.05, torch.empty(Mr, N, device=dev)
        sc, sh = torch.rand(N, device=dev, generator=g) + 0.5, rn(N)
        fn = lambda: X.call("myolo_pwconv1x1_affine_act_fwd", X.ptr(x), X.ptr(w), X.ptr(sc), X.ptr(sh), 2, X.ptr(y), Mr, K, N, ws.data_ptr(), ws.numel(), st)   # noqa: E731
        ms = timeit(fn, a.iters)
        nbytes = 4.0 * (Mr * K + K * N + Mr * N)
        print("pw_smallm M=%d %d->%d: %.4f ms  %.1f TF/s (%.1f%% of 157.3 fp32 MFMA)  %.0f GB/s on %.1f MB" % (Mr, K, N, ms, 2.0 * Mr * K * N / ms / 1e9, 2.0 * Mr * K * N / ms / 1e9 / 1.573, nbytes / ms / 1e6, nbytes / 1e6))
    elif a.which.startswith("roialign"):
        B, H = 32, 28
        R = NR // B
        feat = rn(B, H, H, C)
        c, s = torch.rand(NR, 2, device=dev, generator=g), torch.rand(NR, 2, device=dev, generator=g) * 0.5 + 0.05
        boxes = torch.cat([c - s / 2, c + s / 2], 1).contiguous()
        bind = torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(R).contiguous()
        out = torch.empty(M, C, device=dev)
        nbytes = M * C * 4 + B * H * H * C * 4
        if a.which == "roialign_fwd":
            fn = lambda: X.call("myolo_crop_and_resize_fwd", X.ptr(feat), X.ptr(boxes), X.ptr(bind), X.ptr(out), B, H, H, C, NR, ps, ps, st)   # noqa: E731
        else:
            out.normal_()
            fn = lambda: X.call("myolo_roialign_bwd_grouped", X.ptr(out), X.ptr(boxes), X.ptr(feat), B, H, H, C, R, ps, ps, st)   # noqa: E731
        ms = timeit(fn, a.iters)
        print("%s: %.3f ms  %.0f GB/s (%.1f%% of 8000)" % (a.which, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / 80))
    elif a.which == "bn_apply":
        # pure streaming pass over the mask-head activation tensor: reads M*C*4 bytes, writes the same
        x, y = rn(M, C), torch.empty(M, C, device=dev)
        sc, sh = rn(C), rn(C)
        fn = lambda: X.call("myolo_bn_apply_act", X.ptr(x), X.ptr(sc), X.ptr(sh), X.ptr(y), M, C, 1, st)   # noqa: E731
        ms = timeit(fn, a.iters)
        print("bn_apply M=%d C=%d: %.3f ms  %.0f GB/s (read %d B + write %d B)" % (M, C, ms, 2 * M * C * 4 / ms / 1e6, M * C * 4, M * C * 4))
    elif a.which == "dw":
        # the 14 depthwise layers of the backbone + YOLO head at 224x224, batch 32, alpha 1
        layers = [(112, 32, 1), (112, 64, 2), (56, 64, 1), (56, 128, 2), (28, 256, 1), (28, 256, 1), (28, 512, 2),
                  (14, 512, 1), (14, 512, 1), (14, 512, 1), (14, 512, 1), (14, 512, 1), (14, 512, 2), (7, 1024, 1)]
        tot_ms, tot_b = 0.0, 0.0
        if os.environ.get("KBENCH_DW_ROWS1"):
            X.set_option("dw_rows1", 1)
        if os.environ.get("KBENCH_DW_MIN_WG"):
            X.set_option("dw_min_wg", int(os.environ["KBENCH_DW_MIN_WG"]))
        for H, Cc, s in layers:
            x, w = rn(32, H, H, Cc), rn(3, 3, Cc)
            y = torch.empty(32, H // s, H // s, Cc, device=dev)
            fn = lambda: X.call("myolo_dwconv3x3_fwd", X.ptr(x), X.ptr(w), X.ptr(y), 32, H, H, Cc, s, st)   # noqa: E731
            ms = timeit(fn, a.iters)
            nb = (x.numel() + y.numel()) * 4
            tot_ms += ms
            tot_b += nb
            print("dw %3dx%3dx%4d s%d: %.4f ms %6.0f GB/s" % (H, H, Cc, s, ms, nb / ms / 1e6))
        print("dw total: %.3f ms, %.0f GB/s (%.1f%% of 8000)" % (tot_ms, tot_b / tot_ms / 1e6, tot_b / tot_ms / 1e6 / 80))
    elif a.which == "dw_fused":
        # the depthwise layers AS THE TRAINING STEP RUNS THEM: producer's BatchNorm + ReLU6 applied on load, this conv's BatchNorm statistics from
        # its epilogue (phase 1 of myolo_dwconv3x3_bnstats_fwd; the finish launch is timed separately), inputs rotated through enough
        # buffers (> 600 MB) that no launch finds its input in the 256 MB Infinity Cache
        layers = [(112, 32, 1), (112, 64, 2), (56, 64, 1), (56, 128, 2), (28, 256, 1), (28, 256, 1), (28, 512, 2),
                  (14, 512, 1), (14, 512, 1), (14, 512, 1), (14, 512, 1), (14, 512, 1), (14, 512, 2), (7, 1024, 1)]
        if os.environ.get("KBENCH_DW_MIN_WG"):
            X.set_option("dw_min_wg", int(os.environ["KBENCH_DW_MIN_WG"]))
        tot_ms, tot_b = 0.0, 0.0
        import ctypes
        for H, Cc, s in layers:
            nbytes = 32 * H * H * Cc * 4
            nbuf = max(2, int(640e6 // nbytes) + 1) if os.environ.get("KBENCH_COLD", "1") == "1" else 1
            xs = [rn(32, H, H, Cc) for _ in range(nbuf)]
            w = rn(3, 3, Cc)
            sc, sh = torch.rand(Cc, device=dev) + 0.5, rn(Cc) * 0.1
            ys = [torch.empty(32, H // s, H // s, Cc, device=dev) for _ in range(min(nbuf, 4))]
            gam, bet = torch.ones(Cc, device=dev), torch.zeros(Cc, device=dev)
            outs = [torch.empty(Cc, device=dev) for _ in range(6)]
            wsb = X.dw_bnstats_ws_bytes(32, H, H, Cc, s)
            wsd = torch.empty(wsb, dtype=torch.uint8, device=dev)
            it = [0]

            def fn(phases=1):
                i = it[0]
                it[0] += 1
                X.call("myolo_dwconv3x3_bnstats_fwd", X.ptr(xs[i % nbuf]), X.ptr(sc), X.ptr(sh), 2, X.ptr(w), X.ptr(ys[i % len(ys)]),
                       X.ptr(gam), X.ptr(bet), *[X.ptr(o) for o in outs], 32, H, H, Cc, s, phases, wsd.data_ptr(), wsb, st)
            ms = timeit(lambda: fn(1), a.iters)
            ms_fin = timeit(lambda: fn(2), a.iters)
            nb = (xs[0].numel() + ys[0].numel()) * 4
            tot_ms += ms
            tot_b += nb
            print("dw_fused %3dx%3dx%4d s%d: %.4f ms %6.0f GB/s   (finish launch %.4f ms)" % (H, H, Cc, s, ms, nb / ms / 1e6, ms_fin))
        print("dw_fused total: %.3f ms, %.0f GB/s (%.1f%% of 8000)" % (tot_ms, tot_b / tot_ms / 1e6, tot_b / tot_ms / 1e6 / 80))
    elif a.which == "bn_bwd":
        # training-mode BatchNorm + ReLU6 backward of the trunk's layers (three launches: sums, finish, dx), inputs rotated through > 600 MB:
        # 5 tensor passes (dy, x read twice; dx written once)
        shapes = [("conv1/dw1", 401408, 32), ("pw1", 401408, 64), ("dw2-3/pw2", 100352, 64), ("pw3", 100352, 128), ("dw4", 25088, 128), ("pw4-5/dw5-6", 25088, 256),
                  ("pw6", 25088, 512), ("dw7", 6272, 512)]
        tot = [0.0, 0.0]
        for name, Mr, Cc in shapes:
            nbuf = max(2, int(640e6 // (2 * Mr * Cc * 4)) + 1)
            xs, dys = [rn(Mr, Cc) for _ in range(nbuf)], [rn(Mr, Cc) for _ in range(nbuf)]
            gam, mean, var = torch.rand(Cc, device=dev) + 0.5, rn(Cc) * 0.1, torch.rand(Cc, device=dev) + 0.5
            scale = gam / torch.sqrt(var + 1e-3); shift = -mean * scale
            dx, dg, db = torch.empty(Mr, Cc, device=dev), torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
            it = [0]

            def f():
                i = it[0]; it[0] += 1
                X.call("myolo_bn_act_bwd", X.ptr(dys[i % nbuf]), X.ptr(xs[i % nbuf]), X.ptr(gam), X.ptr(mean), X.ptr(var), X.ptr(scale), X.ptr(shift),
                       X.ptr(dx), X.ptr(dg), X.ptr(db), Mr, Cc, 2, 1, ws.data_ptr(), ws.numel(), st)
            ms = timeit(f, a.iters)
            nb = 5.0 * Mr * Cc * 4
            tot[0] += ms; tot[1] += nb
            print("bn_bwd %-12s M=%6d C=%4d: %.4f ms %6.0f GB/s" % (name, Mr, Cc, ms, nb / ms / 1e6))
        print("bn_bwd total (one of each): %.3f ms, %.0f GB/s" % (tot[0], tot[1] / tot[0] / 1e6))
    elif a.which in ("dw_bwd", "pw_fused"):
        # the trunk's other launches as the training step runs them, inputs rotated through > 600 MB of buffers (no Infinity-Cache hits):
        #   dw_bwd:   depthwise data gradient and weight gradient (input re-normalised on load) of the 14 layers
        #   pw_fused: the 14 pointwise layers (A operand normalised on load, BatchNorm partial sums in the epilogue; phase 1 only)
        layers = [(112, 32, 1, 64), (112, 64, 2, 64), (56, 64, 1, 128), (56, 128, 2, 256), (28, 256, 1, 256), (28, 256, 1, 512), (28, 512, 2, 512),
                  (14, 512, 1, 512), (14, 512, 1, 512), (14, 512, 1, 512), (14, 512, 1, 512), (14, 512, 1, 512), (14, 512, 2, 1024), (7, 1024, 1, 1024)]
        tot = {}
        for li, (H, Cc, s, Co) in enumerate(layers, 1):
            Ho = H // s
            nb_in = 32 * H * H * Cc * 4
            nbuf = max(2, int(640e6 // nb_in) + 1)
            sc, sh = torch.rand(Cc, device=dev) + 0.5, rn(Cc) * 0.1
            it = [0]
            if a.which == "dw_bwd":
                xs = [rn(32, H, H, Cc) for _ in range(nbuf)]
                dys = [rn(32, Ho, Ho, Cc) for _ in range(max(2, min(nbuf, int(640e6 // (32 * Ho * Ho * Cc * 4)) + 1)))]
                w, dx, dwg = rn(3, 3, Cc), torch.empty(32, H, H, Cc, device=dev), torch.empty(3, 3, Cc, device=dev)

                def f_data():
                    i = it[0]; it[0] += 1
                    X.call("myolo_dwconv3x3_bwd_data", X.ptr(dys[i % len(dys)]), X.ptr(w), X.ptr(dx), 32, H, H, Cc, s, st)

                def f_wgrad():
                    i = it[0]; it[0] += 1
                    X.call("myolo_dwconv3x3_bwd_weight_affine_in", X.ptr(xs[i % nbuf]), X.ptr(sc), X.ptr(sh), 2, X.ptr(dys[i % len(dys)]), X.ptr(dwg),
                           32, H, H, Cc, s, ws.data_ptr(), ws.numel(), st)
                for name, fn, nb in (("bwd_data", f_data, (32 * Ho * Ho + 32 * H * H) * Cc * 4), ("bwd_weight", f_wgrad, (32 * Ho * Ho + 32 * H * H) * Cc * 4)):
                    ms = timeit(fn, a.iters)
                    tot.setdefault(name, [0.0, 0.0])
                    tot[name][0] += ms; tot[name][1] += nb
                    print("dw%-2d %-10s %3dx%3dx%4d s%d: %.4f ms %6.0f GB/s" % (li, name, H, H, Cc, s, ms, nb / ms / 1e6))
            else:
                Mr = 32 * Ho * Ho
                nbuf = max(2, int(640e6 // (Mr * Cc * 4)) + 1)
                xs = [rn(Mr, Cc) for _ in range(nbuf)]
                w, y = rn(Cc, Co) * 0.05, torch.empty(Mr, Co, device=dev)
                gam, bet = torch.ones(Co, device=dev), torch.zeros(Co, device=dev)
                outs = [torch.empty(Co, device=dev) for _ in range(6)]
                wsb = X.pw_bnstats_ws_bytes(Mr, Cc, Co)
                wsd = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)

                def f_pw(phases=1):
                    i = it[0]; it[0] += 1
                    X.call("myolo_pwconv1x1_bnstats_fwd", X.ptr(xs[i % nbuf]), X.ptr(sc), X.ptr(sh), 2, X.ptr(w), X.ptr(y), X.ptr(gam), X.ptr(bet),
                           *[X.ptr(o) for o in outs], Mr, Cc, Co, phases, wsd.data_ptr(), wsb, st)
                ms = timeit(lambda: f_pw(1), a.iters)
                ms2 = timeit(lambda: f_pw(2), a.iters)
                fl, nb = 2.0 * Mr * Cc * Co, 4.0 * (Mr * Cc + Mr * Co + Cc * Co)
                tot.setdefault("pw", [0.0, 0.0, 0.0])
                tot["pw"][0] += ms; tot["pw"][1] += nb; tot["pw"][2] += fl
                print("pw%-2d M=%6d %4d->%4d: %.4f ms %6.1f TF/s %6.0f GB/s   (finish launch %.4f ms)" % (li, Mr, Cc, Co, ms, fl / ms / 1e9, nb / ms / 1e6, ms2))
        for k, v in tot.items():
            print("%s %s total: %.3f ms, %.0f GB/s (%.1f%% of 8000)%s" % (a.which, k, v[0], v[1] / v[0] / 1e6, v[1] / v[0] / 1e6 / 80,
                                                                         "  %.1f TF/s" % (v[2] / v[0] / 1e9) if len(v) > 2 else ""))


if __name__ == "__main__":
    main()
