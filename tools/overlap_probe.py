#!/usr/bin/env python
"""Does an HBM-bound Winograd transform of one half of the ROIs run underneath the MFMA-bound multiply of the other half when
the two are issued on different HIP streams?  (tuning probe; numbers go to profiles/r2_notes.md)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
import torch         # noqa: E402
from myolo import _ext as X   # noqa: E402

dev = "cuda:0"
for opt in os.environ.get("KBENCH_OPTIONS", "").split(","):
    if "=" in opt:
        X.set_option(opt.split("=")[0], int(opt.split("=")[1]))
C, ps = 256, 14
NRh = 32 * 147 // 2
n = X.wino_plane_elems(NRh, ps, ps, C)
g = torch.Generator(device=dev).manual_seed(0)
V = [torch.randn(n, device=dev, generator=g) for _ in range(2)]
M = [torch.empty(n, device=dev) for _ in range(2)]
Vn = [torch.empty(n, device=dev) for _ in range(2)]
U = torch.randn(36, C, C, device=dev, generator=g) * 0.02
bias = torch.zeros(C, device=dev)
s = [torch.cuda.Stream(), torch.cuda.Stream()]


def mul(h, st):
    X.call("myolo_wino_multiply", X.ptr(V[h]), X.ptr(U), X.ptr(M[h]), NRh, ps, ps, C, C, st.cuda_stream)


def outin(h, st):
    X.call("myolo_wino_output_input_transform", X.ptr(M[h]), X.ptr(bias), None, None, None, None, X.ptr(Vn[h]), NRh, ps, ps, C, 1, st.cuda_stream)


def timed(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


cur = torch.cuda.current_stream()


def seq():
    for layer in range(3):
        for h in (0, 1):
            mul(h, cur)
            outin(h, cur)


def two():
    for st in s:
        st.wait_stream(cur)
    for layer in range(3):
        for h in (0, 1):
            mul(h, s[h])
            outin(h, s[h])
    for st in s:
        cur.wait_stream(st)


def two_staggered():
    # half 1 starts one stage late so that its transform always faces the other half's multiply
    for st in s:
        st.wait_stream(cur)
    mul(0, s[0])
    for layer in range(3):
        outin(0, s[0])
        mul(1, s[1])
        if layer < 2:
            mul(0, s[0])
        outin(1, s[1])
    for st in s:
        cur.wait_stream(st)


print("only multiplies  : %.3f ms" % timed(lambda: [mul(h, cur) for _ in range(3) for h in (0, 1)]))
print("only transforms  : %.3f ms" % timed(lambda: [outin(h, cur) for _ in range(3) for h in (0, 1)]))
print("sequential       : %.3f ms" % timed(seq))
print("two streams      : %.3f ms" % timed(two))
print("two, staggered   : %.3f ms" % timed(two_staggered))
