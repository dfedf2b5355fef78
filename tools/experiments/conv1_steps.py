"""conv1_fwd_rows_kernel: pairs of output rows per workgroup (C1F_STEPS), training and inference shapes; us per call, cold-ish (inputs rotate over 8 buffers)
   tools/experiments/mk_variant.sh c1f4 mem_kernels.hip -DC1F_STEPS=4 (while C1F_STEPS was an #ifndef macro); MYOLO_LIB=tools/_ab/lib_c1f4.so python tools/experiments/conv1_steps.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mask-yolo_amd"))
import torch
from myolo import _ext as X
dev = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
for N, H in ((32, 224), (4, 416)):
    xs = [torch.rand(N, H, H, 3, device=dev) for _ in range(8)]
    ys = [torch.empty(N, H // 2, H // 2, 32, device=dev) for _ in range(8)]
    w = torch.randn(3, 3, 3, 32, device=dev)
    def run(i):
        X.call("myolo_conv3x3s2_c3_fwd", X.ptr(xs[i % 8]), X.ptr(w), X.ptr(ys[i % 8]), N, H, H, 32, st)
    for i in range(16):
        run(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(200):
        run(i)
    e1.record()
    torch.cuda.synchronize()
    print("%s N=%d %dx%d: %.1f us" % (os.environ.get("MYOLO_LIB", "default"), N, H, H, 1e3 * e0.elapsed_time(e1) / 200))
