import os, sys, ctypes, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
import torch
from myolo import _ext as X
dev = "cuda:0"
NR, C = 4704, 256
pe = X.wino63_plane_elems(NR, C)
Mp, Vn, b, sc, sh = torch.randn(pe, device=dev), torch.empty(pe, device=dev), torch.randn(C, device=dev), torch.randn(C, device=dev), torch.randn(C, device=dev)
st = X.stream()
for _ in range(3):
    X.call("myolo_wino63_output_transform", X.ptr(Mp), X.ptr(b), X.ptr(sc), X.ptr(sh), X.ptr(Vn), NR, C, 1, st)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["MYOLO_LIB"])
n = 8192 * 9 * 8
buf = np.zeros(n, np.uint64)
assert lib.myolo_w63_trace_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(n * 8)) == 0
t = buf[:256 * 4 * 9 * 8].reshape(256, 4, 9, 8).astype(np.int64)
names = ["top->first pass done (waits for the loads)", "->next loads issued, second pass, stores issued"]
d = t[..., 1:3] - t[..., 0:2]
for w in range(9):
    print("wave %d: " % w + "  ".join("%s %.0f" % (names[k], d[:, :, w, k].mean()) for k in range(2)))
it = t[:, 1:, :, 0] - t[:, :-1, :, 0]
print("cycles per unit (top to top): mean %.0f  p10 %.0f p90 %.0f" % (it.mean(), np.percentile(it, 10), np.percentile(it, 90)))
