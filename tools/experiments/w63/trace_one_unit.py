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
    X.call("myolo_wino63_output_input_transform", X.ptr(Mp), X.ptr(b), X.ptr(sc), X.ptr(sh), None, None, X.ptr(Vn), NR, C, 1, st)
torch.cuda.synchronize()
lib = ctypes.CDLL(os.environ["MYOLO_LIB"])
n = 8192 * 9 * 8
buf = np.zeros(n, np.uint64)
rc = lib.myolo_w63_trace_read(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(n * 8))
assert rc == 0
t = buf.reshape(8192, 9, 8).astype(np.int64)
# clock64 = s_memtime shader cycles? report in cycles
t = t[:, :8, :]
d = t[:, :, 1:6] - t[:, :, 0:5]
t = t[:, :8, :]
life = t[:, :, 5] - t[:, :, 0]
sel = slice(2048, 8192)        # steady state
names = ["start->colpass done(loads)", "->front done", "->barrier", "->back issued", "->stores acked"]
for w in range(8):
    print("wave %d: " % w + "  ".join("%s %.0f" % (names[k], d[sel, w, k].mean()) for k in range(5)) + "  | life %.0f" % life[sel, w].mean())
wg_life = (t[:, :, 5].max(1) - t[:, :, 0].min(1))
print("workgroup lifetime mean %.0f cycles (p10 %.0f p90 %.0f)" % (wg_life[sel].mean(), np.percentile(wg_life[sel], 10), np.percentile(wg_life[sel], 90)))
span = t[:, :, 5].max() - t[:, :, 0].min()
print("kernel span %.0f cycles; units 18816 -> %.1f cycles per unit per CU-slot(512)" % (span, span / (18816 / 512.0)))
# concurrency: average number of workgroups alive (first 8192)
ev = np.concatenate([np.stack([t[:, :, 0].min(1), np.ones(8192)], 1), np.stack([t[:, :, 5].max(1), -np.ones(8192)], 1)])
ev = ev[ev[:, 0].argsort()]
alive = np.cumsum(ev[:, 1])
dt = np.diff(ev[:, 0])
print("mean workgroups alive (traced 8192): %.1f" % ((alive[:-1] * dt).sum() / dt.sum()))
hw = t[:, 0, 6]
cu = (hw >> 8) & 0xf; sh_ = (hw >> 12) & 1; se = (hw >> 13) & 7
print("HW_ID sample:", [hex(int(x)) for x in hw[:4]], "xcc", t[:8, 0, 7])
np.save(os.path.join(ROOT, "gpurun_out", "trace.npy"), t)
xcc = t[:, 0, 7]
for x in range(2):
    m = xcc == x
    s0 = t[m][:, :, 0].min(1); e0 = t[m][:, :, 5].max(1)
    span = e0.max() - s0.min()
    print("xcc %d: %d workgroups, span %d ticks, sum of lifetimes %d -> %.1f alive on average (32 CUs)" % (x, m.sum(), span, (e0 - s0).sum(), (e0 - s0).sum() / span))
    cu = ((hw[m] >> 8) & 0xf) | (((hw[m] >> 12) & 0xf) << 4)
    ids = np.unique(cu)
    print("  distinct (se,sh,cu) ids:", len(ids))
    c0 = ids[0]
    mm = cu == c0
    o = np.argsort(s0[mm])
    print("  one CU: starts/ends (rel):", [(int(a - s0.min()), int(b - s0.min())) for a, b in zip(s0[mm][o][:10], e0[mm][o][:10])])
    simd = (t[m][:, :, 6] >> 4) & 3
    print("  simd of waves 0..8 (first 4 wgs):", simd[:4].tolist())

for x in range(1):
    m = xcc == x
    s0 = t[m][:, :, 0].min(1); e0 = t[m][:, :, 5].max(1)
    cu = ((hw[m] >> 8) & 0xf) | (((hw[m] >> 12) & 0xf) << 4)
    tot = 0
    for c0 in np.unique(cu):
        mm = cu == c0
        ev = sorted([(a, 1) for a in s0[mm]] + [(b, -1) for b in e0[mm]])
        c = 0; mx = 0
        for _, dlt in ev:
            c += dlt; mx = max(mx, c)
        tot += mx
    print("xcc %d: mean over CUs of max concurrent workgroups: %.2f" % (x, tot / len(np.unique(cu))))
