"""VERDICT r4 item 8, the measurement: the mask head's conv2-4 + deconv chain (bn1's activation y1 -> V2 -> M2 -> V3 -> M3 -> V4 -> M4 -> y4 ->
deconv + mask) run on SLICES of n ROIs whose Winograd planes fit the 256 MiB Infinity Cache, against the whole batch at once (4704 ROIs: 1.93 GB per
plane set, every stage at the HBM rate).  tools/experiments/mall_probe.py: a ping-pong copy moves 7.2-8.6 TB/s while its working set is <= 256 MiB and
4.5-5.8 TB/s beyond.  Prints ms per full batch (4704 ROIs) for each slice size, stage by stage.
    gpurun -- 'python tools/experiments/mall_slices.py'"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mask-yolo_amd"))
import torch
from myolo import _ext as X

X.load()
X.set_option("wino_x6", 1)
dev = "cuda:0"
NR, C, q, ncls = 32 * 147, 256, 196, 4
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)
y1 = rn(NR * q, C)
w = [rn(3, 3, C, C) * 0.02 for _ in range(3)]
bias, sc, sh = rn(C) * 0.1, torch.rand(C, device=dev, generator=g) + 0.5, rn(C) * 0.1
wd, bd, w2, b2 = rn(2, 2, C, C) * 0.05, rn(C) * 0.1, rn(1, 1, C, ncls) * 0.1, rn(ncls) * 0.1
U = [torch.empty(X.wino63_u_elems(C, C), device=dev) for _ in range(3)]
y4 = torch.empty(NR * q, C, device=dev)
p = torch.empty(NR * 4 * q, ncls, device=dev)
ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
st = X.stream()
wp = X.WeightPrep(dev)                       # the filters prepared once, as in the training step
wp.activate(True)


def chain(n, lo, V, M, tick=None):
    """the chain on ROIs [lo, lo + n)"""
    yo = y1[lo * q:(lo + n) * q]
    ev = []

    def mark():
        if tick is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            ev.append(e)
    mark()
    X.call("myolo_wino63_input_transform", X.ptr(yo), X.ptr(sc), X.ptr(sh), 1, None, None, X.ptr(V), n, C, st)
    mark()
    for i in range(3):
        X.call("myolo_wino63_multiply_w", X.ptr(V), X.ptr(w[i]), X.ptr(U[i]), X.ptr(M), n, C, C, st)
        mark()
        if i < 2:
            X.call("myolo_wino63_output_input_transform", X.ptr(M), X.ptr(bias), X.ptr(sc), X.ptr(sh), None, None, X.ptr(V), n, C, 1, st)
        else:
            X.call("myolo_wino63_output_transform", X.ptr(M), X.ptr(bias), X.ptr(sc), X.ptr(sh), X.ptr(y4[lo * q:(lo + n) * q]), n, C, 1, st)
        mark()
    X.call("myolo_deconv2x2s2_mask_fwd", X.ptr(y4[lo * q:(lo + n) * q]), X.ptr(wd), X.ptr(bd), X.ptr(w2), X.ptr(b2), X.ptr(p[lo * 4 * q:(lo + n) * 4 * q]),
           n, 14, 14, C, C, ncls, ws.data_ptr(), ws.numel(), st)
    mark()
    if tick is not None:
        tick.append(ev)


def run(n, reps=3):
    V = torch.empty(X.wino63_plane_elems(n, C), device=dev)
    M = torch.empty(X.wino63_plane_elems(n, C), device=dev)
    los = list(range(0, NR, n))
    for lo in los:                       # warm-up (records / refreshes the prepared filters on the first pass)
        chain(min(n, NR - lo), lo, V, M)
    wp.refresh()
    for lo in los:
        chain(min(n, NR - lo), lo, V, M)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        for lo in los:
            chain(min(n, NR - lo), lo, V, M)
    e1.record()
    torch.cuda.synchronize()
    total = e0.elapsed_time(e1) / reps
    ticks = []
    for lo in los:
        chain(min(n, NR - lo), lo, V, M, tick=ticks)
    torch.cuda.synchronize()
    names = ["in", "mm2", "b2", "mm3", "b3", "mm4", "out", "deconv"]
    per = [sum(ev[k].elapsed_time(ev[k + 1]) for ev in ticks) for k in range(8)]
    return total, dict(zip(names, per))


print("slice ROIs | plane set MiB | ms per 4704 ROIs | per stage (event-bracketed, a second pass)")
for n in (NR, 1176, 588, 294, 196, 147, 98):
    t, per = run(n)
    print("%6d | %7.0f | %8.3f | %s" % (n, X.wino63_plane_elems(n, C) * 4 / 2 ** 20, t, "  ".join("%s %.2f" % kv for kv in per.items())))
