#!/usr/bin/env python
"""Does the layer-boundary transform (HBM-bound) hide under the multiply (MFMA-bound) of the OTHER half of the ROIs when the two are
launched on two streams?  Times, at the config-2 mask-head shape split in two halves of 2352 ROIs:
  serial    : mm(A) mm(B) boundary(A) boundary(B) on one stream
  overlapped: stream 1: mm(A) ; stream 2: boundary(B) at the same time, and the mirror image
      gpurun -- python tools/overlap_mm_boundary.py"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
import torch         # noqa: E402
from myolo import _ext as X   # noqa: E402
dev = "cuda:0"
for kv in os.environ.get("KBENCH_OPTIONS", "").split(","):       # e.g. KBENCH_OPTIONS=wino_x6=1: the bf16x6 multiply
    if "=" in kv:
        X.set_option(kv.split("=")[0], int(kv.split("=")[1]))
NR, C = 2352, 256
g = torch.Generator(device=dev).manual_seed(0)
rn = lambda *s: torch.randn(*s, device=dev, generator=g)   # noqa: E731
w, b = rn(3, 3, C, C) * 0.02, rn(C)
U = torch.empty(X.wino63_u_elems(C, C), device=dev)
pe = X.wino63_plane_elems(NR, C)
VA, VB, MA, MB, V2A, V2B = [torch.randn(pe, device=dev) * 0.1 for _ in range(6)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
X.call("myolo_wino63_weight_transform", X.ptr(w), X.ptr(U), C, C, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
def mm(V, M, st): X.call("myolo_wino63_multiply", X.ptr(V), X.ptr(U), X.ptr(M), NR, C, C, st.cuda_stream)
def bnd(M, V, st): X.call("myolo_wino63_output_input_transform", X.ptr(M), X.ptr(b), None, None, None, None, X.ptr(V), NR, C, 1, st.cuda_stream)
def timed(fn, iters=30):
    for _ in range(30): fn()          # steady state: the first ~25 launches after idle run through a clock transient (profiles/r3_notes.md)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s1)
    for _ in range(iters): fn()
    e1.record(s1)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters
def serial():
    mm(VA, MA, s1); mm(VB, MB, s1); bnd(MA, V2A, s1); bnd(MB, V2B, s1)
def only_mm():
    mm(VA, MA, s1); mm(VB, MB, s1)
def only_bnd():
    bnd(MA, V2A, s1); bnd(MB, V2B, s1)
def overlapped_free():
    # the two chains mm(A) -> bnd(A) and mm(B) -> bnd(B) on their own streams, no joins in between: the hardware interleaves them
    ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
    mm(VA, MA, s1); bnd(MA, V2A, s1)
    mm(VB, MB, s2); bnd(MB, V2B, s2)
    ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
def overlapped_staggered():
    # B's chain one stage behind A's: mm(A) | then mm(B) beside bnd(A) | then bnd(B)
    mm(VA, MA, s1)
    ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
    mm(VB, MB, s2); bnd(MA, V2A, s1)
    ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
    bnd(MB, V2B, s1)
def overlapped():
    # two rounds: [mm(A) || bnd(B)] then [mm(B) || bnd(A)]; s2 joins s1 at both ends of each round
    for (Vm, Mm, Mb, Vb) in ((VA, MA, MB, V2B), (VB, MB, MA, V2A)):
        ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
        mm(Vm, Mm, s1); bnd(Mb, Vb, s2)
        ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
t_mm, t_b, t_s, t_o = timed(only_mm), timed(only_bnd), timed(serial), timed(overlapped)
t_f, t_g = timed(overlapped_free), timed(overlapped_staggered)
print("steady state: two half multiplies %.3f ms, two half boundaries %.3f ms, serial all four %.3f ms, overlapped in two rounds %.3f ms, two free-running chains %.3f ms, "
      "staggered (mm(B) beside boundary(A)) %.3f ms" % (t_mm, t_b, t_s, t_o, t_f, t_g))
