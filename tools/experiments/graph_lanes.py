"""Graph lanes: inference forwards of two batches in flight (Net.predict_graphed(lane=...) on two streams) against one, at the Rice 416 / bf16
configuration, and what breaks the overlap.  LANE0_DEFAULT=capture_only|replay_only|eager_default|1 uses lane 0 on the default stream first:
   gpurun -- 'for v in "" capture_only replay_only eager_default; do LANE0_DEFAULT=$v python tools/experiments/graph_lanes.py; done'
Round 3, one box: 1000 img/s serial -> 1078 with two lanes; with lane 0's graph CAPTURED while the legacy default stream was current: 997 -> 997
(no overlap at all); captured under a side stream and merely replayed on the default stream: 1000 -> 1078.  Hence Net._capture_predict never
captures with the default stream current (the variants below go through Net._capture_predict and therefore show the fixed behaviour now)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
import torch
from myolo.config import make_config, RiceConfig
from myolo.engine import Net
dev = "cuda:0"
bsz = 4
cfg = make_config(RiceConfig, BATCH_SIZE=bsz, INFERENCE_DTYPE="bf16")
net = Net(cfg, device=dev, seed=0)
xs = [torch.rand(bsz, 416, 416, 3, device=dev) for _ in range(3)]
streams = [net._lane_state(l)["stream"] for l in range(3)]
V = os.environ.get("LANE0_DEFAULT")
if V == "capture_only":          # capture while the default stream is current, never replay there
    net._graphs[(tuple(xs[0].shape), cfg.INFERENCE_DTYPE, net.conv3x3_algo, net.fp32_matmul, net.wino_tiles, 0)] = net._capture_predict(xs[0])
elif V == "replay_only":         # capture under the lane stream, then replay on the default stream
    with torch.cuda.stream(streams[0]):
        net.predict_graphed(xs[0])
    torch.cuda.synchronize()
    for _ in range(3):
        net.predict_graphed(xs[0])
    torch.cuda.synchronize()
elif V == "eager_default":       # only eager forwards on the default stream before
    for _ in range(3):
        net.predict(xs[0])
    torch.cuda.synchronize()
elif V:
    for _ in range(3):
        net.predict_graphed(xs[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(40):
        net.predict_graphed(xs[0])
    torch.cuda.synchronize()
    print("default stream serial: %.3f ms" % (1e3 * (time.perf_counter() - t0) / 40), flush=True)
if V in ("capture_only", "replay_only", "eager_default"):
    pass
for l in range(3):
    with torch.cuda.stream(streams[l]):
        for _ in range(3):
            net.predict_graphed(xs[l], lane=l)
torch.cuda.synchronize()
print([k[-1] for k, v in net._graphs.items() if v[0] is not None], flush=True)
def run(k, steps, mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    keep = []
    for i in range(steps):
        j = i % k
        with torch.cuda.stream(streams[j]):
            o = net.predict_graphed(xs[j], lane=j)
            if mode >= 1:
                o = tuple(t.clone() for t in o)
            if mode >= 2:
                ev = torch.cuda.Event(); ev.record(streams[j]); keep.append((ev, o))
        if mode >= 2 and len(keep) == 2 * k:
            keep.pop(0)[0].synchronize()
    torch.cuda.synchronize()
    return time.perf_counter() - t0
for mode in (0,):
    for k in (1, 2):
        run(k, 6, mode)
        el = run(k, 60, mode)
        print("mode %d in flight %d: %.1f img/s, %.3f ms" % (mode, k, bsz * 60 / el, 1e3 * el / 60), flush=True)
def gen(n, k):
    for i in range(n): yield xs[i % k]
for k in (1, 2):
    for _ in net.predict_stream(gen(6, k), in_flight=k): pass
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in net.predict_stream(gen(60, k), in_flight=k): pass
    torch.cuda.synchronize(); el = time.perf_counter() - t0
    print("predict_stream in flight %d: %.1f img/s, %.3f ms" % (k, bsz * 60 / el, 1e3 * el / 60), flush=True)
