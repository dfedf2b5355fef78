"""One inference batch in flight (Rice-416, bf16 mask head, batch 4; Net.predict_graphed): feature_map's conv beside the YOLO head
(Net.infer_fork_feature_map, opt-in) against the serial order, alternating on one box; and, under rocprofv3 --kernel-trace, the
kernel sequence of one replay of each form.
  gpurun -- 'python tools/experiments/infer_fork.py'                                  -> ms per forward, alternating
  gpurun -- 'bash tools/experiments/infer_fork.sh'                                    -> + gpurun_out/infer_fork/sequence_{fork,serial}.txt
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mask-yolo_amd"))
import torch
from myolo.config import make_config, RiceConfig
from myolo.engine import Net

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg = make_config(RiceConfig, BATCH_SIZE=4, INFERENCE_DTYPE="bf16")
net = Net(cfg, device="cuda:0", seed=0)
x = torch.rand(4, 416, 416, 3, device="cuda:0")
outs = {}
for fork in (1, 0):
    net.infer_fork_feature_map = bool(fork)
    for _ in range(3):
        o = net.predict_graphed(x)
    torch.cuda.synchronize()
    outs[fork] = [t.clone() for t in o]
same = all(torch.equal(a, b) for a, b in zip(outs[1], outs[0]))
print("outputs of the two forms bit-identical:", same)
for r in range(rounds):
    for fork in (1, 0):
        net.infer_fork_feature_map = bool(fork)
        net.predict_graphed(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            net.predict_graphed(x)
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / reps
        print("round %d  fork=%d  %.4f ms / forward  %.1f img/s" % (r, fork, ms, 4e3 / ms), flush=True)
