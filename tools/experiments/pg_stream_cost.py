#!/usr/bin/env python
"""Does torch.distributed's NCCL process group (its internal stream) slow the training step on ONE GPU the way any extra HIP stream does?
   python tools/experiments/pg_stream_cost.py [none|pg|pg_first]      -> ms per step
none: no process group; pg: a 1-rank nccl group initialised and used once (barrier + all_reduce) AFTER the Net exists; pg_first: before.
Round-4 results (ms per step, 224^2 / batch 32): normal-priority side streams (MYOLO_STREAM_PRIORITY=0): none 20.5-20.8, pg 22.0, pg_first 26.3-26.8
(the same with GPU_MAX_HW_QUEUES = 8 / 16 / 24; 22.6 / 22.6 with 4); high-priority side streams (the default since): 20.63 / 20.68 / 20.67."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch                      # noqa: E402
import torch.distributed as dist  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "none"


def pg():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", RANK="0", WORLD_SIZE="1")
    dist.init_process_group("nccl", rank=0, world_size=1)
    t = torch.ones(4, device="cuda:0")
    dist.all_reduce(t)
    dist.barrier()
    torch.cuda.synchronize()


from bench import make_batches    # noqa: E402
from myolo.config import make_config, ShapesConfig   # noqa: E402
from myolo.model import MaskYOLO  # noqa: E402
torch.cuda.set_device(0)
if mode == "pg_first":
    pg()
cfg = make_config(ShapesConfig, BATCH_SIZE=32)
model = MaskYOLO(mode="training", config=cfg, device="cuda:0", seed=0)
net = model.net
dbs = [net.to_device_batch(b) for b in make_batches(cfg, 0, 1, 32, 2)]
if mode == "pg":
    pg()
for i in range(4):
    net.train_step(dbs[i % 2], 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20):
    net.train_step(dbs[i % 2], 1e-3)
torch.cuda.synchronize()
print(mode, "%.2f ms per step" % ((time.perf_counter() - t0) / 20 * 1e3), "| MYOLO_STREAM_PRIORITY =", os.environ.get("MYOLO_STREAM_PRIORITY", "-1 (default)"),
      "| GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"))
if mode != "none":
    dist.destroy_process_group()
