#!/usr/bin/env python
"""Inference throughput at BASELINE.json configs[3] (Rice 416x416, 5 anchors, 28x28 mask head): the device part of
MaskYOLO.detect() -- trunk + decode + ROIAlign of all 845 boxes + mask head -- in fp32 and with the bf16 mask head.
  python tools/bench_infer.py [--batch 4] [--iters 10]
Not the headline metric (bench.py measures the training step); numbers are recorded in profiles/ and BASELINE.md."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
import torch   # noqa: E402
from myolo.config import make_config, RiceConfig   # noqa: E402
from myolo.engine import Net   # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    dev = "cuda:0"
    out = {"workload": "rice416_nbox5_inference", "batch": a.batch}
    for dtype in ("fp32", "bf16"):
        cfg = make_config(RiceConfig, BATCH_SIZE=a.batch, INFERENCE_DTYPE=dtype)
        net = Net(cfg, device=dev, seed=0)
        x = torch.rand(a.batch, 416, 416, 3, device=dev)
        for _ in range(2):
            net.predict(x)
        net.timed_tags = {"mask_conv3x3_fwd", "roialign_fwd"}
        net.timings = {}
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            net.predict(x)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.iters
        conv_ms, _ = net.kernel_ms("mask_conv3x3_fwd")
        roi_ms, _ = net.kernel_ms("roialign_fwd")
        M = a.batch * 845 * 14 * 14
        out[dtype] = {"ms_per_batch": round(ms, 3), "images_per_sec": round(a.batch / ms * 1e3, 2),
                      "mask_conv3x3_ms": round(conv_ms, 3), "mask_conv3x3_tflops": round(2.0 * M * 9 * 256 * 256 / conv_ms / 1e9, 1),
                      "roialign_ms": round(roi_ms, 3)}
        del net
    out["speedup_bf16"] = round(out["bf16"]["images_per_sec"] / out["fp32"]["images_per_sec"], 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
