"""Does a working set that fits the 256 MiB Infinity Cache (MALL) move faster than HBM?  Ping-pong float4 copies A -> B, B -> A (each launch reads
what the previous one wrote) for working sets |A| + |B| from 32 MiB to 4 GiB: read+write GB/s per size.  The question behind it:
would running the mask head's conv2-4 chain on slices of ROIs whose V / M planes fit the cache take the layer boundaries (4.2 ms at the HBM copy
rate) off HBM?      gpurun -- 'python tools/experiments/mall_probe.py'"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mask-yolo_amd"))
import torch
from myolo import _ext as X

X.load()
dev = "cuda:0"
cus = torch.cuda.get_device_properties(dev).multi_processor_count
print("working set (A+B) MiB | GB/s read+write: 8 wg/CU, 1 wg/CU | nt_store 8 wg/CU")
for mib in (32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024, 4096):
    n = (mib << 20) // 2
    a = torch.zeros(n, dtype=torch.uint8, device=dev)
    b = torch.zeros(n, dtype=torch.uint8, device=dev)
    row = []
    for variant, blocks in ((0, 8 * cus), (0, cus), (1, 8 * cus)):
        iters = max(10, min(400, (8 << 30) // n))
        iters -= iters % 2
        for _ in range(4):
            X.call("myolo_stream_copy", a.data_ptr(), b.data_ptr(), n, variant, blocks, X.stream())
            X.call("myolo_stream_copy", b.data_ptr(), a.data_ptr(), n, variant, blocks, X.stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters // 2):
            X.call("myolo_stream_copy", a.data_ptr(), b.data_ptr(), n, variant, blocks, X.stream())
            X.call("myolo_stream_copy", b.data_ptr(), a.data_ptr(), n, variant, blocks, X.stream())
        e1.record()
        torch.cuda.synchronize()
        row.append(2.0 * n * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    print("%6d | %8.0f %8.0f | %8.0f" % (mib, row[0], row[1], row[2]))
    del a, b
