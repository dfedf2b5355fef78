"""What the bf16 matrix pipe SUSTAINS under the package power cap: myolo_mfma_probe (register operands, no memory traffic, eight independent
accumulator blocks per wave, two workgroups of four waves per CU) run back to back for several seconds; TFLOP/s of every 0.25 s window, with the socket
power and shader clock rocm-smi reports at that moment.  bench.py's `mfma_measured_tflops` is the first window's kind of number (a few ms after idle).
    gpurun -- 'python tools/experiments/mfma_sustained.py'"""
import os
import subprocess
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "mask-yolo_amd"))
import torch
from myolo import _ext as X

X.load()
dev = "cuda:0"
cus = torch.cuda.get_device_properties(dev).multi_processor_count
blocks = 2 * cus
out = torch.zeros(blocks * 256, device=dev)


def smi():
    try:
        t = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
        pw = [l.split(":")[-1].strip() for l in t.splitlines() if "Socket Graphics Package Power" in l]
        sc = [l.split("(")[-1].split(")")[0] for l in t.splitlines() if "sclk" in l]
        return (pw[0] if pw else "?"), (sc[0] if sc else "?")
    except Exception as e:
        return "?", "?"


for kind, name, flop, it in ((0, "bf16 32x32x16", 32768.0, 20000), (1, "f32 32x32x2", 4096.0, 10000)):
    per = blocks * 4.0 * it * 8 * flop
    torch.cuda.synchronize()
    time.sleep(1.0)
    print("--- %s: %d workgroups x 4 waves, %d MFMAs x 8 blocks per wave and launch" % (name, blocks, it))
    t_start = time.perf_counter()
    while time.perf_counter() - t_start < 4.0:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 0
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            for _ in range(4):
                X.call("myolo_mfma_probe", kind, it, blocks, out.data_ptr(), X.stream())
            n += 4
            torch.cuda.synchronize() if n % 16 == 0 else None
        e1.record()
        pw, sc = smi()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        print("t = %.2f s: %7.1f TFLOP/s   power %s W   sclk %s" % (time.perf_counter() - t_start, n * per / ms / 1e9, pw, sc))
