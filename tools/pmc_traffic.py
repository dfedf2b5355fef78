#!/usr/bin/env python
"""HBM traffic of one kernel, measured the way /opt/skills/guides/MI355X_MICROARCH.md prescribes: two SEPARATE `rocprofv3 --pmc` passes
(FETCH_SIZE, WRITE_SIZE; --kernel-trace only) on a tools/kbench.py target, steady-state launches only (the last five of 36), corrected for gfx950
(FETCH_SIZE counts half of a wide streaming read): traffic = (2 x FETCH_SIZE + WRITE_SIZE) KB x 1024 per launch.  Prints ONE JSON line.
    python tools/pmc_traffic.py wino63_mm wino_mm_x6_kernel [KBENCH_OPTIONS]
bench.py runs this as a subprocess after its timed region (`roofline.traffic_live`); tools/collect_pmc.sh x6 is the fuller offline version."""
import collections
import csv
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counter, target, kname, opts, timeout):
    out = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    if opts:
        env["KBENCH_OPTIONS"] = opts
    cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
           sys.executable, os.path.join(ROOT, "tools", "kbench.py"), target, "--warm", "30", "--iters", "6"]
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
        kt = [r for r in csv.DictReader(open(os.path.join(out, "p_kernel_trace.csv"))) if kname in r["Kernel_Name"]]
        kt = sorted(kt, key=lambda r: int(r["Start_Timestamp"]))[-5:]
        keep = set(r["Dispatch_Id"] for r in kt)
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(os.path.join(out, "p_counter_collection.csv"))):
            if kname in r["Kernel_Name"] and r["Dispatch_Id"] in keep and r["Counter_Name"] == counter:
                acc[r["Dispatch_Id"]] += float(r["Counter_Value"])
        if not acc:
            raise RuntimeError("no %s rows for %s" % (counter, kname))
        ns = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt) / len(kt)
        return sum(acc.values()) / len(acc), ns, len(acc)
    finally:
        shutil.rmtree(out, ignore_errors=True)


def main():
    target, kname = sys.argv[1], sys.argv[2]
    opts = sys.argv[3] if len(sys.argv) > 3 else ""
    timeout = float(os.environ.get("PMC_TIMEOUT_S", "90"))
    res = {"kbench": target, "kernel": kname, "kbench_options": opts}
    try:
        f, ns_f, n = one_pass("FETCH_SIZE", target, kname, opts, timeout)
        w, ns_w, _ = one_pass("WRITE_SIZE", target, kname, opts, timeout)
        res.update(FETCH_SIZE_KB_per_launch=f, WRITE_SIZE_KB_per_launch=w, launches=n, avg_ns_per_launch=0.5 * (ns_f + ns_w),
                   traffic_bytes_per_launch_corrected=1024.0 * (2 * f + w),
                   method="two separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE, --kernel-trace only), last five of 36 launches, 2 x FETCH_SIZE + WRITE_SIZE (gfx950)")
    except Exception as e:
        res["error"] = "%s: %s" % (type(e).__name__, e)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
