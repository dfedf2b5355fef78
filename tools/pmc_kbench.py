#!/usr/bin/env python
"""Per-kernel HBM traffic and SQ counters of a tools/kbench.py target, grouped by (kernel, grid): the method of
/opt/skills/guides/MI355X_MICROARCH.md -- SEPARATE `rocprofv3 --pmc` passes with --kernel-trace only (FETCH_SIZE | WRITE_SIZE | an SQ set), traffic =
2 x FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of a wide streaming read), KB x 1024.  Means over the last `--last` launches of each group.

    python tools/pmc_kbench.py dw_fused --match dw_rows_kernel --out gpurun_out/r4_pmc_dw.json [--opts dw_legacy=1]
"""
import argparse
import collections
import csv
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SQ = "SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"


def one_pass(counters, target, opts, extra, timeout):
    out = tempfile.mkdtemp(prefix="pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    if opts:
        env["KBENCH_OPTIONS"] = opts
    cmd = ["rocprofv3", "--pmc"] + counters.split() + ["--kernel-trace", "--output-format", "csv", "-d", out, "-o", "p", "--",
                                                       sys.executable, os.path.join(ROOT, "tools", "kbench.py"), target] + extra
    try:
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=timeout, check=True)
        kt = list(csv.DictReader(open(os.path.join(out, "p_kernel_trace.csv"))))
        cc = list(csv.DictReader(open(os.path.join(out, "p_counter_collection.csv"))))
        return kt, cc
    finally:
        shutil.rmtree(out, ignore_errors=True)


def grid_of(r):
    return "%sx%sx%s" % (r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""))


def group(kt, cc, match, last):
    by = collections.defaultdict(list)
    for r in kt:
        if match in r["Kernel_Name"]:
            by[(r["Kernel_Name"], grid_of(r))].append(r)
    vals = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in cc:
        vals[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
    res = {}
    for key, rows in by.items():
        rows = sorted(rows, key=lambda r: int(r["Start_Timestamp"]))[-last:]
        ns = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows) / len(rows)
        acc = collections.defaultdict(float)
        for r in rows:
            for k, v in vals[r["Dispatch_Id"]].items():
                acc[k] += v / len(rows)
        res[key] = (ns, dict(acc), len(rows), int(rows[0]["Start_Timestamp"]))
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("target")
    ap.add_argument("--match", required=True)
    ap.add_argument("--opts", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--last", type=int, default=10)
    ap.add_argument("--iters", type=int, default=12)
    ap.add_argument("--no-sq", action="store_true")
    a = ap.parse_args()
    extra = ["--iters", str(a.iters)]
    timeout = float(os.environ.get("PMC_TIMEOUT_S", "240"))
    passes = {}
    for name, ctr in (("FETCH_SIZE", "FETCH_SIZE"), ("WRITE_SIZE", "WRITE_SIZE")) + (() if a.no_sq else (("SQ", SQ),)):
        kt, cc = one_pass(ctr, a.target, a.opts, extra, timeout)
        passes[name] = group(kt, cc, a.match, a.last)
    rows = []
    for key in sorted(passes["FETCH_SIZE"], key=lambda k: passes["FETCH_SIZE"][k][3]):
        ns_f, vf, n, _ = passes["FETCH_SIZE"][key]
        ns_w, vw, _, _ = passes["WRITE_SIZE"].get(key, (ns_f, {}, 0, 0))
        e = {"kernel": key[0][:160], "grid": key[1], "launches": n, "avg_ns": 0.5 * (ns_f + ns_w),
             "FETCH_SIZE_KB": vf.get("FETCH_SIZE"), "WRITE_SIZE_KB": vw.get("WRITE_SIZE")}
        if e["FETCH_SIZE_KB"] is not None and e["WRITE_SIZE_KB"] is not None:
            e["traffic_bytes_corrected"] = 1024.0 * (2 * e["FETCH_SIZE_KB"] + e["WRITE_SIZE_KB"])
            e["traffic_gbs"] = e["traffic_bytes_corrected"] / e["avg_ns"]
        if "SQ" in passes and key in passes["SQ"]:
            ns_s, vs, _, _ = passes["SQ"][key]
            e["sq"] = vs
            e["sq_avg_ns"] = ns_s
            if vs.get("GRBM_GUI_ACTIVE") and vs.get("SQ_WAVE_CYCLES"):
                e["wait_frac_of_wave_cycles"] = vs.get("SQ_WAIT_ANY", 0.0) / vs["SQ_WAVE_CYCLES"]
                e["valu_frac_of_wave_cycles"] = vs.get("SQ_ACTIVE_INST_VALU", 0.0) / vs["SQ_WAVE_CYCLES"]
        rows.append(e)
    res = {"kbench": a.target, "kbench_options": a.opts, "match": a.match,
           "method": "separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ set), --kernel-trace only; traffic = 2 x FETCH_SIZE + WRITE_SIZE "
                     "(gfx950 correction of MI355X_MICROARCH.md), KB x 1024; means over the last %d launches of each (kernel, grid) group" % a.last,
           "rows": rows}
    txt = json.dumps(res, indent=1)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt)
    for e in rows:
        print("%-60s %-14s %8.1f us  fetch %9.0f KB write %9.0f KB  traffic %6.1f MB  %6.0f GB/s  wait %.2f" % (
            e["kernel"][:60], e["grid"], e["avg_ns"] / 1e3, e["FETCH_SIZE_KB"] or 0, e["WRITE_SIZE_KB"] or 0,
            (e.get("traffic_bytes_corrected") or 0) / 1e6, e.get("traffic_gbs") or 0, e.get("wait_frac_of_wave_cycles") or 0))


if __name__ == "__main__":
    main()
