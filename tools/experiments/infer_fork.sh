#!/bin/bash
# kernel sequence of one graph replay of the inference forward, forked / serial (see infer_fork.py)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/infer_fork
mkdir -p $OUT
python tools/experiments/infer_fork.py 300 3 > $OUT/ab.txt 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/raw -o p -- python tools/experiments/infer_fork.py 4 1 > $OUT/prof.log 2>&1
python - $OUT <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/raw/**/p_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("conv1_fwd_")]
# the script's last 2 x (1 + 4) forwards are replays: fork=1 x5 then fork=0 x5
def dump(a, b, path):
    t0 = int(rows[a]["Start_Timestamp"]); prev = t0
    with open(path, "w") as g:
        g.write("%10s %9s %8s %6s %-14s %s\n" % ("start_us", "dur_us", "gap_us", "queue", "grid", "kernel"))
        for r in rows[a:b]:
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            g.write("%10.1f %9.1f %8.1f %6s %-14s %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3, r.get("Queue_Id", "?"), r["Grid_Size_X"], r["Kernel_Name"][:90]))
            prev = max(prev, e)
        g.write("wall to the end of the last kernel: %.1f us\n" % ((prev - t0) / 1e3))
n = len(starts)
dump(starts[n - 7], starts[n - 6], out + "/sequence_fork.txt")      # a replay from the middle of the fork=1 run
dump(starts[n - 2], starts[n - 1], out + "/sequence_serial.txt")
PY
rm -rf $OUT/raw
cat $OUT/ab.txt | tail -8
tail -3 $OUT/sequence_fork.txt; tail -3 $OUT/sequence_serial.txt
