#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command: per-kernel and per-(kernel, grid) time of the training step.
#   gpurun -- 'bash tools/profile_step.sh r2c'      -> gpurun_out/prof_<tag>/<tag>_bench_kernel_{stats,by_grid}.csv
TAG=${1:-r2}
shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python bench.py --steps 5 --warmup 2 --cpu-images 0 --no-variant --no-extras "$@" > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/bench/bench_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
python - "$OUT/bench/bench_kernel_trace.csv" "$OUT/${TAG}_bench_kernel_by_grid.csv" <<'PY'
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    grid = "x".join(r[k] for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = "x".join(r[k] for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
    acc[(r["Kernel_Name"], grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = sorted(((sum(v), k, v) for k, v in acc.items()), reverse=True)
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Grid_Size(threads)", "Workgroup_Size", "Calls", "TotalNs", "AverageNs", "MinNs", "MaxNs"])
    for tot, k, v in out[:80]:
        w.writerow([k[0], k[1], k[2], len(v), tot, tot / len(v), min(v), max(v)])
PY
python - "$OUT/bench/bench_kernel_trace.csv" > "$OUT/${TAG}_timeline.txt" <<'PY'
# where the wall time of one step goes: segments of the step delimited by marker kernels; for each the wall span, the summed
# kernel time (all streams) and the idle time of the main timeline (no kernel of any stream running)
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
starts = [i for i, e in enumerate(ev) if e[2].startswith("conv1_fwd_")]
if len(starts) >= 3:
    a, b = starts[-2], starts[-1]                       # the last complete step
    step = ev[a:b]
    marks = [("trunk forward (conv1 .. yolo_loss)", "conv1_fwd_", "yolo_loss_kernel"),
             ("mask head forward (.. bce)", "yolo_loss_kernel", "bce_"),
             ("mask head backward (.. ROIAlign bwd)", "bce_", "crop_bwd"),
             ("backbone backward + Adam (.. end of step)", "crop_bwd", None)]
    def first(name, lo=0):
        for i in range(lo, len(step)):
            if name in step[i][2]:
                return i
        return None
    print("step wall %.3f ms, %d kernels, kernel time summed %.3f ms" % ((ev[b][0] - ev[a][0]) / 1e6, len(step), sum(e[1] - e[0] for e in step) / 1e6))
    for title, m0, m1 in marks:
        i0 = first(m0)
        i1 = (first(m1, i0 + 1) if m1 else len(step)) if i0 is not None else None
        if i0 is None or i1 is None:
            print(title, ": marker not found"); continue
        seg = step[i0:i1]
        t0, t1 = seg[0][0], (step[i1][0] if i1 < len(step) else ev[b][0])
        busy, cur_end = 0, t0
        for s0, e0, _ in seg:                           # union of kernel intervals
            if e0 > cur_end:
                busy += e0 - max(s0, cur_end); cur_end = e0
        small = [e0 - s0 for s0, e0, _ in seg if e0 - s0 < 20000]
        print("%-48s wall %7.3f ms  kernels %4d  summed %7.3f ms  idle %6.3f ms  (<20us kernels: %d, %.3f ms)" %
              (title, (t1 - t0) / 1e6, len(seg), sum(e0 - s0 for s0, e0, _ in seg) / 1e6, (t1 - t0 - busy) / 1e6, len(small), sum(small) / 1e6))
        import collections
        acc = collections.defaultdict(lambda: [0, 0])
        for s0, e0, nm in seg:
            k = nm.split("(")[0][:64]
            acc[k][0] += e0 - s0; acc[k][1] += 1
        for k, (t, n) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:12]:
            print("      %-66s %3d x  %8.1f us" % (k, n, t / 1e3))
PY
python - "$OUT/bench/bench_kernel_trace.csv" > "$OUT/${TAG}_step_sequence.txt" <<'PY'
# every kernel of the last complete step in start order: offset from the step's start, duration, gap to the end of the previous
# kernel on the timeline (negative = overlapped with it: another stream), stream / queue id, grid, name
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
starts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("conv1_fwd_")]
if len(starts) >= 3:
    a, b = starts[-2], starts[-1]
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = t0
    print("%10s %9s %8s %6s %-18s %s" % ("start_us", "dur_us", "gap_us", "queue", "grid", "kernel"))
    for r in rows[a:b]:
        s0, e0 = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        grid = "x".join(r[k] for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        print("%10.1f %9.1f %8.1f %6s %-18s %s" % ((s0 - t0) / 1e3, (e0 - s0) / 1e3, (s0 - prev_end) / 1e3, r.get("Queue_Id", "?"), grid, r["Kernel_Name"].split("(")[0][:90]))
        prev_end = max(prev_end, e0)
PY
cat "$OUT/${TAG}_timeline.txt"
rm -rf $OUT/bench
tail -2 $OUT/bench_under_rocprof.log | cut -c1-300
