#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench command: per-kernel and per-(kernel, grid) time of the training step.
#   gpurun -- 'bash tools/profile_step.sh r2c'      -> gpurun_out/prof_<tag>/<tag>_bench_kernel_{stats,by_grid}.csv
TAG=${1:-r2}
shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python bench.py --steps 5 --warmup 2 --cpu-images 0 --no-variant "$@" > $OUT/bench_under_rocprof.log 2>&1
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
rm -rf $OUT/bench
tail -2 $OUT/bench_under_rocprof.log | cut -c1-300
