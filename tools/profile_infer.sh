#!/bin/bash
# rocprofv3 kernel statistics of the inference bench line (bench.py --config rice416-bf16).   gpurun -- 'bash tools/profile_infer.sh r2g'
#   -> gpurun_out/prof_infer_<tag>/<tag>_infer_kernel_stats.csv and a top-kernel table on stdout
TAG=${1:-r2g}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_infer_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/raw -o p -- python bench.py --config rice416-bf16 --steps 20 --cpu-images 0 > $OUT/bench.log 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, glob, shutil, sys
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(out + "/raw/**/p_kernel_stats.csv", recursive=True)[0]
shutil.copy(f, "%s/%s_infer_kernel_stats.csv" % (out, tag))
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:30]:
    print("%-72s %6s x %9.1f us avg  %5.1f%%" % (r["Name"][:72], r["Calls"], float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
tail -c 400 $OUT/bench.log
rm -rf $OUT/raw
