S=$(date +%s)
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "rc=$? wall=$(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/bench_default.json").read().strip().splitlines()[-1])
print(d["metric"], d["value"], d["steps"], d["warmup"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
print(list(d.keys()))
PY
