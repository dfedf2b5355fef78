# what the driver runs at round end, checked the way the driver reads it: ONE stdout line, < 6000 bytes, valid JSON with the contract keys
S=$(date +%s)
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err
echo "rc=$? wall=$(( $(date +%s) - S )) s"
python - <<'PY'
import json
lines=[l for l in open("gpurun_out/bench_default.json").read().splitlines() if l.strip()]
assert len(lines) == 1, "stdout must carry exactly one line, has %d" % len(lines)
assert len(lines[0]) < 6000, "bench line is %d bytes" % len(lines[0])
d=json.loads(lines[0])
for k in ("metric","value","unit","n_gpus","steps","warmup","ms_per_step","higher_is_better","scaling","vs_baseline","dtype","data","config","roofline","cpu_baseline"):
    assert k in d, k
assert d["steps"] == 20 and d["warmup"] == 5 and d["n_gpus"] == 1
print(len(lines[0]), "bytes:", d["metric"], d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
PY
