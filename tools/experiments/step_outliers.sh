for i in $(seq 1 14); do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-variant --no-live-pmc --cpu-images 0 >/dev/null 2>&1
  python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
s,h=d['step_ms'],d['host_step_ms']
print("ms %.3f p50 %.3f max %.2f@%d | host p50 %.2f max %.2f@%d" % (d['ms_per_step'], s['p50'], s['max'], s['max_at_step'], h['p50'], h['max'], h['max_at_step']))
PY
done
