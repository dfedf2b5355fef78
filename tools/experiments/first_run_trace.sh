# does the FIRST full default bench run on a fresh box carry a long step?  prints the per-step trace of two runs back to back
for i in 1 2; do
python bench.py --steps 20 --warmup 5 $@ >/dev/null 2>&1
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
print(d['ms_per_step'], d['step_ms']['max'], d['step_ms']['max_at_step'])
t=d['step_trace']
print('gpu ', t['gpu_ms']); print('host', t['host_ms']); print('npos', t['n_pos_total']); print('resv', t['allocator_reserved_mb'])
PY
done
