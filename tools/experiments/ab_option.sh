# alternating A/B of a library option over whole steps:  ab_option.sh <option> [reps]
O=$1; REPS=${2:-3}
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-variant --no-live-pmc --cpu-images 0"
for rep in $(seq $REPS); do
  for v in 1 0; do
    r=$($B --lib-option $O=$v 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['step_ms']['p50'])")
    echo "$O=$v ms,p50 = $r"
  done
done
