# A/B of one library option on one box:   bash tools/experiments/ab_opt.sh tune0=4096 [reps]
OPT=$1; REPS=${2:-3}
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-variant --no-live-pmc --cpu-images 0"
for rep in $(seq $REPS); do
  for which in default "$OPT"; do
    if [ "$which" = default ]; then EX=""; else EX="--lib-option $which"; fi
    r=$($B $EX 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$which ms=$r"
  done
done
