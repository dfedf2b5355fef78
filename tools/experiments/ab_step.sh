#!/bin/bash
# Same-box A/B of whole training steps: alternates libraries (MYOLO_LIB) over `reps` rounds, prints ms per step of each run.
#   tools/experiments/ab_step.sh "<name>=<lib.so> <name>=<lib.so> ..." [reps] [extra bench.py flags]
LIBS=$1; REPS=${2:-3}; shift 2
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-variant --no-live-pmc --cpu-images 0 $@"
for rep in $(seq $REPS); do
  for nl in $LIBS; do
    n=${nl%%=*}; l=${nl#*=}
    r=$(MYOLO_LIB=$PWD/$l $B 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$n ms=$r"
  done
done
