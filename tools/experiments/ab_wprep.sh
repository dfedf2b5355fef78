python -m pytest tests/test_gpu_step.py -q -x -k "prepared_weights" 2>&1 | grep -a "^E\|FAILED\|passed\|failed" | cut -c1-300 | head
B="python bench.py --steps 20 --warmup 5 --no-extras --no-variant --no-live-pmc --cpu-images 0"
for attr in "" "--net-attr weight_prep=0" "" "--net-attr weight_prep=0"; do
  for fp in 0 20; do
    r=$($B $attr --force-pos $fp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "attr='$attr' force_pos=$fp ms=$r"
  done
done
