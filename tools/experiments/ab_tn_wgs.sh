B="python bench.py --steps 20 --warmup 5 --no-extras --no-variant --no-live-pmc --cpu-images 0"
for opt in "" "--lib-option tn_wgs=224" "--lib-option tn_wgs=208" "--lib-option tn_wgs=192" "--lib-option tn_wgs=160" ""; do
  for fp in 0 20; do
    r=$($B $opt --force-pos $fp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "opt='$opt' force_pos=$fp ms=$r"
  done
done
python -m pytest tests/test_gpu_ops.py -q -x -k "tn or wgrad or weight" 2>&1 | tail -2
