python -m pytest tests/test_gpu_ops.py -q -x -k "positive_index or keeps_the or deconv_mask" 2>&1 | tail -2
python -m pytest tests/test_gpu_step.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -3
B="python bench.py --steps 20 --warmup 5 --no-extras --no-variant --no-live-pmc --cpu-images 0"
for attr in "" "--net-attr keep_deconv_rows=0" "" "--net-attr keep_deconv_rows=0"; do
  for fp in 0 20; do
    r=$($B $attr --force-pos $fp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "attr='$attr' force_pos=$fp ms=$r"
  done
done
