B="timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-variant --no-live-pmc --cpu-images 0"
for rep in 1 2; do
for opt in "" "--lib-option tn_wgs=240" "--lib-option tn_wgs=208" "--lib-option tn_wgs=192" "--lib-option tn_wgs=256"; do
  for fp in 0 20; do
    r=$($B $opt --force-pos $fp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "opt='$opt' force_pos=$fp ms=$r"
  done
done
done
