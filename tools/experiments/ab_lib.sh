# A/B of the working-tree library against tools/experiments/lib_base.so on one box
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-variant --no-live-pmc --cpu-images 0"
L=mask-yolo_amd/myolo/_lib/libmyolo_hip.so
cp $L /tmp/lib_new.so
for rep in 1 2 3; do
for which in new base; do
  if [ $which = new ]; then cp /tmp/lib_new.so $L; else cp tools/experiments/lib_base.so $L; fi
  for fp in 0 20; do
    r=$($B --force-pos $fp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$which force_pos=$fp ms=$r"
  done
done
done
cp /tmp/lib_new.so $L
