# A/B of the working-tree engine.py against tools/experiments/engine_base.py on one box
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-variant --no-live-pmc --cpu-images 0"
cp mask-yolo_amd/myolo/engine.py /tmp/engine_new.py
for rep in 1 2; do
for which in new base; do
  if [ $which = new ]; then cp /tmp/engine_new.py mask-yolo_amd/myolo/engine.py; else cp tools/experiments/engine_base.py mask-yolo_amd/myolo/engine.py; fi
  for fp in 0 20; do
    r=$($B --force-pos $fp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "$which force_pos=$fp ms=$r"
  done
done
done
cp /tmp/engine_new.py mask-yolo_amd/myolo/engine.py
