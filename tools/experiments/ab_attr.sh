# usage: ab_attr.sh "<attr A>" "<attr B>" ...   (each a --net-attr / --lib-option string, "" = default)
B="timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-variant --no-live-pmc --cpu-images 0"
for rep in 1 2; do
for attr in "$@"; do
  for fp in 0 20; do
    r=$($B $attr --force-pos $fp 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
    echo "attr='$attr' force_pos=$fp ms=$r"
  done
done
done
