#!/bin/bash
# Build a variant of libmyolo_hip.so with ONE translation unit recompiled under extra -D flags (A/B runs: MYOLO_LIB=tools/_ab/lib_<name>.so).
#   tools/experiments/mk_variant.sh <name> <unit.hip> [-DFOO=1 ...]
set -e
name=$1; unit=$2; shift 2
R=$(cd "$(dirname "$0")/../.." && pwd)
L=$R/mask-yolo_amd/myolo/_lib
mkdir -p $R/tools/_ab
extra=""
[ "$unit" = mem_kernels.hip ] && extra="-munsafe-fp-atomics"
[ "$unit" = exact_kernels.hip ] && extra="-ffp-contract=off"
o=$R/tools/_ab/${unit%.hip}.$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $extra "$@" -c $R/mask-yolo_amd/csrc/$unit -o $o
objs=""
for u in gemm_kernels bf16_kernels wino_kernels wino_mm wino63_kernels mem_kernels exact_kernels comm_rccl; do
  if [ "$u.hip" = "$unit" ]; then objs="$objs $o"; else objs="$objs $L/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_ab/lib_$name.so $objs -ldl
echo built tools/_ab/lib_$name.so
