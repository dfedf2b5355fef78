#!/bin/bash
# Timing-only ablations of wino_tn_x6_kernel (conv1's weight gradient; results are wrong, durations are what is measured), same method as x6_tune.sh:
# a second library with csrc/wino_mm.hip compiled with -DMM_X6_TUNE, the product's other objects, tools/kbench.py wino63_wgrad in steady state,
# option tune0 = bit mask:   1 no operand traffic (every load out of range)   2 no exact split (one piece stored three times)   4 no MFMAs
#                            8 no LDS writes                                  16 operands from L2 (the first 16 rows of the split every chunk)
#   build:  bash tools/experiments/tn_x6_tune.sh build       run:  gpurun -- 'bash tools/experiments/tn_x6_tune.sh run'
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
L=$ROOT/mask-yolo_amd/myolo/_lib
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMM_X6_TUNE -c $ROOT/mask-yolo_amd/csrc/wino_mm.hip -o $L/wino_mm.tune.o || exit 1
  objs=""; for u in gemm_kernels bf16_kernels wino_kernels wino63_kernels mem_kernels exact_kernels comm_rccl; do objs="$objs $L/$u.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libmyolo_hip_tune.so $objs $L/wino_mm.tune.o -ldl && echo built $L/libmyolo_hip_tune.so
  exit
fi
cd $ROOT
for t in 0 1 2 4 5 6 7 8 16 3 0; do
  MYOLO_LIB=$L/libmyolo_hip_tune.so KBENCH_OPTIONS=wino_x6=1,tune0=$t python tools/kbench.py wino63_wgrad --warm 20 --iters 20 2>&1 | tail -1 | sed "s/^/tune0=$t  /"
done
