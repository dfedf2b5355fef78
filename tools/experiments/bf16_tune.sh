#!/bin/bash
# Timing-only ablations of the bf16 deconv + mask kernel (gemm_bf16_256<PLAIN, DECONV_MASK, LOOPN>; results are wrong, durations are what is measured):
# a second library with csrc/bf16_kernels.hip compiled with -DBF16_TUNE, steady state (30 untimed launches first), option tune0 = bit mask:
#    1  no epilogue (ReLU, 1x1 mask conv, partial-logit stores) after a tap's four k tiles      2  accumulators not re-initialised with the bias
#   build here:  bash tools/experiments/bf16_tune.sh build        run:  gpurun -- 'bash tools/experiments/bf16_tune.sh run'
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
L=$ROOT/mask-yolo_amd/myolo/_lib
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DBF16_TUNE -c $ROOT/mask-yolo_amd/csrc/bf16_kernels.hip -o $L/bf16_kernels.tune.o || exit 1
  objs=""; for u in gemm_kernels wino_mm wino_kernels wino63_kernels mem_kernels exact_kernels comm_rccl; do objs="$objs $L/$u.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libmyolo_hip_tune.so $objs $L/bf16_kernels.tune.o -ldl && echo built $L/libmyolo_hip_tune.so
  exit
fi
cd $ROOT
for t in 0 1 2 3 0; do
  MYOLO_LIB=$L/libmyolo_hip_tune.so KBENCH_OPTIONS=tune0=$t python tools/kbench.py deconv_mask_bf16_fwd --warm 30 --iters 30 2>&1 | tail -1 | sed "s/^/tune0=$t  /"
done
