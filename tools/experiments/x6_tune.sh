#!/bin/bash
# Timing-only ablations of wino_mm_x6_kernel (results are wrong, durations are what is measured): a second library with csrc/wino_mm.hip
# compiled with -DMM_X6_TUNE (the other objects are the product's), then the F(6,3) multiply at the config-2 shape in STEADY STATE
# (60 launches back to back, the mean of the last 30: the first ~25 launches of an MFMA-bound kernel after idle run through a clock
# transient, see profiles/r3_notes.md) with option tune0 = bit mask:
#    1  B fragments always from chunk 0 (L1 hits instead of L2 traffic)      2  no B loads after the prologue
#    4  no exact split of A (one piece stored three times)                   8  no epilogue stores
#   16  A always from the tile's chunk 0 (L2 hits instead of HBM traffic)  32  no barrier, no LDS writes in the loop
#   64  no LDS fragment reads in the loop                                  128  no A loads in the loop
#  256  only three of the six piece products (round 5: the most a three-product scheme could save)
#   build here (no GPU needed):  bash tools/experiments/x6_tune.sh build        run:  gpurun -- 'bash tools/experiments/x6_tune.sh run'
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
L=$ROOT/mask-yolo_amd/myolo/_lib
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DMM_X6_TUNE -c $ROOT/mask-yolo_amd/csrc/wino_mm.hip -o $L/wino_mm.tune.o || exit 1
  objs=""; for u in gemm_kernels bf16_kernels wino_kernels wino63_kernels mem_kernels exact_kernels comm_rccl; do objs="$objs $L/$u.o"; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libmyolo_hip_tune.so $objs $L/wino_mm.tune.o -ldl && echo built $L/libmyolo_hip_tune.so
  exit
fi
cd $ROOT
for t in 0 2 16 31 63 127 255 159 0; do
  MYOLO_LIB=$L/libmyolo_hip_tune.so KBENCH_OPTIONS=wino_x6=1,tune0=$t python tools/kbench.py wino63_mm --warm 30 --iters 30 2>&1 | tail -1 | sed "s/^/tune0=$t  /"
done
