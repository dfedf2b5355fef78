#!/bin/bash
# Everything profiles/<tag>_* is made of, in one gpurun call:   gpurun --timeout 3000 -- 'bash tools/collect_evidence.sh r2g'
#   bench lines (default config with variants + CPU baseline; FP32_MATMUL=bf16x6; rice416-bf16), rocprofv3 kernel stats / by-grid /
#   step timeline for both matmul modes, the per-kernel micro-benchmarks, the MFMA/VALU overlap microbenchmark.
TAG=${1:-r2g}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/evidence_$TAG
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/${TAG}_bench.json
python bench.py --steps 20 --warmup 5 --fp32-matmul bf16x6 --cpu-images 0 2> $OUT/bench_x6.err | tail -1 > $OUT/${TAG}_bench_bf16x6.json
python bench.py --config rice416-bf16 --steps 20 2> $OUT/bench_rice.err | tail -1 > $OUT/${TAG}_bench_rice416_bf16.json
bash tools/profile_step.sh $TAG > /dev/null 2>&1
cp gpurun_out/prof_$TAG/${TAG}_bench_kernel_stats.csv gpurun_out/prof_$TAG/${TAG}_bench_kernel_by_grid.csv gpurun_out/prof_$TAG/${TAG}_timeline.txt $OUT/
bash tools/profile_step.sh ${TAG}x6 --fp32-matmul bf16x6 > /dev/null 2>&1
cp gpurun_out/prof_${TAG}x6/${TAG}x6_bench_kernel_stats.csv gpurun_out/prof_${TAG}x6/${TAG}x6_bench_kernel_by_grid.csv gpurun_out/prof_${TAG}x6/${TAG}x6_timeline.txt $OUT/
{
  for k in wino_fwd wino63_fwd wino_bwd_data wino_bwd_weight conv3x3_fwd deconv_mask_fwd roialign_fwd roialign_bwd dw; do
    python tools/kbench.py $k --iters 10 2>&1 | grep -vE "amdgpu.ids|^$" | tail -16
  done
  echo "--- KBENCH_OPTIONS=wino_x6=1"
  for k in wino_fwd wino63_fwd wino_bwd_data deconv_mask_fwd; do
    KBENCH_OPTIONS=wino_x6=1 python tools/kbench.py $k --iters 10 2>&1 | grep -vE "amdgpu.ids|^$" | tail -1
  done
  echo "--- bf16 inference kernels (default; bf16_no_c3=1 = the nine-fetch implicit GEMM; bf16_no256=1 = the 128^2 kernels; bf16_no_loopn=1)"
  python tools/kbench.py conv3x3_bf16_fwd --iters 20 2>&1 | tail -1
  KBENCH_OPTIONS=bf16_no_c3=1 python tools/kbench.py conv3x3_bf16_fwd --iters 20 2>&1 | tail -1
  KBENCH_OPTIONS=bf16_no256=1 python tools/kbench.py conv3x3_bf16_fwd --iters 20 2>&1 | tail -1
  python tools/kbench.py deconv_mask_bf16_fwd --iters 20 2>&1 | tail -1
  KBENCH_OPTIONS=bf16_no_loopn=1 python tools/kbench.py deconv_mask_bf16_fwd --iters 20 2>&1 | tail -1
  KBENCH_OPTIONS=bf16_no256=1 python tools/kbench.py deconv_mask_bf16_fwd --iters 20 2>&1 | tail -1
  echo "--- tools/overlap_mm_boundary.py"
  python tools/overlap_mm_boundary.py 2>&1 | tail -1
  echo "--- tools/pw_layers.py"
  python tools/pw_layers.py 2>&1 | grep -E "total"
} > $OUT/${TAG}_kbench.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o /tmp/ovl tools/mfma_valu_overlap.hip 2>/dev/null && /tmp/ovl > $OUT/${TAG}_mfma_valu_overlap.txt 2>&1
head -c 400 $OUT/${TAG}_bench.json; echo; head -c 300 $OUT/${TAG}_bench_bf16x6.json; echo; head -c 300 $OUT/${TAG}_bench_rice416_bf16.json; echo
bash tools/profile_infer.sh $TAG > $OUT/infer_top.txt 2>&1
cp gpurun_out/prof_infer_$TAG/${TAG}_infer_kernel_stats.csv $OUT/
cat $OUT/${TAG}_timeline.txt | grep -E "wall" ; cat $OUT/${TAG}x6_timeline.txt | grep -E "wall"; head -8 $OUT/infer_top.txt
