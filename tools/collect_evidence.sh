#!/bin/bash
# Everything profiles/<tag>_* is made of, in one gpurun call:   gpurun --timeout 3000 -- 'bash tools/collect_evidence.sh r2g'
#   bench lines (default config -- FP32_MATMUL=bf16x6 since round 3 -- with variants, extras and the CPU baseline; FP32_MATMUL=native; rice416-bf16), rocprofv3 kernel stats / by-grid /
#   step timeline for both matmul modes, the per-kernel micro-benchmarks, the MFMA/VALU overlap microbenchmark.
TAG=${1:-r3}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/evidence_$TAG
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 2> $OUT/bench.err | tail -1 > $OUT/${TAG}_bench.json
python bench.py --steps 20 --warmup 5 --fp32-matmul native --cpu-images 0 --no-extras 2> $OUT/bench_native.err | tail -1 > $OUT/${TAG}_bench_native.json
python bench.py --config rice416-bf16 --steps 20 2> $OUT/bench_rice.err | tail -1 > $OUT/${TAG}_bench_rice416_bf16.json
bash tools/profile_step.sh $TAG > /dev/null 2>&1
cp gpurun_out/prof_$TAG/${TAG}_bench_kernel_stats.csv gpurun_out/prof_$TAG/${TAG}_bench_kernel_by_grid.csv gpurun_out/prof_$TAG/${TAG}_timeline.txt gpurun_out/prof_$TAG/${TAG}_step_sequence.txt $OUT/
bash tools/profile_step.sh ${TAG}native --fp32-matmul native > /dev/null 2>&1
cp gpurun_out/prof_${TAG}native/${TAG}native_bench_kernel_stats.csv gpurun_out/prof_${TAG}native/${TAG}native_bench_kernel_by_grid.csv gpurun_out/prof_${TAG}native/${TAG}native_timeline.txt $OUT/
{
  for k in wino_fwd wino63_fwd wino63_mm wino63_wgrad wino63_boundary wino_bwd_data wino_bwd_weight conv3x3_fwd deconv_mask_fwd roialign_fwd roialign_bwd dw; do
    python tools/kbench.py $k --warm 30 --iters 20 2>&1 | grep -vE "amdgpu.ids|^$" | tail -16      # steady state: 30 untimed launches first (clock transient after idle, r3_notes.md)
  done
  echo "--- KBENCH_OPTIONS=wino_x6=1"
  for k in wino_fwd wino63_fwd wino63_mm wino63_wgrad wino_bwd_data deconv_mask_fwd; do
    KBENCH_OPTIONS=wino_x6=1 python tools/kbench.py $k --warm 30 --iters 20 2>&1 | grep -vE "amdgpu.ids|^$" | tail -1
  done
  echo "--- bf16 inference kernels (default; bf16_no_c3=1 = the nine-fetch implicit GEMM; bf16_no256=1 = the 128^2 kernels; bf16_no_loopn=1)"
  python tools/kbench.py conv3x3_bf16_fwd --warm 30 --iters 20 2>&1 | tail -1
  KBENCH_OPTIONS=bf16_no_c3=1 python tools/kbench.py conv3x3_bf16_fwd --warm 30 --iters 20 2>&1 | tail -1
  KBENCH_OPTIONS=bf16_no256=1 python tools/kbench.py conv3x3_bf16_fwd --warm 30 --iters 20 2>&1 | tail -1
  python tools/kbench.py deconv_mask_bf16_fwd --warm 30 --iters 20 2>&1 | tail -1
  KBENCH_OPTIONS=bf16_no_loopn=1 python tools/kbench.py deconv_mask_bf16_fwd --warm 30 --iters 20 2>&1 | tail -1
  KBENCH_OPTIONS=bf16_no256=1 python tools/kbench.py deconv_mask_bf16_fwd --warm 30 --iters 20 2>&1 | tail -1
  echo "--- tools/overlap_mm_boundary.py"
  python tools/overlap_mm_boundary.py 2>&1 | tail -1
  KBENCH_OPTIONS=wino_x6=1 python tools/overlap_mm_boundary.py 2>&1 | tail -1
  echo "--- HBM stream copy (hand-written float4 kernel, grid sweep) beside torch copy_"
  python tools/kbench.py copy --iters 5 2>&1 | grep -v amdgpu
  echo "--- matrix-pipe ceiling (myolo_mfma_probe: register operands, no memory traffic)"
  python tools/kbench.py mfma --warm 5 2>&1 | grep -v amdgpu
  echo "--- wino_mm_x6_kernel as a plain GEMM at constant FLOPs over K (tools/experiments/x6_k_sweep.py)"
  python tools/experiments/x6_k_sweep.py 2>&1 | grep -v amdgpu
  echo "--- tools/pw_layers.py"
  python tools/pw_layers.py 2>&1 | grep -E "total"
} > $OUT/${TAG}_kbench.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -o /tmp/ovl tools/mfma_valu_overlap.hip 2>/dev/null && /tmp/ovl > $OUT/${TAG}_mfma_valu_overlap.txt 2>&1
head -c 400 $OUT/${TAG}_bench.json; echo; head -c 300 $OUT/${TAG}_bench_native.json; echo; head -c 300 $OUT/${TAG}_bench_rice416_bf16.json; echo
bash tools/profile_infer.sh $TAG > $OUT/infer_top.txt 2>&1
cp gpurun_out/prof_infer_$TAG/${TAG}_infer_kernel_stats.csv $OUT/
cat $OUT/${TAG}_timeline.txt | grep -E "wall" ; cat $OUT/${TAG}native_timeline.txt | grep -E "wall"; head -8 $OUT/infer_top.txt
