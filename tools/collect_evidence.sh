#!/bin/bash
# The evidence set of a round in one gpurun call:   gpurun --timeout 3000 -- 'bash tools/collect_evidence.sh r5'   (-> gpurun_out/evidence_<tag>/, copy into profiles/)
#   the full default bench line, the step profile (default and 20 forced positives), the kernel micro-benchmarks on cold inputs.
TAG=${1:-r6}
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/evidence_$TAG
mkdir -p $OUT
python bench.py --steps 20 --warmup 5 --detail-name ${TAG}_bench_detail.json 2> $OUT/bench.err | tail -1 > $OUT/${TAG}_bench.json      # the ONE compact line the driver parses
cp gpurun_out/${TAG}_bench_detail.json $OUT/${TAG}_bench_detail.json          # the full object (per-layer table, probes, variants)
rm -f ${TAG}_bench_detail.json
bash tools/profile_step.sh $TAG --steps 10 > /dev/null 2>&1
cp gpurun_out/prof_$TAG/${TAG}_bench_kernel_stats.csv gpurun_out/prof_$TAG/${TAG}_bench_kernel_by_grid.csv gpurun_out/prof_$TAG/${TAG}_timeline.txt gpurun_out/prof_$TAG/${TAG}_step_sequence.txt $OUT/
bash tools/profile_step.sh ${TAG}_pos20 --steps 10 --force-pos 20 > /dev/null 2>&1
cp gpurun_out/prof_${TAG}_pos20/${TAG}_pos20_bench_kernel_stats.csv gpurun_out/prof_${TAG}_pos20/${TAG}_pos20_bench_kernel_by_grid.csv gpurun_out/prof_${TAG}_pos20/${TAG}_pos20_timeline.txt gpurun_out/prof_${TAG}_pos20/${TAG}_pos20_step_sequence.txt $OUT/
{
  echo "--- depthwise forward as the step runs it (cold inputs); then with the round-3 kernel (dw_legacy=1)"
  python tools/kbench.py dw_fused --iters 20 2>&1 | grep -v amdgpu
  KBENCH_OPTIONS=dw_legacy=1 python tools/kbench.py dw_fused --iters 20 2>&1 | grep -v amdgpu | tail -1
  echo "--- depthwise data / weight gradients (cold inputs); then with the round-3 kernels (dw_bwd_legacy=1)"
  python tools/kbench.py dw_bwd --iters 20 2>&1 | grep -v amdgpu
  KBENCH_OPTIONS=dw_bwd_legacy=1 python tools/kbench.py dw_bwd --iters 20 2>&1 | grep -v amdgpu | grep total
  echo "--- BatchNorm + ReLU6 backward of the trunk layers (three launches: sums, finish, dx; cold inputs; in the step the sums of 14 of the 29 come from the depthwise data gradient's epilogue, round 5)"
  python tools/kbench.py bn_bwd --iters 20 2>&1 | grep -v amdgpu
  echo "--- pointwise layers (fp32 MFMA kernels; then wino_x6=1 = the product's FP32_MATMUL=bf16x6)"
  python tools/kbench.py pw_fused --iters 20 2>&1 | grep -v amdgpu
  KBENCH_OPTIONS=wino_x6=1 python tools/kbench.py pw_fused --iters 20 2>&1 | grep -v amdgpu
  echo "--- ROIAlign forward / backward (backward: default, tune0=1 = round-3 pixel order, tune0=4 = 4x4 tiles)"
  python tools/kbench.py roialign_fwd --iters 20 --warm 10 2>&1 | grep -v amdgpu
  python tools/kbench.py roialign_bwd --iters 20 --warm 10 2>&1 | grep -v amdgpu
  KBENCH_OPTIONS=tune0=1 python tools/kbench.py roialign_bwd --iters 20 --warm 10 2>&1 | grep -v amdgpu
  KBENCH_OPTIONS=tune0=4 python tools/kbench.py roialign_bwd --iters 20 --warm 10 2>&1 | grep -v amdgpu
  echo "--- Winograd kernels, steady state (boundary: round 6 persistent kernel; then the round-5 kernel, w63_legacy=1)"
  for k in wino63_mm wino63_wgrad wino63_boundary wino63_lazy; do KBENCH_OPTIONS=wino_x6=1 python tools/kbench.py $k --warm 30 --iters 20 2>&1 | grep -vE "amdgpu.ids|^$" | tail -3; done
  for k in wino63_boundary wino63_lazy; do KBENCH_OPTIONS=wino_x6=1,w63_legacy=1 python tools/kbench.py $k --warm 30 --iters 20 2>&1 | grep -vE "amdgpu.ids|^$" | tail -3; done
  echo "--- deconv + ReLU + 1x1 mask conv, training (bf16x6): round 6 = transposed tile, packed channel sums, in-kernel finish; then deconv_mask_legacy=2 (partials + finish launch) and =1 (the round-5 butterfly epilogue)"
  for o in wino_x6=1 wino_x6=1,deconv_mask_legacy=2 wino_x6=1,deconv_mask_legacy=1 wino_x6=1; do KBENCH_OPTIONS=$o python tools/kbench.py deconv_mask_fwd --warm 30 --iters 20 2>&1 | grep -vE "amdgpu.ids|^$" | tail -1 | sed "s/^/$o  /"; done
  echo "--- the same op of the bf16 inference path at the Rice-416 shape (3380 boxes): round 6 = 1x1 conv on the matrix pipe + in-kernel finish; then bf16_mask_nofin=1 (partials + finish) and bf16_mask_valu=1 (the round-3 VALU epilogue)"
  for o in tune0=0 bf16_mask_nofin=1 bf16_mask_valu=1 tune0=0; do KBENCH_OPTIONS=$o python tools/kbench.py deconv_mask_bf16_fwd --rois 3380 --warm 30 --iters 20 2>&1 | grep deconv_mask | sed "s/^/$o  /"; done
  echo "--- what a plain copy of the boundary kernels' plane sets moves (tools/experiments/vecwidth: persistent nine-wave workgroups, 64 strided planes, 4 / 8 / 16 bytes per lane)"
  hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o /tmp/vecwidth tools/experiments/vecwidth/vecwidth.hip > /dev/null 2>&1 && /tmp/vecwidth | head -13
  echo "--- residency census (tools/experiments/census: workgroups of T threads / V VGPRs / L bytes of LDS a CU holds at once; API = hipOccupancyMaxActiveBlocksPerMultiprocessor)"
  hipcc --offload-arch=gfx950 -O2 -Wno-unused-result -o /tmp/census tools/experiments/census/census.hip > /dev/null 2>&1 && /tmp/census | grep -E "lds  50176|lds   1024"
  echo "--- HBM stream copy"
  python tools/kbench.py copy --iters 5 2>&1 | grep -v amdgpu | head -8
  echo "--- stream / process-group experiment (tools/experiments/pg_stream_cost.py): default = high-priority side streams; MYOLO_STREAM_PRIORITY=0 = rounds 1-3"
  for m in none pg_first pg; do HSA_ENABLE_IPC_MODE_LEGACY=0 python tools/experiments/pg_stream_cost.py $m 2>&1 | grep "ms per step"; done
  for m in none pg_first pg; do MYOLO_STREAM_PRIORITY=0 HSA_ENABLE_IPC_MODE_LEGACY=0 python tools/experiments/pg_stream_cost.py $m 2>&1 | grep "ms per step"; done
} > $OUT/${TAG}_kbench.txt
head -c 600 $OUT/${TAG}_bench.json; echo; grep wall $OUT/${TAG}_timeline.txt; grep wall $OUT/${TAG}_pos20_timeline.txt; tail -8 $OUT/${TAG}_kbench.txt
