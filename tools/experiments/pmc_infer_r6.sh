#!/bin/bash
# counter passes (separate FETCH_SIZE / WRITE_SIZE runs, tools/pmc_traffic.py) of the kernels the round's last session added: the bf16 ROIAlign as a column walk against the
# four-corner form, and pw_smallm_kernel against the split-K pair's GEMM.   gpurun -- 'bash tools/experiments/pmc_infer_r6.sh' -> gpurun_out/r6_pmc_infer.json
cd "$GRAFT_REPO_ROOT"
{
  python tools/kbench.py roialign_bf16_fwd --warm 30 --iters 20 2>&1 | grep -v amdgpu | sed 's/^/# /'
  KBENCH_OPTIONS=crop_bf16_legacy=1 python tools/kbench.py roialign_bf16_fwd --warm 30 --iters 20 2>&1 | grep -v amdgpu | sed 's/^/# legacy: /'
  python tools/kbench.py pw_smallm --warm 30 --iters 20 2>&1 | grep -v amdgpu | sed 's/^/# /'
  KBENCH_OPTIONS=pw_no_smallm=1 python tools/kbench.py pw_smallm --warm 30 --iters 20 2>&1 | grep -v amdgpu | sed 's/^/# split-K pair: /'
} > gpurun_out/r6_pmc_infer.txt
{
  python tools/pmc_traffic.py roialign_bf16_fwd crop_fwd_bf16_walk_kernel
  python tools/pmc_traffic.py roialign_bf16_fwd crop_fwd_bf16_kernel crop_bf16_legacy=1
  python tools/pmc_traffic.py pw_smallm pw_smallm_kernel
  python tools/pmc_traffic.py pw_smallm gemm_nn_fast pw_no_smallm=1
  python tools/pmc_traffic.py pw_smallm splitk_epilogue pw_no_smallm=1
} > gpurun_out/r6_pmc_infer.jsonl
cat gpurun_out/r6_pmc_infer.txt gpurun_out/r6_pmc_infer.jsonl
