#!/bin/bash
# PMC passes for the trunk / ROIAlign kernels north_star names for the HBM target: separate rocprofv3 --pmc runs
# (FETCH_SIZE | WRITE_SIZE | SQ set, --kernel-trace only) on tools/kbench.py targets whose inputs are rotated past the Infinity Cache.
#   gpurun -- 'bash tools/collect_pmc_r4.sh'   -> gpurun_out/pmc_r4/*.json, *.txt
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_r4
mkdir -p $OUT
python tools/pmc_kbench.py dw_fused --match dw_rows_kernel --out $OUT/r4_pmc_dw_fwd.json > $OUT/r4_pmc_dw_fwd.txt 2>&1
python tools/pmc_kbench.py dw_fused --match dw_fwd_kernel --opts dw_legacy=1 --out $OUT/r4_pmc_dw_fwd_round3_kernel.json > $OUT/r4_pmc_dw_fwd_round3_kernel.txt 2>&1
python tools/pmc_kbench.py dw_bwd --match dw_bwd_data_kernel --out $OUT/r4_pmc_dw_bwd_data.json > $OUT/r4_pmc_dw_bwd_data.txt 2>&1
python tools/pmc_kbench.py pw_fused --match gemm_nn_fast --out $OUT/r4_pmc_pw_gemm.json > $OUT/r4_pmc_pw_gemm.txt 2>&1
python tools/pmc_kbench.py pw_fused --match wino_mm_x6_kernel --opts wino_x6=1 --out $OUT/r4_pmc_pw_x6.json > $OUT/r4_pmc_pw_x6.txt 2>&1
python tools/pmc_kbench.py dw_bwd --match dw_rows_kernel --out $OUT/r4_pmc_dw_bwd_data_s1.json > $OUT/r4_pmc_dw_bwd_data_s1.txt 2>&1
python tools/pmc_kbench.py dw_bwd --match dw_rows_wgrad_kernel --out $OUT/r4_pmc_dw_wgrad.json > $OUT/r4_pmc_dw_wgrad.txt 2>&1
python tools/pmc_kbench.py dw_bwd --match dw_wgrad_kernel --opts dw_bwd_legacy=1 --out $OUT/r4_pmc_dw_wgrad_round3_kernel.json > $OUT/r4_pmc_dw_wgrad_round3_kernel.txt 2>&1
python tools/pmc_kbench.py dw_bwd --match dw_bwd_data_kernel --opts dw_bwd_legacy=1 --out $OUT/r4_pmc_dw_bwd_data_round3_kernel.json > $OUT/r4_pmc_dw_bwd_data_round3_kernel.txt 2>&1
KBENCH_OPTIONS= python tools/pmc_kbench.py roialign_bwd --match crop_bwd --opts tune0=1 --out $OUT/r4_pmc_crop_bwd_round3_order.json > $OUT/r4_pmc_crop_bwd_round3_order.txt 2>&1
python tools/pmc_kbench.py roialign_fwd --match crop_fwd --out $OUT/r4_pmc_crop_fwd.json > $OUT/r4_pmc_crop_fwd.txt 2>&1
python tools/pmc_kbench.py roialign_bwd --match crop_bwd --out $OUT/r4_pmc_crop_bwd.json > $OUT/r4_pmc_crop_bwd.txt 2>&1
python tools/merge_pmc_r4.py $OUT profiles/r4_pmc_trunk.json
for f in $OUT/*.txt; do echo "== $f"; cat $f; done
