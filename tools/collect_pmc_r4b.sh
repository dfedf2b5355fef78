#!/bin/bash
# PMC passes for the kernels added late in round 4 (same method as tools/collect_pmc_r4.sh): the stride-2 depthwise data gradient, the register-fed thin pointwise
# forward, the BatchNorm backward launches.   gpurun -- 'bash tools/collect_pmc_r4b.sh'  -> gpurun_out/pmc_r4/*.json; then locally: python tools/merge_pmc_r4.py gpurun_out/pmc_r4 profiles/r4_pmc_trunk.json
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_r4
mkdir -p $OUT
python tools/pmc_kbench.py dw_bwd --match dw_bwd_data_s2 --out $OUT/r4_pmc_dw_bwd_data_s2.json > $OUT/r4_pmc_dw_bwd_data_s2.txt 2>&1
python tools/pmc_kbench.py pw_fused --match pw_fwd_thin --out $OUT/r4_pmc_pw_thin_fwd.json > $OUT/r4_pmc_pw_thin_fwd.txt 2>&1
python tools/pmc_kbench.py bn_bwd --match OpBnBwd --out $OUT/r4_pmc_bn_bwd_sums.json > $OUT/r4_pmc_bn_bwd_sums.txt 2>&1
python tools/pmc_kbench.py bn_bwd --match bn_bwd_dx --out $OUT/r4_pmc_bn_bwd_dx.json > $OUT/r4_pmc_bn_bwd_dx.txt 2>&1
for f in $OUT/r4_pmc_dw_bwd_data_s2.txt $OUT/r4_pmc_pw_thin_fwd.txt $OUT/r4_pmc_bn_bwd_sums.txt $OUT/r4_pmc_bn_bwd_dx.txt; do echo "== $f"; tail -12 $f | cut -c1-220; done
