#!/bin/bash
# PMC passes on the two bf16 inference kernels at the config-2/3 mask-head shape (tools/kbench.py conv3x3_bf16_fwd, deconv_mask_bf16_fwd):
# SQ counters and FETCH_SIZE / WRITE_SIZE, each set in its own run (with --kernel-trace only), effective clock from GRBM_GUI_ACTIVE.   gpurun -- 'bash tools/collect_pmc_bf16.sh r2'
#   -> gpurun_out/pmc_bf16_<tag>/<tag>_pmc_bf16.json
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_bf16_$TAG
mkdir -p $OUT
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"
for k in conv3x3_bf16_fwd deconv_mask_bf16_fwd; do
  for opt in "" "bf16_no_c3=1"; do
    [ "$k" = deconv_mask_bf16_fwd ] && [ -n "$opt" ] && continue
    d=$OUT/${k}_${opt:-default}
    rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d ${d}_1 -o p -- env KBENCH_OPTIONS=$opt python tools/kbench.py $k --iters 3 > /dev/null 2>&1
    rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d ${d}_2 -o p -- env KBENCH_OPTIONS=$opt python tools/kbench.py $k --iters 3 > /dev/null 2>&1
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d ${d}_3 -o p -- env KBENCH_OPTIONS=$opt python tools/kbench.py $k --iters 3 > /dev/null 2>&1
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d ${d}_4 -o p -- env KBENCH_OPTIONS=$opt python tools/kbench.py $k --iters 3 > /dev/null 2>&1
  done
done
python - "$OUT" "$TAG" <<'PY'
import csv, collections, glob, json, os, sys
out, tag = sys.argv[1], sys.argv[2]
res = {"note": "per launch; SQ_* counters summed over the chip, SQ_WAVE_CYCLES-class counters in units of 4 cycles; M = 921984 rows "
               "(4704 ROIs x 14 x 14), 256 channels; clock = GRBM_GUI_ACTIVE / 8 XCDs / duration"}
for d in sorted(glob.glob(out + "/*_1")):
    key = os.path.basename(d)[:-2]
    ent = {}
    for part in ("_1", "_2", "_3", "_4"):
        dd = d[:-2] + part
        try:
            rows = [r for r in csv.DictReader(open(dd + "/p_counter_collection.csv")) if "bf16_256" in r["Kernel_Name"]]
            kt = [r for r in csv.DictReader(open(dd + "/p_kernel_trace.csv")) if "bf16_256" in r["Kernel_Name"]]
        except Exception as e:
            ent["error" + part] = str(e); continue
        acc = collections.defaultdict(float); disp = set()
        for r in rows:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
        n = max(1, len(disp))
        ent["kernel"] = rows[0]["Kernel_Name"][:60] if rows else None
        ent["avg_ns" + part] = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt) / max(1, len(kt))
        for k, v in acc.items(): ent[k] = v / n
    if "GRBM_GUI_ACTIVE" in ent:
        cyc = ent["GRBM_GUI_ACTIVE"] / 8
        ent["effective_clock_GHz"] = cyc / ent["avg_ns_1"]
        ent["mfma_pipe_busy"] = ent["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
        ent["waves_parked_frac"] = ent["SQ_WAIT_ANY"] / ent["SQ_WAVE_CYCLES"]
        ent["waves_issue_stalled_frac"] = ent["SQ_WAIT_INST_ANY"] / ent["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in ent and "WRITE_SIZE" in ent:
        # KB -> bytes; gfx950 counts 128-byte read requests at 64 B (calibrated in tools/collect_pmc.sh against a kernel of known size)
        ent["hbm_traffic_bytes_corrected"] = 1024.0 * (2 * ent["FETCH_SIZE"] + ent["WRITE_SIZE"])
    res[key] = ent
json.dump(res, open("%s/%s_pmc_bf16.json" % (out, tag), "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
rm -rf $OUT/*_1 $OUT/*_2 $OUT/*_3 $OUT/*_4
