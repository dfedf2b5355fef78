#!/bin/bash
# PMC passes (separate from any trace domain other than --kernel-trace, one counter set per pass) on the Winograd conv op
# at the config-2 mask-head shape (tools/kbench.py wino_fwd: input transform, the batched multiply launches, output transform),
# plus a streaming kernel of known size to calibrate FETCH_SIZE / WRITE_SIZE.   gpurun -- 'bash tools/collect_pmc.sh r2'
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/wino_$c -o p -- python tools/kbench.py wino_fwd --iters 3 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal_$c -o p -- python tools/kbench.py roialign_fwd --iters 3 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/wino_sq -o p -- python tools/kbench.py wino_fwd --iters 3 > /dev/null 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, collections, json, sys
sys.path[:0] = [".", "mask-yolo_amd"]
out, tag = sys.argv[1], sys.argv[2]
def per_op(d, kname, launches_per_op):
    """counter totals and duration of `kname` per OP (an op = launches_per_op consecutive launches of that symbol)"""
    rows = [r for r in csv.DictReader(open("%s/%s/p_counter_collection.csv" % (out, d))) if kname in r["Kernel_Name"]]
    acc = collections.defaultdict(float); disp = set()
    for r in rows:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    kt = [r for r in csv.DictReader(open("%s/%s/p_kernel_trace.csv" % (out, d))) if kname in r["Kernel_Name"]]
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt)
    nops = max(1, len(disp) // launches_per_op)
    return {k: v / nops for k, v in acc.items()}, dur / max(1, len(kt) // launches_per_op), nops
NR, C = 4704, 256
ptiles = 484 * NR                                      # mixed F(4,3)/F(2,3) tiling at 14x14 (myolo_wino_plane_elems)
w = {"op": "myolo_conv3x3_wino_fwd NR=4704 14x14 256->256 (tools/kbench.py wino_fwd), mixed tiling: %d point-tiles (484 per ROI)" % ptiles,
     "note": "the multiply stage is THREE batched launches of gemm_nn_fast<PLAIN> per op (point groups with 16, 12 and 9 tiles per ROI: "
             "grids 301056x1x16, 225792x1x16, 169472x1x4 threads); counters and durations below are summed over the three"}
for kname, key, n in (("gemm_nn_fast", "multiply", 3), ("wino_in_kernel", "input_transform", 1), ("wino_out_kernel", "output_transform", 1)):
    e = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v, ns, nops = per_op("wino_" + c, kname, n)
        e[c + "_KB_per_op"] = v.get(c); e["avg_ns_per_op"] = ns; e["ops"] = nops
    e["traffic_bytes_per_launch_corrected"] = 1024.0 * (2 * e["FETCH_SIZE_KB_per_op"] + e["WRITE_SIZE_KB_per_op"])
    w[key] = e
w["multiply"]["algorithmic_bytes"] = float(ptiles) * 512 * 4 + 36 * 256 * 256 * 4
w["multiply"]["algorithmic_flop"] = 2.0 * ptiles * 256 * 256
w["input_transform"]["algorithmic_bytes"] = NR * 196 * 256 * 4 + float(ptiles) * 256 * 4
w["output_transform"]["algorithmic_bytes"] = NR * 196 * 256 * 4 + float(ptiles) * 256 * 4
v, ns, nops = per_op("wino_sq", "gemm_nn_fast", 3)
w["multiply"]["sq_per_op"] = v
if "GRBM_GUI_ACTIVE" in v:
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    w["multiply"]["effective_clock_GHz"] = cyc / ns
    w["multiply"]["mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
    w["multiply"]["mfma_busy_cycles_minimum"] = w["multiply"]["algorithmic_flop"] / (2 * 32 * 32 * 2) * 64      # 64 cycles per 32x32x2 MFMA
cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v, ns, nops = per_op("cal_" + c, "crop_fwd", 1)
    cal[c + "_KB_per_launch"] = v.get(c)
cal["true_bytes"] = {"written": NR * 196 * 256 * 4, "read_unique": 32 * 28 * 28 * 256 * 4}
w["calibration_crop_fwd"] = cal
w["traffic_bytes_per_launch_corrected"] = w["multiply"]["traffic_bytes_per_launch_corrected"]
w["correction"] = "2 x FETCH_SIZE (gfx950 counts 128-B requests at 64 B; see calibration_crop_fwd) + WRITE_SIZE, KB -> bytes"
json.dump(w, open("%s/%s_pmc_wino_multiply.json" % (out, tag), "w"), indent=1)
print(json.dumps(w, indent=1)[:1800])
PY
rm -rf $OUT/wino_FETCH_SIZE $OUT/wino_WRITE_SIZE $OUT/cal_FETCH_SIZE $OUT/cal_WRITE_SIZE $OUT/wino_sq
