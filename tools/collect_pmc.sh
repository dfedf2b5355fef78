#!/bin/bash
# PMC passes (separate from any trace domain other than --kernel-trace, one counter set per pass) on the Winograd conv op
# at the config-2 mask-head shape (tools/kbench.py wino_fwd: input transform, the one-launch multiply, output transform) for both
# ways of forming the fp32 products (FP32_MATMUL native / bf16x6), plus a streaming kernel of known size to calibrate
# FETCH_SIZE / WRITE_SIZE.      gpurun -- 'bash tools/collect_pmc.sh r2'
#   -> gpurun_out/pmc_<tag>/<tag>_pmc_wino_multiply.json, <tag>_pmc_wino_multiply_x6.json
TAG=${1:-r2}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
for x6 in 0 1; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/wino${x6}_$c -o p -- env KBENCH_OPTIONS=wino_x6=$x6 python tools/kbench.py wino_fwd --iters 3 > /dev/null 2>&1
  done
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/wino${x6}_sq -o p -- env KBENCH_OPTIONS=wino_x6=$x6 python tools/kbench.py wino_fwd --iters 3 > /dev/null 2>&1
done
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal_$c -o p -- python tools/kbench.py roialign_fwd --iters 3 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/w63_$c -o p -- python tools/kbench.py wino63_fwd --iters 3 > /dev/null 2>&1
done
rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/w63_sq -o p -- python tools/kbench.py wino63_fwd --iters 3 > /dev/null 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, collections, json, sys
sys.path[:0] = [".", "mask-yolo_amd"]
out, tag = sys.argv[1], sys.argv[2]
def per_op(d, kname, launches_per_op=1):
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
cal = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v, ns, nops = per_op("cal_" + c, "crop_fwd")
    cal[c + "_KB_per_launch"] = v.get(c)
cal["true_bytes"] = {"written": NR * 196 * 256 * 4, "read_unique": 32 * 28 * 28 * 256 * 4}
for x6, mmk, fname in ((0, "wino_mm_kernel", "%s_pmc_wino_multiply.json"), (1, "wino_mm_x6_kernel", "%s_pmc_wino_multiply_x6.json")):
    w = {"op": "myolo_conv3x3_wino_fwd NR=4704 14x14 256->256 (tools/kbench.py wino_fwd, KBENCH_OPTIONS=wino_x6=%d), mixed tiling: %d "
               "point-tiles (484 per ROI)" % (x6, ptiles),
         "note": "the multiply stage is ONE launch of %s covering the 36 per-point GEMMs (grid 17788 workgroups of 128 x 256 outputs)" % mmk}
    for kname, key in ((mmk, "multiply"), ("wino_in_kernel", "input_transform"), ("wino_out_kernel", "output_transform")):
        e = {}
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            v, ns, nops = per_op("wino%d_%s" % (x6, c), kname)
            e[c + "_KB_per_op"] = v.get(c); e["avg_ns_per_op"] = ns; e["ops"] = nops
        e["traffic_bytes_per_launch_corrected"] = 1024.0 * (2 * e["FETCH_SIZE_KB_per_op"] + e["WRITE_SIZE_KB_per_op"])
        w[key] = e
    m = w["multiply"]
    m["algorithmic_bytes"] = float(ptiles) * 512 * 4 + 36 * 256 * 256 * (6 if x6 else 4)
    m["algorithmic_flop_fp32"] = 2.0 * ptiles * 256 * 256
    w["input_transform"]["algorithmic_bytes"] = NR * 196 * 256 * 4 + float(ptiles) * 256 * 4
    w["output_transform"]["algorithmic_bytes"] = NR * 196 * 256 * 4 + float(ptiles) * 256 * 4
    v, ns, nops = per_op("wino%d_sq" % x6, mmk)
    m["sq_per_op"] = v
    if "GRBM_GUI_ACTIVE" in v:
        cyc = v["GRBM_GUI_ACTIVE"] / 8
        m["effective_clock_GHz"] = cyc / ns
        m["mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
        # one 32x32x2 fp32 MFMA = 4096 flop in 64 cycles; one 32x32x16 bf16 MFMA = 32768 flop in 32 cycles, six per 16-deep fp32 block
        n_mfma = m["algorithmic_flop_fp32"] / 4096 if not x6 else 6 * m["algorithmic_flop_fp32"] / 32768
        m["mfma_busy_cycles_minimum"] = n_mfma * (64 if not x6 else 32)
    w["calibration_crop_fwd"] = cal
    w["traffic_bytes_per_launch_corrected"] = m["traffic_bytes_per_launch_corrected"]
    w["correction"] = "2 x FETCH_SIZE (gfx950 counts 128-B requests at 64 B; see calibration_crop_fwd) + WRITE_SIZE, KB -> bytes"
    json.dump(w, open(("%s/" + fname) % (out, tag), "w"), indent=1)
    print(json.dumps(m, indent=1)[:1200])
# the same multiply kernel on the F(6,3)/F(4,3) tiling (400 point-tiles per ROI, 64 planes in three runs)
p63 = 400 * NR
w = {"op": "myolo_wino63_* NR=4704 14x14 256->256 (tools/kbench.py wino63_fwd): %d point-tiles (400 per ROI)" % p63, "multiply": {}}
m = w["multiply"]
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v, ns, nops = per_op("w63_" + c, "wino_mm_kernel")
    m[c + "_KB_per_op"] = v.get(c); m["avg_ns_per_op"] = ns; m["ops"] = nops
m["traffic_bytes_per_launch_corrected"] = 1024.0 * (2 * m["FETCH_SIZE_KB_per_op"] + m["WRITE_SIZE_KB_per_op"])
m["algorithmic_bytes"] = float(p63) * 512 * 4 + 64 * 256 * 256 * 4
m["algorithmic_flop_fp32"] = 2.0 * p63 * 256 * 256
v, ns, nops = per_op("w63_sq", "wino_mm_kernel")
m["sq_per_op"] = v
if "GRBM_GUI_ACTIVE" in v:
    cyc = v["GRBM_GUI_ACTIVE"] / 8
    m["effective_clock_GHz"] = cyc / ns
    m["mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc
    m["mfma_busy_cycles_minimum"] = m["algorithmic_flop_fp32"] / 4096 * 64
for kname, key in (("wino63_boundary_kernel<1, 0>", "input_transform"), ("wino63_boundary_kernel<0, 1>", "output_transform")):
    e = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            v, ns, nops = per_op("w63_" + c, kname)
            e[c + "_KB_per_op"] = v.get(c); e["avg_ns_per_op"] = ns
        except Exception:
            pass
    w[key] = e
w["traffic_bytes_per_launch_corrected"] = m["traffic_bytes_per_launch_corrected"]
json.dump(w, open("%s/%s_pmc_wino63_multiply.json" % (out, tag), "w"), indent=1)
print(json.dumps(m, indent=1)[:900])
PY
rm -rf $OUT/wino0_* $OUT/wino1_* $OUT/cal_FETCH_SIZE $OUT/cal_WRITE_SIZE $OUT/w63_*
