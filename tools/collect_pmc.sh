#!/bin/bash
# rocprofv3 --pmc passes, one script for every kernel set (rounds 1-4 had one script per round).  Counters are ALWAYS collected in their own runs with
# --kernel-trace only (never with sys / hip / memory-copy traces), one counter set per pass, FETCH_SIZE and WRITE_SIZE in separate passes, traffic =
# 2 x FETCH_SIZE + WRITE_SIZE in KB x 1024 (gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md).
#     gpurun --timeout 2400 -- 'bash tools/collect_pmc.sh <set> [tag]'      sets:
#       wino       round 2: the Winograd conv op at the config-2 mask-head shape (wino_fwd), native and bf16x6 products, + a calibration copy -> <tag>_pmc_wino_multiply{,_x6}.json
#       bf16       round 2: the two bf16 inference kernels (conv3x3_bf16_fwd, deconv_mask_bf16_fwd), SQ + FETCH/WRITE -> <tag>_pmc_bf16.json
#       x6         round 3: F(6,3)-tiling multiply / weight gradient / layer boundary with bf16x6 products, steady state -> <tag>_pmc_x6.json
#       trunk      round 4: trunk and ROIAlign kernel families (depthwise fwd / data / weight gradient, pointwise, crop fwd / bwd) -> profiles/<tag>_pmc_trunk.json via tools/merge_pmc.py
#       trunk_late round 4: stride-2 depthwise data gradient, thin pointwise forward, BatchNorm backward launches
SET=${1:?usage: collect_pmc.sh <set> [tag]}
TAG=${2:-r5}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"

pmc_wino() {
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
for kname, key in (("wino63_boundary_kernel<1, 0", "input_transform"), ("wino63_boundary_kernel<0, 1", "output_transform")):
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
}

pmc_bf16() {
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
}

pmc_x6() {
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
for k in wino63_mm wino63_wgrad wino63_boundary; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${k}_$c -o p -- env KBENCH_OPTIONS=wino_x6=1 python tools/kbench.py $k --warm 30 --iters 6 > /dev/null 2>&1
  done
  rocprofv3 --pmc $SQ --kernel-trace --output-format csv -d $OUT/${k}_sq -o p -- env KBENCH_OPTIONS=wino_x6=1 python tools/kbench.py $k --warm 30 --iters 6 > /dev/null 2>&1
done
python - "$OUT" "$TAG" <<'PY'
import csv, collections, json, sys
out, tag = sys.argv[1], sys.argv[2]
def per_launch(d, kname):
    # steady state only: the LAST five launches of the kernel (the first ~25 after idle run through a clock transient)
    kt = [r for r in csv.DictReader(open("%s/%s/p_kernel_trace.csv" % (out, d))) if kname in r["Kernel_Name"]]
    kt = sorted(kt, key=lambda r: int(r["Start_Timestamp"]))[-5:]
    keep = set(r["Dispatch_Id"] for r in kt)
    rows = [r for r in csv.DictReader(open("%s/%s/p_counter_collection.csv" % (out, d))) if kname in r["Kernel_Name"] and r["Dispatch_Id"] in keep]
    acc = collections.defaultdict(float); disp = set()
    for r in rows:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    dur = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt)
    n = max(1, len(disp))
    return {k: v / n for k, v in acc.items()}, dur / max(1, len(kt)), n
NR, C = 4704, 256
pe = 400 * NR * C
res = {"shape": "NR = 4704 ROIs (32 x 147), 14x14, 256 -> 256 channels, F(6,3)/F(4,3) tiling: 400 point-tiles per ROI, 64 planes; KBENCH_OPTIONS=wino_x6=1",
       "steady_state": "30 untimed launches first; counters and durations of the last five launches only (round 3: the earlier passes measured launches 3-5 after idle, inside the clock transient)",
       "method": "separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ set), --kernel-trace only; traffic = 2 x FETCH_SIZE + WRITE_SIZE "
                 "(gfx950: FETCH_SIZE counts half of a wide streaming read, MI355X_MICROARCH.md), KB -> bytes x 1024"}
for target, kname, key, alg_bytes, alg_flop in (
        ("wino63_mm", "wino_mm_x6_kernel", "multiply_x6", 2.0 * pe * 4 + 64 * C * C * 6, 2.0 * 400 * NR * C * C),
        ("wino63_wgrad", "wino_tn_x6_kernel", "weight_gradient_x6", 2.0 * pe * 4, 2.0 * 400 * NR * C * C),
        ("wino63_boundary", "wino63_boundary_kernel<0, 0", "boundary_M_to_V", 2.0 * pe * 4, 0.0)):
    e = {"kernel": kname, "kbench": target, "algorithmic_bytes": alg_bytes, "algorithmic_flop_fp32": alg_flop}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        try:
            v, ns, n = per_launch("%s_%s" % (target, c), kname)
            e[c + "_KB_per_launch"] = v.get(c); e["avg_ns_per_launch"] = ns; e["launches"] = n
        except Exception as ex:
            e[c + "_error"] = str(ex)
    if "FETCH_SIZE_KB_per_launch" in e and "WRITE_SIZE_KB_per_launch" in e and e["FETCH_SIZE_KB_per_launch"] is not None:
        e["traffic_bytes_per_launch_corrected"] = 1024.0 * (2 * e["FETCH_SIZE_KB_per_launch"] + e["WRITE_SIZE_KB_per_launch"])
        e["traffic_over_algorithmic"] = e["traffic_bytes_per_launch_corrected"] / alg_bytes
    try:
        v, ns, n = per_launch("%s_sq" % target, kname)
        e["sq"] = v; e["sq_avg_ns_per_launch"] = ns
        if v.get("GRBM_GUI_ACTIVE") and ns:
            cyc = v["GRBM_GUI_ACTIVE"] / 8                      # the counter sums the 8 XCDs
            e["effective_clock_ghz"] = cyc / ns
            e["mfma_pipe_busy"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024 / cyc      # 256 CUs x 4 SIMDs
            if alg_flop:          # six 32x32x16 bf16 MFMAs (32768 flop, 32 cycles each) per 16-deep fp32 block of 32x32
                e["mfma_busy_cycles_minimum"] = 6 * alg_flop / 32768 * 32
    except Exception as ex:
        e["sq_error"] = str(ex)
    res[key] = e
json.dump(res, open("%s/%s_pmc_x6.json" % (out, tag), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "sq"} if isinstance(v, dict) else v for k, v in res.items()}, indent=1)[:3000])
PY
find $OUT -name "*.csv" -size +20M -delete
}

pmc_trunk() {
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
python tools/pmc_kbench.py dw_fused --match dw_rows_kernel --out $OUT/${TAG}_pmc_dw_fwd.json > $OUT/${TAG}_pmc_dw_fwd.txt 2>&1
python tools/pmc_kbench.py dw_fused --match dw_fwd_kernel --opts dw_legacy=1 --out $OUT/${TAG}_pmc_dw_fwd_round3_kernel.json > $OUT/${TAG}_pmc_dw_fwd_round3_kernel.txt 2>&1
python tools/pmc_kbench.py dw_bwd --match dw_bwd_data_kernel --out $OUT/${TAG}_pmc_dw_bwd_data.json > $OUT/${TAG}_pmc_dw_bwd_data.txt 2>&1
python tools/pmc_kbench.py pw_fused --match gemm_nn_fast --out $OUT/${TAG}_pmc_pw_gemm.json > $OUT/${TAG}_pmc_pw_gemm.txt 2>&1
python tools/pmc_kbench.py pw_fused --match wino_mm_x6_kernel --opts wino_x6=1 --out $OUT/${TAG}_pmc_pw_x6.json > $OUT/${TAG}_pmc_pw_x6.txt 2>&1
python tools/pmc_kbench.py dw_bwd --match dw_rows_kernel --out $OUT/${TAG}_pmc_dw_bwd_data_s1.json > $OUT/${TAG}_pmc_dw_bwd_data_s1.txt 2>&1
python tools/pmc_kbench.py dw_bwd --match dw_rows_wgrad_kernel --out $OUT/${TAG}_pmc_dw_wgrad.json > $OUT/${TAG}_pmc_dw_wgrad.txt 2>&1
python tools/pmc_kbench.py dw_bwd --match dw_wgrad_kernel --opts dw_bwd_legacy=1 --out $OUT/${TAG}_pmc_dw_wgrad_round3_kernel.json > $OUT/${TAG}_pmc_dw_wgrad_round3_kernel.txt 2>&1
python tools/pmc_kbench.py dw_bwd --match dw_bwd_data_kernel --opts dw_bwd_legacy=1 --out $OUT/${TAG}_pmc_dw_bwd_data_round3_kernel.json > $OUT/${TAG}_pmc_dw_bwd_data_round3_kernel.txt 2>&1
KBENCH_OPTIONS= python tools/pmc_kbench.py roialign_bwd --match crop_bwd --opts tune0=1 --out $OUT/${TAG}_pmc_crop_bwd_round3_order.json > $OUT/${TAG}_pmc_crop_bwd_round3_order.txt 2>&1
python tools/pmc_kbench.py roialign_fwd --match crop_fwd --out $OUT/${TAG}_pmc_crop_fwd.json > $OUT/${TAG}_pmc_crop_fwd.txt 2>&1
python tools/pmc_kbench.py roialign_bwd --match crop_bwd --out $OUT/${TAG}_pmc_crop_bwd.json > $OUT/${TAG}_pmc_crop_bwd.txt 2>&1
python tools/merge_pmc.py $OUT gpurun_out/${TAG}_pmc_trunk.json   # (copy into profiles/ afterwards: only gpurun_out/ comes back from the GPU box)
for f in $OUT/*.txt; do echo "== $f"; cat $f; done
}

pmc_trunk_late() {
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
python tools/pmc_kbench.py dw_bwd --match dw_bwd_data_s2 --out $OUT/${TAG}_pmc_dw_bwd_data_s2.json > $OUT/${TAG}_pmc_dw_bwd_data_s2.txt 2>&1
python tools/pmc_kbench.py pw_fused --match pw_fwd_thin --out $OUT/${TAG}_pmc_pw_thin_fwd.json > $OUT/${TAG}_pmc_pw_thin_fwd.txt 2>&1
python tools/pmc_kbench.py bn_bwd --match OpBnBwd --out $OUT/${TAG}_pmc_bn_bwd_sums.json > $OUT/${TAG}_pmc_bn_bwd_sums.txt 2>&1
python tools/pmc_kbench.py bn_bwd --match bn_bwd_dx --out $OUT/${TAG}_pmc_bn_bwd_dx.json > $OUT/${TAG}_pmc_bn_bwd_dx.txt 2>&1
for f in $OUT/${TAG}_pmc_dw_bwd_data_s2.txt $OUT/${TAG}_pmc_pw_thin_fwd.txt $OUT/${TAG}_pmc_bn_bwd_sums.txt $OUT/${TAG}_pmc_bn_bwd_dx.txt; do echo "== $f"; tail -12 $f | cut -c1-220; done
}

case "$SET" in
  wino) pmc_wino ;;
  bf16) pmc_bf16 ;;
  x6) pmc_x6 ;;
  trunk) pmc_trunk ;;
  trunk_late) pmc_trunk_late ;;
  *) echo "unknown set $SET"; exit 2 ;;
esac
