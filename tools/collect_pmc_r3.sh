#!/bin/bash
# PMC passes for the round-3 kernels (separate rocprofv3 --pmc runs with --kernel-trace only, one counter set per pass, as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes): the F(6,3)-tiling multiply with bf16x6 products (the dominant kernel of the
# default step), the bf16x6 weight-gradient product, the layer-boundary transform.   gpurun -- 'bash tools/collect_pmc_r3.sh r3'
#   -> gpurun_out/pmc_<tag>/<tag>_pmc_x6.json
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
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
        ("wino63_boundary", "wino63_boundary_kernel<0, 0>", "boundary_M_to_V", 2.0 * pe * 4, 0.0)):
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
