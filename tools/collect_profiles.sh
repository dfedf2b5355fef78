#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ (copy the summaries into profiles/ afterwards).
#   gpurun -- 'bash tools/collect_profiles.sh r1'
# Counters are collected in their own passes (never combined with sys/hip traces), as MI355X_MICROARCH.md prescribes:
# FETCH_SIZE and WRITE_SIZE cannot share a pass (TCC slots).
TAG=${1:-r1}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
# 1. the bench command itself: per-kernel time
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -o bench -- python bench.py --steps 5 --warmup 2 --cpu-images 0 > $OUT/bench_under_rocprof.log 2>&1
cp $OUT/bench/bench_kernel_stats.csv $OUT/${TAG}_bench_kernel_stats.csv
# per (kernel, grid) averages from the same trace: the stats file pools every launch of a symbol, this one
# isolates e.g. the four mask-head conv forwards (grid 14406 x 256) that bench.py times live with HIP events
python - "$OUT/bench/bench_kernel_trace.csv" "$OUT/${TAG}_bench_kernel_by_grid.csv" <<'PY'
import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(list)
for r in rows:
    grid = "x".join(r[k] for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
    wg = "x".join(r[k] for k in ("Workgroup_Size_X", "Workgroup_Size_Y", "Workgroup_Size_Z"))
    acc[(r["Kernel_Name"], grid, wg)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
out = sorted(((sum(v), k, v) for k, v in acc.items()), reverse=True)
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["Kernel_Name", "Grid_Size(threads)", "Workgroup_Size", "Calls", "TotalNs", "AverageNs", "MinNs", "MaxNs"])
    for tot, k, v in out[:60]:
        w.writerow([k[0], k[1], k[2], len(v), tot, tot / len(v), min(v), max(v)])
PY
# 2. un-profiled bench line (the number of record)
python bench.py --steps 10 --warmup 3 2> /dev/null | tail -1 > $OUT/${TAG}_bench.json
# 3. single-kernel micro-benchmarks
for k in wino_fwd wino_bwd_data wino_bwd_weight conv3x3_fwd conv3x3_bwd_data conv3x3_bwd_weight deconv_fwd roialign_fwd roialign_bwd conv3x3_bf16_fwd deconv_bf16_fwd dw; do python tools/kbench.py $k --iters 20 2>&1 | grep -v amdgpu.ids; done > $OUT/${TAG}_kbench.txt
# 4. PMC passes on the dominant kernel (+ a pure streaming kernel to calibrate FETCH_SIZE / WRITE_SIZE units)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o p -- python tools/kbench.py conv3x3_fwd --iters 3 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/wino_$c -o p -- python tools/kbench.py wino_fwd --iters 3 > /dev/null 2>&1
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/cal_$c -o p -- python tools/kbench.py roialign_fwd --iters 3 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_sq -o p -- python tools/kbench.py conv3x3_fwd --iters 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/wino_sq -o p -- python tools/kbench.py wino_fwd --iters 3 > /dev/null 2>&1
python - "$OUT" "$TAG" <<'PY'
import csv, collections, json, sys
out, tag = sys.argv[1], sys.argv[2]
def per_launch(d, kname):
    rows = [r for r in csv.DictReader(open("%s/%s/p_counter_collection.csv" % (out, d))) if kname in r["Kernel_Name"]]
    acc = collections.defaultdict(float); disp = set()
    for r in rows:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    kt = [r for r in csv.DictReader(open("%s/%s/p_kernel_trace.csv" % (out, d))) if kname in r["Kernel_Name"]]
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in kt]
    return {k: v / max(1, len(disp)) for k, v in acc.items()}, sum(dur) / max(1, len(dur)), len(disp)
res = {"kernel": "gemm_nn_fast<CONV3> M=921984 K=2304 N=256 (mask-head 3x3 conv fwd)"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    v, ns, n = per_launch("pmc_" + c, "gemm_nn_fast")
    res[c + "_KB_per_launch"] = v.get(c); res["avg_ns_" + c] = ns
    v, ns, n = per_launch("cal_" + c, "crop_fwd")
    res["calibration_crop_fwd_" + c + "_KB_per_launch"] = v.get(c)
res["calibration_crop_fwd_true_bytes"] = {"written": 4704 * 196 * 256 * 4, "read_unique": 32 * 28 * 28 * 256 * 4}
v, ns, n = per_launch("pmc_sq", "gemm_nn_fast")
res["sq_per_launch"] = v; res["avg_ns_sq"] = ns
xcds, simds = 8, 1024
if "GRBM_GUI_ACTIVE" in v:
    cyc = v["GRBM_GUI_ACTIVE"] / xcds
    res["effective_clock_GHz"] = cyc / ns
    res["mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / cyc
res["traffic_bytes_per_launch_corrected"] = 1024.0 * (2 * res["FETCH_SIZE_KB_per_launch"] + res["WRITE_SIZE_KB_per_launch"])
res["correction"] = "2 x FETCH_SIZE (gfx950 counts 128-B requests at 64 B; checked on crop_fwd above) + WRITE_SIZE, KB -> bytes"
json.dump(res, open("%s/%s_pmc_conv3x3_fwd.json" % (out, tag), "w"), indent=1)
print(json.dumps(res, indent=1))
# Winograd forward: the multiply stage (batched GEMM) and the two transforms, same passes
w = {"op": "myolo_conv3x3_wino_fwd NR=4704 14x14 256->256 (tools/kbench.py wino_fwd)"}
for kname, key in (("gemm_nn_fast", "multiply"), ("wino_in_kernel", "input_transform"), ("wino_out_kernel", "output_transform")):
    e = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        v, ns, n = per_launch("wino_" + c, kname)
        e[c + "_KB_per_launch"] = v.get(c); e["avg_ns"] = ns
    e["traffic_bytes_per_launch_corrected"] = 1024.0 * (2 * e["FETCH_SIZE_KB_per_launch"] + e["WRITE_SIZE_KB_per_launch"])
    w[key] = e
T = 4704 * 16
w["multiply"]["algorithmic_bytes"] = 36.0 * T * 512 * 4 + 36 * 256 * 256 * 4
w["input_transform"]["algorithmic_bytes"] = 4704 * 196 * 256 * 4 + 36.0 * T * 256 * 4
w["output_transform"]["algorithmic_bytes"] = 4704 * 196 * 256 * 4 + 36.0 * T * 256 * 4
v, ns, n = per_launch("wino_sq", "gemm_nn_fast")
w["multiply"]["sq_per_launch"] = v
if "GRBM_GUI_ACTIVE" in v:
    cyc = v["GRBM_GUI_ACTIVE"] / xcds
    w["multiply"]["effective_clock_GHz"] = cyc / ns
    w["multiply"]["mfma_util"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / cyc
w["traffic_bytes_per_launch_corrected"] = w["multiply"]["traffic_bytes_per_launch_corrected"]
json.dump(w, open("%s/%s_pmc_wino_multiply.json" % (out, tag), "w"), indent=1)
print(json.dumps(w, indent=1))
PY
head -8 $OUT/${TAG}_bench_kernel_by_grid.csv | cut -c1-170
cat $OUT/${TAG}_kbench.txt | grep -v "^dw  \|^dw 1"
cat $OUT/${TAG}_bench.json | cut -c1-400
python bench.py --config rice416-bf16 --steps 20 2>/dev/null | tail -1 > $OUT/${TAG}_bench_infer.json; cut -c1-300 $OUT/${TAG}_bench_infer.json
# the repository-HEAD head (N_BOX=5, R=245) and the direct-convolution form, for BASELINE.md
python bench.py --steps 10 --warmup 3 --cpu-images 0 --nbox 5 2> /dev/null | tail -1 > $OUT/${TAG}_bench_nbox5.json; cut -c1-260 $OUT/${TAG}_bench_nbox5.json
python bench.py --steps 10 --warmup 3 --cpu-images 0 --conv3x3 direct --no-variant 2> /dev/null | tail -1 > $OUT/${TAG}_bench_direct.json; cut -c1-260 $OUT/${TAG}_bench_direct.json
for k in deconv_mask_fwd; do python tools/kbench.py $k --iters 20 2>&1 | grep -v amdgpu.ids; done >> $OUT/${TAG}_kbench.txt
