#!/usr/bin/env python
"""gpurun_out/pmc_<tag>/<tag>_pmc_*.json (tools/collect_pmc.sh trunk / trunk_late) -> profiles/<tag>_pmc_trunk.json: one object per kernel family with the algorithmic bytes of
SURVEY 8(d) beside the counters.  Depthwise rows are matched to layers by launch order (tools/kbench.py walks dw1 .. dw14; identical
(kernel, grid) pairs share a row)."""
import glob
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
B = 32
dw = [(112, 32, 1), (112, 64, 2), (56, 64, 1), (56, 128, 2), (28, 256, 1), (28, 256, 1), (28, 512, 2)] + [(14, 512, 1)] * 5 + [(14, 512, 2), (7, 1024, 1)]
dw_bytes = [B * (h * h + (h // s) * (h // s)) * c * 4.0 for h, c, s in dw]
out = {"method": "tools/pmc_kbench.py: separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ set) with --kernel-trace only on tools/kbench.py targets whose "
                 "inputs rotate through > 600 MB of buffers (no Infinity-Cache hits); traffic = 2 x FETCH_SIZE + WRITE_SIZE, KB x 1024 (gfx950 correction of "
                 "MI355X_MICROARCH.md); per-launch means over the last launches of each (kernel, grid) group",
       "depthwise_layer_bytes": {"dw%d" % (i + 1): b for i, b in enumerate(dw_bytes)},
       "families": {}}
notes = {
    "dw_fwd": "depthwise forward as the training step runs it (producer BatchNorm on load + statistics epilogue), dw_rows_kernel (round 4)",
    "dw_fwd_round3_kernel": "the same with option dw_legacy=1: round 3's register-tiled dw_fwd_kernel",
    "dw_bwd_data_s1": "depthwise data gradient, stride 1: dw_rows_kernel with the filter rotated (round 4)",
    "dw_bwd_data": "depthwise data gradient, stride 2 layers (gather kernel; stride-1 layers no longer use it)",
    "dw_bwd_data_round3_kernel": "option dw_bwd_legacy=1: round 3's gather kernel on every layer",
    "dw_wgrad": "depthwise weight gradient, dw_rows_wgrad_kernel (round 4)",
    "dw_wgrad_round3_kernel": "option dw_bwd_legacy=1: round 3's dw_wgrad_kernel",
    "pw_gemm": "pointwise layers on gemm_nn_fast<PLAIN> (fp32 MFMA; every layer when wino_x6 is off, the < 256-channel ones in the product)",
    "pw_x6": "pointwise layers with >= 256 channels on wino_mm_x6_kernel<PLAIN, PW> (option wino_x6=1 = the product's FP32_MATMUL='bf16x6')",
    "crop_fwd": "ROIAlign forward, stand-alone crop_fwd_kernel (the step fuses it into conv1's Winograd input transform); algorithmic 0.970 GB",
    "crop_bwd": "ROIAlign backward, 2 x 2 pixel quads in XCD-contiguous order (round 4); algorithmic 0.944 GB read + 0.026 GB written",
    "crop_bwd_round3_order": "option tune0=1: round 3's workgroup = four consecutive pixels, plain order",
    "dw_bwd_data_s2": "depthwise data gradient, stride 2: dw_bwd_data_s2_kernel (late round 4: a dy pixel's thread writes its 2 x 2 block of dx); algorithmic = dy + dx of dw2 / dw4 / dw7 / dw13",
    "pw_thin_fwd": "conv_pw_1 / conv_pw_2 forward on pw_fwd_thin_kernel (late round 4: register-fed fp32 MFMA, BatchNorm of the input in registers, statistics epilogue); algorithmic = x + y: 154 / 51 MB",
    "bn_bwd_sums": "BatchNorm + ReLU6 backward, pass 1 (colreduce_kernel<OpBnBwd>: reads dy and x) at the trunk shapes of tools/kbench.py bn_bwd; algorithmic = 2 x M x C x 4 bytes",
    "bn_bwd_dx": "BatchNorm + ReLU6 backward, pass 2 (bn_bwd_dx_kernel: reads dy and x, writes dx); algorithmic = 3 x M x C x 4 bytes",
}
for f in sorted(glob.glob(os.path.join(src, "*.json"))):
    name = os.path.basename(f)[:-5]
    if "_pmc_" not in name:
        continue
    key = name.split("_pmc_", 1)[1]              # family name without the round tag
    r = json.load(open(f))
    rows = r["rows"]
    if key.startswith("dw_fwd"):
        # launch order dw1.. ; groups in first-seen order
        order = {"dw_fwd": [[0], [1], [2, 4, 5], [3, 6], [7, 8, 9, 10, 11], [12], [13]]}.get(key)
        if order and len(order) == len(rows):
            for e, ls in zip(rows, order):
                e["layers"] = ["dw%d" % (i + 1) for i in ls]
                e["algorithmic_bytes_per_launch"] = sum(dw_bytes[i] for i in ls) / len(ls)
                if e.get("traffic_bytes_corrected"):
                    e["traffic_over_algorithmic"] = e["traffic_bytes_corrected"] / e["algorithmic_bytes_per_launch"]
                e["algorithmic_gbs"] = e["algorithmic_bytes_per_launch"] / e["avg_ns"]
                e["frac_of_8tbs"] = e["algorithmic_gbs"] / 8000.0
    if key.startswith("crop"):
        alg = 0.970e9 if "fwd" in key else 0.970e9
        for e in rows:
            e["algorithmic_bytes_per_launch"] = alg
            if e.get("traffic_bytes_corrected"):
                e["traffic_over_algorithmic"] = e["traffic_bytes_corrected"] / alg
            e["frac_of_8tbs"] = alg / e["avg_ns"] / 8000.0
    out["families"][name] = {"what": notes.get(key, ""), "kbench": r["kbench"], "kbench_options": r["kbench_options"], "rows": rows}
os.makedirs(os.path.dirname(os.path.abspath(dst)), exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)
print("wrote", dst, "families:", len(out["families"]))
