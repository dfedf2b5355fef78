#!/usr/bin/env python
"""Appends / rewrites the results section of a round in BASELINE.md from the committed evidence profiles/<tag>_bench_detail.json (the full object
bench.py writes beside its compact line), so that every number in the table is one bench.py printed.
    python tools/gen_baseline_table.py r5 [section number, default 8] [free text appended to the provenance line]
(sections 5-7 = rounds 2-4, written by earlier forms of this script from profiles/r2g_*, r3c_*, r4_bench.json)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
sec = int(sys.argv[2]) if len(sys.argv) > 2 else 8
spread = sys.argv[3] if len(sys.argv) > 3 else ""
rnd = tag[1:2]
p = os.path.join(ROOT, "BASELINE.md")
s = open(p).read()
d = json.load(open(os.path.join(ROOT, "profiles", "%s_bench_detail.json" % tag)))
r, v = d["roofline"], d["variants"]
sw = v["n_pos_sweep"]
hb = d.get("hbm_copy_measured_gbs", {})
inf = d.get("inference_rice416_bf16", {})
nb5 = d.get("secondary_nbox5", {})
mf = r["pointwise"].get("mfma_bound_layers", {})
rows = "\n".join("| %s | %s | %.1f | %.0f | %.1f | %s | %.2f | %.1f | %.2f |" % (l["layer"], l["shape"], l["ms"] * 1e3, l["gbs"], l["tflops"], l["roof"], l["frac"],
                                                                           l.get("ms_net", 0.0) * 1e3, l.get("frac_net", 0.0))
                 for l in r.get("trunk_layers", []))
new = '''## %(sec)d. Results (round %(rnd)s, measured on 1x MI355X by ONE `python bench.py --steps 20 --warmup 5`; evidence `profiles/%(tag)s_*`, notes `profiles/r%(rnd)s_notes.md`)

(generated from `profiles/%(tag)s_bench_detail.json` by `tools/gen_baseline_table.py`; %(spread)s)

| Object of the line | img/s | ms | what it is |
|---|---|---|---|
| **headline** `value` (`config.fp32_products = "%(fp)s"`, `dtype f32`) | **%(val).1f** | **%(ms).2f** (p10 %(p10).2f / p50 %(p50).2f / p90 %(p90).2f) | Shapes 224x224, batch 32, N_BOX=3 (R=147), forward + backward + Adam; round 5: 1620 / 19.75, round 4: 1596 / 20.05 (best box 1630 / 19.63), round 3: 1487.8 / 21.51, round 2: 1139.2 / 28.09, round 1: 863.4 / 37.06 |
| `train_api.train` = `MaskYOLO.train()` on a 512-image ShapesDataset | %(tav).1f | %(tams).2f | the drop-in call (model.py:943-1060): host BatchGenerator on a prefetch thread, pinned byte staging, lazy losses |
| `train_api.train_shapes_stream` | %(tsv).1f | %(tsms).2f | inputs produced on the device |
| `train_api.reference_same_state` | %(rsv).1f | %(rsms).2f | `Net.train_step` on resident batches right after those calls, same weights (%(rsn).2f positives per image; the headline's random-init net: %(hn)s): the like-for-like reference of the public calls |
| `comm_overlap_probe_ms.ms_per_step_with_probe` | | %(cpms).2f | the same step with the gradient buckets (five from round 6, three before) all-reduced on the copy stream as backward completes them (1-rank RCCL communicator through the C-ABI) |
| `variants.fp32_products_native` | %(nat_v).1f | %(nat_ms).2f | the same step with every product on `v_mfma_f32_32x32x2_f32` |
| `variants.dense_mask_backward` | %(dn_v).1f | %(dn_ms).2f | structural zeros of the mask-head backward not exploited |
| `variants.mask_head_forward_on_positives_only` | %(po_v).1f | %(po_ms).2f | opt-in, DESIGN 4b; never the headline |
| `variants.n_pos_sweep` k = 5 / 10 / 20 | %(s5v).0f / %(s10v).0f / %(s20v).0f | %(s5).2f / %(s10).2f / %(s20).2f | first k proposals of every image forced onto a ground-truth box: the band of a trained net |
| `secondary_nbox5` | %(n5v).1f | %(n5ms).2f | repository-HEAD head, N_BOX=5, R=245 |
| `inference_rice416_bf16` | %(iv).1f | %(ims).2f | BASELINE configs[3]: Rice 416x416, batch 4, bf16 mask head, hipGraph replays, `config.in_flight` batches in flight (`Net.predict_stream`; three by default) |
| `inference_rice416_bf16.one_in_flight` | %(i1v).1f | %(i1ms).2f | the same forwards strictly one after the other (what rounds 1-2 reported) |
| `inference_rice416_bf16.detect_many` | %(dmv).1f | | the public call: `MaskYOLO.detect_many` on uint8 images -- upload, the same graphs, detect()'s selection and unmolding per image |
| `cpu_baseline` | %(cpu).2f | | torch-CPU fp32 restatement, %(cores)d threads, 32-image training step |

Dominant kernel (`roofline`): %(kname)s: **%(kms).3f ms per launch = %(ach).0f TFLOP/s of bf16 piece products = %(frac).3f of 2.5 PFLOP/s** (`frac_composite` %(fcomp).3f against max(flop / peak, bytes / measured copy rate))
(%(eq).0f fp32-equivalent TFLOP/s; the fp32 MFMA peak is 157.3); whole conv op %(cop).2f ms; HBM traffic %(traf)s.
Measured HBM copy bandwidth (`hbm_copy_measured_gbs`, GB/s): float4 copy %(c1).0f at 8 workgroups per CU, **%(c2).0f at one workgroup per CU**;
read-only %(c3).0f / %(c4).0f; write-only %(c5).0f / %(c6).0f. Measured matrix-pipe rate with register operands (`mfma_measured_tflops`): bf16 %(mfb).0f, fp32 %(mff32).0f TFLOP/s. Winograd layer boundary (`hbm_stages`): %(bms).3f ms = %(bgb).0f GB/s.
Depthwise (14 layers, in-step, HIP-event brackets around each fused launch): %(dwms).3f ms = %(dwf).2f of 8 TB/s on SURVEY 8(d) bytes; minus the event
brackets' own cost %(dwn).3f ms = %(dwfn).2f (rocprofv3 kernel time of the same launches: `profiles/%(tag)s_bench_kernel_by_grid.csv`). ROIAlign forward (fused into
conv1's input transform): %(roims).3f ms = %(roif).2f on 8(d) bytes, %(roiw).0f GB/s on the bytes it writes. Pointwise (14 layers): %(pwms).3f ms;
MFMA-bound layers %(mff).2f of 157.3 TF/s in fp32-equivalent flops.

Per-layer trunk table (`roofline.trunk_layers`; ms = HIP events around the layer's forward call in the step: conv + its BatchNorm statistics):

(`us net` / `frac net`: the same bracket minus `roofline.event_bracket_ms` = %(brus).1f us, what a pair of timing events around a 4-byte fill kernel reads)

| layer | shape | us | GB/s | TFLOP/s | roof | frac | us net | frac net |
|---|---|---|---|---|---|---|---|---|
%(rows)s
''' % dict(tag=tag, sec=sec, rnd=rnd, spread=spread, tav=d.get('train_api', {}).get('train', {}).get('images_per_sec', 0.0), tams=d.get('train_api', {}).get('train', {}).get('ms_per_step', 0.0),
           rsv=(d.get('train_api', {}).get('reference_same_state') or {}).get('images_per_sec', 0.0), rsms=(d.get('train_api', {}).get('reference_same_state') or {}).get('ms_per_step', 0.0),
           rsn=(d.get('train_api', {}).get('reference_same_state') or {}).get('n_pos_mean', 0.0), hn='%.2f' % d['config'].get('n_pos_mean', 0.0),
           tsv=d.get('train_api', {}).get('train_shapes_stream', {}).get('images_per_sec', 0.0), tsms=d.get('train_api', {}).get('train_shapes_stream', {}).get('ms_per_step', 0.0),
           cpms=d.get('comm_overlap_probe_ms', {}).get('ms_per_step_with_probe', 0.0), fcomp=r.get('frac_composite', 0.0), fp=d["config"].get("fp32_products"), val=d["value"], ms=d["ms_per_step"], p10=d["step_ms"]["p10"], p50=d["step_ms"]["p50"], p90=d["step_ms"]["p90"],
           nat_v=v["fp32_products_native"]["value"], nat_ms=v["fp32_products_native"]["ms_per_step"],
           dn_v=v["dense_mask_backward"]["value"], dn_ms=v["dense_mask_backward"]["ms_per_step"],
           po_v=v["mask_head_forward_on_positives_only"]["value"], po_ms=v["mask_head_forward_on_positives_only"]["ms_per_step"],
           s5v=sw["n_pos_5"]["images_per_sec"], s10v=sw["n_pos_10"]["images_per_sec"], s20v=sw["n_pos_20"]["images_per_sec"],
           s5=sw["n_pos_5"]["ms_per_step"], s10=sw["n_pos_10"]["ms_per_step"], s20=sw["n_pos_20"]["ms_per_step"],
           n5v=nb5.get("value", 0.0), n5ms=nb5.get("ms_per_step", 0.0), iv=inf.get("value", 0.0), ims=inf.get("ms_per_step", 0.0),
           dmv=(inf.get("detect_many") or {}).get("images_per_sec", 0.0),
           i1v=inf.get("one_in_flight", {}).get("value", 0.0), i1ms=inf.get("one_in_flight", {}).get("ms_per_step", 0.0),
           mfb=d.get("mfma_measured_tflops", {}).get("bf16_32x32x16", 0.0), mff32=d.get("mfma_measured_tflops", {}).get("f32_32x32x2", 0.0),
           cpu=d["cpu_baseline"]["value"], cores=d["cpu_baseline"]["cores"],
           kname=r["kernel"].split(":")[0], kms=r["avg_launch_ms"], ach=r["achieved"], frac=r["frac"], eq=r.get("fp32_equivalent_tflops") or 0.0,
           cop=r["conv_op"]["avg_ms"], traf=("%.2f GB per launch (PMC, %s)" % (r["traffic"] / 1e9, r.get("traffic_source", "profiles/r3_pmc_x6.json")[:60])) if r.get("traffic") else "n/a",
           c1=hb.get("float4", 0), c2=hb.get("float4_1wg_per_cu", 0), c3=hb.get("read_only", 0), c4=hb.get("read_only_1wg_per_cu", 0),
           c5=hb.get("write_only", 0), c6=hb.get("write_only_1wg_per_cu", 0),
           bms=r["hbm_stages"][-1]["avg_ms"], bgb=r["hbm_stages"][-1]["achieved"],
           dwms=r["depthwise"]["avg_ms"], dwf=r["depthwise"]["frac"], dwn=r["depthwise"].get("avg_ms_net", 0.0), dwfn=r["depthwise"].get("frac_net", 0.0),
           brus=1e3 * r.get("event_bracket_ms", 0.0), roims=r["roialign"]["avg_ms"], roif=r["roialign"]["frac"],
           roiw=r["roialign"].get("achieved_on_written_bytes") or 0.0, pwms=r["pointwise"]["avg_ms"], mff=mf.get("frac_of_fp32_mfma_peak", 0.0), rows=rows)
head = "## %d. Results (round %s" % (sec, rnd)
if head in s:
    s = s[:s.index(head)]
s = s.rstrip() + "\n\n" + new
open(p, "w").write(s)
print("BASELINE.md section %d written from" % sec, tag)
