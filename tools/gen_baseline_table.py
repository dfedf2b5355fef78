#!/usr/bin/env python
"""Rewrites BASELINE.md section 5 (round-2 results) from the committed evidence in profiles/r2g_*.json, so that every number in the
table is one bench.py printed.   python tools/gen_baseline_table.py"""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p = os.path.join(ROOT, "BASELINE.md")
s = open(p).read()
i = s.index('## 5. Results (round 2')
L = lambda n: json.load(open(os.path.join(ROOT, "profiles", n)))
d, x, rc = L("r2g_bench.json"), L("r2g_bench_bf16x6.json"), L("r2g_bench_rice416_bf16.json")
v, xv, rr, r = d['variants'], x['variants'], rc['roofline'], d['roofline']
new = '''## 5. Results (round 2, measured on 1× MI355X by `bench.py`; evidence under `profiles/r2*`, notes in `profiles/r2_notes.md`)

(this table is generated from `profiles/r2g_*.json` by `tools/gen_baseline_table.py`)

| Config | img/s | step ms | dominant kernel (`roofline`) | whole 3×3 conv op | other `roofline` objects | CPU restatement |
|---|---|---|---|---|---|---|
| Shapes 224², B=32, fp32, N_BOX=3 (R=147), all-ROI forward (**headline**; `profiles/r2g_bench.json`) | **%.1f** (round 1: 857–865) | **%.2f** | `wino_mm_kernel`: ONE launch of the 64 per-point GEMMs of the Winograd multiply on the F(6,3)/F(4,3) tiling (14 = 6+4+4: 400 point-tiles per ROI, 0.247 TFLOP): **%.2f ms = %.0f TFLOP/s = %.1f %%** of the 157.3 TFLOP/s fp32 MFMA peak; HBM traffic 3.90 GB vs 3.87 GB algorithmic (`r2_pmc_wino63_multiply.json`); `SQ_VALU_MFMA_BUSY_CYCLES` = the minimum for that work; MFMA pipe busy 82 %% at an effective 2.02 GHz. (Same kernel on the F(4,3)/F(2,3) tiling, 484 point-tiles: 2.51 ms = 119 TFLOP/s = 75.7 %%; rocBLAS `bmm` on those shapes: 112–114 TFLOP/s) | %.2f ms (round 1: 4.33) = %.0f direct-equivalent TFLOP/s | depthwise (14 layers, in-step events) %.2f ms; stand-alone 0.148 ms = 4.5 TB/s = 57 %% of 8 TB/s (round 1: 43 %%); ROIAlign fwd fused into conv1's input transform %.2f ms (= %.0f %% of 8 TB/s on SURVEY §8(d) bytes; the kernel writes the 2.0× larger Winograd image at 4.2 TB/s), stand-alone 0.17–0.21 ms = 57–70 %%; ROIAlign bwd 0.41 ms = 2.4 TB/s (round 1: 0.73 ms); pointwise (14 layers) %.2f ms = %.0f TFLOP/s = %.0f %% of the fp32 MFMA peak; Winograd layer boundary %.2f ms = %.2f TB/s (a plain device copy: 4.9 TB/s) | **%.2f img/s**: 32-image training step of the torch-CPU fp32 restatement, 16 threads (all usable cores of an EPYC 9575F), median of 2 after 1 warm-up |
| same, `FP32_MATMUL="bf16x6"` (opt-in, DESIGN §3 / §8: six exact bf16 piece products per fp32 product; `profiles/r2g_bench_bf16x6.json`, also `variants.winograd_multiply_bf16x6` of the headline run: %.1f ms) | **%.1f** | **%.2f** | `wino_mm_x6_kernel`: %.2f ms = %.0f TFLOP/s of bf16 piece products = %.0f %% of 2.5 PFLOP/s (%.0f fp32-equivalent TFLOP/s); bf16 pipe busy 65 %% at an effective 1.67 GHz (PMC on the 484-point-tile launch) | %.2f ms; fused deconv+mask GEMM 2.43 ms (native 3.99) | | |
| headline config, dense mask-head backward (`variants.dense_mask_backward`) | %.1f | %.1f | | | | |
| headline config, positives-only forward (`variants.mask_head_forward_on_positives_only`, opt-in, DESIGN §4b) | %.1f | %.1f (bf16x6: %.1f) | | | | |
| headline config, first k proposals of every image forced onto a ground-truth box (`variants.n_pos_sweep`) | %s at k = 5 / 10 / 20 | %s | the compacted mask-head backward costs ≈0.3 ms per positive per image; the headline batch (random-init net) has %.2f positives per image | | | |
| Rice 416², 5 anchors, inference, batch 4, bf16 mask head (`python bench.py --config rice416-bf16`; `profiles/r2g_bench_rice416_bf16.json`) | %.0f | %.2f | %s: %.3f ms = %.0f TFLOP/s = %.1f %% of 2.5 PFLOP/s dense bf16 | | | |

History of the single-GPU headline this round (same workload; all parity suites green at every step): 859.7 img/s (round-1 kernels) →
954.9 (mixed F(4,3)/F(2,3) Winograd tiling) → 969.5 (reduction-finish kernels with 8 loads in flight, ROIAlign backward with
lane-parallel box / sample tests, depthwise vertical strips) → 995–1004 (`wino_mm_kernel`: k-contiguous operands, `ds_read_b128`
fragments, one launch for all point groups; split-K for the 7×7 pointwise layers) → 1049–1067 (conv2-4 on the F(6,3)/F(4,3) tiling:
17 %% fewer products and plane bytes) → **1103–1118** (conv1 forward, data and weight gradient on it as well) — and 1372–1385 with the
opt-in bf16x6 product formation on top.
''' % (d['value'], d['ms_per_step'], r['avg_launch_ms'], r['achieved'], 100 * r['frac'],
       r['conv_op']['avg_ms'], r['conv_op']['direct_equivalent_tflops'],
       r['depthwise']['avg_ms'], r['roialign']['avg_ms'], 100 * r['roialign']['frac'], r['pointwise']['avg_ms'], r['pointwise']['achieved_tflops'],
       100 * r['pointwise']['frac_of_fp32_mfma_peak'], r['hbm_stages'][-1]['avg_ms'], r['hbm_stages'][-1]['achieved'] / 1000, d['cpu_baseline']['value'],
       v['winograd_multiply_bf16x6']['ms_per_step'], x['value'], x['ms_per_step'], x['roofline']['avg_launch_ms'], x['roofline']['achieved'],
       100 * x['roofline']['frac'], x['roofline']['fp32_equivalent_tflops'], x['roofline']['conv_op']['avg_ms'],
       v['dense_mask_backward']['value'], v['dense_mask_backward']['ms_per_step'],
       v['mask_head_forward_on_positives_only']['value'], v['mask_head_forward_on_positives_only']['ms_per_step'],
       xv['mask_head_forward_on_positives_only']['ms_per_step'],
       " / ".join("%.0f" % v['n_pos_sweep']['n_pos_%d' % k]['images_per_sec'] for k in (5, 10, 20)),
       " / ".join("%.1f" % v['n_pos_sweep']['n_pos_%d' % k]['ms_per_step'] for k in (5, 10, 20)), d['config']['n_pos_mean'],
       rc['value'], rc['ms_per_step'], rr['kernel'].split(' ')[0], rr['avg_launch_ms'], rr['achieved'], 100 * rr['frac'])
open(p, 'w').write(s[:i] + new)
print("BASELINE.md section 5 rewritten: %.1f img/s, %.2f ms" % (d['value'], d['ms_per_step']))
