#!/usr/bin/env python
"""End-to-end learning check (no oracle involved): overfit ONE fixed batch of 8 Shapes images with the training step, then
run detect() on those images with the trained weights.  Expected on an MI355X (25 s): the mask loss falls from 0.69 to about
0.01, every detection has the class of a ground-truth instance and its pasted mask overlaps that instance with IoU > 0.8.
  python tools/overfit_check.py [WINOGRAD_TILES=f43|f63] [FP32_MATMUL=native|bf16x6] [SEED=n]       (config overrides; default: the config's)
"""
import sys
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
import numpy as np
from myolo.config import make_config, ShapesConfig
from myolo.model import MaskYOLO
from myolo.shapes import make_shapes_samples
from myolo.myolo_utils import BatchGenerator
B = 8
for kv in os.environ.get("MYOLO_LIB_OPTIONS", "").split(","):          # kernel selections for A/B runs, e.g. MYOLO_LIB_OPTIONS=tune0=23552
    if "=" in kv:
        from myolo import _ext as _X
        _X.load(); _X.set_option(kv.split("=")[0], int(kv.split("=")[1]))
over = dict(a.split("=", 1) for a in sys.argv[1:])
SEED = int(over.pop("SEED", 1))                # weight-init seed
cfg = make_config(ShapesConfig, IMAGE_SHAPE=[224, 224, 3], BATCH_SIZE=B, **over)
print("WINOGRAD_TILES=%s FP32_MATMUL=%s" % (cfg.WINOGRAD_TILES, cfg.FP32_MATMUL))
samples = make_shapes_samples(B, cfg)
batch, _ = BatchGenerator(samples, cfg, 'training', shuffle=False, norm=True)[0]
m = MaskYOLO(mode="training", config=cfg, seed=SEED)
m.set_trainable(".*"); m.compile(1e-3, 0.9)
db = m.net.to_device_batch(batch)
for i in range(1500):
    out = m.net.train_step(db, 1e-3 if i < 1000 else 3e-4)
    if i % 100 == 99:
        print("step %d yolo %.4f mask %.4f npos %d" % (i + 1, float(out["yolo_terms"][0]), float(out["mask_terms"][0]), int(out["n_pos"].sum())))
m.save_weights("/tmp/overfit.npz")
inf = MaskYOLO(mode="inference", config=cfg)
inf.load_weights("/tmp/overfit.npz")
for k in range(3):
    img = samples[k][0]
    res = inf.detect(img.astype(np.uint8), cs_threshold=0.35)[0]
    gt_cls = [int(c) for c in samples[k][1]]
    gt_masks = samples[k][3]
    ious = []
    for j in range(res["full_masks"].shape[2]):
        pm = res["full_masks"][:, :, j]
        best = max((np.logical_and(pm, gt_masks[:, :, g]).sum() / max(1, np.logical_or(pm, gt_masks[:, :, g]).sum())) for g in range(gt_masks.shape[2]))
        ious.append(round(float(best), 2))
    print("image %d: GT classes %s | detected classes %s scores %s best mask IoU %s" % (k, gt_cls, [int(c) for c in res["class_ids"]],
          ["%.2f" % s for s in res["confidence_scores"]], ious))
