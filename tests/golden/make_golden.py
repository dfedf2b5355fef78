#!/usr/bin/env python
"""Generates tests/golden/hotpath_small.npz -- inputs and expected outputs of the bit-exact and the
small floating-point pieces of the hot path, produced by the CPU oracle (oracle/np_ops.py).

The reference cannot be imported or run (TensorFlow 1.x / Keras 2.x / keras_applications are not
installable here and are un-pinned), so these vectors are NOT outputs of the
reference: they freeze the oracle's restatement (itself pinned by tests/test_oracle_kat.py) so that a
later change to either the oracle or the HIP kernels is caught.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]

from oracle import np_ops as O                       # noqa: E402
from myolo.config import make_config, ShapesConfig   # noqa: E402
from myolo.shapes import make_shapes_samples         # noqa: E402


def build():
    rng = np.random.default_rng(20260928)
    cfg = make_config(ShapesConfig, IMAGE_SHAPE=[128, 128, 3], BATCH_SIZE=2)
    G, A, C, T, R = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER, cfg.TRAIN_ROIS_PER_IMAGE
    B = 2
    out = {}
    samples = make_shapes_samples(B, cfg, start_index=5)
    enc = O.encode_batch(samples, cfg)
    images, true_boxes, y_true, gt_ids, gt_boxes, gt_masks = enc
    out.update(true_boxes=true_boxes.astype(np.float32), y_true=y_true.astype(np.float32), gt_ids=gt_ids, gt_boxes=gt_boxes,
               gt_masks_packed=np.packbits(gt_masks, axis=None), gt_masks_shape=np.array(gt_masks.shape))
    # logits that put some proposals on the ground-truth boxes (so positives exist)
    yp = (rng.standard_normal((B, G, G, A, 5 + C)) * 0.7).astype(np.float32)
    for b in range(B):
        for r, c, a in zip(*np.nonzero(y_true[b, ..., 4])):
            t = y_true[b, r, c, a]
            fx, fy = np.clip(t[0] - c, 0.05, 0.95), np.clip(t[1] - r, 0.05, 0.95)
            yp[b, r, c, a, 0:2] = [np.log(fx / (1 - fx)), np.log(fy / (1 - fy))]
            yp[b, r, c, a, 2:4] = [np.log(t[2] / cfg.ANCHORS[2 * a]), np.log(t[3] / cfg.ANCHORS[2 * a + 1])]
    out["y_pred"] = yp
    out["proposals"] = O.yolo_decode(yp, cfg.ANCHORS, G)
    out["detections"] = O.yolo_detections(yp, cfg.ANCHORS, G)
    rois, cls, masks, npos = O.mask_targets(out["proposals"], gt_ids, gt_boxes, gt_masks, cfg)
    assert npos.sum() >= 2
    out.update(rois=rois, target_class_ids=cls, target_masks_packed=np.packbits(masks.astype(bool), axis=None),
               target_masks_shape=np.array(masks.shape), n_pos=npos)
    yl = O.yolo_loss(y_true, yp, true_boxes, cfg, want_grad=True)
    out["yolo_terms"] = np.array([yl[k] for k in ("loss", "loss_xy", "loss_wh", "loss_conf", "loss_class", "recall", "n_coord", "n_conf")], np.float32)
    out["yolo_grad"] = yl["grad"]
    feat = rng.standard_normal((B, 16, 16, 8)).astype(np.float32)
    boxes = O.roi_boxes_to_crop_order(rois.reshape(-1, 4), cfg.ROI_BOX_ORDER)[:24]
    bind = np.repeat(np.arange(B), R).astype(np.int32)[:24]
    out.update(feat=feat, crop_boxes=boxes, crop_bind=bind, crop_out=O.crop_and_resize(feat, boxes, bind, (14, 14)))
    return out


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hotpath_small.npz")
    np.savez_compressed(path, **build())
    print(path, os.path.getsize(path), "bytes")
