"""Committed golden vectors (tests/golden/hotpath_small.npz, made by tests/golden/make_golden.py):
the oracle must keep reproducing them (CPU), and the HIP kernels must match them (GPU) -- bit-exact for
decode / detections / ROI partition / class ids / mask targets, 1e-4 for the float pieces."""
import os

import numpy as np
import pytest

from oracle import np_ops as O
from myolo.config import make_config, ShapesConfig

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_small.npz")


def load():
    g = dict(np.load(PATH))
    g["gt_masks"] = np.unpackbits(g["gt_masks_packed"])[:int(np.prod(g["gt_masks_shape"]))].reshape(g["gt_masks_shape"]).astype(bool)
    g["target_masks"] = np.unpackbits(g["target_masks_packed"])[:int(np.prod(g["target_masks_shape"]))] \
        .reshape(g["target_masks_shape"]).astype(np.float32)
    return g, make_config(ShapesConfig, IMAGE_SHAPE=[128, 128, 3], BATCH_SIZE=2)


def test_oracle_reproduces_golden_vectors():
    g, cfg = load()
    G = cfg.GRID_W
    assert np.array_equal(O.yolo_decode(g["y_pred"], cfg.ANCHORS, G), g["proposals"])
    assert np.array_equal(O.yolo_detections(g["y_pred"], cfg.ANCHORS, G), g["detections"])
    rois, cls, masks, npos = O.mask_targets(g["proposals"], g["gt_ids"], g["gt_boxes"], g["gt_masks"], cfg)
    assert np.array_equal(rois, g["rois"]) and np.array_equal(cls, g["target_class_ids"])
    assert np.array_equal(masks, g["target_masks"]) and np.array_equal(npos, g["n_pos"])
    tb = g["true_boxes"]
    yl = O.yolo_loss(g["y_true"], g["y_pred"], tb, cfg, want_grad=True)
    np.testing.assert_allclose([yl[k] for k in ("loss", "loss_xy", "loss_wh", "loss_conf", "loss_class", "recall", "n_coord", "n_conf")],
                               g["yolo_terms"], rtol=1e-6)
    np.testing.assert_allclose(yl["grad"], g["yolo_grad"], rtol=1e-5, atol=1e-8)
    assert np.array_equal(O.crop_and_resize(g["feat"], g["crop_boxes"], g["crop_bind"], (14, 14)), g["crop_out"])


@pytest.mark.gpu
def test_hip_kernels_match_golden_vectors():
    import torch
    from myolo import _ext as X
    g, cfg = load()
    dev = "cuda:0"
    keep = []

    def dt(a):
        t = torch.as_tensor(np.ascontiguousarray(a), device=dev)
        keep.append(t)
        return t

    B, G, A, C, T, R = 2, cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER, cfg.TRAIN_ROIS_PER_IMAGE
    anchors = dt(np.asarray(cfg.ANCHORS, np.float32))
    prop = torch.empty(B, R, 4, device=dev)
    det = torch.empty(B, R, 6, device=dev)
    X.call("myolo_yolo_decode", X.ptr(dt(g["y_pred"])), X.ptr(anchors), X.ptr(prop), B, G, A, C, X.stream())
    X.call("myolo_yolo_detections", X.ptr(dt(g["y_pred"])), X.ptr(anchors), X.ptr(det), B, G, A, C, X.stream())
    assert np.array_equal(prop.cpu().numpy(), g["proposals"]) and np.array_equal(det.cpu().numpy(), g["detections"])
    rois, tcls = torch.empty(B, R, 4, device=dev), torch.empty(B, R, dtype=torch.int32, device=dev)
    tm, npos = torch.empty(B, R, 28, 28, device=dev), torch.empty(B, dtype=torch.int32, device=dev)
    X.call("myolo_mask_targets", X.ptr(prop), X.ptr(dt(g["gt_ids"])), X.ptr(dt(g["gt_boxes"])), X.ptr(dt(g["gt_masks"].view(np.uint8))),
           X.ptr(rois), X.ptr(tcls), X.ptr(tm), X.ptr(npos), B, R, T, 128, 128, 28, 28, X.stream())
    assert np.array_equal(rois.cpu().numpy(), g["rois"]) and np.array_equal(tcls.cpu().numpy(), g["target_class_ids"])
    assert np.array_equal(tm.cpu().numpy(), g["target_masks"]) and np.array_equal(npos.cpu().numpy(), g["n_pos"])
    terms, grad = torch.empty(8, device=dev), torch.empty(*g["y_pred"].shape, device=dev)
    ws = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
    X.call("myolo_yolo_loss", X.ptr(dt(g["y_true"])), X.ptr(dt(g["y_pred"])), X.ptr(dt(g["true_boxes"].reshape(B, T, 4))), X.ptr(anchors),
           X.ptr(dt(cfg.CLASS_WEIGHTS)), cfg.OBJECT_SCALE, cfg.NO_OBJECT_SCALE, cfg.COORD_SCALE, cfg.CLASS_SCALE, 1.0,
           X.ptr(terms), X.ptr(grad), B, G, A, C, T, ws.data_ptr(), ws.numel(), X.stream())
    np.testing.assert_allclose(terms.cpu().numpy(), g["yolo_terms"], rtol=1e-4)
    assert np.abs(grad.cpu().numpy() - g["yolo_grad"]).max() <= 1e-4 * np.abs(g["yolo_grad"]).max()
    nb = g["crop_boxes"].shape[0]
    out = torch.empty(nb, 14, 14, 8, device=dev)
    X.call("myolo_crop_and_resize_fwd", X.ptr(dt(g["feat"])), X.ptr(dt(g["crop_boxes"])), X.ptr(dt(g["crop_bind"])), X.ptr(out),
           2, 16, 16, 8, nb, 14, 14, X.stream())
    assert np.abs(out.cpu().numpy() - g["crop_out"]).max() <= 1e-5
