"""Known-answer tests that pin the CPU oracle (oracle/np_ops.py, oracle/np_model.py).

The reference has no tests, golden vectors or runnable kernels, so the
oracle is pinned by (1) the hand-derived KATs below -- each states the closed form it checks --
(2) agreement of its analytic backward with an independent torch-autograd composition in float64,
(3) property tests.  Never checked against TensorFlow itself (not installable); third-party pins: tests/test_third_party_kats.py, tests/test_keras_h5_and_skimage.py."""
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle import np_ops as O
from oracle import np_model
from myolo.config import make_config, ShapesConfig

F32 = np.float32


# ---------------------------------------------------------------- deterministic math
def test_det_expf_within_2ulp_of_libm():
    x = np.linspace(-80, 80, 200001).astype(F32)
    got, ref = O.det_expf(x), np.exp(x.astype(np.float64))
    ulp = np.spacing(ref.astype(F32)).astype(np.float64)
    assert np.abs(got - ref).max() / 1.0 >= 0 and (np.abs(got.astype(np.float64) - ref) / ulp).max() <= 2.0
    assert O.det_expf(F32(0)) == 1.0 and O.det_sigmoid(F32(0)) == 0.5


def test_round_half_even():
    assert np.array_equal(O.round_half_even(np.array([0.5, 1.5, 2.5, -0.5, 0.49999997, 0.50000006], F32)),
                          np.array([0, 2, 2, -0, 0, 1], F32))


# ---------------------------------------------------------------- decode (model.py:1442-1473, 1493-1538)
def test_zero_logit_decode():
    G, A, C = 7, 3, 4
    anchors = [1.27273, 1.277385, 2.47446, 2.56253, 4.03843, 4.07434]
    yp = np.zeros((1, G, G, A, 5 + C), F32)
    det = O.yolo_detections(yp, anchors, G).reshape(G, G, A, 6)
    for row in range(G):
        for col in range(G):
            for a in range(A):
                cx, cy = (col + 0.5) / G, (row + 0.5) / G
                w, h = anchors[2 * a] / G, anchors[2 * a + 1] / G       # both axes / GRID_W
                np.testing.assert_allclose(det[row, col, a, :4], [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], rtol=2e-6)
                assert det[row, col, a, 4] == 0.5 and det[row, col, a, 5] == 0.0
    assert np.array_equal(O.yolo_decode(yp, anchors, G).reshape(G, G, A, 4), det[..., :4])


def test_decode_flatten_order_is_row_col_box():
    G, A, C = 4, 3, 2
    yp = np.zeros((1, G, G, A, 5 + C), F32)
    yp[0, 2, 1, 1, 5 + 1] = 3.0       # class 1 wins at (row 2, col 1, box 1)
    det = O.yolo_detections(yp, [1, 1, 2, 2, 3, 3], G)[0]
    assert det[(2 * G + 1) * A + 1, 5] == 1.0 and det[:, 5].sum() == 1.0


# ---------------------------------------------------------------- crop_and_resize (model.py:385-387)
def _ramp(H, W):
    return (np.arange(H)[:, None] * 10 + np.arange(W)[None, :]).astype(F32)[None, :, :, None]


def test_crop_identity_and_midpoints():
    img = _ramp(3, 3)
    box = np.array([[0, 0, 1, 1]], F32)
    assert np.array_equal(O.crop_and_resize(img, box, [0], (3, 3)), img)
    out = O.crop_and_resize(img, box, [0], (5, 5))[0, :, :, 0]
    exp = (np.arange(5)[:, None] * 5.0 + np.arange(5)[None, :] * 0.5).astype(F32)     # samples every half pixel
    assert np.array_equal(out, exp)


def test_crop_extrapolation_and_reference_eyeball_case():
    """deprecated/test_mask.py:39: boxes [[0,0,0,0],[0.2,0.6,1.3,0.9]]."""
    img = _ramp(11, 11)
    boxes = np.array([[0, 0, 0, 0], [0.2, 0.6, 1.3, 0.9]], F32)
    out = O.crop_and_resize(img, boxes, [0, 0], (12, 12))[..., 0]
    assert np.all(out[0] == img[0, 0, 0, 0])                 # degenerate box: every sample at pixel (0,0)
    in_y = 0.2 * 10 + np.arange(12) * ((1.3 - 0.2) * 10 / 11)
    outside = in_y > 10
    assert outside.any() and np.all(out[1][outside] == 0) and np.all(out[1][~outside] != 0)


def test_crop_box_order_is_y_first():
    """non-square ROI on an asymmetric ramp: box columns are (y1,x1,y2,x2)."""
    img = _ramp(5, 9)
    out = O.crop_and_resize(img, np.array([[0.0, 0.5, 1.0, 1.0]], F32), [0], (2, 2))[0, :, :, 0]
    assert np.array_equal(out, np.array([[4, 8], [44, 48]], F32))
    # model.py:385-387 quirk: an [x1,y1,x2,y2] ROI is read as [y1,x1,y2,x2]
    roi_xyxy = np.array([[0.5, 0.0, 1.0, 1.0]], F32)
    quirk = O.crop_and_resize(img, O.roi_boxes_to_crop_order(roi_xyxy, "xyxy_as_yxyx"), [0], (2, 2))[0, :, :, 0]
    fixed = O.crop_and_resize(img, O.roi_boxes_to_crop_order(roi_xyxy, "yxyx"), [0], (2, 2))[0, :, :, 0]
    assert np.array_equal(fixed, out) and not np.array_equal(quirk, out)


def test_crop_bwd_is_transpose_of_fwd():
    rng = np.random.default_rng(0)
    img = rng.standard_normal((2, 6, 7, 3)).astype(F32)
    boxes = np.array([[0.1, 0.2, 0.8, 0.9], [-0.2, 0.1, 0.6, 1.2], [0.3, 0.3, 0.3, 0.3]], F32)
    bind = [0, 1, 1]
    g = rng.standard_normal((3, 4, 4, 3)).astype(F32)
    lhs = float((O.crop_and_resize(img, boxes, bind, (4, 4)).astype(np.float64) * g).sum())
    rhs = float((O.crop_and_resize_bwd_image(g, boxes, bind, img.shape).astype(np.float64) * img).sum())
    assert abs(lhs - rhs) < 1e-4 * max(1, abs(lhs))


# ---------------------------------------------------------------- conv / depthwise / BN
def test_dw_stride2_pads_bottom_right():
    x = np.zeros((1, 4, 4, 1), F32)
    x[0, 3, 3, 0] = 1
    w = np.arange(9, dtype=F32).reshape(3, 3, 1)
    y = O.dwconv3x3(x, w, 2)
    assert y.shape == (1, 2, 2, 1) and y[0, 1, 1, 0] == w[1, 1, 0] and y[0, 0, 0, 0] == 0
    x[:] = 0
    x[0, 0, 0, 0] = 1                     # top-left pixel is tap (0,0) of output (0,0): no top/left padding
    assert O.dwconv3x3(x, w, 2)[0, 0, 0, 0] == w[0, 0, 0]


def test_conv1_geometry():
    x = np.zeros((1, 4, 4, 3), F32)
    x[0, 0, 0, :] = 1
    w = np.zeros((3, 3, 3, 1), F32)
    w[1, 1, :, 0] = 1                    # centre tap
    y = O.conv2d(x, w, stride=2, pads=O.conv1_pads())
    assert y.shape == (1, 2, 2, 1) and y[0, 0, 0, 0] == 3 and y.sum() == 3


def _numgrad(f, x, eps=1e-5):
    g = np.zeros_like(x)
    it = np.nditer(x, flags=["multi_index"])
    for _ in it:
        i = it.multi_index
        o = x[i]
        x[i] = o + eps
        a = f()
        x[i] = o - eps
        b = f()
        x[i] = o
        g[i] = (a - b) / (2 * eps)
    return g


def test_bn_train_backward_finite_difference():
    O.set_precision(np.float64)
    try:
        rng = np.random.default_rng(1)
        x = rng.standard_normal((7, 3))
        g, b = rng.standard_normal(3) + 1, rng.standard_normal(3)
        dy = rng.standard_normal((7, 3))
        y, cache = O.bn_train(x, g, b)
        dx, dg, db = O.bn_train_bwd(cache, g, dy)
        np.testing.assert_allclose(dx, _numgrad(lambda: float((O.bn_train(x, g, b)[0] * dy).sum()), x), atol=1e-6)
        np.testing.assert_allclose(dg, _numgrad(lambda: float((O.bn_train(x, g, b)[0] * dy).sum()), g), atol=1e-6)
        np.testing.assert_allclose(db, dy.sum(0), atol=1e-9)
    finally:
        O.set_precision(np.float32)


def test_bn_moving_update_keras_formula():
    mm, mv = O.bn_moving_update(np.zeros(1, F32), np.ones(1, F32), np.array([2.0], F32), np.array([4.0], F32), 10, fused_tf=False)
    assert abs(mm[0] - 0.02) < 1e-7
    assert abs(mv[0] - (0.99 + 0.01 * 4.0 * 10 / (10 - 1.001))) < 1e-6
    # default: Keras' factor on top of tf.nn.fused_batch_norm's Bessel-corrected batch variance (see tests/test_third_party_kats.py)
    _, mv = O.bn_moving_update(np.zeros(1, F32), np.ones(1, F32), np.array([2.0], F32), np.array([4.0], F32), 10)
    assert abs(mv[0] - (0.99 + 0.01 * 4.0 * (10 / 9.0) * 10 / (10 - 1.001))) < 1e-6


def test_deconv_layout():
    x = np.zeros((1, 1, 1, 2), F32)
    x[0, 0, 0] = [1, 10]
    w = np.arange(2 * 2 * 3 * 2, dtype=F32).reshape(2, 2, 3, 2)          # [ky,kx,co,ci]
    y = O.deconv2x2s2(x, w, np.zeros(3, F32))
    for ky in range(2):
        for kx in range(2):
            assert np.array_equal(y[0, ky, kx], w[ky, kx] @ x[0, 0, 0])


# ---------------------------------------------------------------- YOLO loss (model.py:86-242)
def _cfg(**kw):
    return make_config(ShapesConfig, **kw)


def test_yolo_loss_perfect_prediction_closed_form():
    cfg = _cfg()
    G, A, C, T = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER
    yt = np.zeros((1, G, G, A, 5 + C), F32)
    yp = np.zeros((1, G, G, A, 5 + C), F32)
    tb = np.zeros((1, 1, 1, 1, T, 4), F32)
    row, col, a = 3, 2, 1
    box = [col + 0.5, row + 0.5, cfg.ANCHORS[2 * a], cfg.ANCHORS[2 * a + 1]]       # sigmoid(0)=.5, exp(0)*anchor
    yt[0, row, col, a, :4] = box
    yt[0, row, col, a, 4] = 1
    yt[0, row, col, a, 5 + 2] = 1
    tb[0, 0, 0, 0, 0] = box
    yp[..., 4] = -30.0                              # every other cell: confidence ~ 0
    yp[0, row, col, a, 4] = 30.0                    # the object cell: confidence ~ 1 == IoU 1
    yp[0, row, col, a, 5 + 2] = 30.0
    out = O.yolo_loss(yt, yp, tb, cfg)
    assert out["loss_xy"] == 0 and out["loss_wh"] == 0
    assert out["n_coord"] == 1 and out["n_class"] == 1
    # all other predictions have IoU<0.6 with the single true box except same-cell anchors of similar size:
    assert float(out["loss_conf"]) < 1e-12 and float(out["loss_class"]) < 1e-9 and out["recall"] > 0.999
    # now confidence 0.5 everywhere: loss_conf = sum((tconf - .5)^2 * mask)/n_conf/2 in closed form
    yp[..., 4] = 0.0
    out = O.yolo_loss(yt, yp, tb, cfg)
    n_conf = float(out["n_conf"])
    expected = ((1 - 0.5) ** 2 * cfg.OBJECT_SCALE + (n_conf - 1) * 0.25 * cfg.NO_OBJECT_SCALE) / (n_conf + 1e-6) / 2
    assert abs(float(out["loss_conf"]) - expected) < 1e-6


def test_yolo_loss_gradient_matches_finite_difference():
    O.set_precision(np.float64)
    try:
        cfg = _cfg(IMAGE_SHAPE=[64, 64, 3])
        rng = np.random.default_rng(2)
        G, A, C, T = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER
        yp = rng.standard_normal((2, G, G, A, 5 + C)) * 0.5
        yt = np.zeros_like(yp)
        tb = np.zeros((2, 1, 1, 1, T, 4))
        for b, (r, c, a) in enumerate([(0, 1, 2), (1, 0, 0)]):
            box = [c + 0.4, r + 0.6, 1.5, 1.2]
            yt[b, r, c, a, :4] = box
            yt[b, r, c, a, 4] = 1
            yt[b, r, c, a, 5 + 1 + b] = 1
            tb[b, 0, 0, 0, 0] = box
        g = O.yolo_loss(yt, yp, tb, cfg, want_grad=True)["grad"]
        num = _numgrad(lambda: float(O.yolo_loss(yt, yp, tb, cfg)["loss"]), yp, eps=1e-6)
        np.testing.assert_allclose(g, num, atol=2e-6)
    finally:
        O.set_precision(np.float32)


def test_yolo_loss_warmup_branch_closed_form_and_gradient():
    """model.py:193-207 (taken while seen < WARM_UP_BATCHES): with sigmoid(0) = .5 and exp(0) * anchor every EMPTY predictor sits exactly on
    its warm-up target (cell centre, anchor size), so only the one object predictor contributes to the coordinate terms -- with weight 1, not
    COORD_SCALE, over ALL G*G*A predictors; confidence / class terms equal the plain branch's.  Then the analytic gradient of the branch
    against finite differences in float64, and against torch autograd of the same graph (oracle/torch_ref.py)."""
    cfg = _cfg(COORD_SCALE=4.0)
    G, A, C, T = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER
    yt = np.zeros((1, G, G, A, 5 + C), F32)
    yp = np.zeros((1, G, G, A, 5 + C), F32)
    tb = np.zeros((1, 1, 1, 1, T, 4), F32)
    row, col, a = 3, 2, 1
    box = [col + 0.25, row + 0.5, cfg.ANCHORS[2 * a] * 2, cfg.ANCHORS[2 * a + 1]]
    yt[0, row, col, a, :4] = box
    yt[0, row, col, a, 4] = 1
    yt[0, row, col, a, 5 + 2] = 1
    tb[0, 0, 0, 0, 0] = box
    plain = O.yolo_loss(yt, yp, tb, cfg)
    warm = O.yolo_loss(yt, yp, tb, cfg, warmup=True)
    n_all = G * G * A
    assert plain["n_coord"] == 1 and warm["n_coord"] == n_all
    assert abs(float(warm["loss_xy"]) - 0.25 ** 2 / (n_all + 1e-6) / 2) < 1e-9
    assert abs(float(warm["loss_wh"]) - cfg.ANCHORS[2 * a] ** 2 / (n_all + 1e-6) / 2) < 1e-7
    assert abs(float(plain["loss_xy"]) - 0.25 ** 2 * 4.0 / (1 + 1e-6) / 2) < 1e-7          # the plain branch weighs the object cell by COORD_SCALE
    assert warm["loss_conf"] == plain["loss_conf"] and warm["loss_class"] == plain["loss_class"] and warm["n_conf"] == plain["n_conf"]
    O.set_precision(np.float64)
    try:
        cfg = _cfg(IMAGE_SHAPE=[64, 64, 3], COORD_SCALE=2.0)
        rng = np.random.default_rng(7)
        G, A, C, T = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER
        yp = rng.standard_normal((2, G, G, A, 5 + C)) * 0.5
        yt = np.zeros_like(yp)
        tb = np.zeros((2, 1, 1, 1, T, 4))
        for b, (r, c, a) in enumerate([(0, 1, 2), (1, 0, 0)]):
            box = [c + 0.4, r + 0.6, 1.5, 1.2]
            yt[b, r, c, a, :4] = box
            yt[b, r, c, a, 4] = 1
            yt[b, r, c, a, 5 + 1 + b] = 1
            tb[b, 0, 0, 0, 0] = box
        g = O.yolo_loss(yt, yp, tb, cfg, want_grad=True, warmup=True)["grad"]
        num = _numgrad(lambda: float(O.yolo_loss(yt, yp, tb, cfg, warmup=True)["loss"]), yp, eps=1e-6)
        np.testing.assert_allclose(g, num, atol=2e-6)
        import torch
        from oracle import torch_ref as TR
        ypt = torch.tensor(yp, dtype=torch.float64, requires_grad=True)
        loss_t, _ = TR.yolo_loss_t(torch.tensor(yt), ypt, torch.tensor(tb), cfg, warmup=True)
        loss_t.backward()
        assert abs(float(loss_t) - float(O.yolo_loss(yt, yp, tb, cfg, warmup=True)["loss"])) < 1e-12
        np.testing.assert_allclose(g, ypt.grad.numpy(), atol=1e-12)
        assert abs(float(O.yolo_loss(yt, yp, tb, cfg, warmup=True)["loss"]) - float(O.yolo_loss(yt, yp, tb, cfg)["loss"])) > 1e-3
    finally:
        O.set_precision(np.float32)


# ---------------------------------------------------------------- targets (model.py:457-602)
def test_mask_targets_empty_gt_all_negative_and_padded_quirk():
    cfg = _cfg(IMAGE_SHAPE=[64, 64, 3])
    R, T = cfg.TRAIN_ROIS_PER_IMAGE, cfg.TRUE_BOX_BUFFER
    rng = np.random.default_rng(3)
    c, s = rng.random((R, 2)), rng.random((R, 2)) * 0.4 + 0.05
    prop = np.concatenate([c - s / 2, c + s / 2], 1).astype(F32)[None]
    rois, cls, masks, npos = O.mask_targets(prop, np.zeros((1, T), np.int32), np.zeros((1, T, 4), np.int32),
                                            np.zeros((1, 64, 64, T), bool), cfg)
    assert npos[0] == 0 and cls.sum() == 0 and masks.sum() == 0
    assert np.array_equal(rois[0], prop[0])              # all negative -> original order
    # zero-padded pixel rows are NOT trimmed: norm_boxes turns them into [0,0,-1/(W-1),-1/(H-1)] (model.py:819, 1418)
    gtn = O.norm_boxes(np.zeros((T, 4), np.int32), 64, 64)
    assert np.abs(gtn).sum(1).min() > 0
    assert O.mask_bce(masks, cls, np.full((1, R, 28, 28, cfg.NUM_CLASSES), 0.3, F32)) == 0


def test_mask_targets_positive_first_and_mask_crop():
    cfg = _cfg(IMAGE_SHAPE=[64, 64, 3])
    R, T = cfg.TRAIN_ROIS_PER_IMAGE, cfg.TRUE_BOX_BUFFER
    gt_boxes = np.zeros((1, T, 4), np.int32)
    gt_boxes[0, 0] = [16, 8, 48, 40]                      # x1,y1,x2,y2
    gt_ids = np.zeros((1, T), np.int32)
    gt_ids[0, 0] = 2
    gt_masks = np.zeros((1, 64, 64, T), bool)
    gt_masks[0, 8:40, 16:48, 0] = True
    prop = np.tile(np.array([[0.9, 0.9, 0.95, 0.95]], F32), (R, 1))[None]
    g = O.norm_boxes(gt_boxes[0, :1], 64, 64)[0]
    prop[0, 5] = g                                         # exact match -> IoU 1
    prop[0, 9] = g + np.array([0.02, 0.0, 0.02, 0.0], F32)
    rois, cls, masks, npos = O.mask_targets(prop, gt_ids, gt_boxes, gt_masks, cfg)
    assert npos[0] == 2 and list(cls[0, :3]) == [2, 2, 0]
    assert np.array_equal(rois[0, 0], prop[0, 5]) and np.array_equal(rois[0, 1], prop[0, 9]) and np.array_equal(rois[0, 2], prop[0, 0])
    assert masks[0, 0].min() == 1.0                        # crop of the box interior is all ones
    assert set(np.unique(masks)) <= {0.0, 1.0} and masks[0, 2:].sum() == 0


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2 ** 31 - 1))
def test_partition_properties(seed):
    cfg = _cfg(IMAGE_SHAPE=[64, 64, 3])
    R, T = cfg.TRAIN_ROIS_PER_IMAGE, cfg.TRUE_BOX_BUFFER
    rng = np.random.default_rng(seed)
    c, s = rng.random((R, 2)), rng.random((R, 2)) * 0.6 + 0.05
    prop = np.concatenate([c - s / 2, c + s / 2], 1).astype(F32)
    n = int(rng.integers(0, 4))
    gt_boxes = np.zeros((T, 4), np.int32)
    gt_ids = np.zeros(T, np.int32)
    for k in range(n):
        x1, y1 = rng.integers(0, 40, 2)
        gt_boxes[k] = [x1, y1, x1 + rng.integers(8, 24), y1 + rng.integers(8, 24)]
        gt_ids[k] = rng.integers(1, 4)
        prop[rng.integers(0, R)] = O.norm_boxes(gt_boxes[k:k + 1], 64, 64)[0]
    masks = np.zeros((64, 64, T), bool)
    rois, cls, tm, npos = O.mask_targets_one(prop, gt_ids, O.norm_boxes(gt_boxes, 64, 64), masks, cfg)
    assert rois.shape == (R, 4) and sorted(map(tuple, rois)) == sorted(map(tuple, prop))      # a permutation
    assert (cls[:npos] > 0).all() and (cls[npos:] == 0).all()
    iou = O.overlaps(prop, O.norm_boxes(gt_boxes, 64, 64))
    assert np.all((iou >= 0) & (iou <= 1.0000001))


# ---------------------------------------------------------------- mask BCE / Adam
def test_mask_bce_closed_form():
    tm = np.zeros((1, 2, 2, 2), F32)
    tm[0, 0, 0, 0] = 1
    ids = np.array([[3, 0]], np.int32)
    pr = np.full((1, 2, 2, 2, 4), 0.25, F32)
    loss, g = O.mask_bce(tm, ids, pr, want_grad=True)
    exp = (-np.log(0.25) - 3 * np.log(0.75)) / 4
    assert abs(float(loss) - exp) < 1e-6
    assert g[0, 1].sum() == 0 and np.count_nonzero(g[0, 0, :, :, :3]) == 0       # only the class-3 channel of the positive ROI
    np.testing.assert_allclose(g[0, 0, :, :, 3], [[(0.25 - 1) / (0.25 * 0.75) / 4, 0.25 / (0.25 * 0.75) / 4]] * 1 +
                               [[0.25 / (0.25 * 0.75) / 4] * 2], rtol=1e-5)


def test_adam_three_steps_by_hand():
    p, m, v = np.array([1.0], F32), np.zeros(1, F32), np.zeros(1, F32)
    gs, lr, b1, b2, eps = [0.5, -0.25, 1.0], 1e-3, 0.9, 0.999, 1e-8
    pe, me, ve = 1.0, 0.0, 0.0
    for t, g in enumerate(gs, 1):
        p, m, v = O.adam_step(p, np.array([g], F32), m, v, t)
        me = b1 * me + (1 - b1) * g
        ve = b2 * ve + (1 - b2) * g * g
        pe -= lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * me / (np.sqrt(ve) + eps)
    assert abs(float(p[0]) - pe) < 1e-7


# ---------------------------------------------------------------- host encoding (myolo_utils.py:727-860, 247-271)
def test_encode_box_kat():
    """box [50,60,120,160] at 224/G=7 -> cell (row 3, col 2), [2.65625, 3.4375, 2.1875, 3.125]."""
    cfg = _cfg()
    masks = np.zeros((224, 224, 1), bool)
    masks[60:160, 50:120, 0] = True
    img = np.zeros((224, 224, 3), np.uint8)
    out = O.encode_batch([(img, np.array([2], np.int32), np.array([[50, 60, 120, 160]], np.int32), masks)], cfg)
    y_true, tb = out[2], out[1]
    cell = y_true[0, 3, 2]
    a = int(np.argmax(cell[:, 4]))
    assert cell[:, 4].sum() == 1 and y_true[..., 4].sum() == 1
    assert np.array_equal(cell[a, :4], [2.65625, 3.4375, 2.1875, 3.125]) and cell[a, 5 + 2] == 1
    # best anchor by hand: IoU of (2.1875 x 3.125) with the three Shapes anchors, origin-anchored
    ious = [min(2.1875, aw) * min(3.125, ah) / (2.1875 * 3.125 + aw * ah - min(2.1875, aw) * min(3.125, ah))
            for aw, ah in np.asarray(cfg.ANCHORS).reshape(-1, 2)]
    assert a == int(np.argmax(ious)) == 1
    assert np.array_equal(tb[0, 0, 0, 0, 0], [2.65625, 3.4375, 2.1875, 3.125]) and tb[0, 0, 0, 0, 1:].sum() == 0


def test_extract_bboxes_kat():
    m = np.zeros((12, 12, 2), bool)
    m[3:6, 7:10, 0] = True
    assert np.array_equal(O.extract_bboxes(m), [[7, 3, 10, 6], [0, 0, 0, 0]])


# ---------------------------------------------------------------- whole step: analytic backward == autograd (float64)
def test_full_step_backward_matches_torch_autograd_f64():
    import torch
    from myolo.shapes import make_shapes_samples
    from myolo.myolo_utils import BatchGenerator
    from oracle.torch_ref import TorchRef
    cfg = make_config(ShapesConfig, IMAGE_SHAPE=[96, 96, 3], ALPHA=0.25, BATCH_SIZE=2)
    samples = make_shapes_samples(2, cfg)
    batch, _ = BatchGenerator(samples, cfg, 'training', shuffle=False, norm=True)[0]
    P = np_model.init_params(cfg, seed=0, bias_scale=0.1)
    O.set_precision(np.float64)
    try:
        P64 = {k: v.astype(np.float64) for k, v in P.items()}
        b64 = [b.astype(np.float64) if b.dtype == np.float32 else b for b in batch]
        out = np_model.train_step_fwd_bwd(P64, b64, cfg)
    finally:
        O.set_precision(np.float32)
    ref = TorchRef(P, cfg, torch.float64).train_step(batch)
    assert out["n_pos"].sum() >= 1, "case must exercise the mask loss"
    assert abs(float(out["loss"]) - ref["loss"]) < 1e-7
    for k, g in ref["grads"].items():
        if np.abs(g).max() > 1e-12:
            assert float(np.abs(out["grads"][k] - g).max() / np.abs(g).max()) < 1e-5, k


def test_parameter_count_matches_survey():
    """7,314,481 trainable (N_BOX=5, C=4) / 7,296,031 (N_BOX=3, C=4) at alpha=1."""
    from myolo.config import ShapesHeadConfig
    for base, n in ((ShapesConfig, 7296031), (ShapesHeadConfig, 7314481)):
        cfg = make_config(base)
        P = np_model.init_params(cfg)
        assert sum(P[k].size for k in np_model.trainable_names(P)) == n
