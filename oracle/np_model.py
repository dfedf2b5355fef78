"""
ORACLE (test infrastructure, NOT product code) -- the whole Mask-YOLO training step and
inference forward, composed from oracle/np_ops.py.  PARITY PARTLY PINNED (see np_ops.py header).

Follows MaskYOLO.build model.py:787-941 (training branch :872-904, inference :922-936),
mobilenet_graph :55-79, yolo_branch_graph :249-278, build_mask_graph :668-715,
compile :1062-1094 (loss = mean(yolo_sum_loss)*1 + mean(myolo_mask_loss)*1, Adam).
"""
import numpy as np
from . import np_ops as O


BACKBONE_BLOCKS = [(64, 1), (64, 2), (128, 1), (256, 2), (256, 1), (512, 1)]          # model.py:68-77
YOLO_BLOCKS = [(512, 2), (512, 1), (512, 1), (512, 1), (512, 1), (512, 1), (1024, 2), (1024, 1)]  # :256-268
MASK_FILTERS = 256                                                                     # :688-711


def layer_table(cfg):
    """Ordered (name, kind, shape...) list in graph-definition order."""
    a = cfg.ALPHA
    C = cfg.NUM_CLASSES
    t = [("conv1", "conv", (3, 3, 3, int(32 * a))), ("conv1_bn", "bn", int(32 * a))]
    cin = int(32 * a)
    bid = 1
    for f, s in BACKBONE_BLOCKS:
        co = int(f * a)
        t += [("conv_dw_%d" % bid, "dw", (3, 3, cin)), ("conv_dw_%d_bn" % bid, "bn", cin),
              ("conv_pw_%d" % bid, "conv", (1, 1, cin, co)), ("conv_pw_%d_bn" % bid, "bn", co)]
        cin = co
        bid += 1
    c4 = cin
    t += [("feature_map", "convb", (3, 3, c4, cfg.TOP_FEATURE_MAP_DEPTH))]
    for f, s in YOLO_BLOCKS:
        co = int(f * a)
        t += [("conv_dw_%d" % bid, "dw", (3, 3, cin)), ("conv_dw_%d_bn" % bid, "bn", cin),
              ("conv_pw_%d" % bid, "conv", (1, 1, cin, co)), ("conv_pw_%d_bn" % bid, "bn", co)]
        cin = co
        bid += 1
    t += [("conv_23", "convb", (1, 1, cin, cfg.N_BOX * (5 + C)))]
    cm = cfg.TOP_FEATURE_MAP_DEPTH
    for i in range(1, 5):
        t += [("myolo_mask_conv%d" % i, "convb", (3, 3, cm, MASK_FILTERS)), ("myolo_mask_bn%d" % i, "bn", MASK_FILTERS)]
        cm = MASK_FILTERS
    t += [("myolo_mask_deconv", "deconv", (2, 2, MASK_FILTERS, MASK_FILTERS)),
          ("myolo_mask", "convb", (1, 1, MASK_FILTERS, C))]
    return t


def _glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(O.F32)


def init_params(cfg, seed=0, bias_scale=0.0):
    """glorot_uniform kernels, zero biases, BN gamma=1 beta=0 mean=0 var=1 (Keras defaults).
    bias_scale>0 perturbs biases / BN affine so tests exercise them."""
    rng = np.random.default_rng(seed)
    P = {}
    for name, kind, shp in layer_table(cfg):
        if kind in ("conv", "convb"):
            kh, kw, ci, co = shp
            P[name + "/kernel"] = _glorot(rng, shp, kh * kw * ci, kh * kw * co)
            if kind == "convb":
                P[name + "/bias"] = (rng.standard_normal(co) * bias_scale).astype(O.F32)
        elif kind == "dw":
            kh, kw, c = shp
            P[name + "/depthwise_kernel"] = _glorot(rng, shp, kh * kw * c, kh * kw)
        elif kind == "deconv":
            kh, kw, co, ci = shp
            P[name + "/kernel"] = _glorot(rng, shp, kh * kw * co, kh * kw * ci)
            P[name + "/bias"] = (rng.standard_normal(co) * bias_scale).astype(O.F32)
        elif kind == "bn":
            c = shp
            P[name + "/gamma"] = (1 + rng.standard_normal(c) * bias_scale).astype(O.F32)
            P[name + "/beta"] = (rng.standard_normal(c) * bias_scale).astype(O.F32)
            P[name + "/moving_mean"] = (rng.standard_normal(c) * bias_scale).astype(O.F32)
            P[name + "/moving_variance"] = (1 + np.abs(rng.standard_normal(c)) * bias_scale).astype(O.F32)
    return P


def trainable_names(P):
    return [k for k in P if not (k.endswith("moving_mean") or k.endswith("moving_variance"))]


# ---------------------------------------------------------------------------
class Tape(object):
    """Forward pass that records what backward needs."""

    def __init__(self, P, cfg, training=True):
        self.P, self.cfg, self.training = P, cfg, training
        self.c = {}
        self.moving = {}

    # -- BN + activation ----------------------------------------------------
    def bn_act(self, name, x, act, frozen=False):
        P = self.P
        g, b = P[name + "/gamma"], P[name + "/beta"]
        self.c[name + "/x"] = x          # pre-BN tensor (tests teacher-force the GPU backward's activation masks with it)
        if self.training and not frozen:
            y, cache = O.bn_train(x, g, b)
            n = x.size // x.shape[-1]
            self.moving[name] = O.bn_moving_update(P[name + "/moving_mean"], P[name + "/moving_variance"],
                                                   cache[2], cache[3], n)
            self.c[name] = ("train", cache)
        else:
            y, cache = O.bn_infer(x, g, b, P[name + "/moving_mean"], P[name + "/moving_variance"])
            self.c[name] = ("infer", cache)
        a = O.relu6(y) if act == "relu6" else O.relu(y)
        self.c[name + "/act"] = (act, a)
        return a

    def bn_act_bwd(self, name, da, G):
        act, a = self.c[name + "/act"]
        dy = O.relu6_bwd(a, da) if act == "relu6" else O.relu_bwd(a, da)
        mode, cache = self.c[name]
        g = self.P[name + "/gamma"]
        if mode == "train":
            dx, dg, db = O.bn_train_bwd(cache, g, dy)
        else:
            dx, dg, db = O.bn_infer_bwd(cache, g, dy)
        G[name + "/gamma"] = dg
        G[name + "/beta"] = db
        return dx

    # -- depthwise-separable block ------------------------------------------
    def dw_block(self, bid, x, stride):
        P = self.P
        self.c["dw%d/x" % bid] = x
        y = O.dwconv3x3(x, P["conv_dw_%d/depthwise_kernel" % bid], stride)
        a = self.bn_act("conv_dw_%d_bn" % bid, y, "relu6")
        self.c["pw%d/x" % bid] = a
        y = O.conv2d(a, P["conv_pw_%d/kernel" % bid])
        return self.bn_act("conv_pw_%d_bn" % bid, y, "relu6")

    def dw_block_bwd(self, bid, da, stride, G):
        P = self.P
        dy = self.bn_act_bwd("conv_pw_%d_bn" % bid, da, G)
        dx, dw, _ = O.conv2d_bwd(self.c["pw%d/x" % bid], P["conv_pw_%d/kernel" % bid], dy)
        G["conv_pw_%d/kernel" % bid] = dw
        dy = self.bn_act_bwd("conv_dw_%d_bn" % bid, dx, G)
        dx, dw = O.dwconv3x3_bwd(self.c["dw%d/x" % bid], P["conv_dw_%d/depthwise_kernel" % bid], dy, stride)
        G["conv_dw_%d/depthwise_kernel" % bid] = dw
        return dx

    # -- trunk ---------------------------------------------------------------
    def trunk(self, images):
        P = self.P
        self.c["conv1/x"] = images
        y = O.conv2d(images, P["conv1/kernel"], stride=2, pads=O.conv1_pads())
        x = self.bn_act("conv1_bn", y, "relu6")
        bid = 1
        for f, s in BACKBONE_BLOCKS:
            x = self.dw_block(bid, x, s)
            bid += 1
        C4 = x
        self.c["feature_map/x"] = C4
        Fm = O.conv2d(C4, P["feature_map/kernel"], pads=O.same_pads_3x3(), bias=P["feature_map/bias"])
        for f, s in YOLO_BLOCKS:
            x = self.dw_block(bid, x, s)
            bid += 1
        self.c["conv_23/x"] = x
        y = O.conv2d(x, P["conv_23/kernel"], bias=P["conv_23/bias"])
        cfg = self.cfg
        yolo_out = y.reshape(y.shape[0], cfg.GRID_H, cfg.GRID_W, cfg.N_BOX, 5 + cfg.NUM_CLASSES)
        return C4, Fm, yolo_out

    def trunk_bwd(self, dF, dyolo, G):
        P = self.P
        B = dyolo.shape[0]
        dy = dyolo.reshape(B, self.cfg.GRID_H, self.cfg.GRID_W, -1)
        dx, dw, db = O.conv2d_bwd(self.c["conv_23/x"], P["conv_23/kernel"], dy)
        G["conv_23/kernel"], G["conv_23/bias"] = dw, db
        bid = 14
        for f, s in reversed(YOLO_BLOCKS):
            dx = self.dw_block_bwd(bid, dx, s, G)
            bid -= 1
        dC4 = dx
        dx, dw, db = O.conv2d_bwd(self.c["feature_map/x"], P["feature_map/kernel"], dF, pads=O.same_pads_3x3())
        G["feature_map/kernel"], G["feature_map/bias"] = dw, db
        dC4 = dC4 + dx
        dx = dC4
        for f, s in reversed(BACKBONE_BLOCKS):
            dx = self.dw_block_bwd(bid, dx, s, G)
            bid -= 1
        dy = self.bn_act_bwd("conv1_bn", dx, G)
        _, dw, _ = O.conv2d_bwd(self.c["conv1/x"], P["conv1/kernel"], dy, stride=2, pads=O.conv1_pads(), need_dx=False)
        G["conv1/kernel"] = dw

    # -- mask head -----------------------------------------------------------
    def mask_head(self, Fm, rois):
        """rois [B,R,4]; returns pred masks [B,R,28,28,C] (post-sigmoid)."""
        P, cfg = self.P, self.cfg
        B, R = rois.shape[:2]
        boxes = O.roi_boxes_to_crop_order(rois.reshape(-1, 4), cfg.ROI_BOX_ORDER)
        bidx = np.repeat(np.arange(B), R)
        ps = cfg.MASK_POOL_SIZE
        x = O.crop_and_resize(Fm, boxes, bidx, (ps, ps))
        self.c["roi"] = (boxes, bidx, Fm.shape)
        self.c["roi/out"] = x
        for i in range(1, 5):
            self.c["mconv%d/x" % i] = x
            y = O.conv2d(x, P["myolo_mask_conv%d/kernel" % i], pads=O.same_pads_3x3(),
                         bias=P["myolo_mask_conv%d/bias" % i])
            # bn1: no training= argument (model.py:690) -> batch stats in training;
            # bn2-4: training=train_bn=False (model.py:696,702,708) -> frozen
            x = self.bn_act("myolo_mask_bn%d" % i, y, "relu", frozen=(i > 1))
        self.c["deconv/x"] = x
        d = O.relu(O.deconv2x2s2(x, P["myolo_mask_deconv/kernel"], P["myolo_mask_deconv/bias"]))
        self.c["deconv/out"] = d
        z = O.conv2d(d, P["myolo_mask/kernel"], bias=P["myolo_mask/bias"])
        p = O.sigmoid(z)
        self.c["mask/p"] = p
        return p.reshape(B, R, p.shape[1], p.shape[2], p.shape[3])

    def mask_head_bwd(self, dp, G):
        P = self.P
        p = self.c["mask/p"]
        dz = (dp.reshape(p.shape).astype(np.float64) * p * (1 - p)).astype(O.F32)
        dd, dw, db = O.conv2d_bwd(self.c["deconv/out"], P["myolo_mask/kernel"], dz)
        G["myolo_mask/kernel"], G["myolo_mask/bias"] = dw, db
        dd = O.relu_bwd(self.c["deconv/out"], dd)
        dx, dw, db = O.deconv2x2s2_bwd(self.c["deconv/x"], P["myolo_mask_deconv/kernel"], dd)
        G["myolo_mask_deconv/kernel"], G["myolo_mask_deconv/bias"] = dw, db
        for i in range(4, 0, -1):
            dy = self.bn_act_bwd("myolo_mask_bn%d" % i, dx, G)
            dx, dw, db = O.conv2d_bwd(self.c["mconv%d/x" % i], P["myolo_mask_conv%d/kernel" % i], dy,
                                      pads=O.same_pads_3x3())
            G["myolo_mask_conv%d/kernel" % i], G["myolo_mask_conv%d/bias" % i] = dw, db
        boxes, bidx, fshape = self.c["roi"]
        return O.crop_and_resize_bwd_image(dx, boxes, bidx, fshape)


def train_step_fwd_bwd(P, batch, cfg, warmup=False):
    """One training forward+backward (no optimiser).  batch = the six arrays of model.py:896-897.
    Returns dict(outputs..., loss, grads, moving).  warmup: the loss's warm-up branch (np_ops.yolo_loss; the caller keeps `seen`)."""
    images, true_boxes, y_true, gt_ids, gt_boxes, gt_masks = batch
    T = Tape(P, cfg, training=True)
    C4, Fm, yolo_out = T.trunk(images.astype(O.F32))
    proposals = O.yolo_decode(yolo_out, cfg.ANCHORS, cfg.GRID_W)
    rois, tcls, tmask, npos = O.mask_targets(proposals, gt_ids, gt_boxes, gt_masks, cfg)
    pred = T.mask_head(Fm, rois)
    yl = O.yolo_loss(y_true, yolo_out, true_boxes, cfg, want_grad=True, warmup=warmup)
    ml, dpred = O.mask_bce(tmask, tcls, pred, want_grad=True)
    w1 = O.F32(cfg.LOSS_WEIGHTS.get("yolo_sum_loss", 1.))
    w2 = O.F32(cfg.LOSS_WEIGHTS.get("myolo_mask_loss", 1.))
    loss = O.F32(yl["loss"] * w1 + ml * w2)
    G = {}
    dF = T.mask_head_bwd(dpred * w2, G)
    T.trunk_bwd(dF, yl["grad"] * w1, G)
    return dict(yolo_output=yolo_out, yolo_proposals=proposals, output_rois=rois, myolo_mask=pred,
                target_class_ids=tcls, target_mask=tmask, n_pos=npos,
                yolo_sum_loss=yl["loss"], mask_loss=ml, loss=loss, yolo_terms=yl,
                feature_map=Fm, C4=C4, grads=G, moving=T.moving, tape=T)


def val_step_fwd(P, batch, cfg, warmup=False):
    """Validation forward of fit_generator (model.py:1053-1054: validation_data=val_generator): the training graph
    (model.py:872-904) evaluated in Keras' test phase -- K.learning_phase() = 0, so EVERY BatchNormalization, bn1 of the
    mask head included (model.py:690 has no training= argument and therefore follows the learning phase), normalises with
    its moving statistics; nothing is updated.  Returns the loss terms Keras averages into val_loss."""
    images, true_boxes, y_true, gt_ids, gt_boxes, gt_masks = batch
    T = Tape(P, cfg, training=False)
    C4, Fm, yolo_out = T.trunk(images.astype(O.F32))
    proposals = O.yolo_decode(yolo_out, cfg.ANCHORS, cfg.GRID_W)
    rois, tcls, tmask, npos = O.mask_targets(proposals, gt_ids, gt_boxes, gt_masks, cfg)
    pred = T.mask_head(Fm, rois)
    yl = O.yolo_loss(y_true, yolo_out, true_boxes, cfg, want_grad=False, warmup=warmup)
    ml = O.mask_bce(tmask, tcls, pred, want_grad=False)
    ml = ml[0] if isinstance(ml, tuple) else ml
    w1 = O.F32(cfg.LOSS_WEIGHTS.get("yolo_sum_loss", 1.))
    w2 = O.F32(cfg.LOSS_WEIGHTS.get("myolo_mask_loss", 1.))
    return dict(yolo_output=yolo_out, output_rois=rois, target_class_ids=tcls, myolo_mask=pred, n_pos=npos,
                yolo_sum_loss=yl["loss"], mask_loss=ml, loss=O.F32(yl["loss"] * w1 + ml * w2), yolo_terms=yl)


def yolo_step_fwd_bwd(P, batch, cfg, warmup=False):
    """'yolo' mode training step (model.py:906-920, compile :1084-1085): loss = mean(yolo_sum_loss) only."""
    images, true_boxes, y_true = batch[:3]
    T = Tape(P, cfg, training=True)
    C4, Fm, yolo_out = T.trunk(images.astype(O.F32))
    yl = O.yolo_loss(y_true, yolo_out, true_boxes, cfg, want_grad=True, warmup=warmup)
    w1 = O.F32(cfg.LOSS_WEIGHTS.get("yolo_sum_loss", 1.))
    G = {}
    T.trunk_bwd(np.zeros_like(Fm), yl["grad"] * w1, G)
    return dict(yolo_output=yolo_out, yolo_sum_loss=yl["loss"], loss=O.F32(yl["loss"] * w1), grads=G)


def inference_fwd(P, images, cfg):
    """inference graph model.py:922-936: outputs [yolo_output, detections, myolo_mask]."""
    T = Tape(P, cfg, training=False)
    C4, Fm, yolo_out = T.trunk(images.astype(O.F32))
    det = O.yolo_detections(yolo_out, cfg.ANCHORS, cfg.GRID_W)
    pred = T.mask_head(Fm, det[..., :4])
    return dict(yolo_output=yolo_out, detections=det, myolo_mask=pred, feature_map=Fm)


def adam_update(P, G, state, t, lr):
    """Keras Adam over every trainable tensor; state = {name: (m, v)}."""
    for k in trainable_names(P):
        m, v = state.get(k, (np.zeros_like(P[k]), np.zeros_like(P[k])))
        P[k], m, v = O.adam_step(P[k], G[k], m, v, t, lr=lr)
        state[k] = (m, v)
    return P, state
