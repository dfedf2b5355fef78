"""
ORACLE (test infrastructure, NOT product code) -- numpy restatement of every
arithmetic op on the Mask-YOLO forward/backward hot path.

PARITY PARTLY PINNED: the reference delegates all arithmetic to TensorFlow-1.x /
Keras-2.x / keras_applications (un-vendored, un-pinned, not installable here;
SURVEY.md section 8(c)) and has no tests, golden vectors or weights.  This file
restates the documented behaviour of those ops -- never checked against TensorFlow
itself.  It is pinned by vectors quoted from the dependencies' own test suites
(tests/test_third_party_kats.py: TF crop_and_resize_op_test.cc, adam_test.py,
fused_batch_norm x Keras' moving-variance factor, keras_applications padding,
tf.round), by fixtures generated with the real scikit-image / h5py of this image's
Anaconda Python (tests/golden/make_*_fixture.py: oracle/np_post.py's resize), by
the hand-derived known-answer tests in tests/test_oracle_kat.py and by agreement
with an independent torch-CPU autograd composition (oracle/torch_ref.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  The product (mask-yolo_amd/) never does.

Every function cites the reference line(s) it follows (paths relative to
/root/reference/).  Layout everywhere: NHWC float32, Keras kernel layouts
(Conv [kh,kw,Cin,Cout]; Depthwise [kh,kw,C]; Conv2DTranspose [kh,kw,Cout,Cin]).
"""
import numpy as np

F32 = np.float32


def set_precision(dt):
    """Tests only: rebind the working dtype (np.float64 turns this file into the
    'float64 truth' variant used to pin the analytic backward against torch autograd)."""
    global F32, BN_EPS, BN_MOMENTUM, BCE_EPS
    F32 = dt
    BN_EPS = dt(np.float32(1e-3)) if dt is np.float32 else dt(1e-3)
    BN_MOMENTUM = dt(0.99)
    BCE_EPS = dt(np.float32(1e-7)) if dt is np.float32 else dt(1e-7)


# ---------------------------------------------------------------------------
# Deterministic fp32 exp / sigmoid shared (as a *specification*) with the HIP
# kernels in mask-yolo_amd/csrc/exact_math.h: every operation below is a single
# IEEE-754 binary32 operation in this exact order (no fused multiply-add), so
# CPU and GPU agree bit-for-bit and the integer decisions downstream
# (IoU >= 0.5 partition, argmax) are bit-exact.  Cephes expf coefficients;
# max error vs libm expf <= 2 ulp (tests/test_oracle_kat.py).
# ---------------------------------------------------------------------------
def det_expf(x):
    if F32 is not np.float32:
        return np.exp(np.asarray(x, dtype=F32))
    x = np.asarray(x, dtype=F32)
    x = np.minimum(np.maximum(x, F32(-86.0)), F32(88.0))
    n = np.rint(x * F32(1.44269504))
    r = x - n * F32(0.693359375)
    r = r - n * F32(-2.12194440e-4)
    p = np.full_like(r, F32(1.9875691500e-4))
    p = p * r + F32(1.3981999507e-3)
    p = p * r + F32(8.3334519073e-3)
    p = p * r + F32(4.1665795894e-2)
    p = p * r + F32(1.6666665459e-1)
    p = p * r + F32(5.0000001201e-1)
    y = (p * (r * r) + r) + F32(1.0)
    scale = ((n.astype(np.int32) + 127) << 23).astype(np.int32).view(F32)
    return (y * scale).astype(F32)


def det_sigmoid(x):
    x = np.asarray(x, dtype=F32)
    return (F32(1.0) / (F32(1.0) + det_expf(-x))).astype(F32)


# ---------------------------------------------------------------------------
# activations
# ---------------------------------------------------------------------------
def relu6(x):
    """model.py:38-39  backend.relu(x, max_value=6)."""
    return np.minimum(np.maximum(x, F32(0)), F32(6))


def relu6_bwd(y, dy):
    """gradient passes where 0 < y < 6 (tf.nn.relu6 grad)."""
    return dy * ((y > 0) & (y < 6))


def relu(x):
    return np.maximum(x, F32(0))


def relu_bwd(y, dy):
    return dy * (y > 0)


# ---------------------------------------------------------------------------
# dense convolution (Conv2D) -- model.py:45-50 (conv1), :848 (feature_map),
# :271 (conv_23), :688-709 (mask convs), :713 (myolo_mask)
# ---------------------------------------------------------------------------
def _patches(xp, kh, kw, stride, Ho, Wo):
    N, Hp, Wp, C = xp.shape
    s = xp.strides
    return np.lib.stride_tricks.as_strided(
        xp, shape=(N, Ho, Wo, kh, kw, C),
        strides=(s[0], s[1] * stride, s[2] * stride, s[1], s[2], s[3]), writeable=False)


def conv2d(x, w, stride=1, pads=(0, 0, 0, 0), bias=None, acc=None):
    """x [N,H,W,Ci], w [kh,kw,Ci,Co]; pads=(top,bottom,left,right) zero padding, then VALID."""
    acc = acc or F32
    kh, kw, Ci, Co = w.shape
    t, b, l, r = pads
    xp = np.pad(x, ((0, 0), (t, b), (l, r), (0, 0)))
    N, Hp, Wp, _ = xp.shape
    Ho = (Hp - kh) // stride + 1
    Wo = (Wp - kw) // stride + 1
    P = _patches(xp, kh, kw, stride, Ho, Wo).reshape(N * Ho * Wo, kh * kw * Ci)
    y = (P.astype(acc) @ w.reshape(kh * kw * Ci, Co).astype(acc)).astype(F32)
    if bias is not None:
        y = y + bias.astype(F32)
    return y.reshape(N, Ho, Wo, Co)


def conv2d_bwd(x, w, dy, stride=1, pads=(0, 0, 0, 0), acc=None, need_dx=True):
    """returns dx, dw, db for conv2d above."""
    acc = acc or F32
    kh, kw, Ci, Co = w.shape
    t, b, l, r = pads
    xp = np.pad(x, ((0, 0), (t, b), (l, r), (0, 0)))
    N, Hp, Wp, _ = xp.shape
    Ho, Wo = dy.shape[1], dy.shape[2]
    P = _patches(xp, kh, kw, stride, Ho, Wo).reshape(N * Ho * Wo, kh * kw * Ci)
    dy2 = dy.reshape(N * Ho * Wo, Co)
    dw = (P.astype(acc).T @ dy2.astype(acc)).astype(F32).reshape(kh, kw, Ci, Co)
    db = dy2.astype(acc).sum(0).astype(F32)
    dx = None
    if need_dx:
        dP = (dy2.astype(acc) @ w.reshape(kh * kw * Ci, Co).astype(acc).T).astype(F32)
        dP = dP.reshape(N, Ho, Wo, kh, kw, Ci)
        dxp = np.zeros_like(xp)
        for i in range(kh):
            for j in range(kw):
                dxp[:, i:i + stride * Ho:stride, j:j + stride * Wo:stride, :] += dP[:, :, :, i, j, :]
        dx = dxp[:, t:Hp - b, l:Wp - r, :]
    return dx, dw, db


def conv1_pads():
    """model.py:45 ZeroPadding2D((1,1)) then 3x3 stride-2 VALID."""
    return (1, 1, 1, 1)


def same_pads_3x3():
    return (1, 1, 1, 1)


# ---------------------------------------------------------------------------
# depthwise 3x3 (keras_applications.mobilenet._depthwise_conv_block, imported
# at model.py:19, used at model.py:68-77,256-268):
#   stride 1 -> DepthwiseConv2D 3x3 'same';
#   stride 2 -> ZeroPadding2D(((0,1),(0,1))) + 'valid'  (keras_applications>=1.0.5)
# ---------------------------------------------------------------------------
def dw_pads(stride):
    return (1, 1, 1, 1) if stride == 1 else (0, 1, 0, 1)


def dwconv3x3(x, w, stride):
    """x [N,H,W,C], w [3,3,C]."""
    t, b, l, r = dw_pads(stride)
    xp = np.pad(x, ((0, 0), (t, b), (l, r), (0, 0)))
    N, Hp, Wp, C = xp.shape
    Ho = (Hp - 3) // stride + 1
    Wo = (Wp - 3) // stride + 1
    y = np.zeros((N, Ho, Wo, C), F32)
    for i in range(3):
        for j in range(3):
            y += xp[:, i:i + stride * Ho:stride, j:j + stride * Wo:stride, :] * w[i, j]
    return y


def dwconv3x3_bwd(x, w, dy, stride):
    t, b, l, r = dw_pads(stride)
    xp = np.pad(x, ((0, 0), (t, b), (l, r), (0, 0)))
    N, Hp, Wp, C = xp.shape
    Ho, Wo = dy.shape[1], dy.shape[2]
    dxp = np.zeros_like(xp)
    dw = np.zeros_like(w)
    for i in range(3):
        for j in range(3):
            sl = (slice(None), slice(i, i + stride * Ho, stride), slice(j, j + stride * Wo, stride), slice(None))
            dxp[sl] += dy * w[i, j]
            dw[i, j] = (xp[sl].astype(np.float64) * dy).sum((0, 1, 2)).astype(F32)
    return dxp[:, t:Hp - b, l:Wp - r, :], dw


# ---------------------------------------------------------------------------
# BatchNormalization(axis=-1), eps=1e-3, momentum=0.99 (Keras defaults; model.py:51,
# keras_applications blocks, model.py:690-708)
# ---------------------------------------------------------------------------
BN_EPS = F32(1e-3)
BN_MOMENTUM = F32(0.99)


def bn_train(x, gamma, beta):
    """batch statistics over all axes but the last; biased variance for normalisation."""
    x2 = x.reshape(-1, x.shape[-1]).astype(np.float64)
    mean = x2.mean(0)
    var = x2.var(0)
    rstd = 1.0 / np.sqrt(var + float(BN_EPS))
    xhat = ((x2 - mean) * rstd)
    y = (xhat * gamma + beta).astype(F32).reshape(x.shape)
    return y, (xhat.astype(F32).reshape(x.shape), rstd.astype(F32), mean.astype(F32), var.astype(F32))


def bn_train_bwd(cache, gamma, dy):
    xhat, rstd, _, _ = cache
    C = dy.shape[-1]
    d2 = dy.reshape(-1, C).astype(np.float64)
    xh = xhat.reshape(-1, C).astype(np.float64)
    n = d2.shape[0]
    dbeta = d2.sum(0)
    dgamma = (d2 * xh).sum(0)
    dx = (gamma * rstd / n) * (n * d2 - dbeta - xh * dgamma)
    return dx.astype(F32).reshape(dy.shape), dgamma.astype(F32), dbeta.astype(F32)


# Which Keras / TensorFlow pair the moving-variance update restates (the reference pins neither: model.py:27-28 only
# assert TF >= 1.3 and Keras >= 2.0.8; `get_keras_submodule`, model.py:18, implies Keras 2.2.x + keras_applications >= 1.0.5).
#   True  (default): Keras 2.2.x on TensorFlow 1.x.  Every BatchNormalization of this graph sees a 4-D NHWC tensor with
#          axis=-1 (TimeDistributed reshapes to [B*R,14,14,C]), which keras/backend/tensorflow_backend.py
#          normalize_batch_in_training sends to tf.nn.fused_batch_norm; its `batch_variance` output is the Bessel-corrected
#          estimate (tensorflow/core/kernels/fused_batch_norm_op.cc: variance * rest_size / (rest_size - 1)), and Keras'
#          BatchNormalization.call multiplies whatever the backend returns by sample_size / (sample_size - (1 + epsilon)).
#   False: the non-fused backend path: Keras' factor on the biased variance only.
BN_FUSED_TF_VARIANCE = True


def bn_moving_update(moving_mean, moving_var, mean, var, n, fused_tf=None):
    """Keras 2.2 BatchNormalization.call: moving = moving*m + batch*(1-m), with the batch variance rescaled by
    n/(n-(1+eps)) before the update -- applied to TF's fused-path (Bessel-corrected) variance when fused_tf."""
    if fused_tf is None:
        fused_tf = BN_FUSED_TF_VARIANCE
    if fused_tf and n > 1:
        var = var * (F32(n) / (F32(n) - F32(1.0)))
    var_u = var * (F32(n) / (F32(n) - (F32(1.0) + BN_EPS)))
    mm = moving_mean * BN_MOMENTUM + mean * (F32(1) - BN_MOMENTUM)
    mv = moving_var * BN_MOMENTUM + var_u * (F32(1) - BN_MOMENTUM)
    return mm.astype(F32), mv.astype(F32)


def bn_infer(x, gamma, beta, moving_mean, moving_var):
    """inference / frozen mode (model.py:696,702,708: training=train_bn=False)."""
    rstd = (1.0 / np.sqrt(moving_var.astype(np.float64) + float(BN_EPS)))
    xhat = (x.astype(np.float64) - moving_mean) * rstd
    return (xhat * gamma + beta).astype(F32), (xhat.astype(F32), rstd.astype(F32))


def bn_infer_bwd(cache, gamma, dy):
    xhat, rstd = cache
    C = dy.shape[-1]
    dgamma = (dy.reshape(-1, C).astype(np.float64) * xhat.reshape(-1, C)).sum(0).astype(F32)
    dbeta = dy.reshape(-1, C).astype(np.float64).sum(0).astype(F32)
    dx = (dy * (gamma * rstd)).astype(F32)
    return dx, dgamma, dbeta


# ---------------------------------------------------------------------------
# Conv2DTranspose(256,(2,2),strides=2) -- model.py:711-712.  kernel [2,2,Cout,Cin]
# ---------------------------------------------------------------------------
def deconv2x2s2(x, w, bias):
    N, H, W, Ci = x.shape
    Co = w.shape[2]
    y = np.zeros((N, 2 * H, 2 * W, Co), F32)
    x2 = x.reshape(-1, Ci)
    for ky in range(2):
        for kx in range(2):
            y[:, ky::2, kx::2, :] = (x2 @ w[ky, kx].T).reshape(N, H, W, Co)
    return y + bias


def deconv2x2s2_bwd(x, w, dy):
    N, H, W, Ci = x.shape
    Co = w.shape[2]
    x2 = x.reshape(-1, Ci)
    dx = np.zeros_like(x2)
    dw = np.zeros_like(w)
    for ky in range(2):
        for kx in range(2):
            d = dy[:, ky::2, kx::2, :].reshape(-1, Co)
            dx += d @ w[ky, kx]
            dw[ky, kx] = d.T @ x2
    db = dy.reshape(-1, Co).sum(0)
    return dx.reshape(x.shape), dw, db


# ---------------------------------------------------------------------------
# tf.image.crop_and_resize (bilinear, extrapolation_value=0) -- the effective op of
# PyramidROIAlign (model.py:385-387) and of the mask-target crop (model.py:581-583).
# Restates TF-1.x core/kernels/crop_and_resize_op.cc, every op in float32.
# ---------------------------------------------------------------------------
def _crop_coords(lo, hi, size, crop):
    """returns in_coord [nb,crop] float32 following the TF kernel's operation order."""
    lo = lo.astype(F32)
    hi = hi.astype(F32)
    if crop > 1:
        scale = (hi - lo) * F32(size - 1) / F32(crop - 1)
        idx = np.arange(crop, dtype=F32)[None, :]
        return (lo[:, None] * F32(size - 1) + idx * scale[:, None]).astype(F32)
    return (F32(0.5) * (lo + hi) * F32(size - 1))[:, None].astype(F32)


def crop_and_resize(image, boxes, box_ind, crop_hw):
    """image [B,H,W,C]; boxes [nb,4]=(y1,x1,y2,x2) normalised; returns [nb,ch,cw,C]."""
    B, H, W, C = image.shape
    ch, cw = crop_hw
    nb = boxes.shape[0]
    out = np.zeros((nb, ch, cw, C), F32)
    if nb == 0:
        return out
    in_y = _crop_coords(boxes[:, 0], boxes[:, 2], H, ch)
    in_x = _crop_coords(boxes[:, 1], boxes[:, 3], W, cw)
    vy = ~((in_y < 0) | (in_y > F32(H - 1)))
    vx = ~((in_x < 0) | (in_x > F32(W - 1)))
    ty = np.floor(in_y)
    by = np.ceil(in_y)
    ly = (in_y - ty).astype(F32)
    lx0 = np.floor(in_x)
    rx0 = np.ceil(in_x)
    lx = (in_x - lx0).astype(F32)
    tyi = np.clip(ty, 0, H - 1).astype(np.int64)
    byi = np.clip(by, 0, H - 1).astype(np.int64)
    lxi = np.clip(lx0, 0, W - 1).astype(np.int64)
    rxi = np.clip(rx0, 0, W - 1).astype(np.int64)
    bi = np.asarray(box_ind, dtype=np.int64)[:, None, None]
    tl = image[bi, tyi[:, :, None], lxi[:, None, :]]
    tr = image[bi, tyi[:, :, None], rxi[:, None, :]]
    bl = image[bi, byi[:, :, None], lxi[:, None, :]]
    br = image[bi, byi[:, :, None], rxi[:, None, :]]
    lxe = lx[:, None, :, None]
    lye = ly[:, :, None, None]
    top = tl + (tr - tl) * lxe
    bot = bl + (br - bl) * lxe
    val = (top + (bot - top) * lye).astype(F32)
    valid = (vy[:, :, None] & vx[:, None, :])[..., None]
    return np.where(valid, val, F32(0)).astype(F32)


def crop_and_resize_bwd_image(dout, boxes, box_ind, image_shape):
    """gradient wrt image (TF CropAndResizeGradImage): scatter-add of the 4 bilinear weights."""
    B, H, W, C = image_shape
    nb, ch, cw, _ = dout.shape
    dimg = np.zeros(image_shape, np.float64)
    if nb == 0:
        return dimg.astype(F32)
    in_y = _crop_coords(boxes[:, 0], boxes[:, 2], H, ch)
    in_x = _crop_coords(boxes[:, 1], boxes[:, 3], W, cw)
    vy = ~((in_y < 0) | (in_y > F32(H - 1)))
    vx = ~((in_x < 0) | (in_x > F32(W - 1)))
    ty = np.floor(in_y)
    ly = (in_y - ty).astype(F32)
    lx0 = np.floor(in_x)
    lx = (in_x - lx0).astype(F32)
    tyi = np.clip(ty, 0, H - 1).astype(np.int64)
    byi = np.clip(np.ceil(in_y), 0, H - 1).astype(np.int64)
    lxi = np.clip(lx0, 0, W - 1).astype(np.int64)
    rxi = np.clip(np.ceil(in_x), 0, W - 1).astype(np.int64)
    valid = (vy[:, :, None] & vx[:, None, :])[..., None]
    g = np.where(valid, dout, 0).astype(np.float64)
    lye = ly[:, :, None, None].astype(np.float64)
    lxe = lx[:, None, :, None].astype(np.float64)
    dtop = (1 - lye) * g
    dbot = lye * g
    bi = np.broadcast_to(np.asarray(box_ind, dtype=np.int64)[:, None, None], (nb, ch, cw))
    TY = np.broadcast_to(tyi[:, :, None], (nb, ch, cw))
    BY = np.broadcast_to(byi[:, :, None], (nb, ch, cw))
    LX = np.broadcast_to(lxi[:, None, :], (nb, ch, cw))
    RX = np.broadcast_to(rxi[:, None, :], (nb, ch, cw))
    np.add.at(dimg, (bi, TY, LX), (1 - lxe) * dtop)
    np.add.at(dimg, (bi, TY, RX), lxe * dtop)
    np.add.at(dimg, (bi, BY, LX), (1 - lxe) * dbot)
    np.add.at(dimg, (bi, BY, RX), lxe * dbot)
    return dimg.astype(F32)


def roi_boxes_to_crop_order(rois, order="xyxy_as_yxyx"):
    """model.py:385-387 hands [xmin,ymin,xmax,ymax] to crop_and_resize unchanged, which reads
    columns as (y1,x1,y2,x2): the default reproduces that.  'yxyx' is the corrected order."""
    if order == "xyxy_as_yxyx":
        return rois
    return rois[..., [1, 0, 3, 2]]


# ---------------------------------------------------------------------------
# YOLO decode -- DecodeYOLOLayer.call model.py:1442-1473; DetectionsLayer.call :1493-1538
# ---------------------------------------------------------------------------
def cell_grid(G):
    """model.py:1445-1449: [...,0] = column index, [...,1] = row index."""
    col = np.tile(np.arange(G, dtype=F32)[None, :], (G, 1))
    row = col.T
    return np.stack([col, row], -1)[None, :, :, None, :]   # [1,G,G,1,2]


def yolo_decode(y_pred, anchors, G):
    """y_pred [B,G,G,A,5+C] -> proposals [B,G*G*A,4] = [xmin,ymin,xmax,ymax] (normalised).
    Both axes divided by GRID_W (model.py:1454,1459)."""
    y_pred = y_pred.astype(F32)
    A = y_pred.shape[3]
    anc = np.asarray(anchors, F32).reshape(1, 1, 1, A, 2)
    xy = (det_sigmoid(y_pred[..., 0:2]) + cell_grid(G)).astype(F32)
    xy = xy / F32(G)
    wh = (det_expf(y_pred[..., 2:4]) * anc).astype(F32)
    wh = wh / F32(G)
    half = wh / F32(2.0)
    mins = xy - half
    maxs = xy + half
    out = np.concatenate([mins, maxs], -1).astype(F32)
    return out.reshape(y_pred.shape[0], -1, 4)


def yolo_detections(y_pred, anchors, G):
    """-> [B,R,6] = [xmin,ymin,xmax,ymax, sigmoid(conf), float(argmax class logits)] model.py:1527-1536."""
    boxes = yolo_decode(y_pred, anchors, G)
    conf = det_sigmoid(y_pred[..., 4].astype(F32)).reshape(y_pred.shape[0], -1, 1)
    cls = np.argmax(y_pred[..., 5:], -1).astype(F32).reshape(y_pred.shape[0], -1, 1)
    return np.concatenate([boxes, conf, cls], -1).astype(F32)


# ---------------------------------------------------------------------------
# YOLOv2 loss -- yolo_custom_loss model.py:86-242.
# Forward returns the 7 scalars the reference tf.Print()s; grad() is the analytic
# gradient wrt y_pred *as TF autodiff would give it*: through the IoU in true_box_conf
# (model.py:146 has no stop_gradient) and with tf.maximum/minimum sending the gradient to
# the first argument on ties.
# ---------------------------------------------------------------------------
def _iou_centre(pxy, pwh, txy, twh):
    """intersection-over-union of centre/size boxes; returns iou and pieces for the gradient."""
    pmin = pxy - pwh / F32(2.0)
    pmax = pxy + pwh / F32(2.0)
    tmin = txy - twh / F32(2.0)
    tmax = txy + twh / F32(2.0)
    imin = np.maximum(pmin, tmin)
    imax = np.minimum(pmax, tmax)
    d = imax - imin
    iwh = np.maximum(d, F32(0.0))
    inter = iwh[..., 0] * iwh[..., 1]
    tarea = twh[..., 0] * twh[..., 1]
    parea = pwh[..., 0] * pwh[..., 1]
    union = parea + tarea - inter
    iou = inter / union
    return iou, (pmin, pmax, tmin, tmax, d, iwh, inter, union)


def yolo_loss(y_true, y_pred, true_boxes, cfg, want_grad=False, warmup=False):
    """cfg needs ANCHORS, N_BOX, GRID_W, *_SCALE, CLASS_WEIGHTS.  warmup=True is the tf.cond branch of model.py:193-207
    (taken while `seen` -- incremented once per evaluation of the loss, model.py:194 -- is below WARM_UP_BATCHES; dead at the
    default WARM_UP_BATCHES = 0, config.py:38): predictors without a ground-truth box are pulled to their cell centre and their
    anchor's size, and EVERY predictor's coordinate terms count with weight 1 (tf.ones_like(coord_mask), not COORD_SCALE).  The
    confidence / class terms and the IoU they use are formed before the branch and do not change."""
    y_true = y_true.astype(F32)
    y_pred = y_pred.astype(F32)
    tb = true_boxes.astype(F32)
    B, G, _, A, D = y_pred.shape
    anc = np.asarray(cfg.ANCHORS, F32).reshape(1, 1, 1, A, 2)
    sxy = det_sigmoid(y_pred[..., 0:2])
    pxy = (sxy + cell_grid(G)).astype(F32)
    pwh = (det_expf(y_pred[..., 2:4]) * anc).astype(F32)
    pcf = det_sigmoid(y_pred[..., 4])
    logits = y_pred[..., 5:]
    txy = y_true[..., 0:2]
    twh = y_true[..., 2:4]
    t4 = y_true[..., 4]
    iou1, pieces1 = _iou_centre(pxy, pwh, txy, twh)
    tconf = (iou1 * t4).astype(F32)
    tcls = np.argmax(y_true[..., 5:], -1)
    coord_mask = (t4 * F32(cfg.COORD_SCALE))[..., None]
    # IoU against every buffered true box (model.py:159-184)
    iou2, _ = _iou_centre(pxy[..., None, :], pwh[..., None, :], tb[..., 0:2], tb[..., 2:4])
    best = iou2.max(-1)
    conf_mask = (best < F32(0.6)).astype(F32) * (F32(1) - t4) * F32(cfg.NO_OBJECT_SCALE) + t4 * F32(cfg.OBJECT_SCALE)
    cw = np.asarray(cfg.CLASS_WEIGHTS, F32)
    class_mask = (t4 * cw[tcls] * F32(cfg.CLASS_SCALE)).astype(F32)
    if warmup:                                                    # model.py:193-207
        no_boxes = (coord_mask < F32(cfg.COORD_SCALE) / F32(2)).astype(F32)
        txy = (txy + (F32(0.5) + cell_grid(G)) * no_boxes).astype(F32)
        twh = (twh + np.ones_like(twh) * anc * no_boxes).astype(F32)
        coord_mask = np.ones_like(coord_mask)
    n_coord = F32((coord_mask > 0).sum())
    n_conf = F32((conf_mask > 0).sum())
    n_cls = F32((class_mask > 0).sum())
    eps = F32(1e-6)
    f64 = np.float64
    loss_xy = F32((np.square(txy - pxy) * coord_mask).astype(f64).sum()) / (n_coord + eps) / F32(2)
    loss_wh = F32((np.square(twh - pwh) * coord_mask).astype(f64).sum()) / (n_coord + eps) / F32(2)
    loss_conf = F32((np.square(tconf - pcf) * conf_mask).astype(f64).sum()) / (n_conf + eps) / F32(2)
    m = logits.max(-1, keepdims=True)
    ex = np.exp((logits - m).astype(f64))
    lse = np.log(ex.sum(-1)) + m[..., 0]
    ce = (lse - np.take_along_axis(logits, tcls[..., None], -1)[..., 0]).astype(F32)
    loss_cls = F32((ce * class_mask).astype(f64).sum()) / (n_cls + eps)
    loss = F32(loss_xy + loss_wh + loss_conf + loss_cls)
    nb_true = t4.sum()
    nb_pred = ((tconf > 0.5).astype(F32) * (pcf > 0.3).astype(F32)).sum()
    recall = F32(nb_pred / (nb_true + eps))
    out = dict(loss=loss, loss_xy=F32(loss_xy), loss_wh=F32(loss_wh), loss_conf=F32(loss_conf),
               loss_class=F32(loss_cls), recall=recall,
               n_coord=n_coord, n_conf=n_conf, n_class=n_cls)
    if not want_grad:
        return out
    # ---- analytic gradient wrt y_pred -------------------------------------
    g = np.zeros_like(y_pred, dtype=f64)
    d_pxy = -(txy - pxy).astype(f64) * coord_mask / f64(n_coord + eps)          # [.,2]
    d_pwh = -(twh - pwh).astype(f64) * coord_mask / f64(n_coord + eps)
    d_tconf = (tconf - pcf).astype(f64) * conf_mask / f64(n_conf + eps)
    d_pcf = -d_tconf
    d_iou = d_tconf * t4
    pmin, pmax, tmin, tmax, d, iwh, inter, union = [a.astype(f64) for a in pieces1]
    # iou = I/U, U = pw*ph + T - I
    dI = d_iou / union + d_iou * inter / (union * union)       # dIoU/dI incl. the -I inside U
    dU_p = -d_iou * inter / (union * union)                    # multiplies d(pw*ph)
    # I = iw*ih
    d_iwh = np.stack([dI * iwh[..., 1], dI * iwh[..., 0]], -1)
    d_d = d_iwh * (d >= 0)                                      # tf.maximum(d, 0.): first arg wins ties
    d_imax = d_d
    d_imin = -d_d
    d_pmax = d_imax * (pmax <= tmax)                            # tf.minimum(pmax, tmax)
    d_pmin = d_imin * (pmin >= tmin)                            # tf.maximum(pmin, tmin)
    d_pxy = d_pxy + d_pmax + d_pmin
    d_pwh = d_pwh + (d_pmax - d_pmin) / 2.0
    pw = pwh.astype(f64)
    d_pwh = d_pwh + np.stack([dU_p * pw[..., 1], dU_p * pw[..., 0]], -1)
    s = sxy.astype(f64)
    g[..., 0:2] = d_pxy * s * (1 - s)
    g[..., 2:4] = d_pwh * pw
    c = pcf.astype(f64)
    g[..., 4] = d_pcf * c * (1 - c)
    sm = ex / ex.sum(-1, keepdims=True)
    onehot = np.zeros_like(sm)
    np.put_along_axis(onehot, tcls[..., None], 1.0, -1)
    g[..., 5:] = (sm - onehot) * (class_mask.astype(f64) / f64(n_cls + eps))[..., None]
    out['grad'] = g.astype(F32)
    return out


# ---------------------------------------------------------------------------
# mask-target assignment -- norm_boxes_graph model.py:1394-1408, trim_zeros_graph :1411-1420,
# overlaps_graph :420-454, detect_mask_target_graph :457-602, batch_slice myolo_utils.py:929-963
# ---------------------------------------------------------------------------
def norm_boxes(boxes_px, H, W):
    """model.py:1405-1408 (called at :819-820 with shape=(H,W) split as 'w,h': square images)."""
    scale = np.array([H - 1, W - 1, H - 1, W - 1], F32)   # (w,h,w,h) as the reference splits it
    shift = np.array([0, 0, 1, 1], F32)
    return ((boxes_px.astype(F32) - shift) / scale).astype(F32)


def overlaps(b1, b2):
    """IoU matrix [len(b1), len(b2)], boxes (x1,y1,x2,y2); no +1; model.py:438-451."""
    b1 = b1.astype(F32)[:, None, :]
    b2 = b2.astype(F32)[None, :, :]
    x1 = np.maximum(b1[..., 0], b2[..., 0])
    y1 = np.maximum(b1[..., 1], b2[..., 1])
    x2 = np.minimum(b1[..., 2], b2[..., 2])
    y2 = np.minimum(b1[..., 3], b2[..., 3])
    inter = np.maximum(x2 - x1, F32(0)) * np.maximum(y2 - y1, F32(0))
    a1 = (b1[..., 3] - b1[..., 1]) * (b1[..., 2] - b1[..., 0])
    a2 = (b2[..., 3] - b2[..., 1]) * (b2[..., 2] - b2[..., 0])
    union = a1 + a2 - inter
    with np.errstate(divide='ignore', invalid='ignore'):
        return (inter / union).astype(F32)


def round_half_even(x):
    """tf.round (model.py:589)."""
    return np.rint(x).astype(F32)


def mask_targets_one(proposals, gt_class_ids, gt_boxes_norm, gt_masks, cfg):
    """One image (model.py:457-602).  proposals [R,4] (x1,y1,x2,y2) normalised; gt_boxes_norm
    [T,4] ALREADY through norm_boxes (zero-padded pixel rows therefore become
    [0,0,-1/(W-1),-1/(H-1)] and are NOT trimmed -- faithful to model.py:819-820 + :487);
    gt_masks [H,W,T] bool.  Returns rois [R,4] f32, class_ids [R] i32, masks [R,mh,mw] f32 0/1,
    n_pos."""
    R = cfg.TRAIN_ROIS_PER_IMAGE
    mh, mw = cfg.MASK_SHAPE
    non_zero = np.abs(gt_boxes_norm).sum(1) != 0                  # trim_zeros_graph
    keep = np.where(non_zero)[0]
    gtb = gt_boxes_norm[keep]
    gtc = gt_class_ids[keep]
    ov = overlaps(proposals, gtb)                                 # [R, G]
    if ov.shape[1] > 0:
        iou_max = ov.max(1)
    else:
        iou_max = np.full((proposals.shape[0],), -np.inf, F32)
    pos_idx = np.where(iou_max >= F32(0.5))[0]
    neg_idx = np.where(iou_max < F32(0.5))[0]
    pos = proposals[pos_idx]
    neg = proposals[neg_idx]
    if ov.shape[1] > 0 and len(pos_idx) > 0:
        assign = np.argmax(ov[pos_idx], 1)
    else:
        assign = np.zeros((0,), np.int64)
    cls = gtc[assign].astype(np.int32)
    m = gt_masks[:, :, keep]                                      # [H,W,G]
    roi_masks = np.transpose(m, (2, 0, 1))[assign][..., None].astype(F32)   # [P,H,W,1]
    boxes = pos[:, [1, 0, 3, 2]]                                  # model.py:558-559 -> (y1,x1,y2,x2)
    masks = crop_and_resize(roi_masks, boxes, np.arange(len(assign)), (mh, mw))[..., 0]
    masks = round_half_even(masks)
    rois = np.concatenate([pos, neg], 0).astype(F32)
    P = max(R - rois.shape[0], 0)
    N = neg.shape[0]
    rois = np.pad(rois, ((0, P), (0, 0)))
    cls = np.pad(cls, (0, N + P))
    masks = np.pad(masks, ((0, N + P), (0, 0), (0, 0)))
    return rois.astype(F32), cls.astype(np.int32), masks.astype(F32), len(pos_idx)


def mask_targets(proposals, gt_class_ids, gt_boxes_px, gt_masks, cfg):
    """DetectMaskTargetLayer (model.py:635-649) = per-image slices stacked (batch_slice)."""
    H, W = cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1]
    gtn = norm_boxes(gt_boxes_px, H, W)
    outs = [mask_targets_one(proposals[b], gt_class_ids[b], gtn[b], gt_masks[b], cfg)
            for b in range(proposals.shape[0])]
    rois = np.stack([o[0] for o in outs])
    cls = np.stack([o[1] for o in outs])
    masks = np.stack([o[2] for o in outs])
    npos = np.array([o[3] for o in outs], np.int32)
    return rois, cls, masks, npos


# ---------------------------------------------------------------------------
# mask BCE loss -- myolo_mask_loss_graph model.py:718-754, K.binary_crossentropy
# ---------------------------------------------------------------------------
BCE_EPS = F32(1e-7)


def mask_bce(target_masks, target_class_ids, pred_masks, want_grad=False):
    """target_masks [B,R,h,w]; ids [B,R]; pred_masks [B,R,h,w,C] (post-sigmoid).
    Returns loss (mean over positives x h x w; 0 if none) and d loss / d pred_masks."""
    ids = target_class_ids.reshape(-1)
    tm = target_masks.reshape((-1,) + target_masks.shape[2:])
    pm = pred_masks.reshape((-1,) + pred_masks.shape[2:])
    pos = np.where(ids > 0)[0]
    grad = np.zeros_like(pm, dtype=F32)
    if len(pos) == 0:
        return (F32(0.0), grad.reshape(pred_masks.shape)) if want_grad else F32(0.0)
    yt = tm[pos].astype(np.float64)
    yp = pm[pos, :, :, ids[pos]].astype(F32)
    p = np.clip(yp, BCE_EPS, F32(1) - BCE_EPS).astype(np.float64)
    z = np.log(p / (1 - p))
    l = np.maximum(z, 0) - z * yt + np.log1p(np.exp(-np.abs(z)))
    n = l.size
    loss = F32(l.mean())
    if not want_grad:
        return loss
    inside = (yp >= BCE_EPS) & (yp <= F32(1) - BCE_EPS)
    sig = 1.0 / (1.0 + np.exp(-z))
    dz = (sig - yt) / n
    dp = dz / (p * (1 - p)) * inside
    grad[pos, :, :, ids[pos]] = dp.astype(F32)
    return loss, grad.reshape(pred_masks.shape)


def sigmoid(x):
    return (1.0 / (1.0 + np.exp(-x.astype(np.float64)))).astype(F32)


# ---------------------------------------------------------------------------
# Keras Adam (model.py:1071-1075): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t*m/(sqrt(v)+eps)
# ---------------------------------------------------------------------------
def adam_step(p, g, m, v, t, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    lr_t = F32(lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
    m = (F32(b1) * m + F32(1 - b1) * g).astype(F32)
    v = (F32(b2) * v + F32(1 - b2) * g * g).astype(F32)
    p = (p - lr_t * m / (np.sqrt(v) + F32(eps))).astype(F32)
    return p, m, v


# ---------------------------------------------------------------------------
# host-side target encoding -- BatchGenerator.__getitem__ myolo_utils.py:727-860,
# bbox_iou/_interval_overlap :186-244, extract_bboxes :247-271.  Literal loop restatement.
# ---------------------------------------------------------------------------
def _interval_overlap(a, b):
    x1, x2 = a
    x3, x4 = b
    if x3 < x1:
        return 0 if x4 < x1 else min(x2, x4) - x1
    return 0 if x2 < x3 else min(x2, x4) - x3


def _bbox_iou_wh(w1, h1, w2, h2):
    iw = _interval_overlap([0, w1], [0, w2])
    ih = _interval_overlap([0, h1], [0, h2])
    inter = iw * ih
    union = w1 * h1 + w2 * h2 - inter
    return float(inter) / union


def extract_bboxes(mask):
    """myolo_utils.py:247-271 -> [N,(x1,y1,x2,y2)] int32, x2/y2 exclusive."""
    boxes = np.zeros([mask.shape[-1], 4], dtype=np.int32)
    for i in range(mask.shape[-1]):
        m = mask[:, :, i]
        hz = np.where(np.any(m, axis=0))[0]
        vt = np.where(np.any(m, axis=1))[0]
        if hz.shape[0]:
            x1, x2 = hz[[0, -1]]
            y1, y2 = vt[[0, -1]]
            x2 += 1
            y2 += 1
        else:
            x1, x2, y1, y2 = 0, 0, 0, 0
        boxes[i] = np.array([x1, y1, x2, y2])
    return boxes


def encode_batch(samples, cfg):
    """samples: list of (image uint8 [H,W,3], class_ids [n], boxes [n,4] int x1y1x2y2, masks [H,W,n] bool).
    Returns the six arrays of myolo_utils.py:853-854 (norm=True)."""
    Bn = len(samples)
    H, W = cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1]
    G, A, C, T = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER
    images = np.zeros((Bn, H, W, 3), np.float32)
    y_true = np.zeros((Bn, cfg.GRID_H, cfg.GRID_W, A, 5 + C))
    true_boxes = np.zeros((Bn, 1, 1, 1, T, 4))
    gt_ids = np.zeros((Bn, T), np.int32)
    gt_boxes = np.zeros((Bn, T, 4), np.int32)
    gt_masks = np.zeros((Bn, H, W, cfg.MAX_GT_INSTANCES), bool)
    anchors = [(cfg.ANCHORS[2 * i], cfg.ANCHORS[2 * i + 1]) for i in range(len(cfg.ANCHORS) // 2)]
    for n, (image, ids, boxes, masks) in enumerate(samples):
        tbi = 0
        for i in range(boxes.shape[0]):
            xmin, ymin, xmax, ymax = [boxes[i][k] for k in range(4)]
            cx = .5 * (xmin + xmax) / (float(W) / cfg.GRID_W)
            cy = .5 * (ymin + ymax) / (float(H) / cfg.GRID_H)
            gx = int(np.floor(cx))
            gy = int(np.floor(cy))
            if gx < cfg.GRID_W and gy < cfg.GRID_H:
                cw_ = (xmax - xmin) / (float(W) / cfg.GRID_W)
                ch_ = (ymax - ymin) / (float(H) / cfg.GRID_H)
                box = [cx, cy, cw_, ch_]
                best, max_iou = -1, -1
                for j, (aw, ah) in enumerate(anchors):
                    iou = _bbox_iou_wh(cw_, ch_, aw, ah)
                    if max_iou < iou:
                        best, max_iou = j, iou
                y_true[n, gy, gx, best, 0:4] = box
                y_true[n, gy, gx, best, 4] = 1.
                y_true[n, gy, gx, best, 5 + ids[i]] = 1
                true_boxes[n, 0, 0, 0, tbi] = box
                tbi = (tbi + 1) % T
        images[n] = image / 255.
        gt_ids[n, :ids.shape[0]] = ids
        gt_boxes[n, :boxes.shape[0]] = boxes
        gt_masks[n, :, :, :masks.shape[-1]] = masks
    return [images, true_boxes, y_true, gt_ids, gt_boxes, gt_masks]
