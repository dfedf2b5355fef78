"""
ORACLE (test infrastructure, NOT product code) -- second, independent CPU composition of the
Mask-YOLO training step in torch-CPU with *autograd* (no hand-written backward).  Used
(1) to pin oracle/np_model.py's analytic backward, (2) as the fast CPU baseline that bench.py
times on the GPU box's host cores ("cpu_baseline", kind "port").  PARITY PARTLY PINNED (np_ops.py).

Reference lines followed: same as oracle/np_model.py (model.py:38-79, 86-242, 249-278,
385-387, 457-602, 668-754, 1062-1094).  Integer-valued pieces (decode -> targets) are taken
from np_ops so both oracles see identical ROIs.
"""
import numpy as np
import torch
import torch.nn.functional as Fn

from . import np_ops as O
from .np_model import BACKBONE_BLOCKS, YOLO_BLOCKS, trainable_names

BN_EPS = 1e-3


def _t(a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dtype)


def _conv(x, k, stride=1, pad=None, bias=None):
    """x NCHW; k keras [kh,kw,ci,co]."""
    w = k.permute(3, 2, 0, 1)
    if pad is not None:
        x = Fn.pad(x, pad)      # (l, r, t, b)
    return Fn.conv2d(x, w, bias=bias, stride=stride)


def _dw(x, k, stride):
    C = k.shape[2]
    w = k.permute(2, 0, 1).unsqueeze(1)      # [C,1,3,3]
    x = Fn.pad(x, (1, 1, 1, 1) if stride == 1 else (0, 1, 0, 1))
    return Fn.conv2d(x, w, stride=stride, groups=C)


def _bn(x, g, b, mm, mv, train):
    if train:
        mean = x.mean((0, 2, 3), keepdim=True)
        var = x.var((0, 2, 3), unbiased=False, keepdim=True)
    else:
        mean, var = mm.view(1, -1, 1, 1), mv.view(1, -1, 1, 1)
    return (x - mean) / torch.sqrt(var + BN_EPS) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def _relu6(x):
    return torch.clamp(x, 0, 6)


def crop_and_resize_t(img, boxes, bidx, ch, cw):
    """img [B,C,H,W] torch; boxes np [nb,4] (y1,x1,y2,x2); differentiable wrt img."""
    B, C, H, W = img.shape
    nb = boxes.shape[0]
    in_y = O._crop_coords(boxes[:, 0], boxes[:, 2], H, ch)
    in_x = O._crop_coords(boxes[:, 1], boxes[:, 3], W, cw)
    vy = ~((in_y < 0) | (in_y > np.float32(H - 1)))
    vx = ~((in_x < 0) | (in_x > np.float32(W - 1)))
    ty = np.clip(np.floor(in_y), 0, H - 1).astype(np.int64)
    by = np.clip(np.ceil(in_y), 0, H - 1).astype(np.int64)
    lx = np.clip(np.floor(in_x), 0, W - 1).astype(np.int64)
    rx = np.clip(np.ceil(in_x), 0, W - 1).astype(np.int64)
    wy = torch.from_numpy((in_y - np.floor(in_y)).astype(np.float64)).to(img.dtype)[:, None, :, None]
    wx = torch.from_numpy((in_x - np.floor(in_x)).astype(np.float64)).to(img.dtype)[:, None, None, :]
    bi = torch.from_numpy(np.asarray(bidx, np.int64))[:, None, None]
    TY, BY = torch.from_numpy(ty)[:, :, None], torch.from_numpy(by)[:, :, None]
    LX, RX = torch.from_numpy(lx)[:, None, :], torch.from_numpy(rx)[:, None, :]
    im = img.permute(0, 2, 3, 1)           # [B,H,W,C]
    tl = im[bi, TY, LX].permute(0, 3, 1, 2)
    tr = im[bi, TY, RX].permute(0, 3, 1, 2)
    bl = im[bi, BY, LX].permute(0, 3, 1, 2)
    br = im[bi, BY, RX].permute(0, 3, 1, 2)
    top = tl + (tr - tl) * wx
    bot = bl + (br - bl) * wx
    val = top + (bot - top) * wy
    valid = torch.from_numpy((vy[:, :, None] & vx[:, None, :]))[:, None]
    return torch.where(valid, val, torch.zeros((), dtype=img.dtype))


def yolo_loss_t(y_true, y_pred, true_boxes, cfg, warmup=False):
    """model.py:86-242 in torch ops (autograd supplies the gradient); warmup: the branch of model.py:193-207 (see np_ops.yolo_loss)."""
    dt = y_pred.dtype
    B, G, _, A, D = y_pred.shape
    anc = torch.tensor(np.asarray(cfg.ANCHORS, np.float64).reshape(1, 1, 1, A, 2), dtype=dt)
    grid = _t(O.cell_grid(G), dt)
    pxy = torch.sigmoid(y_pred[..., 0:2]) + grid
    pwh = torch.exp(y_pred[..., 2:4]) * anc
    pcf = torch.sigmoid(y_pred[..., 4])
    logits = y_pred[..., 5:]
    txy, twh, t4 = y_true[..., 0:2], y_true[..., 2:4], y_true[..., 4]

    def iou(pxy, pwh, txy, twh):
        pmin, pmax = pxy - pwh / 2, pxy + pwh / 2
        tmin, tmax = txy - twh / 2, txy + twh / 2
        iwh = torch.clamp(torch.minimum(pmax, tmax) - torch.maximum(pmin, tmin), min=0)
        inter = iwh[..., 0] * iwh[..., 1]
        return inter / (pwh[..., 0] * pwh[..., 1] + twh[..., 0] * twh[..., 1] - inter)

    tconf = iou(pxy, pwh, txy, twh) * t4
    tcls = torch.argmax(y_true[..., 5:], -1)
    coord_mask = (t4 * cfg.COORD_SCALE).unsqueeze(-1)
    best = iou(pxy.unsqueeze(4), pwh.unsqueeze(4), true_boxes[..., 0:2], true_boxes[..., 2:4]).max(-1).values
    conf_mask = (best < 0.6).to(dt) * (1 - t4) * cfg.NO_OBJECT_SCALE + t4 * cfg.OBJECT_SCALE
    cw = torch.tensor(np.asarray(cfg.CLASS_WEIGHTS, np.float64), dtype=dt)
    class_mask = t4 * cw[tcls] * cfg.CLASS_SCALE
    if warmup:
        no_boxes = (coord_mask < cfg.COORD_SCALE / 2.).to(dt)
        txy = txy + (0.5 + grid) * no_boxes
        twh = twh + torch.ones_like(twh) * anc * no_boxes
        coord_mask = torch.ones_like(coord_mask)
    n_coord = (coord_mask > 0).to(dt).sum()
    n_conf = (conf_mask > 0).to(dt).sum()
    n_cls = (class_mask > 0).to(dt).sum()
    l_xy = ((txy - pxy) ** 2 * coord_mask).sum() / (n_coord + 1e-6) / 2
    l_wh = ((twh - pwh) ** 2 * coord_mask).sum() / (n_coord + 1e-6) / 2
    l_cf = ((tconf - pcf) ** 2 * conf_mask).sum() / (n_conf + 1e-6) / 2
    ce = Fn.cross_entropy(logits.reshape(-1, logits.shape[-1]), tcls.reshape(-1), reduction='none').reshape(tcls.shape)
    l_cl = (ce * class_mask).sum() / (n_cls + 1e-6)
    return l_xy + l_wh + l_cf + l_cl, (l_xy, l_wh, l_cf, l_cl)


def mask_bce_t(tmask, tcls, pred):
    """model.py:718-754.  pred [N,C,h,w] post-sigmoid torch; tmask np [N,h,w]; tcls np [N]."""
    pos = np.where(tcls > 0)[0]
    if len(pos) == 0:
        return pred.sum() * 0
    yt = _t(tmask[pos], pred.dtype)
    yp = pred[torch.from_numpy(pos), torch.from_numpy(tcls[pos].astype(np.int64))]
    eps = float(np.float32(1e-7)) if pred.dtype == torch.float32 else 1e-7
    p = torch.clamp(yp, eps, 1 - eps)
    z = torch.log(p / (1 - p))
    return (torch.clamp(z, min=0) - z * yt + torch.log1p(torch.exp(-z.abs()))).mean()


class TorchRef(object):
    def __init__(self, P_np, cfg, dtype=torch.float64, capture=False):
        """capture=True: keep every BatchNorm's input (NHWC numpy, keyed by the BN layer name) and the deconv output ("deconv/out")
        of the last forward in self.cap -- the tensors every ReLU / ReLU6 decision of the backward is read from
        (tests: test_gradients_with_oracle_activation_masks_hold_maxnorm at config 2)."""
        self.cfg, self.dtype = cfg, dtype
        self.capture, self.cap = capture, {}
        self.P = {k: _t(v, dtype) for k, v in P_np.items()}
        self.train_names = trainable_names(P_np)
        for k in self.train_names:
            self.P[k].requires_grad_(True)

    def _keep(self, name, x):
        if self.capture:
            self.cap[name] = x.detach().permute(0, 2, 3, 1).contiguous().numpy()
        return x

    def _block(self, x, bid, stride, train):
        P = self.P
        n = "conv_dw_%d" % bid
        x = self._keep(n + "_bn", _dw(x, P[n + "/depthwise_kernel"], stride))
        x = _relu6(_bn(x, P[n + "_bn/gamma"], P[n + "_bn/beta"], P[n + "_bn/moving_mean"], P[n + "_bn/moving_variance"], train))
        n = "conv_pw_%d" % bid
        x = self._keep(n + "_bn", _conv(x, P[n + "/kernel"]))
        return _relu6(_bn(x, P[n + "_bn/gamma"], P[n + "_bn/beta"], P[n + "_bn/moving_mean"], P[n + "_bn/moving_variance"], train))

    def trunk(self, images, train):
        P, cfg = self.P, self.cfg
        x = _t(images, self.dtype).permute(0, 3, 1, 2)
        x = self._keep("conv1_bn", _conv(x, P["conv1/kernel"], stride=2, pad=(1, 1, 1, 1)))
        x = _relu6(_bn(x, P["conv1_bn/gamma"], P["conv1_bn/beta"], P["conv1_bn/moving_mean"], P["conv1_bn/moving_variance"], train))
        bid = 1
        for f, s in BACKBONE_BLOCKS:
            x = self._block(x, bid, s, train)
            bid += 1
        C4 = x
        Fm = _conv(C4, P["feature_map/kernel"], pad=(1, 1, 1, 1), bias=P["feature_map/bias"])
        for f, s in YOLO_BLOCKS:
            x = self._block(x, bid, s, train)
            bid += 1
        y = _conv(x, P["conv_23/kernel"], bias=P["conv_23/bias"])
        B = y.shape[0]
        yolo_out = y.permute(0, 2, 3, 1).reshape(B, cfg.GRID_H, cfg.GRID_W, cfg.N_BOX, 5 + cfg.NUM_CLASSES)
        return C4, Fm, yolo_out

    def mask_head(self, Fm, rois, train):
        P, cfg = self.P, self.cfg
        B, R = rois.shape[:2]
        boxes = O.roi_boxes_to_crop_order(rois.reshape(-1, 4), cfg.ROI_BOX_ORDER)
        bidx = np.repeat(np.arange(B), R)
        ps = cfg.MASK_POOL_SIZE
        x = crop_and_resize_t(Fm, boxes, bidx, ps, ps)
        for i in range(1, 5):
            n = "myolo_mask_conv%d" % i
            b = "myolo_mask_bn%d" % i
            x = self._keep(b, _conv(x, P[n + "/kernel"], pad=(1, 1, 1, 1), bias=P[n + "/bias"]))
            x = torch.relu(_bn(x, P[b + "/gamma"], P[b + "/beta"], P[b + "/moving_mean"], P[b + "/moving_variance"],
                               train and i == 1))
        w = P["myolo_mask_deconv/kernel"].permute(3, 2, 0, 1)      # [Cin,Cout,kh,kw]
        x = self._keep("deconv/out", torch.relu(Fn.conv_transpose2d(x, w, bias=P["myolo_mask_deconv/bias"], stride=2)))      # post-ReLU, as the engine tapes it
        z = _conv(x, P["myolo_mask/kernel"], bias=P["myolo_mask/bias"])
        return torch.sigmoid(z)                                    # [N,C,h,w]

    def train_step(self, batch, backward=True):
        cfg = self.cfg
        images, true_boxes, y_true, gt_ids, gt_boxes, gt_masks = batch
        for k in self.train_names:
            self.P[k].grad = None
        C4, Fm, yolo_out = self.trunk(images, True)
        yo_np = yolo_out.detach().to(torch.float32).numpy()
        proposals = O.yolo_decode(yo_np, cfg.ANCHORS, cfg.GRID_W)
        rois, tcls, tmask, npos = O.mask_targets(proposals, gt_ids, gt_boxes, gt_masks, cfg)
        pred = self.mask_head(Fm, rois, True)
        yl, terms = yolo_loss_t(_t(y_true, self.dtype), yolo_out, _t(true_boxes, self.dtype), cfg)
        ml = mask_bce_t(tmask.reshape((-1,) + tmask.shape[2:]), tcls.reshape(-1), pred)
        loss = yl * cfg.LOSS_WEIGHTS.get("yolo_sum_loss", 1.) + ml * cfg.LOSS_WEIGHTS.get("myolo_mask_loss", 1.)
        out = dict(loss=float(loss.detach()), yolo_sum_loss=float(yl.detach()), mask_loss=float(ml.detach()),
                   yolo_output=yo_np, output_rois=rois, target_class_ids=tcls,
                   myolo_mask=pred.detach().permute(0, 2, 3, 1).numpy(),
                   feature_map=Fm.detach().permute(0, 2, 3, 1).numpy())
        if backward:
            loss.backward()
            out["grads"] = {k: self.P[k].grad.detach().numpy() for k in self.train_names if self.P[k].grad is not None}
        return out

    def adam(self, state, t, lr):
        """Keras Adam on the torch params in place."""
        b1, b2, eps = 0.9, 0.999, 1e-8
        lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
        with torch.no_grad():
            for k in self.train_names:
                p = self.P[k]
                if p.grad is None:
                    continue
                m, v = state.get(k, (torch.zeros_like(p), torch.zeros_like(p)))
                m = b1 * m + (1 - b1) * p.grad
                v = b2 * v + (1 - b2) * p.grad * p.grad
                p -= lr_t * m / (v.sqrt() + eps)
                state[k] = (m, v)
        return state
