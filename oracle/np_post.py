"""
ORACLE (test infrastructure, NOT product code) -- literal restatement of the reference's inference
post-processing.  `skimage.transform.resize` (used by unmold_mask,
myolo_utils.py:903, through the wrapper at :433-447) is PINNED: the restatement below (order-1 interpolation, pixel-centre
alignment, zero border of mode='constant', clip=True to the input range, float32 in a fixed operation order so that the GPU
kernel can reproduce the thresholded masks bit for bit) is checked against outputs of the real scikit-image 0.18.3 found in this
image's Anaconda Python (tests/golden/make_skimage_fixture.py -> skimage_resize_fixture.npz).

Follows: decode_one_yolo_output myolo_utils.py:36-85; NMB myolo_utils.py:88-113; bbox_iou / bbox_iou_2 /
_interval_overlap myolo_utils.py:186-244; unmold_mask myolo_utils.py:883-912; MaskYOLO.decode_masks
model.py:1330-1391; the selection logic of MaskYOLO.detect model.py:1290-1304 (without the debug override at :1306).
"""
import numpy as np

F32 = np.float32


def _interval_overlap(a, b):
    x1, x2 = a
    x3, x4 = b
    if x3 < x1:
        return 0 if x4 < x1 else min(x2, x4) - x1
    return 0 if x2 < x3 else min(x2, x4) - x3


def bbox_iou(b1, b2):
    """boxes as (xmin, ymin, xmax, ymax)."""
    iw = _interval_overlap([b1[0], b1[2]], [b2[0], b2[2]])
    ih = _interval_overlap([b1[1], b1[3]], [b2[1], b2[3]])
    inter = iw * ih
    return float(inter) / ((b1[2] - b1[0]) * (b1[3] - b1[1]) + (b2[2] - b2[0]) * (b2[3] - b2[1]) - inter)


def decode_one_yolo_output(netout, anchors, nb_class, obj_threshold=0.3, nms_threshold=0.3):
    """-> list of (xmin, ymin, xmax, ymax, confidence, classes[nb_class]) after per-class greedy NMS.
    PINNED by tests/golden/ref_host_decode.npz (outputs of the reference's own function, numpy 1.26.4): array steps in the
    input's dtype (float32 network output, model.py:1224), per-element steps in float64 -- the scalar promotion of the numpy
    generation the reference ran with (round 4 finding: rounds 1-3 converted everything to float64 first, which kept 1-2 boxes
    more or fewer near the NMS threshold)."""
    netout = np.asarray(netout)
    netout = np.array(netout, dtype=netout.dtype if netout.dtype in (np.float32, np.float64) else np.float64)
    gh, gw, nb = netout.shape[:3]
    one = netout.dtype.type(1)
    netout[..., 4] = one / (one + np.exp(-netout[..., 4]))
    x = netout[..., 5:] - np.max(netout[..., 5:])
    if np.min(x) < -100.:
        x = x / np.min(x) * netout.dtype.type(-100.)
    e = np.exp(x)
    netout[..., 5:] = netout[..., 4][..., None] * (e / e.sum(-1, keepdims=True))
    netout[..., 5:] *= netout[..., 5:] > netout.dtype.type(obj_threshold)
    boxes = []
    for row in range(gh):
        for col in range(gw):
            for b in range(nb):
                classes = netout[row, col, b, 5:]
                if np.sum(classes) > 0:
                    tx, ty, tw, th = netout[row, col, b, :4]
                    ex, ey, ew, eh = (np.float64(np.exp(-tx)), np.float64(np.exp(-ty)), np.float64(np.exp(tw)), np.float64(np.exp(th)))
                    cx = (col + 1. / (1. + ex)) / gw
                    cy = (row + 1. / (1. + ey)) / gh
                    w = np.float64(anchors[2 * b + 0]) * ew / gw
                    h = np.float64(anchors[2 * b + 1]) * eh / gh
                    boxes.append([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2, netout[row, col, b, 4], classes])
    for c in range(nb_class):
        order = list(reversed(np.argsort([bx[5][c] for bx in boxes])))
        for i in range(len(order)):
            bi = order[i]
            if boxes[bi][5][c] == 0:
                continue
            for j in range(i + 1, len(order)):
                bj = order[j]
                if bbox_iou(boxes[bi][:4], boxes[bj][:4]) >= nms_threshold:
                    boxes[bj][5][c] = 0
    return [bx for bx in boxes if np.float64(bx[5][int(np.argmax(bx[5]))]) > obj_threshold]


def nmb(boxes, class_ids, indices, image_shape, nms_threshold=0.3):
    """myolo_utils.py:88-113 -- suppress later same-class boxes overlapping an earlier one (every pair is tested,
    also pairs whose first member was itself suppressed, as in the reference).  PINNED by tests/golden/ref_host_nmb.npz;
    float32 box elements times the Python ints of image_shape are float64 products (numpy < 2.0 scalar promotion)."""
    w, h = image_shape[0], image_shape[1]
    remove = []
    B = np.asarray(boxes, dtype=np.float64)
    for i in range(len(indices)):
        for j in range(i + 1, len(indices)):
            a = [B[i][0] * w, B[i][1] * h, B[i][2] * w, B[i][3] * h]
            b = [B[j][0] * w, B[j][1] * h, B[j][2] * w, B[j][3] * h]
            if bbox_iou(a, b) >= nms_threshold and class_ids[i] == class_ids[j]:
                remove.append(j)
    return np.delete(indices, remove)


def box_to_pixels(bbox, image_shape):
    """the clamped integer window of unmold_mask (myolo_utils.py:892-900); float32 product, truncation."""
    w, h = image_shape[0], image_shape[1]
    x1, y1, x2, y2 = [F32(v) for v in bbox]
    px1 = min(max(0, int(x1 * F32(w))), w)
    px2 = min(max(1, int(x2 * F32(w))), w)
    py1 = min(max(0, int(y1 * F32(h))), h)
    py2 = min(max(1, int(y2 * F32(h))), h)
    return px1, py1, px2, py2


def resize_bilinear_f32(mask, out_h, out_w):
    """skimage.transform.resize(mask, (out_h, out_w), order=1, mode='constant', cval=0, clip=True, anti_aliasing=False) -- the
    call the reference makes through its wrapper (myolo_utils.py:433-447, 903): pixel centres aligned
    (src = (dst + 0.5) * in/out - 0.5), bilinear, and samples OUTSIDE the mask read cval = 0 (not the edge value), so an
    up-scaled mask fades towards 0 over its outermost half source pixel -- and clip=True then clips the result to the input's
    [min, max], which undoes that fade for a mask whose minimum is already above it.  Every operation float32 in this order (the HIP kernel
    uses the same expressions); pinned against scikit-image 0.18.3 outputs in tests/golden/skimage_resize_fixture.npz."""
    mask = np.asarray(mask, F32)
    h, w = mask.shape
    sy, sx = F32(h) / F32(out_h), F32(w) / F32(out_w)
    ys = ((np.arange(out_h, dtype=F32) + F32(0.5)) * sy - F32(0.5)).astype(F32)
    xs = ((np.arange(out_w, dtype=F32) + F32(0.5)) * sx - F32(0.5)).astype(F32)
    y0, x0 = np.floor(ys).astype(np.int64), np.floor(xs).astype(np.int64)
    wy = (ys - y0.astype(F32))[:, None].astype(F32)
    wx = (xs - x0.astype(F32))[None, :].astype(F32)
    P = np.zeros((h + 2, w + 2), F32)                    # constant-0 border: index -1 -> 0, index h -> h + 1
    P[1:-1, 1:-1] = mask
    tl, tr = P[y0 + 1][:, x0 + 1], P[y0 + 1][:, x0 + 2]
    bl, br = P[y0 + 2][:, x0 + 1], P[y0 + 2][:, x0 + 2]
    top = tl + (tr - tl) * wx
    bot = bl + (br - bl) * wx
    out = (top + (bot - top) * wy).astype(F32)
    return np.clip(out, mask.min(), mask.max()).astype(F32)          # clip=True: to the value range of the input


def unmold_mask(mask, bbox, image_shape):
    px1, py1, px2, py2 = box_to_pixels(bbox, image_shape)
    m = resize_bilinear_f32(mask, max(1, py2 - py1), max(1, px2 - px1)) >= F32(0.5)
    full = np.zeros(tuple(image_shape[:2]), bool)
    full[py1:py2, px1:px2] = m[:max(0, py2 - py1), :max(0, px2 - px1)]
    return full


def decode_masks(detections, myolo_mask, image_shape):
    """model.py:1330-1391 for one image: detections [N,6], myolo_mask [N,h,w,C]."""
    N = len(detections)
    boxes = detections[:N, :4]
    scores = detections[:N, 4]
    class_ids = detections[:N, 5].astype(np.int32)
    masks = myolo_mask[np.arange(N), :, :, class_ids]
    excl = np.where((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) <= 0)[0]
    if excl.shape[0] > 0:
        boxes, class_ids = np.delete(boxes, excl, axis=0), np.delete(class_ids, excl, axis=0)
        scores, masks = np.delete(scores, excl, axis=0), np.delete(masks, excl, axis=0)
        N = class_ids.shape[0]
    full = [unmold_mask(masks[i], boxes[i], image_shape) for i in range(N)]
    full = np.stack(full, axis=-1) if full else np.empty(tuple(image_shape[:2]) + (0,))
    return boxes, class_ids, scores, full


def detect_post(detections, myolo_mask, image_shape, cs_threshold=0.35):
    """model.py:1290-1321 minus the debug override at :1306 (`nmb_indices = [109, 130]`)."""
    boxes, class_ids, scores, full = decode_masks(detections, myolo_mask, image_shape)
    top10 = np.argsort(scores)[::-1][:10]
    kept = np.array([i for i in top10 if scores[i] >= cs_threshold], dtype=np.int64)
    idx = nmb(boxes[kept], class_ids[kept], kept, image_shape, nms_threshold=0.7) if len(kept) else kept
    idx = np.asarray(idx, dtype=np.int64)
    # model.py:1307 scales the kept boxes to pixels (`i * 224`, the reference's only image size): x by the width, y by the height
    scale = np.array([image_shape[1], image_shape[0], image_shape[1], image_shape[0]], dtype=boxes.dtype)
    return dict(bboxes=boxes[idx] * scale, class_ids=class_ids[idx], confidence_scores=scores[idx], full_masks=full[:, :, idx])
