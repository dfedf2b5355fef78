"""
Host-side data layer of the Mask-YOLO hot path (numpy): target encoding and the small
post-processing helpers.  Same names / argument meaning as the reference's
``myolo/myolo_utils.py`` for the rows SURVEY.md section 8 marks in scope:

  BatchGenerator.__getitem__  myolo_utils.py:727-860   (a19)
  extract_bboxes              myolo_utils.py:247-271   (a20)
  load_image_gt               myolo_utils.py:274-366   (no resize/augment: out of scope)
  BoundBox/bbox_iou           myolo_utils.py:161-244
  decode_one_yolo_output/NMB/unmold_mask  :36-113,883-912  ("next" rows, section 8(f))

Differences: image size comes from config.IMAGE_SHAPE instead of the
hard-coded 224 (myolo_utils.py:737,744); no cv2 drawing branch (norm=False path).
"""
import numpy as np


class BoundBox:
    def __init__(self, xmin, ymin, xmax, ymax, c=None, classes=None):
        self.xmin, self.ymin, self.xmax, self.ymax = xmin, ymin, xmax, ymax
        self.c = c
        self.classes = classes
        self.label = -1
        self.score = -1

    def get_label(self):
        if self.label == -1:
            self.label = np.argmax(self.classes)
        return self.label

    def get_score(self):
        if self.score == -1:
            self.score = self.classes[self.get_label()]
        return self.score


def _interval_overlap(interval_a, interval_b):
    x1, x2 = interval_a
    x3, x4 = interval_b
    if x3 < x1:
        if x4 < x1:
            return 0
        return min(x2, x4) - x1
    if x2 < x3:
        return 0
    return min(x2, x4) - x3


def bbox_iou(box1, box2):
    iw = _interval_overlap([box1.xmin, box1.xmax], [box2.xmin, box2.xmax])
    ih = _interval_overlap([box1.ymin, box1.ymax], [box2.ymin, box2.ymax])
    inter = iw * ih
    w1, h1 = box1.xmax - box1.xmin, box1.ymax - box1.ymin
    w2, h2 = box2.xmax - box2.xmin, box2.ymax - box2.ymin
    return float(inter) / (w1 * h1 + w2 * h2 - inter)


def bbox_iou_2(box1, box2, image_shape):
    """myolo_utils.py:201-228.  The rows detect() passes are float32 (model.py:1304) and the reference multiplies their
    elements by the Python ints of image_shape: a float64 product under the scalar promotion of every numpy before 2.0
    (the ones the reference ran with), a float32 one from 2.0 on.  float() pins the former, whatever numpy runs this
    (tests/golden/ref_host_boxes.npz holds the reference's own outputs)."""
    w, h = image_shape[0], image_shape[1]
    a = BoundBox(float(box1[0]) * w, float(box1[1]) * h, float(box1[2]) * w, float(box1[3]) * h)
    b = BoundBox(float(box2[0]) * w, float(box2[1]) * h, float(box2[2]) * w, float(box2[3]) * h)
    return bbox_iou(a, b)


def extract_bboxes(mask):
    """mask [H,W,N] -> [N,(x1,y1,x2,y2)] int32; x2,y2 exclusive (myolo_utils.py:247-271)."""
    n = mask.shape[-1]
    boxes = np.zeros([n, 4], dtype=np.int32)
    cols = np.any(mask, axis=0)            # [W,N]
    rows = np.any(mask, axis=1)            # [H,N]
    for i in range(n):
        hz = np.where(cols[:, i])[0]
        if hz.shape[0]:
            vt = np.where(rows[:, i])[0]
            boxes[i] = (hz[0], vt[0], hz[-1] + 1, vt[-1] + 1)
    return boxes


def resize(image, output_shape, order=1, mode='constant', cval=0, clip=True, preserve_range=False, anti_aliasing=False,
           anti_aliasing_sigma=None):
    """The reference's scikit-image wrapper (myolo_utils.py:433-455) restated for the one way the path calls it: order 1, mode 'constant', cval 0,
    clip, no anti-aliasing -- skimage.transform.resize as a per-channel bilinear warp with pixel centres aligned (src = (dst + .5) * in / out - .5),
    samples outside the image reading cval, computed in float64, clipped to the input's range.  Pinned by tests/golden/ref_load_image_gt.npz
    (the reference's wrapper executed with scikit-image 0.18.3): float results to 1e-10, the uint8 image of resize_image to 1 count on < 0.1 % of
    the pixels (skimage estimates its affine matrix by least squares, so a value that is an integer up to rounding may truncate either way)."""
    if order != 1 or mode != 'constant' or anti_aliasing:
        raise NotImplementedError("resize: only order=1, mode='constant', anti_aliasing=False (what myolo_utils.py:388,903 use)")
    img = np.asarray(image)
    a = img.astype(np.float64)
    if not preserve_range:
        if img.dtype == np.uint8:
            a = a / 255.
        elif img.dtype == bool:
            a = img.astype(np.float64)
    h, w = a.shape[:2]
    oh, ow = int(output_shape[0]), int(output_shape[1])
    flat = a.ndim == 2
    if flat:
        a = a[:, :, None]
    ys = (np.arange(oh, dtype=np.float64) + 0.5) * (h / float(oh)) - 0.5
    xs = (np.arange(ow, dtype=np.float64) + 0.5) * (w / float(ow)) - 0.5
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    y1, x1 = np.ceil(ys).astype(int), np.ceil(xs).astype(int)
    dy, dx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    P = np.full((h + 2, w + 2, a.shape[2]), float(cval))
    P[1:-1, 1:-1] = a

    def px(yi, xi):
        return P[np.clip(yi + 1, 0, h + 1)][:, np.clip(xi + 1, 0, w + 1)]
    top = (1 - dx) * px(y0, x0) + dx * px(y0, x1)
    bot = (1 - dx) * px(y1, x0) + dx * px(y1, x1)
    out = (1 - dy) * top + dy * bot
    if clip:
        out = np.clip(out, a.min(), a.max())
    return out[:, :, 0] if flat else out


def resize_image(image, net_image_shape):
    """-> (image resized to net_image_shape[:2] in its own dtype, [scale_y, scale_x]) (myolo_utils.py:369-392)."""
    image_dtype = image.dtype
    h, w = image.shape[:2]
    scale = [net_image_shape[0] / h, net_image_shape[1] / w]
    if scale != [1, 1]:
        image = resize(image, (round(h * scale[0]), round(w * scale[1])), preserve_range=True)
    return image.astype(image_dtype), scale


def resize_mask(mask, scale):
    """scipy.ndimage.zoom(mask, [scale_y, scale_x, 1], order=0) restated (myolo_utils.py:395-411): output extent round(n * scale), output
    index o reads input index floor(o * (n - 1) / (out - 1) + .5) (corner-aligned nearest neighbour)."""
    h, w = mask.shape[:2]
    oh, ow = int(round(h * scale[0])), int(round(w * scale[1]))

    def idx(n, o):
        if o <= 1:
            return np.zeros(max(o, 0), int)
        return np.floor(np.arange(o) * ((n - 1) / float(o - 1)) + 0.5).astype(int)
    return mask[idx(h, oh)][:, idx(w, ow)]


def load_image_gt(dataset, config, image_id, augment=False, augmentation=None, use_mini_mask=False):
    """-> image, class_ids, bbox [n,(x1,y1,x2,y2)], mask [H,W,n] (myolo_utils.py:274-366): load, resize image and masks to
    config.IMAGE_SHAPE, drop the instances whose mask came out empty, tight boxes.  `augment=True` is the reference's deprecated random horizontal flip; imgaug
    augmentation and mini-masks are image-file I/O options outside the hot path and are refused.  Pinned by tests/golden/ref_load_image_gt.npz (the reference's function executed)."""
    if augmentation is not None or use_mini_mask:
        raise NotImplementedError("imgaug augmentation / mini-masks are outside the hot path")
    image = dataset.load_image(image_id)
    mask, class_ids = dataset.load_mask(image_id)
    if list(image.shape[:2]) != list(config.IMAGE_SHAPE[:2]):
        image, scale = resize_image(image, config.IMAGE_SHAPE)
        mask = resize_mask(mask, scale)
    if augment:
        # the deprecated random horizontal flip (myolo_utils.py:306-311): ONE draw of the global `random` module per image, as the reference makes it
        import logging
        import random
        logging.warning("'augment' is deprecated. Use 'augmentation' instead.")
        if random.randint(0, 1):
            image = np.fliplr(image)
            mask = np.fliplr(mask)
    keep = np.sum(mask, axis=(0, 1)) > 0
    mask = mask[:, :, keep]
    class_ids = class_ids[keep]
    bbox = extract_bboxes(mask)
    return image, class_ids, bbox, mask


class BatchGenerator(object):
    """Target encoding + batching (myolo_utils.py:689-860).  __getitem__ returns
    ([images f32, true_boxes f64, y_true f64, gt_class_ids i32, gt_boxes i32, gt_masks bool], [])
    in 'training' mode, the first three in 'yolo' mode."""

    def __init__(self, all_info, config, mode, shuffle=True, jitter=False, norm=False, rng=None):
        assert mode in ['yolo', 'training']
        self.config = config
        self.mode = mode
        self.all_info = all_info
        self.shuffle = shuffle
        self.jitter = jitter
        self.norm = norm
        a = np.asarray(config.ANCHORS, dtype=np.float64).reshape(-1, 2)
        self.anchor_w, self.anchor_h = a[:, 0], a[:, 1]
        if shuffle:
            (rng or np.random).shuffle(self.all_info)

    def __len__(self):
        return int(np.ceil(float(len(self.all_info)) / self.config.BATCH_SIZE))

    def num_classes(self):
        return self.config.NUM_CLASSES

    def size(self):
        return len(self.all_info)

    def _best_anchor(self, w, h):
        """argmax IoU of origin-anchored boxes, first maximum wins (myolo_utils.py:795-809)."""
        iw = np.minimum(w, self.anchor_w)
        ih = np.minimum(h, self.anchor_h)
        inter = iw * ih
        iou = inter / (w * h + self.anchor_w * self.anchor_h - inter)
        return int(np.argmax(iou))

    def batch_bounds(self, idx):
        """[l_bound, r_bound) of batch idx; the last batch is wrapped back to full size (myolo_utils.py:728-735)."""
        cfg = self.config
        l_bound = idx * cfg.BATCH_SIZE
        r_bound = (idx + 1) * cfg.BATCH_SIZE
        if r_bound > len(self.all_info):
            r_bound = len(self.all_info)
            l_bound = max(0, r_bound - cfg.BATCH_SIZE)
        return l_bound, r_bound

    def __getitem__(self, idx):
        cfg = self.config
        l_bound, r_bound = self.batch_bounds(idx)
        n = r_bound - l_bound
        H, W = cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1]
        T = cfg.TRUE_BOX_BUFFER
        images = np.zeros((n, H, W, 3), dtype=np.float32)
        y_true = np.zeros((n, cfg.GRID_H, cfg.GRID_W, cfg.N_BOX, 4 + 1 + cfg.NUM_CLASSES))
        true_boxes = np.zeros((n, 1, 1, 1, T, 4))
        gt_ids = np.zeros((n, T), dtype=np.int32)
        gt_boxes_b = np.zeros((n, T, 4), dtype=np.int32)
        gt_masks_b = np.zeros((n, H, W, cfg.MAX_GT_INSTANCES), dtype=bool)
        self._encode(l_bound, r_bound, images, true_boxes, y_true, gt_ids, gt_boxes_b, gt_masks_b)
        if self.mode == 'yolo':
            return [images, true_boxes, y_true], []
        return [images, true_boxes, y_true, gt_ids, gt_boxes_b, gt_masks_b], []

    def fill(self, idx, out):
        """__getitem__(idx)'s arrays written INTO caller-provided buffers `out` (same order; any float dtype for the three float
        arrays, uint8 or bool masks; true_boxes may be [n,T,4]) -- MaskYOLO.train() hands in the engine's pinned staging buffers, so a
        batch is encoded straight into the memory the H2D copy reads (no 35 MB of fresh arrays and no second host copy per step).
        Same values as __getitem__ after the cast the upload applies anyway."""
        l_bound, r_bound = self.batch_bounds(idx)
        for a in out[1:]:
            a.fill(0)                                # (images are overwritten completely)
        tb = out[1].reshape(r_bound - l_bound, 1, 1, 1, -1, 4)
        if self.mode == 'yolo':
            self._encode(l_bound, r_bound, out[0], tb, out[2], None, None, None)
        else:
            self._encode(l_bound, r_bound, out[0], tb, out[2], out[3], out[4], out[5])
        return out

    def _encode(self, l_bound, r_bound, images, true_boxes, y_true, gt_ids, gt_boxes_b, gt_masks_b):
        """the body of myolo_utils.py:748-844 on zero-initialised outputs."""
        cfg = self.config
        H, W = cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1]
        T = cfg.TRUE_BOX_BUFFER
        cell_w = float(W) / cfg.GRID_W
        cell_h = float(H) / cfg.GRID_H
        for k, inst in enumerate(self.all_info[l_bound:r_bound]):
            image, class_ids, boxes, masks = inst[0], inst[1], inst[2], inst[3]
            if boxes.shape[0] > T:
                ids = np.random.choice(np.arange(boxes.shape[0]), T, replace=False)
                class_ids, boxes, masks = class_ids[ids], boxes[ids], masks[:, :, ids]
            tbi = 0
            for i in range(boxes.shape[0]):
                xmin, ymin, xmax, ymax = (boxes[i][0], boxes[i][1], boxes[i][2], boxes[i][3])
                cx = .5 * (xmin + xmax) / cell_w
                cy = .5 * (ymin + ymax) / cell_h
                gx, gy = int(np.floor(cx)), int(np.floor(cy))
                if gx < cfg.GRID_W and gy < cfg.GRID_H:
                    bw = (xmax - xmin) / cell_w
                    bh = (ymax - ymin) / cell_h
                    box = [cx, cy, bw, bh]
                    a = self._best_anchor(bw, bh)
                    y_true[k, gy, gx, a, 0:4] = box
                    y_true[k, gy, gx, a, 4] = 1.
                    y_true[k, gy, gx, a, 5 + class_ids[i]] = 1
                    true_boxes[k, 0, 0, 0, tbi] = box
                    tbi = (tbi + 1) % T
            if images.dtype == np.uint8:
                # a byte batch (Net.stage_batch): the `/ 255.` happens on the device (myolo_u8_to_unit_f32, same bits)
                assert self.norm and image.dtype == np.uint8, "byte staging needs uint8 images and norm=True"
                images[k] = image
            elif self.norm and image.dtype == np.uint8 and images.dtype == np.float32:
                # image / 255. (float64) stored into the float32 batch (myolo_utils.py:824) = one correctly rounded float32 per byte
                # value: a 256-entry table gives the same bits without the float64 temporary
                np.take(_U8_OVER_255, image, out=images[k])
            else:
                images[k] = image / 255. if self.norm else image
            if gt_ids is not None:
                gt_ids[k, :class_ids.shape[0]] = class_ids
                gt_boxes_b[k, :boxes.shape[0]] = boxes
                for j in range(masks.shape[-1]):         # (channel by channel: load_mask stacks the instances channel-major, one
                    gt_masks_b[k, :, :, j] = masks[:, :, j]   #  contiguous plane each -- twice as fast as the strided block assignment)


_U8_OVER_255 = (np.arange(256) / 255.).astype(np.float32)


# ---------------------------------------------------------------------------
# "next" rows: inference post-processing on the host
# ---------------------------------------------------------------------------
def _sigmoid(x):
    return 1. / (1. + np.exp(-x))


def _softmax(x, axis=-1, t=-100.):
    x = x - np.max(x)
    if np.min(x) < t:
        x = x / np.min(x) * t
    e_x = np.exp(x)
    return e_x / e_x.sum(axis, keepdims=True)


def decode_one_yolo_output(netout, anchors, nb_class, obj_threshold=0.3, nms_threshold=0.3):
    """myolo_utils.py:36-85: per-class greedy NMS on the decoded YOLO grid.  -> list of BoundBox.

    Arithmetic types as the reference had them (pinned by tests/golden/ref_host_decode.npz, outputs of the reference's own
    function): the whole-array steps (:42-44) stay in the dtype of `netout` -- float32 for a network output (model.py:1224) --
    and so do the exp() calls on its elements; everything done to those elements one at a time with Python scalars
    (:56-60, the IoUs, the final `> obj_threshold`) is float64, which is what numpy < 2.0 promotes
    `np.float32 <op> Python float` to.  float() below spells that out so that numpy >= 2.0 (which would stay in float32) gives
    the same boxes.  Unlike the reference (:42-44 write into the caller's array) the input is left untouched."""
    netout = np.array(netout, dtype=netout.dtype if getattr(netout, "dtype", None) in (np.float32, np.float64) else np.float64)
    grid_h, grid_w, nb_box = netout.shape[:3]
    boxes = []
    netout[..., 4] = _sigmoid(netout[..., 4])
    netout[..., 5:] = netout[..., 4][..., np.newaxis] * _softmax(netout[..., 5:])
    netout[..., 5:] *= netout[..., 5:] > obj_threshold
    for row in range(grid_h):
        for col in range(grid_w):
            for b in range(nb_box):
                classes = netout[row, col, b, 5:]
                if np.sum(classes) > 0:
                    x, y, w, h = netout[row, col, b, :4]
                    x = (col + 1. / (1. + float(np.exp(-x)))) / grid_w
                    y = (row + 1. / (1. + float(np.exp(-y)))) / grid_h
                    w = float(anchors[2 * b + 0]) * float(np.exp(w)) / grid_w
                    h = float(anchors[2 * b + 1]) * float(np.exp(h)) / grid_h
                    boxes.append(BoundBox(x - w / 2, y - h / 2, x + w / 2, y + h / 2, netout[row, col, b, 4], classes))
    for c in range(nb_class):
        order = list(reversed(np.argsort([box.classes[c] for box in boxes])))
        for i in range(len(order)):
            bi = order[i]
            if boxes[bi].classes[c] == 0:
                continue
            for j in range(i + 1, len(order)):
                bj = order[j]
                if bbox_iou(boxes[bi], boxes[bj]) >= nms_threshold:
                    boxes[bj].classes[c] = 0
    return [box for box in boxes if float(box.get_score()) > obj_threshold]


def NMB(boxes, class_ids, indices, image_shape, nms_threshold=0.3):
    """myolo_utils.py:88-113: same-class suppression over score-ordered indices."""
    remove = []
    for i in range(len(indices)):
        for j in range(i + 1, len(indices)):
            if bbox_iou_2(boxes[i], boxes[j], image_shape) >= nms_threshold and class_ids[i] == class_ids[j]:
                remove.append(j)
    return np.delete(indices, remove)


def _resize_bilinear(mask, out_h, out_w):
    """skimage.transform.resize(order=1, mode='constant', cval=0, anti_aliasing=False) as the reference's wrapper calls it
    (myolo_utils.py:433-447, 903): pixel centres aligned, bilinear, samples outside the mask read 0.  float32; same definition
    as the GPU kernel myolo_unmold_masks."""
    f = np.float32
    mask = np.asarray(mask, f)
    h, w = mask.shape
    ys = ((np.arange(out_h, dtype=f) + f(0.5)) * (f(h) / f(out_h)) - f(0.5)).astype(f)
    xs = ((np.arange(out_w, dtype=f) + f(0.5)) * (f(w) / f(out_w)) - f(0.5)).astype(f)
    y0, x0 = np.floor(ys).astype(int), np.floor(xs).astype(int)
    wy, wx = (ys - y0.astype(f))[:, None], (xs - x0.astype(f))[None, :]
    P = np.zeros((h + 2, w + 2), f)
    P[1:-1, 1:-1] = mask
    tl, tr = P[y0 + 1][:, x0 + 1], P[y0 + 1][:, x0 + 2]
    bl, br = P[y0 + 2][:, x0 + 1], P[y0 + 2][:, x0 + 2]
    top = tl + (tr - tl) * wx
    bot = bl + (br - bl) * wx
    return np.clip((top + (bot - top) * wy).astype(f), mask.min(), mask.max()).astype(f)      # clip=True


def unmold_mask(mask, bbox, image_shape):
    """myolo_utils.py:883-912: resize a 28x28 mask to its box, threshold 0.5, paste."""
    threshold = np.float32(0.5)
    w, h = image_shape[0], image_shape[1]
    x1, y1, x2, y2 = [np.float32(v) for v in bbox]
    x1 = min(max(0, int(x1 * np.float32(w))), w)
    x2 = min(max(1, int(x2 * np.float32(w))), w)
    y1 = min(max(0, int(y1 * np.float32(h))), h)
    y2 = min(max(1, int(y2 * np.float32(h))), h)
    m = _resize_bilinear(mask, max(1, y2 - y1), max(1, x2 - x1))
    m = np.where(m >= threshold, 1, 0).astype(bool)
    full = np.zeros(image_shape[:2], dtype=bool)
    full[y1:y2, x1:x2] = m[:max(0, y2 - y1), :max(0, x2 - x1)]
    return full
