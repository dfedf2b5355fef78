"""
Synthetic Shapes dataset -- the inputs BASELINE.json's metric is quoted on.

Restates example/shapes/dataset_shapes.py:53-180 (ShapesDataset: random_image :158-180,
random_shape :137-156, draw_shape :121-135, load_mask :102-119) without cv2 / mrcnn:
the un-vendored ``mrcnn.utils.Dataset`` base and ``mrcnn.utils.non_max_suppression``
(dataset_shapes.py:6,53,178) are restated inline.  The reference is unseeded; here every
image is a pure function of (seed, image index) so that all ranks / runs agree.
"""
import math
import random

import numpy as np


def _nms(boxes, scores, threshold):
    """matterport mrcnn.utils.non_max_suppression restated: visit by score descending, drop
    boxes whose IoU with the kept one exceeds threshold.  boxes [N,(a1,b1,a2,b2)]."""
    if boxes.shape[0] == 0:
        return np.zeros((0,), np.int32)
    b = boxes.astype(np.float32)
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    ixs = scores.argsort()[::-1]
    pick = []
    while len(ixs) > 0:
        i = ixs[0]
        pick.append(i)
        rest = ixs[1:]
        y1 = np.maximum(b[i, 0], b[rest, 0])
        y2 = np.minimum(b[i, 2], b[rest, 2])
        x1 = np.maximum(b[i, 1], b[rest, 1])
        x2 = np.minimum(b[i, 3], b[rest, 3])
        inter = np.maximum(x2 - x1, 0) * np.maximum(y2 - y1, 0)
        iou = inter / (area[i] + area[rest] - inter)
        ixs = rest[iou <= threshold]
    return np.array(pick, dtype=np.int32)


class ShapesDataset(object):
    """Squares / circles / triangles on a flat background, generated on the fly."""

    def __init__(self, seed=1234):
        self.seed = seed
        self.image_info = []
        self.class_info = [{"source": "", "id": 0, "name": "BG"}]
        self.image_ids = []
        self.class_names = []
        self.source_class_ids = {}

    # ---- mrcnn.utils.Dataset surface used by load_image_gt (myolo_utils.py:299-300,358-359)
    def add_class(self, source, class_id, class_name):
        self.class_info.append({"source": source, "id": class_id, "name": class_name})

    def add_image(self, source, image_id, path, **kwargs):
        info = {"id": image_id, "source": source, "path": path}
        info.update(kwargs)
        self.image_info.append(info)

    def prepare(self):
        self.num_classes = len(self.class_info)
        self.class_ids = np.arange(self.num_classes)
        self.class_names = [c["name"] for c in self.class_info]
        self.num_images = len(self.image_info)
        self.image_ids = np.arange(self.num_images)
        self.source_class_ids = {"shapes": list(range(self.num_classes)), "": [0]}

    # ---- dataset_shapes.py:59-79
    def load_shapes(self, count, height, width, start_index=0):
        self.add_class("shapes", 1, "square")
        self.add_class("shapes", 2, "circle")
        self.add_class("shapes", 3, "triangle")
        for i in range(count):
            rng = random.Random(self.seed + start_index + i)
            bg_color, shapes = self.random_image(height, width, rng)
            self.add_image("shapes", image_id=i, path=None, width=width, height=height,
                           bg_color=bg_color, shapes=shapes)

    def load_image(self, image_id):
        info = self.image_info[image_id]
        bg = np.array(info['bg_color']).reshape([1, 1, 3])
        image = np.ones([info['height'], info['width'], 3], dtype=np.uint8) * bg.astype(np.uint8)
        for shape, color, dims in info['shapes']:
            image = self.draw_shape(image, shape, dims, color)
        return image

    def load_mask(self, image_id):
        info = self.image_info[image_id]
        shapes = info['shapes']
        count = len(shapes)
        mask = np.zeros([info['height'], info['width'], count], dtype=np.uint8)
        for i, (shape, _, dims) in enumerate(shapes):
            mask[:, :, i:i + 1] = self.draw_shape(mask[:, :, i:i + 1].copy(), shape, dims, 1)
        # occlusions: later shapes hide earlier ones (dataset_shapes.py:112-116)
        occlusion = np.logical_not(mask[:, :, -1]).astype(np.uint8)
        for i in range(count - 2, -1, -1):
            mask[:, :, i] = mask[:, :, i] * occlusion
            occlusion = np.logical_and(occlusion, np.logical_not(mask[:, :, i]))
        class_ids = np.array([self.class_names.index(s[0]) for s in shapes])
        return mask.astype(bool), class_ids.astype(np.int32)

    @staticmethod
    def draw_shape(image, shape, dims, color):
        """dataset_shapes.py:121-135 without cv2: square = inclusive [x-s,x+s]; circle =
        dx^2+dy^2 <= s^2; triangle = int-truncated vertices, edge-function fill."""
        x, y, s = dims
        H, W = image.shape[:2]
        yy, xx = np.mgrid[0:H, 0:W]
        if shape == 'square':
            m = (xx >= x - s) & (xx <= x + s) & (yy >= y - s) & (yy <= y + s)
        elif shape == 'circle':
            m = (xx - x) ** 2 + (yy - y) ** 2 <= s * s
        elif shape == 'triangle':
            k = s / math.sin(math.radians(60))
            pts = np.array([(x, y - s), (x - k, y + s), (x + k, y + s)]).astype(np.int32)
            (ax, ay), (bx, by), (cx, cy) = pts

            def edge(px, py, qx, qy):
                return (xx - px) * (qy - py) - (yy - py) * (qx - px)
            e0, e1, e2 = edge(ax, ay, bx, by), edge(bx, by, cx, cy), edge(cx, cy, ax, ay)
            m = ((e0 >= 0) & (e1 >= 0) & (e2 >= 0)) | ((e0 <= 0) & (e1 <= 0) & (e2 <= 0))
        else:
            raise ValueError(shape)
        image = image.copy()
        image[m] = color
        return image

    @staticmethod
    def random_shape(height, width, rng):
        shape = rng.choice(["square", "circle", "triangle"])
        color = tuple([rng.randint(0, 255) for _ in range(3)])
        buffer = 20 if height >= 80 else max(2, height // 5)   # reference: 20 (needs H >= 80)
        y = rng.randint(buffer, height - buffer - 1)
        x = rng.randint(buffer, width - buffer - 1)
        s = rng.randint(buffer, height // 4)
        return shape, color, (x, y, s)

    def random_image(self, height, width, rng):
        bg_color = np.array([rng.randint(0, 255) for _ in range(3)])
        shapes, boxes = [], []
        N = rng.randint(1, 4)
        for _ in range(N):
            shape, color, dims = self.random_shape(height, width, rng)
            shapes.append((shape, color, dims))
            x, y, s = dims
            boxes.append([x - s, y - s, x + s, y + s])
        keep = _nms(np.array(boxes), np.arange(N), 0.3)
        shapes = [s for i, s in enumerate(shapes) if i in keep]
        return bg_color, shapes


def make_shapes_samples(count, cfg, seed=1234, start_index=0):
    """-> list of (image uint8, class_ids, boxes int32 x1y1x2y2, masks bool) like the
    train_info rows MaskYOLO.train builds (model.py:995-999)."""
    from .myolo_utils import load_image_gt
    ds = ShapesDataset(seed)
    ds.load_shapes(count, cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1], start_index=start_index)
    ds.prepare()
    return [list(load_image_gt(ds, cfg, i)) for i in range(count)]


class ShapesProducer(object):
    """GPU producer of Shapes training batches.  The random shape specifications come
    from the same seeded generator as ShapesDataset (so image g of the stream is identical to the host pipeline's);
    rasterisation, the empty-instance filter, extract_bboxes and BatchGenerator's target encoding run on the device
    (libmyolo_hip.so: myolo_shapes_batch) and the result is the device batch dict Net.forward_backward consumes."""

    SHAPE_INTS, MAX_SHAPES = 13, 4
    _TYPES = {"square": 1, "circle": 2, "triangle": 3}

    def __init__(self, cfg, seed=1234, device="cuda:0"):
        import torch
        from . import _ext as X
        X.load()
        self.cfg, self.seed = cfg, seed
        self.dev = torch.device(device)
        self._ds = ShapesDataset(seed)
        self.stride = 4 + self.MAX_SHAPES * self.SHAPE_INTS
        self.lut = torch.tensor((np.arange(256) / 255.).astype(np.float32), device=self.dev)
        self.anchors = torch.tensor(np.asarray(cfg.ANCHORS, np.float64), device=self.dev)
        self.ws = torch.empty(1 << 20, dtype=torch.uint8, device=self.dev)
        self._pin, self._pin_i = None, 0

    def specs(self, indices):
        H, W = self.cfg.IMAGE_SHAPE[0], self.cfg.IMAGE_SHAPE[1]
        out = np.zeros((len(indices), self.stride), np.int32)
        for k, g in enumerate(indices):
            bg, shapes = self._ds.random_image(H, W, random.Random(self.seed + int(g)))
            out[k, 0:3] = bg
            out[k, 3] = len(shapes)
            for j, (shape, color, (x, y, s_)) in enumerate(shapes):
                o = 4 + j * self.SHAPE_INTS
                out[k, o:o + 7] = (self._TYPES[shape],) + tuple(color) + (x, y, s_)
                if shape == "triangle":
                    kk = s_ / math.sin(math.radians(60))
                    out[k, o + 7:o + 13] = np.array([x, y - s_, x - kk, y + s_, x + kk, y + s_]).astype(np.int32)
        return out

    def batch(self, indices, stream=None, consumer=None):
        """-> dict(images, true_boxes, y_true, gt_ids, gt_boxes, gt_masks) of device tensors.
        stream (a torch.cuda.Stream): produce there instead of on the current stream -- the specifications go up from a pinned ring buffer without
        blocking the host (a pageable `.to(device)` waits for everything queued on the current stream: the whole previous step), the batch carries
        the event `_ready` that Net.forward_backward waits for, and its tensors are registered with `consumer` (the stream that will read them)."""
        import torch
        from . import _ext as X
        if stream is not None:
            with torch.cuda.stream(stream):
                d = self.batch(indices)
                ev = torch.cuda.Event()
                ev.record(stream)
            if consumer is not None:
                for t in d.values():
                    t.record_stream(consumer)
            d["_ready"] = ev
            return d
        cfg = self.cfg
        B = len(indices)
        H, W = cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1]
        G, A, C, T = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER
        sp = self.specs(indices)
        if self._pin is None or self._pin[0][0].shape != sp.shape:
            self._pin = [(torch.empty(sp.shape, dtype=torch.int32, pin_memory=True), torch.cuda.Event()) for _ in range(4)]
            self._pin_i = 0
        host, used = self._pin[self._pin_i]
        self._pin_i = (self._pin_i + 1) % len(self._pin)
        used.synchronize()                       # the copy that last read this slot (four batches ago) is done
        host.numpy()[...] = sp
        spec = host.to(self.dev, non_blocking=True)
        used.record(torch.cuda.current_stream())
        d = dict(images=torch.empty(B, H, W, 3, device=self.dev),
                 gt_masks=torch.empty(B, H, W, T, dtype=torch.uint8, device=self.dev),
                 gt_boxes=torch.empty(B, T, 4, dtype=torch.int32, device=self.dev),
                 gt_ids=torch.empty(B, T, dtype=torch.int32, device=self.dev),
                 y_true=torch.empty(B, G, G, A, 5 + C, device=self.dev),
                 true_boxes=torch.empty(B, T, 4, device=self.dev))
        X.call("myolo_shapes_batch", X.ptr(spec), self.stride, X.ptr(self.anchors), X.ptr(self.lut), X.ptr(d["images"]),
               X.ptr(d["gt_masks"]), X.ptr(d["gt_boxes"]), X.ptr(d["gt_ids"]), X.ptr(d["y_true"]), X.ptr(d["true_boxes"]),
               B, H, W, self.MAX_SHAPES, T, G, A, C, self.ws.data_ptr(), self.ws.numel(), X.stream())
        self._keep = spec          # the kernels read it asynchronously
        return d
