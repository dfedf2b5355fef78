"""Rows a20 (extract_bboxes), f1 (decode_one_yolo_output / NMB / bbox_iou*) and the anchor choice of a19 against fixtures
produced by EXECUTING THE REFERENCE'S OWN FUNCTIONS (tests/golden/make_ref_host_fixtures.py: definitions taken from
/root/reference/myolo/myolo_utils.py:21-113,161-271 with `ast`, run with numpy 1.26.4).  Only the .npz files are read here.

Checked: the oracle (oracle/np_post.py, oracle/np_ops.py) AND the product's host functions (myolo/myolo_utils.py).
Decisions (boxes kept, labels, indices, integer boxes) bit-exact; float64 values to 1e-12 relative (numpy builds differ by
1 ulp in float64 exp); pure +,-,*,/ results (IoUs) exactly equal.
"""
import os

import numpy as np
import pytest

from oracle import np_ops, np_post
from myolo import myolo_utils as mutils

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(G, name))


def _unpack(fx, i):
    shape = tuple(int(v) for v in fx["mask_%02d_shape" % i])
    n = int(np.prod(shape))
    bits = np.unpackbits(fx["mask_%02d_bits" % i])[:n].reshape(shape)
    return bits.astype(str(fx["mask_%02d_dtype" % i]))


def test_fixtures_say_where_they_come_from():
    for name in ("ref_host_extract_bboxes.npz", "ref_host_boxes.npz", "ref_host_decode.npz", "ref_host_nmb.npz",
                 "ref_host_shapes.npz", "ref_load_image_gt.npz"):
        prov = str(_load(name)["provenance"])
        assert "/root/reference/myolo/myolo_utils.py" in prov and "numpy 1.26" in prov, prov
    prov = str(_load("ref_shapes_draws.npz")["provenance"])
    assert "/root/reference/example/shapes/dataset_shapes.py" in prov and "numpy 1.26" in prov


# ---------------------------------------------------------------------------------------------------- a20
def test_extract_bboxes_matches_reference_output():
    fx = _load("ref_host_extract_bboxes.npz")
    total = 0
    for i in range(int(fx["n_stacks"])):
        m = _unpack(fx, i)
        want = fx["bbox_%02d" % i]
        for fn in (np_ops.extract_bboxes, mutils.extract_bboxes):
            got = fn(m)
            assert got.dtype == np.int32 and got.shape == want.shape
            np.testing.assert_array_equal(got, want)
        total += m.shape[-1]
    assert total >= 200


# ---------------------------------------------------------------------------------------------------- box helpers
def test_interval_overlap_and_iou_match_reference_output():
    fx = _load("ref_host_boxes.npz")
    iv = fx["interval_in"]
    for fn in (np_post._interval_overlap, mutils._interval_overlap, np_ops._interval_overlap):
        got = np.array([fn([r[0], r[1]], [r[2], r[3]]) for r in iv], np.float64)
        np.testing.assert_array_equal(got, fx["interval_out"])
    a, b = fx["iou_a"], fx["iou_b"]
    got = np.array([np_post.bbox_iou(p, q) for p, q in zip(a, b)])
    np.testing.assert_array_equal(got, fx["iou_out"])
    got = np.array([mutils.bbox_iou(mutils.BoundBox(*p), mutils.BoundBox(*q)) for p, q in zip(a, b)])
    np.testing.assert_array_equal(got, fx["iou_out"])
    assert (fx["iou_out"] == 1.0).sum() >= 60 and (fx["iou_out"] == 0.0).sum() >= 150       # identical / touching / disjoint hit


@pytest.mark.parametrize("tag,shape", [("224", (224, 224, 3)), ("416x320", (416, 320, 3))])
def test_bbox_iou_2_matches_reference_output_also_for_float32_rows(tag, shape):
    """detect() hands float32 detection rows to NMB (model.py:1304); under the numpy the reference ran with, float32 scalar x
    Python int is a float64 product -- the product converts explicitly so that numpy >= 2 gives the same number."""
    fx = _load("ref_host_boxes.npz")
    a, b = fx["iou_a"], fx["iou_b"]
    got = np.array([mutils.bbox_iou_2(p, q, shape) for p, q in zip(a, b)])
    np.testing.assert_array_equal(got, fx["iou2_out_" + tag])
    a32, b32 = a.astype(np.float32), b.astype(np.float32)
    got = np.array([mutils.bbox_iou_2(p, q, shape) for p, q in zip(a32, b32)])
    np.testing.assert_array_equal(got, fx["iou2_f32_out_" + tag])


# ---------------------------------------------------------------------------------------------------- a19: anchor choice
@pytest.mark.parametrize("name", ["shapes", "rice", "food"])
def test_best_anchor_is_first_maximum_of_the_reference_ious(name):
    fx = _load("ref_host_boxes.npz")
    wh, anc, ious = fx["anchor_wh"], fx["anchors_" + name], fx["anchor_iou_" + name]

    class Cfg:
        ANCHORS = list(anc)
        BATCH_SIZE = 1
    gen = mutils.BatchGenerator([], Cfg, "yolo", shuffle=False)
    for (w, h), row in zip(wh, ious):
        best, mx = -1, -1
        for j, v in enumerate(row):                  # the strict `<` scan of myolo_utils.py:801-809 over reference IoUs
            if mx < v:
                best, mx = j, v
        assert gen._best_anchor(w, h) == best
        got = [np_ops._bbox_iou_wh(w, h, anc[2 * j], anc[2 * j + 1]) for j in range(len(anc) // 2)]
        np.testing.assert_array_equal(np.array(got), row)


# ---------------------------------------------------------------------------------------------------- f1: decode
def test_sigmoid_softmax_match_reference_output():
    fx = _load("ref_host_decode.npz")
    with np.errstate(over="ignore"):
        np.testing.assert_allclose(mutils._sigmoid(fx["sigmoid_in"]), fx["sigmoid_out"], rtol=1e-14, atol=1e-300)
        np.testing.assert_array_equal(mutils._sigmoid(fx["sigmoid_in_f32"]), fx["sigmoid_out_f32"])
    for s, want, want32 in zip(fx["softmax_in"], fx["softmax_out"], fx["softmax_out_f32"]):
        np.testing.assert_allclose(mutils._softmax(s), want, rtol=1e-13)
        got32 = mutils._softmax(s.astype(np.float32))
        assert got32.dtype == np.float32
        np.testing.assert_allclose(got32, want32, rtol=3e-7)


def _rows(boxes, ncls):
    if boxes and not hasattr(boxes[0], "xmin"):      # oracle: list rows
        return np.array([[b[0], b[1], b[2], b[3], b[4]] + list(b[5]) for b in boxes], np.float64).reshape(len(boxes), 5 + ncls)
    return np.array([[b.xmin, b.ymin, b.xmax, b.ymax, b.c] + list(b.classes) for b in boxes], np.float64).reshape(len(boxes), 5 + ncls)


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_decode_one_yolo_output_matches_reference_output(which):
    fx = _load("ref_host_decode.npz")
    fn = mutils.decode_one_yolo_output if which == "product" else np_post.decode_one_yolo_output
    n = int(fx["n_dec"])
    assert n >= 50
    kept_total = 0
    for k in range(n):
        net = fx["dec_%02d_in" % k]
        ncls, obj_t, nms_t = fx["dec_%02d_par" % k]
        ncls = int(ncls)
        before = net.copy()
        with np.errstate(over="ignore", invalid="ignore"):
            boxes = fn(net, list(fx["dec_%02d_anchors" % k]), ncls, obj_threshold=float(obj_t), nms_threshold=float(nms_t))
        want = fx["dec_%02d_rows" % k]
        assert len(boxes) == want.shape[0], "case %d (%s): kept %d boxes, the reference kept %d" % (k, net.dtype, len(boxes), want.shape[0])
        got = _rows(boxes, ncls)
        # which classes survived NMS in which box: exact
        np.testing.assert_array_equal(got[:, 5:] > 0, want[:, 5:] > 0)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-300)
        if which == "product":
            np.testing.assert_array_equal(np.array([b.get_label() for b in boxes], np.int64), fx["dec_%02d_label" % k])
            np.testing.assert_allclose(np.array([b.get_score() for b in boxes], np.float64), fx["dec_%02d_score" % k], rtol=1e-12)
            np.testing.assert_array_equal(net, before)        # the product does not scribble on the caller's array
        kept_total += len(boxes)
    assert kept_total > 5000


# ---------------------------------------------------------------------------------------------------- f1: NMB
@pytest.mark.parametrize("which", ["product", "oracle"])
def test_nmb_matches_reference_output(which):
    fx = _load("ref_host_nmb.npz")
    fn = mutils.NMB if which == "product" else np_post.nmb
    n = int(fx["n_nmb"])
    suppressed = 0
    for k in range(n):
        par = fx["nmb_%02d_par" % k]
        shape = [int(par[1]), int(par[2]), int(par[3])]
        idx = fx["nmb_%02d_idx" % k]
        got = fn(fx["nmb_%02d_boxes" % k], fx["nmb_%02d_cls" % k], idx.copy(), shape, nms_threshold=float(par[0]))
        want = fx["nmb_%02d_out" % k]
        np.testing.assert_array_equal(np.asarray(got, np.int64), want)
        suppressed += len(idx) - len(want)
    assert suppressed >= 40


# ---------------------------------------------------------------------------------------------------- Shapes stream (host)
def test_shapes_stream_host_boxes_match_reference_output():
    """the masks of the product's host Shapes generator are the fixture's inputs (same bits under this numpy), and
    load_image_gt's boxes on them are what the reference's extract_bboxes returned."""
    from myolo.config import ShapesConfig, make_config
    from myolo.shapes import make_shapes_samples
    fx = _load("ref_host_shapes.npz")
    cfg = make_config(ShapesConfig, BATCH_SIZE=4)
    assert list(cfg.IMAGE_SHAPE[:2]) == list(fx["hw"])
    counts = fx["counts"]
    samples = make_shapes_samples(len(counts), cfg, seed=int(fx["seed"]))
    bits = np.unpackbits(fx["mask_bits"])
    o_bit, o_box = 0, 0
    for g, (_, ids, boxes, masks) in enumerate(samples):
        n = int(counts[g])
        assert masks.shape == (224, 224, n)
        want = bits[o_bit:o_bit + 224 * 224 * n].reshape(224, 224, n).astype(bool)
        np.testing.assert_array_equal(masks, want)
        np.testing.assert_array_equal(boxes, fx["boxes"][o_box:o_box + n])
        np.testing.assert_array_equal(ids, fx["class_ids"][o_box:o_box + n])
        o_bit += 224 * 224 * n
        o_box += n
    assert o_box == fx["boxes"].shape[0] >= 100


# ---------------------------------------------------------------------------------------------------- GPU: device producer
@pytest.mark.gpu
def test_gpu_shapes_producer_boxes_match_reference_output():
    """the DEVICE producer (myolo_shapes_batch through the C-ABI: rasterise, drop empty instances, tight boxes, target
    encoding) against what the reference's extract_bboxes / bbox_iou returned for the same 64 Shapes images: masks, boxes and
    class ids bit-exact; every box sits in the anchor slot that is the first maximum of the reference's IoUs."""
    import torch
    from myolo.config import ShapesConfig, make_config
    from myolo.shapes import ShapesProducer
    fx = _load("ref_host_shapes.npz")
    counts = fx["counts"]
    n_img = len(counts)
    cfg = make_config(ShapesConfig, BATCH_SIZE=n_img)
    d = ShapesProducer(cfg, seed=int(fx["seed"])).batch(list(range(n_img)))
    torch.cuda.synchronize()
    gt_masks = d["gt_masks"].cpu().numpy().astype(bool)
    gt_boxes, gt_ids, y_true = d["gt_boxes"].cpu().numpy(), d["gt_ids"].cpu().numpy(), d["y_true"].cpu().numpy()
    bits = np.unpackbits(fx["mask_bits"])
    o_bit, o_box, placed = 0, 0, 0
    cell = 224.0 / cfg.GRID_W
    for g in range(n_img):
        n = int(counts[g])
        want = bits[o_bit:o_bit + 224 * 224 * n].reshape(224, 224, n).astype(bool)
        np.testing.assert_array_equal(gt_masks[g, :, :, :n], want)
        assert not gt_masks[g, :, :, n:].any()
        np.testing.assert_array_equal(gt_boxes[g, :n], fx["boxes"][o_box:o_box + n])
        assert not gt_boxes[g, n:].any() and not gt_ids[g, n:].any()
        np.testing.assert_array_equal(gt_ids[g, :n], fx["class_ids"][o_box:o_box + n])
        cells = {}
        for i in range(n):                                   # later boxes overwrite earlier ones in the same slot (:812-814)
            x1, y1, x2, y2 = [int(v) for v in fx["boxes"][o_box + i]]
            gx, gy = int(np.floor(.5 * (x1 + x2) / cell)), int(np.floor(.5 * (y1 + y2) / cell))
            row = fx["anchor_iou"][o_box + i]
            best, mx = -1, -1
            for j, v in enumerate(row):
                if mx < v:
                    best, mx = j, v
            cells[(gy, gx, best)] = int(fx["class_ids"][o_box + i])
        obj = np.argwhere(y_true[g, ..., 4] == 1)
        assert sorted(map(tuple, obj)) == sorted(cells.keys())
        for (gy, gx, a), cid in cells.items():
            assert y_true[g, gy, gx, a, 5 + cid] == 1
            placed += 1
        o_bit += 224 * 224 * n
        o_box += n
    assert placed >= 100


@pytest.mark.gpu
def test_gpu_detect_selection_equals_reference_nmb_on_network_outputs():
    """detect() (model.py:1290-1304) on real network outputs: the indices it keeps are those the reference-pinned NMB /
    bbox_iou_2 give on the device's float32 detections -- the selection runs on the host in the reference and here; this
    checks the whole chain from the GPU detections to the kept set through the product's public call."""
    import torch
    from myolo.config import ShapesConfig, make_config
    from myolo.model import MaskYOLO
    cfg = make_config(ShapesConfig, BATCH_SIZE=1)
    model = MaskYOLO("inference", cfg, seed=3)
    rng = np.random.default_rng(5)
    img = (rng.random((224, 224, 3)) * 255).astype(np.uint8)
    res = model.detect(img, cs_threshold=0.0)[0]
    x = torch.as_tensor((img[None] / 255.).astype(np.float32), device=model.net.dev)
    _, det, _ = model.net.predict(x)
    det = det[0].cpu().numpy()
    boxes, scores, cls = det[:, :4], det[:, 4], det[:, 5].astype(np.int32)
    keep = np.where((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) > 0)[0]
    boxes, scores, cls = boxes[keep], scores[keep], cls[keep]
    top = np.argsort(scores)[::-1][:10]
    idx = np_post.nmb(boxes[top], cls[top], top, cfg.IMAGE_SHAPE, nms_threshold=0.7)
    np.testing.assert_array_equal(res["confidence_scores"], scores[idx])
    np.testing.assert_array_equal(res["class_ids"], cls[idx])
    assert res["full_masks"].shape == (224, 224, len(idx))


# ---------------------------------------------------------------------------------------------------- round 5: config, load_image_gt, draws
# tests/golden/make_ref_pipeline_fixtures.py: myolo/config.py IMPORTED, load_image_gt + resize wrappers and ShapesDataset.random_shape /
# random_image EXECUTED from the reference files (scikit-image 0.18.3, scipy 1.7.1, numpy 1.26.4).  Only the fixtures are read here.
def _ref_config():
    import json

    def dec(v):
        if isinstance(v, dict) and "__ndarray__" in v:
            return np.array(v["__ndarray__"], dtype=v["dtype"])
        if isinstance(v, dict) and "__tuple__" in v:
            return tuple(dec(x) for x in v["__tuple__"])
        if isinstance(v, dict):
            return {k: dec(x) for k, x in v.items()}
        if isinstance(v, list):
            return [dec(x) for x in v]
        return v
    with open(os.path.join(G, "ref_config.json")) as f:
        return dec(json.load(f))


def _same(a, b):
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
        a, b = np.asarray(a), np.asarray(b)
        return a.dtype == b.dtype and np.array_equal(a, b)
    return type(a) is type(b) and a == b


def test_config_equals_the_imported_reference_config():
    """EVERY public attribute of the reference's Config class (config.py:15-257, imported as it is) exists on the product's Config with the
    same type and value -- on the class and on an instance -- except the three the product documents as deliberately recomputed
    (SURVEY appendix A): none for the base class (its derived fields agree).  Same for ShapesConfig, whose checked-in class inherits
    N_BOX = 5 with three anchors (dataset_shapes.py:39 vs config.py:30): there the product's self-consistent head differs in exactly
    N_BOX / TRAIN_ROIS_PER_IMAGE / the length of CLASS_WEIGHTS, and ShapesHeadConfig reproduces the checked-in head (N_BOX 5, config.py:28 anchors)."""
    from myolo.config import Config, ShapesConfig, ShapesHeadConfig
    ref = _ref_config()
    assert "/root/reference/myolo/config.py" in ref["provenance"] and "dataset_shapes.py" in ref["provenance"]
    assert len(ref["Config_class"]) >= 45
    for where, mine in (("Config_class", Config), ("Config_instance", Config())):
        for k, v in ref[where].items():
            assert hasattr(mine, k), "Config.%s is missing" % k
            assert _same(getattr(mine, k), v), (where, k, getattr(mine, k), v)
    sc = ShapesConfig()
    differ = {k for k, v in ref["ShapesConfig_instance"].items() if not _same(getattr(sc, k), v)}
    # appendix A's two fixes: the checked-in class inherits N_BOX = 5 beside three anchors, and CLASS_WEIGHTS sized by the BASE class's NUM_CLASSES
    assert differ == {"N_BOX", "TRAIN_ROIS_PER_IMAGE", "CLASS_WEIGHTS"}, differ
    assert ref["ShapesConfig_instance"]["CLASS_WEIGHTS"].shape == (2,) and ref["ShapesConfig_instance"]["NUM_CLASSES"] == 4
    assert sc.CLASS_WEIGHTS.shape == (4,) and sc.CLASS_WEIGHTS.dtype == np.float32 and (sc.CLASS_WEIGHTS == 1).all()
    assert (ref["ShapesConfig_instance"]["N_BOX"], ref["ShapesConfig_instance"]["TRAIN_ROIS_PER_IMAGE"]) == (5, 245)
    assert len(ref["ShapesConfig_instance"]["ANCHORS"]) == 6                       # the inconsistency the product's finalize() refuses
    assert sc.N_BOX == 3 and sc.TRAIN_ROIS_PER_IMAGE == 147 and sc.ANCHORS == ref["ShapesConfig_instance"]["ANCHORS"]
    hc = ShapesHeadConfig()
    assert hc.N_BOX == ref["ShapesConfig_instance"]["N_BOX"] and hc.TRAIN_ROIS_PER_IMAGE == ref["ShapesConfig_instance"]["TRAIN_ROIS_PER_IMAGE"]
    assert hc.ANCHORS == ref["Config_class"]["ANCHORS"]
    for k in ("NAME", "LABELS", "NUM_CLASSES", "BATCH_SIZE", "IMAGES_PER_GPU", "GPU_COUNT", "USE_MINI_MASK", "IMAGE_MIN_DIM", "IMAGE_MAX_DIM"):
        assert _same(getattr(ShapesConfig, k), ref["ShapesConfig_class"][k]), k


def _gt_case(fx, tag):
    h, w, start = [int(v) for v in fx[tag + "_hw_start"]]
    counts = fx[tag + "_counts"]
    bits = np.unpackbits(fx[tag + "_mask_bits"])
    return h, w, start, counts, bits


@pytest.mark.parametrize("tag", ["native", "resized"])
def test_load_image_gt_matches_reference_output(tag):
    """a19's inputs: the product's load_image_gt (+ resize_image / resize_mask / resize when the dataset's images are not network-sized)
    against what the REFERENCE's load_image_gt (myolo_utils.py:274-366, with real scikit-image / scipy.ndimage) returned for the same
    dataset: class ids, boxes, masks bit-exact; images bit-exact at the native size, within 1 count on < 0.1 % of the pixels when resized."""
    from myolo.config import ShapesConfig
    from myolo.shapes import ShapesDataset
    fx = _load("ref_load_image_gt.npz")
    assert "load_image_gt" in str(fx["provenance"]) and "scikit-image 0.18" in str(fx["provenance"])
    h, w, start, counts, bits = _gt_case(fx, tag)
    cfg = ShapesConfig()
    ds = ShapesDataset(int(fx["seed"]))
    ds.load_shapes(len(counts), h, w, start_index=start)
    ds.prepare()
    o_bit = o_box = 0
    off = 0
    for g in range(len(counts)):
        image, class_ids, bbox, mask = mutils.load_image_gt(ds, cfg, g)
        n = int(counts[g])
        want_img = fx[tag + "_images"][g]
        assert image.dtype == want_img.dtype and image.shape == want_img.shape
        if tag == "native":
            np.testing.assert_array_equal(image, want_img)
        else:
            d = np.abs(image.astype(np.int32) - want_img.astype(np.int32))
            assert d.max() <= 1
            off += int((d > 0).sum())
        assert mask.dtype == bool and mask.shape == (224, 224, n)
        np.testing.assert_array_equal(mask, bits[o_bit:o_bit + 224 * 224 * n].reshape(224, 224, n).astype(bool))
        np.testing.assert_array_equal(bbox, fx[tag + "_boxes"][o_box:o_box + n])
        assert bbox.dtype == np.int32
        np.testing.assert_array_equal(class_ids, fx[tag + "_class_ids"][o_box:o_box + n])
        o_bit += 224 * 224 * n
        o_box += n
    assert o_box == len(fx[tag + "_boxes"]) >= 25
    assert off <= 5e-3 * fx[tag + "_images"].size           # flat colours sit exactly on integers: (1-d)*v + d*v truncates either way


def test_load_image_gt_random_flip_matches_reference_output():
    """load_image_gt(augment=True), the deprecated random horizontal flip (myolo_utils.py:306-311): under the same random.seed the product draws what
    the reference drew and returns its image, class ids, boxes and masks bit for bit (24 images, 12 of them flipped in the fixture)."""
    import logging
    import random
    from myolo.config import ShapesConfig
    from myolo.shapes import ShapesDataset
    fx = _load("ref_load_image_gt.npz")
    h, w, start, counts, bits = _gt_case(fx, "flip")
    drawn = fx["flip_drawn"]
    assert 0 < drawn.sum() < len(drawn)
    ds = ShapesDataset(int(fx["seed"]))
    ds.load_shapes(len(counts), h, w, start_index=start)
    ds.prepare()
    cfg = ShapesConfig()
    o_bit = o_box = 0
    logging.disable(logging.WARNING)
    try:
        for g in range(len(counts)):
            random.seed(int(fx["flip_seed0"]) + g)
            image, class_ids, bbox, mask = mutils.load_image_gt(ds, cfg, g, augment=True)
            n = int(counts[g])
            np.testing.assert_array_equal(image, fx["flip_images"][g])
            np.testing.assert_array_equal(mask, bits[o_bit:o_bit + 224 * 224 * n].reshape(224, 224, n).astype(bool))
            np.testing.assert_array_equal(bbox, fx["flip_boxes"][o_box:o_box + n])
            np.testing.assert_array_equal(class_ids, fx["flip_class_ids"][o_box:o_box + n])
            plain = mutils.load_image_gt(ds, cfg, g)[0]
            assert np.array_equal(image, plain[:, ::-1] if drawn[g] else plain)
            o_bit += 224 * 224 * n
            o_box += n
    finally:
        logging.disable(logging.NOTSET)


def test_resize_wrappers_match_reference_output():
    """the scikit-image wrapper's arguments (order 1, constant 0 outside, clip, preserve_range, no anti-aliasing) and scipy's order-0 zoom, on
    inputs that are not piecewise constant: float result to 1e-10, the uint8 round trip to 1 count on < 0.1 %, the zoomed masks bit-exact."""
    fx = _load("ref_load_image_gt.npz")
    got = mutils.resize(fx["wrap_in"], (64, 80), preserve_range=True)
    assert got.shape == fx["wrap_out_float"].shape and np.abs(got - fx["wrap_out_float"]).max() < 1e-10
    img, scale = mutils.resize_image(fx["wrap_u8_in"], [96, 96, 3])
    assert img.dtype == np.uint8 and np.array_equal(np.asarray(scale, np.float64), fx["wrap_u8_scale"])
    d = np.abs(img.astype(np.int32) - fx["wrap_u8_out"].astype(np.int32))
    assert d.max() <= 1 and (d > 0).sum() <= 1e-3 * d.size
    m = np.unpackbits(fx["zoom_in_bits"])[:37 * 53 * 4].reshape(37, 53, 4).astype(bool)
    z = mutils.resize_mask(m, scale)
    assert z.dtype == bool
    np.testing.assert_array_equal(z, fx["zoom_out"])
    with pytest.raises(NotImplementedError):
        mutils.resize(fx["wrap_in"], (8, 8), order=3)


def test_random_shape_and_draw_order_match_reference_output():
    """ShapesDataset.random_shape (dataset_shapes.py:137-156) under random.seed(k), and the draw order of random_image (:158-180: background, N,
    then N x random_shape) with the boxes / scores / threshold it hands to the un-vendored mrcnn non_max_suppression -- the reference's own
    methods executed.  The product's generator consumes the stream in exactly that order; what it keeps after its restated suppression is a
    sub-sequence of the reference's pre-suppression list (the suppression itself is mrcnn's, not pinned here)."""
    import random
    from myolo.shapes import ShapesDataset
    fx = _load("ref_shapes_draws.npz")
    assert "dataset_shapes.py" in str(fx["provenance"])
    names = ["square", "circle", "triangle"]
    for h, w, k, t, c0, c1, c2, x, y, s in fx["random_shape"].tolist():
        shape, color, dims = ShapesDataset.random_shape(h, w, random.Random(k))
        assert (names.index(shape), tuple(color), tuple(dims)) == (t, (c0, c1, c2), (x, y, s)), (h, w, k)
    rows = fx["random_image_shapes"].tolist()
    o = 0
    ds = ShapesDataset(0)
    kept_total = 0
    for h, w, seed, n, b0, b1, b2 in fx["random_image_meta"].tolist():
        assert 1 <= n <= 4
        before = rows[o:o + n]
        o += n
        for t, c0, c1, c2, x, y, s, x1, y1, x2, y2 in before:
            assert [x1, y1, x2, y2] == [x - s, y - s, x + s, y + s]                    # the boxes handed to the suppression (:170-171)
        bg, shapes = ds.random_image(h, w, random.Random(seed))
        assert [int(v) for v in bg] == [b0, b1, b2]
        mine = [[names.index(sh), c[0], c[1], c[2], d[0], d[1], d[2]] for sh, c, d in shapes]
        spec = [r[:7] for r in before]
        it = iter(spec)
        assert all(m in it for m in mine), (seed, mine, spec)                          # an ordered sub-sequence of the draws
        assert spec[-1] in mine                                                        # the highest score (= last drawn) always survives
        kept_total += len(mine)
    assert o == len(rows) and kept_total < len(rows)


@pytest.mark.gpu
def test_gpu_shapes_producer_matches_reference_load_image_gt():
    """the DEVICE producer's whole batch against what the reference's load_image_gt returned for the same 64 images: pixels (image / 255. stored in
    float32, BatchGenerator's norm, myolo_utils.py:842) bit-exact, masks, boxes and class ids bit-exact."""
    import torch
    from myolo.config import ShapesConfig, make_config
    from myolo.shapes import ShapesProducer
    fx = _load("ref_load_image_gt.npz")
    h, w, start, counts, bits = _gt_case(fx, "native")
    n_img = len(counts)
    cfg = make_config(ShapesConfig, BATCH_SIZE=n_img)
    d = ShapesProducer(cfg, seed=int(fx["seed"])).batch(list(range(start, start + n_img)))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(d["images"].cpu().numpy(), (fx["native_images"] / 255.).astype(np.float32))
    gt_masks, gt_boxes, gt_ids = d["gt_masks"].cpu().numpy().astype(bool), d["gt_boxes"].cpu().numpy(), d["gt_ids"].cpu().numpy()
    o_bit = o_box = 0
    for g in range(n_img):
        n = int(counts[g])
        np.testing.assert_array_equal(gt_masks[g, :, :, :n], bits[o_bit:o_bit + 224 * 224 * n].reshape(224, 224, n).astype(bool))
        np.testing.assert_array_equal(gt_boxes[g, :n], fx["native_boxes"][o_box:o_box + n])
        np.testing.assert_array_equal(gt_ids[g, :n], fx["native_class_ids"][o_box:o_box + n])
        o_bit += 224 * 224 * n
        o_box += n
