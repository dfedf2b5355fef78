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
                 "ref_host_shapes.npz"):
        prov = str(_load(name)["provenance"])
        assert "/root/reference/myolo/myolo_utils.py" in prov and "numpy 1.26" in prov, prov


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
