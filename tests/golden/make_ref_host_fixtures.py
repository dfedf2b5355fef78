"""Generate tests/golden/ref_host_*.npz by EXECUTING THE REFERENCE'S OWN numpy host functions.

Run in the build container only (needs /root/reference; the GPU box gets the .npz, never this path):

    /opt/conda/bin/python3.9 tests/golden/make_ref_host_fixtures.py        # numpy 1.26.4: LEGACY scalar promotion, see below

`/root/reference/myolo/myolo_utils.py` cannot be imported as a module (its header imports mrcnn, tensorflow, keras, cv2 and
imgaug, none of which exist here), but the functions below need nothing except numpy.  Their definitions are picked out of
the parsed file (`ast`), compiled from the reference file where it lies -- nothing of the reference's text is written anywhere --
and executed with `np` = the real numpy:

    _sigmoid, _softmax, decode_one_yolo_output   myolo_utils.py:21-85
    NMB                                          myolo_utils.py:88-113
    BoundBox, bbox_iou, bbox_iou_2,
    _interval_overlap                            myolo_utils.py:161-244
    extract_bboxes                               myolo_utils.py:247-271

Left out on purpose: BatchGenerator.__getitem__ (:727-860) and unmold_mask (:883-912) use `np.bool`, which numpy >= 1.24 no
longer has (both interpreters of this image carry newer ones), and patching numpy would be a stand-in.  The anchor choice inside
__getitem__ (:795-809) is nevertheless made of pinned parts: the fixture stores bbox_iou(BoundBox(0,0,w,h), anchor) of the
reference for every (box, anchor) pair, and the strict `<` scan of :805 is "first maximum".

Which numpy: the reference dates from 2018-19 (TensorFlow 1.x, i.e. numpy <= 1.19) and mixes float32 network outputs with Python
scalars.  Until numpy 2.0 (NEP 50) `np.float32(a) * 224`, `1. + np.float32(a)` and `np.float32(a) > 0.3` are float64 operations;
from 2.0 on they are float32.  The committed fixtures are generated under the image's numpy 1.26.4 (/opt/conda, legacy promotion =
the rule of every numpy the reference could have run with); the product's host functions reproduce that rule with explicit
conversions, whatever numpy executes them.  (Run under numpy 2.2.6 the same reference code makes the same decisions on these
inputs but returns box coordinates that differ in the 7th digit -- float32 instead of float64 scalar arithmetic.)  float64 `exp`
differs by 1 ulp between the two numpy builds (SIMD kernels), float32 `exp` does not: tests compare decisions (which boxes, labels,
indices) exactly and float values to 1e-12 relative.

Inputs are seeded; outputs are data (arrays).  The .npz files are the fixtures; this script is their provenance.
"""
import ast
import io
import os
import sys

import numpy as np

REF = "/root/reference/myolo/myolo_utils.py"
HERE = os.environ.get("REF_FIXTURE_OUT") or os.path.dirname(os.path.abspath(__file__))     # override: cross-check under another numpy
WANT = ["_sigmoid", "_softmax", "decode_one_yolo_output", "NMB", "BoundBox", "bbox_iou", "bbox_iou_2",
        "_interval_overlap", "extract_bboxes"]


def load_reference_functions():
    with open(REF) as f:
        tree = ast.parse(f.read(), filename=REF)
    keep = [n for n in tree.body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in WANT]
    assert sorted(n.name for n in keep) == sorted(WANT), [n.name for n in keep]
    mod = ast.Module(body=keep, type_ignores=[])
    ns = {"np": np}
    exec(compile(mod, REF, "exec"), ns)
    lines = {n.name: (n.lineno, n.end_lineno) for n in keep}
    return ns, lines


# ----------------------------------------------------------------------------------------------------------------------
def masks_cases(rng):
    """>= 200 instance masks in stacks [H,W,n]: empty, full, single pixels (corners, centre), one row / column, rectangles,
    scattered blobs, masks touching each border."""
    stacks = []
    for (H, W) in [(224, 224), (128, 128), (56, 40), (7, 13), (1, 1), (416, 416)]:
        ms = []
        ms.append(np.zeros((H, W), bool))                               # empty
        ms.append(np.ones((H, W), bool))                                # full
        for (y, x) in [(0, 0), (0, W - 1), (H - 1, 0), (H - 1, W - 1), (H // 2, W // 2)]:
            m = np.zeros((H, W), bool); m[y, x] = True; ms.append(m)     # 1 pixel
        m = np.zeros((H, W), bool); m[H // 3, :] = True; ms.append(m)    # one row
        m = np.zeros((H, W), bool); m[:, W // 4] = True; ms.append(m)    # one column
        for _ in range(14 if H * W > 100 else 3):                       # rectangles
            y1, y2 = sorted(rng.integers(0, H + 1, 2)); x1, x2 = sorted(rng.integers(0, W + 1, 2))
            m = np.zeros((H, W), bool); m[y1:y2, x1:x2] = True; ms.append(m)
        for _ in range(10 if H * W > 100 else 2):                       # scattered pixels (non-convex support)
            m = rng.random((H, W)) < rng.choice([0.0005, 0.01, 0.3]); ms.append(m)
        for _ in range(6 if H * W > 100 else 1):                        # disc
            cy, cx, r = rng.integers(0, H), rng.integers(0, W), rng.integers(1, max(2, min(H, W) // 2))
            yy, xx = np.mgrid[:H, :W]
            ms.append((yy - cy) ** 2 + (xx - cx) ** 2 <= r * r)
        stacks.append(np.stack(ms, axis=-1))
    # dtype variants the callers really hand over: uint8 masks (dataset_shapes.py:109) and a 0-instance stack
    stacks.append(stacks[1].astype(np.uint8))
    stacks.append(np.zeros((32, 32, 0), bool))
    return stacks


def box_pairs(rng, n):
    """pairs of (xmin, ymin, xmax, ymax) with positive areas: random, identical, nested, touching edges, disjoint."""
    a = np.zeros((n, 4)); b = np.zeros((n, 4))
    for i in range(n):
        def rb():
            x1, y1 = rng.random(2) * 0.8
            return np.array([x1, y1, x1 + 0.02 + rng.random() * 0.6, y1 + 0.02 + rng.random() * 0.6])
        p = rb(); q = rb()
        k = i % 8
        if k == 0:
            q = p.copy()                                                 # identical
        elif k == 1:
            q = np.array([p[2], p[1], p[2] + 0.25, p[3]])                # touching on x: overlap 0 through the `min - x3` branch
        elif k == 2:
            q = np.array([p[0] + 0.25 * (p[2] - p[0]), p[1] + 0.25 * (p[3] - p[1]),
                          p[2] - 0.25 * (p[2] - p[0]), p[3] - 0.25 * (p[3] - p[1])])   # nested
        elif k == 3:
            q = p + np.array([2.0, 0, 2.0, 0])                           # disjoint on x
        elif k == 4:
            q = p + np.array([0, -3.0, 0, -3.0])                         # disjoint on y, other side
        a[i], b[i] = p, q
    return a, b


def main():
    ns, lines = load_reference_functions()
    R = ns
    rng = np.random.default_rng(20260928)
    prov = "executed from %s with numpy %s, python %s; lines %s" % (
        REF, np.__version__, sys.version.split()[0], sorted(lines.items()))

    # ---- extract_bboxes ------------------------------------------------------------------------------------------
    out = {}
    stacks = masks_cases(rng)
    nmask = 0
    for i, st in enumerate(stacks):
        bb = R["extract_bboxes"](st)
        assert bb.dtype == np.int32
        out["mask_%02d_shape" % i] = np.array(st.shape, np.int64)
        out["mask_%02d_dtype" % i] = np.array(str(st.dtype))
        out["mask_%02d_bits" % i] = np.packbits(st.astype(bool).reshape(-1))
        out["bbox_%02d" % i] = bb
        nmask += st.shape[-1]
    out["n_stacks"] = np.array(len(stacks))
    out["provenance"] = np.array(prov)
    np.savez_compressed(os.path.join(HERE, "ref_host_extract_bboxes.npz"), **out)
    print("extract_bboxes: %d masks in %d stacks" % (nmask, len(stacks)))

    # ---- _interval_overlap / bbox_iou / bbox_iou_2 / best-anchor IoUs ---------------------------------------------
    out = {"provenance": np.array(prov)}
    iv = rng.random((400, 4)) * 4 - 1
    iv[:, :2].sort(axis=1); iv[:, 2:].sort(axis=1)
    iv[::7, 2] = iv[::7, 1]                    # x3 == x2 (touching)
    iv[3::7, 3] = iv[3::7, 0]                  # x4 == x1
    iv[5::7, 2:] = iv[5::7, :2]                # identical
    out["interval_in"] = iv
    out["interval_out"] = np.array([R["_interval_overlap"]([r[0], r[1]], [r[2], r[3]]) for r in iv], np.float64)
    a, b = box_pairs(rng, 512)
    out["iou_a"], out["iou_b"] = a, b
    out["iou_out"] = np.array([R["bbox_iou"](R["BoundBox"](*p), R["BoundBox"](*q)) for p, q in zip(a, b)], np.float64)
    for tag, shape in (("224", (224, 224, 3)), ("416x320", (416, 320, 3))):
        out["iou2_out_" + tag] = np.array([R["bbox_iou_2"](p, q, shape) for p, q in zip(a, b)], np.float64)
        a32, b32 = a.astype(np.float32), b.astype(np.float32)       # detect() hands float32 rows (model.py:1304)
        v = [R["bbox_iou_2"](p, q, shape) for p, q in zip(a32, b32)]
        out["iou2_f32_out_" + tag] = np.array(v, np.float64)
    # anchor choice of BatchGenerator.__getitem__ :795-809 -- IoU of origin-anchored boxes, in grid units
    anchor_sets = {
        "shapes": [1.27273, 1.277385, 2.47446, 2.56253, 4.03843, 4.07434],                       # dataset_shapes.py:39
        "rice": [2.09, 2.48, 2.59, 3.01, 3.60, 3.64, 5.25, 4.56, 6.21, 6.25],                    # example/rice/anchors_5.txt
        "food": [1.27, 1.31, 1.95, 1.85, 2.40, 2.72, 3.20, 3.32, 5.06, 5.05],                    # example/food/anchors_5.txt
    }
    wh_px = np.concatenate([rng.integers(1, 225, (300, 2)), np.array([[224, 224], [1, 1], [32, 32], [224, 1], [1, 224]])])
    wh = wh_px / (224.0 / 7)                    # (xmax - xmin) / (IMAGE_SHAPE / GRID), :792-793
    out["anchor_wh"] = wh
    for name, anc in anchor_sets.items():
        anchors = [R["BoundBox"](0, 0, anc[2 * i], anc[2 * i + 1]) for i in range(len(anc) // 2)]      # :707-708
        ious = np.array([[R["bbox_iou"](R["BoundBox"](0, 0, w, h), an) for an in anchors] for (w, h) in wh], np.float64)
        out["anchors_" + name] = np.array(anc, np.float64)
        out["anchor_iou_" + name] = ious
    np.savez_compressed(os.path.join(HERE, "ref_host_boxes.npz"), **out)
    print("boxes: %d intervals, %d pairs, %d anchor boxes" % (len(iv), len(a), len(wh)))

    # ---- _sigmoid / _softmax / decode_one_yolo_output -------------------------------------------------------------
    out = {"provenance": np.array(prov)}
    xs = np.concatenate([rng.normal(0, 4, 200), [0.0, -50.0, 50.0, -745.0, 700.0]])
    out["sigmoid_in"] = xs
    out["sigmoid_out"] = R["_sigmoid"](xs)
    out["sigmoid_in_f32"] = xs.astype(np.float32)
    with np.errstate(over="ignore"):
        out["sigmoid_out_f32"] = R["_sigmoid"](xs.astype(np.float32))
    sm = rng.normal(0, 3, (6, 7, 7, 3, 4))
    sm[1] *= 40                                       # spread > 100: the rescale branch (`np.min(x) < t`)
    sm[2] = 0.0
    out["softmax_in"] = sm
    out["softmax_out"] = np.stack([R["_softmax"](s) for s in sm])
    out["softmax_out_f32"] = np.stack([R["_softmax"](s.astype(np.float32)) for s in sm])
    cases = []
    for k in range(64):
        G = 7 if k % 2 == 0 else 13
        nb, ncls, anc = [(3, 4, anchor_sets["shapes"]), (5, 2, anchor_sets["rice"]), (5, 2, anchor_sets["food"]),
                         (3, 3, anchor_sets["shapes"])][k % 4]
        dt = np.float32 if k % 3 else np.float64      # keras' predict() returns float32 (model.py:1224); float64 also pinned
        net = rng.normal(0, 1.0, (G, G, nb, 5 + ncls))
        net[..., 4] = rng.normal(-2.0 + 0.5 * (k % 5), 2.0, (G, G, nb))       # from few to many confident cells
        net[..., 5:] *= 1 + (k % 7)
        if k % 8 == 1:                                # near-duplicates in other cells / anchors
            net[1, 1] = net[0, 0]
            net[1, 1, :, 4] += 1e-3
            net[2, 2, 1] = net[2, 2, 0]
            net[2, 2, 1, 4] -= 1e-3
        if k % 8 == 2:                                # a cluster of near-identical confident boxes in one cell
            net[3, 3, :, :4] = 0.05 * rng.normal(0, 1, (nb, 4))
            net[3, 3, :, 4] = 6.0 + 0.1 * np.arange(nb)
            net[3, 3, :, 5:] = np.array([9.0] + [0.0] * (ncls - 1))
        if k % 8 == 3:
            net[..., 4] = -20.0                       # nothing survives
        if k % 8 == 4:
            net[..., 4] = rng.normal(3.0, 0.5, (G, G, nb))      # everything confident: dense NMS
        obj_t, nms_t = [(0.3, 0.3), (0.35, 0.3), (0.5, 0.45), (0.05, 0.7)][k % 4]       # defaults; infer_yolo's (model.py:1227-1231)
        net = net.astype(dt)
        # The reference orders boxes with np.argsort (:68), which is not a stable sort: among DIFFERENT boxes with exactly
        # equal non-zero class scores, who suppresses whom depends on the numpy build.  Such inputs have no defined answer,
        # so the random cases must not contain them (zeros tie harmlessly: :73-74 skips them).
        with np.errstate(over="ignore", invalid="ignore"):
            sc = R["_sigmoid"](net[..., 4])[..., None] * R["_softmax"](net[..., 5:])
        sc = sc * (sc > obj_t)
        for c in range(ncls):
            nz = sc[..., c][sc[..., c] != 0]
            assert len(np.unique(nz)) == len(nz), "case %d class %d: tied scores" % (k, c)
        with np.errstate(over="ignore", invalid="ignore"):
            boxes = R["decode_one_yolo_output"](net.copy(), anc, ncls, obj_threshold=obj_t, nms_threshold=nms_t)
        rows = np.array([[bx.xmin, bx.ymin, bx.xmax, bx.ymax, bx.c] + list(bx.classes) for bx in boxes], np.float64)
        rows = rows.reshape(len(boxes), 5 + ncls)
        labels = np.array([bx.get_label() for bx in boxes], np.int64)
        scores = np.array([bx.get_score() for bx in boxes], np.float64)
        out["dec_%02d_in" % k] = net
        out["dec_%02d_anchors" % k] = np.array(anc, np.float64)
        out["dec_%02d_par" % k] = np.array([ncls, obj_t, nms_t], np.float64)
        out["dec_%02d_rows" % k] = rows
        out["dec_%02d_label" % k] = labels
        out["dec_%02d_score" % k] = scores
        cases.append(len(boxes))
    # thresholds hit EXACTLY: class scores are conf * softmax; one class and conf logit +inf-like gives score 1.0 exactly and
    # `> obj_threshold` with obj_threshold = 1.0 must drop it; identical boxes give IoU == 1.0 == nms_threshold (>= keeps suppressing)
    net = np.full((7, 7, 3, 6), -30.0)
    net[..., :4] = 0.0
    net[2, 3, 0] = [0.2, -0.1, 0.3, 0.1, 40.0, 0.0]
    net[2, 3, 1] = [0.2, -0.1, 0.3, 0.1, 40.0, 0.0]
    net[2, 3, 1, 2:4] += np.log(anchor_sets["shapes"][0] / anchor_sets["shapes"][2]), np.log(anchor_sets["shapes"][1] / anchor_sets["shapes"][3])
    net[5, 5, 2] = [0.0, 0.0, 0.0, 0.0, 40.0, 0.0]
    for k, (obj_t, nms_t) in enumerate([(0.3, 1.0), (1.0, 0.3), (0.999, 0.3)], start=64):
        boxes = R["decode_one_yolo_output"](net.copy(), anchor_sets["shapes"], 1, obj_threshold=obj_t, nms_threshold=nms_t)
        rows = np.array([[bx.xmin, bx.ymin, bx.xmax, bx.ymax, bx.c] + list(bx.classes) for bx in boxes], np.float64).reshape(len(boxes), 6)
        out["dec_%02d_in" % k] = net
        out["dec_%02d_anchors" % k] = np.array(anchor_sets["shapes"], np.float64)
        out["dec_%02d_par" % k] = np.array([1, obj_t, nms_t], np.float64)
        out["dec_%02d_rows" % k] = rows
        out["dec_%02d_label" % k] = np.array([bx.get_label() for bx in boxes], np.int64)
        out["dec_%02d_score" % k] = np.array([bx.get_score() for bx in boxes], np.float64)
        cases.append(len(boxes))
    out["n_dec"] = np.array(len(cases))
    np.savez_compressed(os.path.join(HERE, "ref_host_decode.npz"), **out)
    print("decode_one_yolo_output: %d cases, boxes kept per case %s" % (len(cases), cases))

    # ---- NMB ------------------------------------------------------------------------------------------------------
    out = {"provenance": np.array(prov)}
    n_cases = 0
    for k in range(96):
        n = [0, 1, 2, 5, 10, 10][k % 6]                   # detect() passes at most 10 (model.py:1292-1304)
        dt = np.float32 if k % 2 == 0 else np.float64
        ctr = rng.random((n, 2)) * 0.6 + 0.2
        half = rng.random((n, 2)) * 0.25 + 0.02
        bx = np.concatenate([ctr - half, ctr + half], axis=1)
        if n >= 5 and k % 4 == 1:                         # chains: i suppresses j, j (already listed) "suppresses" l again
            bx[1] = bx[0] + 0.01
            bx[2] = bx[1] + 0.01
            bx[3] = bx[0]
        if n >= 2 and k % 4 == 2:
            bx[1] = bx[0]                                 # exact tie: IoU 1
        cls = rng.integers(1, 3 if k % 3 else 2, n)
        idx = rng.permutation(147)[:n].astype(np.int64)
        thr = [0.7, 0.3, 0.5, 1.0][k % 4]                 # 0.7 is detect()'s (model.py:1304)
        shape = [(224, 224, 3), (416, 416, 3), (128, 128, 3)][k % 3]
        res = R["NMB"](bx.astype(dt), cls, idx.copy(), shape, nms_threshold=thr)
        out["nmb_%02d_boxes" % k] = bx.astype(dt)
        out["nmb_%02d_cls" % k] = cls.astype(np.int64)
        out["nmb_%02d_idx" % k] = idx
        out["nmb_%02d_par" % k] = np.array([thr, shape[0], shape[1], shape[2]], np.float64)
        out["nmb_%02d_out" % k] = np.asarray(res, np.int64)
        n_cases += 1
    out["n_nmb"] = np.array(n_cases)
    np.savez_compressed(os.path.join(HERE, "ref_host_nmb.npz"), **out)
    print("NMB: %d cases" % n_cases)

    # ---- the Shapes stream: reference extract_bboxes / bbox_iou on the masks the training path really sees -----------
    # Inputs come from the product's host generator (myolo.shapes.ShapesDataset, numpy only; image g = f(seed, g)); the
    # GPU test then asks the DEVICE producer (myolo_shapes_batch) for the same images and compares with what the
    # reference's functions returned here.
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "mask-yolo_amd"))
    from myolo.shapes import ShapesDataset
    out = {"provenance": np.array(prov), "seed": np.array(1234), "hw": np.array([224, 224])}
    n_img = 64
    ds = ShapesDataset(1234)
    ds.load_shapes(n_img, 224, 224)
    ds.prepare()
    anc = anchor_sets["shapes"]
    anchors = [R["BoundBox"](0, 0, anc[2 * i], anc[2 * i + 1]) for i in range(3)]
    counts, bits, boxes_all, ids_all, iou_all = [], [], [], [], []
    for g in range(n_img):
        mask, class_ids = ds.load_mask(g)
        keep = np.sum(mask, axis=(0, 1)) > 0                     # load_image_gt's empty-instance filter (myolo_utils.py:346-349)
        mask, class_ids = mask[:, :, keep], class_ids[keep]
        bb = R["extract_bboxes"](mask)
        counts.append(mask.shape[-1])
        bits.append(np.packbits(mask.astype(bool).reshape(-1)))
        boxes_all.append(bb)
        ids_all.append(class_ids.astype(np.int32))
        for (x1, y1, x2, y2) in bb:                              # :792-809 with IMAGE_SHAPE 224, GRID 7
            w, h = (x2 - x1) / (float(224) / 7), (y2 - y1) / (float(224) / 7)
            iou_all.append([R["bbox_iou"](R["BoundBox"](0, 0, w, h), an) for an in anchors])
    out["counts"] = np.array(counts, np.int64)
    out["mask_bits"] = np.concatenate(bits)
    out["boxes"] = np.concatenate(boxes_all).astype(np.int32)
    out["class_ids"] = np.concatenate(ids_all)
    out["anchor_iou"] = np.array(iou_all, np.float64)
    np.savez_compressed(os.path.join(HERE, "ref_host_shapes.npz"), **out)
    print("shapes: %d images, %d instances" % (n_img, int(sum(counts))))


if __name__ == "__main__":
    main()
