"""Generate tests/golden/ref_config.json, ref_load_image_gt.npz and ref_shapes_draws.npz by EXECUTING THE REFERENCE'S OWN host code
(second set, round 5; the first set is make_ref_host_fixtures.py -- same method, same interpreter):

    /opt/conda/bin/python3.9 tests/golden/make_ref_pipeline_fixtures.py      # numpy 1.26.4, scikit-image 0.18.3, scipy 1.7.1

Runs in the build container only (needs /root/reference; only the fixtures travel to the GPU box).  Nothing of the reference's text is
written anywhere: modules are imported / function definitions are compiled from the parsed files where they lie.

  ref_config.json         `myolo/config.py` imports only numpy (config.py:8) and is IMPORTED as it is: every public attribute of class
                          Config and of an instance.  `ShapesConfig` (example/shapes/dataset_shapes.py:14-50) sits in a file whose header
                          imports cv2 / mrcnn: its ClassDef is compiled from the parsed file with `Config` = the imported class.
  ref_load_image_gt.npz   load_image_gt (myolo_utils.py:274-366, augment=False, augmentation=None, use_mini_mask=False) with its helpers
                          resize_image :369-392, resize_mask :395-411, resize :433-455 (the scikit-image wrapper) and extract_bboxes
                          :247-271, executed with the real scikit-image / scipy.ndimage, on the product's ShapesDataset (numpy only;
                          image g = f(seed, g)) and the reference's ShapesConfig: 64 images at the native 224 x 224 (scale [1, 1]: no image
                          resize, scipy zoom by exactly 1) and 16 images generated at 160 x 128 and resized to 224 x 224 (scale [1.4, 1.75]:
                          skimage.transform.resize order 1 / mode constant / cval 0 / clip / preserve_range / no anti-aliasing on the
                          image, scipy.ndimage.zoom order 0 on the masks).
  ref_shapes_draws.npz    ShapesDataset.random_shape (dataset_shapes.py:137-156), a pure function of the `random` module's state, under
                          random.seed(k); and the draw order of random_image (:158-180) up to its call of the un-vendored
                          mrcnn.utils.non_max_suppression: the call's ARGUMENTS (boxes, scores = arange(N), threshold 0.3) are recorded
                          and every box is kept, so the returned list is all N shapes in draw order.  The suppression itself (mrcnn) is
                          NOT pinned by this file -- the product restates it (myolo/shapes.py::_nms) against its published algorithm.
"""
import ast
import importlib.util
import json
import logging
import os
import random
import sys
from distutils.version import LooseVersion

import numpy as np
import scipy
import scipy.ndimage
import skimage
import skimage.transform

REF_ROOT = "/root/reference"
REF_UTILS = REF_ROOT + "/myolo/myolo_utils.py"
REF_CONFIG = REF_ROOT + "/myolo/config.py"
REF_SHAPES = REF_ROOT + "/example/shapes/dataset_shapes.py"
HERE = os.environ.get("REF_FIXTURE_OUT") or os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "mask-yolo_amd"))


def jsonable(v):
    if isinstance(v, np.ndarray):
        return {"__ndarray__": v.tolist(), "dtype": str(v.dtype)}
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, tuple):
        return {"__tuple__": [jsonable(x) for x in v]}
    if isinstance(v, list):
        return [jsonable(x) for x in v]
    if isinstance(v, dict):
        return {str(k): jsonable(x) for k, x in v.items()}
    return v


def public_attrs(obj):
    return {a: jsonable(getattr(obj, a)) for a in sorted(dir(obj)) if not a.startswith("__") and not callable(getattr(obj, a))}


def import_reference_config():
    spec = importlib.util.spec_from_file_location("ref_myolo_config", REF_CONFIG)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pick(path, names, ns, inside_class=None):
    """compile the named top-level definitions (or methods of `inside_class`) of the parsed reference file into ns."""
    with open(path) as f:
        tree = ast.parse(f.read(), filename=path)
    body = tree.body
    if inside_class:
        body = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == inside_class][0].body
    keep = [n for n in body if isinstance(n, (ast.FunctionDef, ast.ClassDef)) and n.name in names]
    assert sorted(n.name for n in keep) == sorted(names), (sorted(n.name for n in keep), names)
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return {n.name: (n.lineno, n.end_lineno) for n in keep}


def main():
    versions = "python %s, numpy %s, scikit-image %s, scipy %s" % (sys.version.split()[0], np.__version__, skimage.__version__, scipy.__version__)

    # ---- config ----------------------------------------------------------------------------------------------------
    cfgmod = import_reference_config()
    ns = {"Config": cfgmod.Config, "np": np}
    lines = pick(REF_SHAPES, ["ShapesConfig"], ns)
    RefShapesConfig = ns["ShapesConfig"]
    out = {"provenance": "imported %s; ShapesConfig compiled from %s lines %s; %s" % (REF_CONFIG, REF_SHAPES, lines["ShapesConfig"], versions),
           "Config_class": public_attrs(cfgmod.Config), "Config_instance": public_attrs(cfgmod.Config()),
           "ShapesConfig_class": public_attrs(RefShapesConfig), "ShapesConfig_instance": public_attrs(RefShapesConfig())}
    with open(os.path.join(HERE, "ref_config.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
        f.write("\n")
    print("config: %d attributes of Config, %d of ShapesConfig" % (len(out["Config_class"]), len(out["ShapesConfig_class"])))

    # ---- load_image_gt -----------------------------------------------------------------------------------------------
    R = {"np": np, "scipy": scipy, "skimage": skimage, "LooseVersion": LooseVersion, "logging": logging, "random": random}
    lines = pick(REF_UTILS, ["load_image_gt", "resize_image", "resize_mask", "resize", "extract_bboxes", "minimize_mask"], R)
    from myolo.shapes import ShapesDataset
    rcfg = RefShapesConfig()
    assert list(rcfg.IMAGE_SHAPE) == [224, 224, 3]
    prov = "executed from %s lines %s with config = the reference's ShapesConfig(); %s" % (REF_UTILS, sorted(lines.items()), versions)
    out = {"provenance": np.array(prov), "seed": np.array(1234)}
    for tag, (n, h, w, start) in {"native": (64, 224, 224, 0), "resized": (16, 160, 128, 1000)}.items():
        ds = ShapesDataset(1234)
        ds.load_shapes(n, h, w, start_index=start)
        ds.prepare()
        imgs, cls, boxes, bits, counts = [], [], [], [], []
        for g in range(n):
            image, class_ids, bbox, mask = R["load_image_gt"](ds, rcfg, g, augment=False, augmentation=None, use_mini_mask=rcfg.USE_MINI_MASK)
            assert image.dtype == np.uint8 and image.shape == (224, 224, 3), (image.dtype, image.shape)
            assert mask.dtype == bool and mask.shape[:2] == (224, 224) and bbox.dtype == np.int32
            imgs.append(image)
            cls.append(class_ids.astype(np.int32))
            boxes.append(bbox)
            counts.append(mask.shape[-1])
            bits.append(np.packbits(mask.reshape(-1)))
        out[tag + "_hw_start"] = np.array([h, w, start], np.int64)
        out[tag + "_images"] = np.stack(imgs)
        out[tag + "_counts"] = np.array(counts, np.int64)
        out[tag + "_class_ids"] = np.concatenate(cls)
        out[tag + "_boxes"] = np.concatenate(boxes).astype(np.int32)
        out[tag + "_mask_bits"] = np.concatenate(bits)
        print("load_image_gt[%s]: %d images %dx%d -> 224x224, %d instances" % (tag, n, h, w, sum(counts)))
    # augment=True (the deprecated random horizontal flip, myolo_utils.py:306-311): one draw of the global `random` module per image, seeded here per image
    ds = ShapesDataset(1234)
    ds.load_shapes(24, 224, 224, start_index=2000)
    ds.prepare()
    logging.disable(logging.WARNING)                    # (the reference warns "'augment' is deprecated" on every call)
    imgs, cls, boxes, bits, counts, flips = [], [], [], [], [], []
    for g in range(24):
        random.seed(7000 + g)
        flips.append(random.randint(0, 1))
        random.seed(7000 + g)
        image, class_ids, bbox, mask = R["load_image_gt"](ds, rcfg, g, augment=True, augmentation=None, use_mini_mask=False)
        imgs.append(image); cls.append(class_ids.astype(np.int32)); boxes.append(bbox); counts.append(mask.shape[-1])
        bits.append(np.packbits(np.ascontiguousarray(mask).reshape(-1)))
    logging.disable(logging.NOTSET)
    out["flip_hw_start"] = np.array([224, 224, 2000], np.int64)
    out["flip_seed0"] = np.array(7000)
    out["flip_drawn"] = np.array(flips, np.int64)
    out["flip_images"] = np.stack(imgs)
    out["flip_counts"] = np.array(counts, np.int64)
    out["flip_class_ids"] = np.concatenate(cls)
    out["flip_boxes"] = np.concatenate(boxes).astype(np.int32)
    out["flip_mask_bits"] = np.concatenate(bits)
    print("load_image_gt[augment=True]: 24 images, %d flipped" % sum(flips))
    # the wrapper's own arguments, on something that is not piecewise constant: resize() of a random float image and of a ramp
    rng = np.random.default_rng(5)
    a = rng.random((37, 53, 3)) * 255.0
    out["wrap_in"] = a
    out["wrap_out_float"] = R["resize"](a, (64, 80), preserve_range=True)
    au8 = a.astype(np.uint8)
    img2, scale2 = R["resize_image"](au8, [96, 96, 3])
    out["wrap_u8_in"] = au8
    out["wrap_u8_out"] = img2
    out["wrap_u8_scale"] = np.array(scale2, np.float64)
    m = rng.random((37, 53, 4)) < 0.3
    out["zoom_in_bits"] = np.packbits(m.reshape(-1))
    out["zoom_out"] = R["resize_mask"](m, scale2)
    np.savez_compressed(os.path.join(HERE, "ref_load_image_gt.npz"), **out)

    # ---- random_shape / random_image draws -----------------------------------------------------------------------------
    calls = []

    class _Rec(object):
        @staticmethod
        def non_max_suppression(boxes, scores, threshold):       # records what the reference passes; keeps everything (see the module docstring)
            calls.append((np.array(boxes), np.array(scores), float(threshold)))
            return np.arange(len(boxes))
    S = {"np": np, "random": random, "utils": _Rec, "math": __import__("math")}
    lines = pick(REF_SHAPES, ["random_shape", "random_image"], S, inside_class="ShapesDataset")

    class _Self(object):
        def random_shape(self, height, width):
            return S["random_shape"](self, height, width)
    names = ["square", "circle", "triangle"]
    out = {"provenance": np.array("executed from %s lines %s; %s" % (REF_SHAPES, sorted(lines.items()), versions))}
    rows = []
    for (h, w) in [(224, 224), (128, 128), (416, 416), (160, 128)]:
        for k in range(128):
            random.seed(k)
            shape, color, (x, y, s) = S["random_shape"](None, h, w)
            rows.append([h, w, k, names.index(shape), color[0], color[1], color[2], x, y, s])
    out["random_shape"] = np.array(rows, np.int64)
    img_rows, img_meta = [], []
    for (h, w) in [(224, 224), (128, 128)]:
        for k in range(128):
            random.seed(1234 + k)
            del calls[:]
            bg, shapes = S["random_image"](_Self(), h, w)
            assert len(calls) == 1
            bx, sc, thr = calls[0]
            assert thr == 0.3 and np.array_equal(sc, np.arange(len(shapes)))
            img_meta.append([h, w, 1234 + k, len(shapes), int(bg[0]), int(bg[1]), int(bg[2])])
            for (shape, color, (x, y, s)), b in zip(shapes, bx):
                img_rows.append([names.index(shape), color[0], color[1], color[2], x, y, s] + [int(v) for v in b])
    out["random_image_meta"] = np.array(img_meta, np.int64)
    out["random_image_shapes"] = np.array(img_rows, np.int64)
    np.savez_compressed(os.path.join(HERE, "ref_shapes_draws.npz"), **out)
    print("draws: %d random_shape, %d random_image (%d shapes before suppression)" % (len(rows), len(img_meta), len(img_rows)))


if __name__ == "__main__":
    main()
