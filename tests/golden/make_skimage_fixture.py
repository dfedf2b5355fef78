#!/usr/bin/env python
"""Generates tests/golden/skimage_resize_fixture.npz with the REAL scikit-image (the reference's dependency for unmold_mask,
myolo_utils.py:433-447, 903).  scikit-image is not installed for the project interpreter; the image ships an Anaconda Python
that has it:
    /opt/conda/bin/python3.9 tests/golden/make_skimage_fixture.py        (scikit-image 0.18.3, numpy 1.26)
The call is exactly the reference wrapper's for skimage >= 0.14:
    skimage.transform.resize(image, output_shape, order=1, mode='constant', cval=0, clip=True, preserve_range=False,
                             anti_aliasing=False, anti_aliasing_sigma=None)
Inputs: seeded random 28x28 masks in [0,1] plus two structured ones; output shapes cover up-scaling (box larger than the mask),
down-scaling, identity, ragged and 1-pixel boxes."""
import os

import numpy as np
import skimage
import skimage.transform

rng = np.random.default_rng(20260928)
shapes = [(100, 57), (28, 28), (10, 13), (224, 224), (5, 3), (1, 1), (40, 28), (29, 56), (1, 30), (63, 2), (90, 120)]
out = {"skimage_version": np.array(skimage.__version__), "shapes": np.array(shapes)}
for k, (oh, ow) in enumerate(shapes):
    if k == 1:
        m = np.zeros((28, 28), np.float32)
        m[4:20, 6:25] = 1.0                                  # a hard-edged blob
    elif k == 3:
        yy, xx = np.mgrid[0:28, 0:28]
        m = (1.0 / (1.0 + np.exp(-(9.0 - np.hypot(yy - 13.5, xx - 13.5))))).astype(np.float32)   # a soft disc touching nothing
        m[:, 0] = 0.9                                        # ... and a bright border column (the border semantics matter)
    elif k == 10:
        m = (0.55 + 0.4 * rng.random((28, 28))).astype(np.float32)   # everywhere >= 0.5: clip=True keeps the rim above the threshold
    else:
        m = rng.random((28, 28)).astype(np.float32)
    r = skimage.transform.resize(m, (oh, ow), order=1, mode='constant', cval=0, clip=True, preserve_range=False,
                                 anti_aliasing=False, anti_aliasing_sigma=None)
    out["mask_%d" % k] = m
    out["resized_%d" % k] = np.asarray(r, np.float32)
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "skimage_resize_fixture.npz")
np.savez_compressed(path, **out)
print(path, os.path.getsize(path), "bytes; scikit-image", skimage.__version__)
