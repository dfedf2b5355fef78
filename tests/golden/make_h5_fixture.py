#!/usr/bin/env python
"""Generates tests/golden/keras_weights_small.h5 (+ keras_weights_small_expected.npz) with the REAL h5py / libhdf5:
    /opt/conda/bin/python3.9 tests/golden/make_h5_fixture.py          (h5py 3.3.0, HDF5 1.10.6; the project interpreter has no h5py)
The file has exactly the layout keras.engine.saving.save_weights_to_hdf5_group (Keras 2.2.4) produces -- the format the
reference's checkpoints have (model.py:1024-1027 ModelCheckpoint(save_weights_only=True); read back by model.py:1157-1196):
    f.attrs['layer_names']   = [layer.name.encode('utf8') ...]          f.attrs['backend'], f.attrs['keras_version'] (bytes)
    g = f.create_group(layer.name);  g.attrs['weight_names'] = [w.name.encode('utf8') ...]   (e.g. b'conv1/kernel:0')
    g.create_dataset(name, val.shape, dtype=val.dtype)[:] = val        (-> nested group 'conv1' inside group 'conv1')
A nested Model ('yolo_model', model.py:851-852) is ONE layer group holding its inner layers' weights.
Only a subset of the network's layers is written (MobileNet alpha 0.25 shapes; load_weights(by_name=True) loads what is there):
a conv + its BatchNorm, one depthwise-separable block, the nested yolo_model with one block and conv_23, and two mask-head layers.
One dataset is written chunked (no filter) and the attributes of one group exceed nothing special: plain libver='earliest'."""
import os

import h5py
import numpy as np

rng = np.random.default_rng(20260928)
a = 0.25
c0, c1 = int(32 * a), int(64 * a)
c6, c7 = int(512 * a), int(512 * a)
C, NBOX = 4, 3


def bn(c):
    return [("gamma:0", (1 + 0.1 * rng.standard_normal(c))), ("beta:0", 0.1 * rng.standard_normal(c)),
            ("moving_mean:0", 0.1 * rng.standard_normal(c)), ("moving_variance:0", 1 + 0.1 * rng.random(c))]


layers = [
    ("conv1", [("conv1/kernel:0", rng.standard_normal((3, 3, 3, c0)))]),
    ("conv1_bn", [("conv1_bn/" + n, v) for n, v in bn(c0)]),
    ("conv_dw_1", [("conv_dw_1/depthwise_kernel:0", rng.standard_normal((3, 3, c0, 1)))]),
    ("conv_dw_1_bn", [("conv_dw_1_bn/" + n, v) for n, v in bn(c0)]),
    ("conv_pw_1", [("conv_pw_1/kernel:0", rng.standard_normal((1, 1, c0, c1)))]),
    ("conv_pw_1_bn", [("conv_pw_1_bn/" + n, v) for n, v in bn(c1)]),
    ("yolo_model", [("conv_dw_7/depthwise_kernel:0", rng.standard_normal((3, 3, c6, 1)))] +
                   [("conv_dw_7_bn/" + n, v) for n, v in bn(c6)] +
                   [("conv_pw_7/kernel:0", rng.standard_normal((1, 1, c6, c7)))] +
                   [("conv_23/kernel:0", rng.standard_normal((1, 1, int(1024 * a), NBOX * (5 + C)))),
                    ("conv_23/bias:0", rng.standard_normal(NBOX * (5 + C)))]),
    ("myolo_mask_bn1", [("myolo_mask_bn1/" + n, v) for n, v in bn(256)]),
    ("myolo_mask", [("myolo_mask/kernel:0", rng.standard_normal((1, 1, 256, C))), ("myolo_mask/bias:0", rng.standard_normal(C))]),
    ("input_image", []),                    # Keras writes a group for weight-less layers too (empty weight_names)
]
here = os.path.dirname(os.path.abspath(__file__))
path = os.path.join(here, "keras_weights_small.h5")
expected = {}
with h5py.File(path, "w") as f:             # libver default: 'earliest'
    # h5py 2.x (the Keras 2.2 era) stored a list of bytes as a fixed-length string array; h5py >= 3 stores it as
    # variable-length strings.  Both occur here: layer_names fixed (explicit 'S' dtype), weight_names / backend variable.
    f.attrs["layer_names"] = np.array([n.encode("utf8") for n, _ in layers], dtype="S")
    f.attrs["backend"] = "tensorflow".encode("utf8")
    f.attrs["keras_version"] = "2.2.4".encode("utf8")
    for lname, items in layers:
        g = f.create_group(lname)
        g.attrs["weight_names"] = [n.encode("utf8") for n, _ in items]
        for n, v in items:
            v = np.asarray(v, np.float32)
            if n == "conv_pw_7/kernel:0":
                d = g.create_dataset(n, v.shape, dtype=v.dtype, chunks=(1, 1, 32, 64))     # a chunked (unfiltered) dataset
            else:
                d = g.create_dataset(n, v.shape, dtype=v.dtype)
            d[:] = v
            expected[lname + "|" + n] = v
np.savez_compressed(os.path.join(here, "keras_weights_small_expected.npz"), **expected)
print(path, os.path.getsize(path), "bytes; h5py", h5py.__version__, "hdf5", h5py.version.hdf5_version)
