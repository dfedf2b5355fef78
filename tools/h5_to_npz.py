#!/usr/bin/env python
"""Convert a reference Keras weight file (what model.py:1024-1027's ModelCheckpoint(save_weights_only=True) writes and
model.py:1157-1196 load_weights reads) into the .npz container of myolo.model.MaskYOLO.load_weights, and back.

  python tools/h5_to_npz.py mask_yolo_shapes_0005.h5 weights.npz          # no h5py needed (myolo/h5lite.py)
  python tools/h5_to_npz.py --to-h5 weights.npz keras_weights.h5          # writing needs h5py on the user's side
(MaskYOLO.load_weights also takes the .h5 directly.)

Keras weight-file schema (keras/engine/saving.py, Keras >= 2.0.8 as pinned by model.py:28):
  root.attrs['layer_names'] -> one group per layer; group.attrs['weight_names'] -> datasets named
  '<layer>/<weight>:0'.  A nested Model (the reference's 'yolo_model', model.py:851-852) is ONE layer group whose
  weight_names are the inner layers' ('conv_dw_7/depthwise_kernel:0', ...).
Array layouts are Keras' (HWIO conv kernels, [kh,kw,C,1] depthwise kernels, [kh,kw,Cout,Cin] Conv2DTranspose
kernels); the only reshape is the depthwise kernel's trailing multiplier axis.  The mapping itself
(`keras_weights_to_state` / `state_to_keras_weights`) lives in myolo/keras_io.py, has no h5py dependency and is unit-tested."""
import argparse
import sys

import numpy as np

import os

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mask-yolo_amd"))
from myolo.keras_io import (NESTED_MODEL, INNER_OF_NESTED, keras_weights_to_state, nested_layers,       # noqa: E402,F401
                            state_to_keras_weights, read_keras_h5)


def read_h5(path):
    """every '<layer>/<weight>:0' array of a Keras weight file (pure-Python HDF5 reader, no h5py needed)"""
    return read_keras_h5(path)[0]


def write_h5(path, groups):
    import h5py
    with h5py.File(path, "w") as f:
        f.attrs["layer_names"] = [n.encode() for n in groups]
        f.attrs["backend"] = b"tensorflow"
        for lname, items in groups.items():
            g = f.create_group(lname)
            g.attrs["weight_names"] = [n.encode() for n, _ in items]
            for n, a in items:
                g.create_dataset(n, data=a)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("dst")
    ap.add_argument("--to-h5", action="store_true")
    a = ap.parse_args()
    try:
        if a.to_h5:
            data = np.load(a.src)
            write_h5(a.dst, state_to_keras_weights({k: data[k] for k in data.files}))
        else:
            np.savez(a.dst, **keras_weights_to_state(read_h5(a.src)))
    except ImportError:
        sys.exit("writing .h5 needs h5py on the machine doing the conversion (it is not part of the MI355X image)")


if __name__ == "__main__":
    main()
