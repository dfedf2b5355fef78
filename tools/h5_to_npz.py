#!/usr/bin/env python
"""Convert a reference Keras weight file (what model.py:1024-1027's ModelCheckpoint(save_weights_only=True) writes and
model.py:1157-1196 load_weights reads) into the .npz container of myolo.model.MaskYOLO.load_weights, and back.

  python tools/h5_to_npz.py mask_yolo_shapes_0005.h5 weights.npz          # needs h5py on the user's side
  python tools/h5_to_npz.py --to-h5 weights.npz keras_weights.h5

Keras weight-file schema (keras/engine/saving.py, Keras >= 2.0.8 as pinned by model.py:28):
  root.attrs['layer_names'] -> one group per layer; group.attrs['weight_names'] -> datasets named
  '<layer>/<weight>:0'.  A nested Model (the reference's 'yolo_model', model.py:851-852) is ONE layer group whose
  weight_names are the inner layers' ('conv_dw_7/depthwise_kernel:0', ...).
Array layouts are Keras' (HWIO conv kernels, [kh,kw,C,1] depthwise kernels, [kh,kw,Cout,Cin] Conv2DTranspose
kernels); the only reshape is the depthwise kernel's trailing multiplier axis.  The mapping itself
(`keras_weights_to_state` / `state_to_keras_weights`) has no h5py dependency and is unit-tested."""
import argparse
import sys

import numpy as np

NESTED_MODEL = "yolo_model"          # model.py:851-852
INNER_OF_NESTED = ("conv_dw_%d", "conv_dw_%d_bn", "conv_pw_%d", "conv_pw_%d_bn")


def keras_weights_to_state(named_arrays):
    """{'<layer>/<weight>:0': array} (flattened over all layer groups) -> {'<layer>/<weight>': array} in the
    myolo layout.  Unknown suffixes are kept; ':0' device suffixes are dropped."""
    sd = {}
    for name, arr in named_arrays.items():
        key = name.split(":")[0]
        parts = key.split("/")
        if len(parts) > 2:               # 'yolo_model/conv_dw_7/depthwise_kernel' style (tf.keras variants)
            key = "/".join(parts[-2:])
        a = np.asarray(arr, np.float32)
        if key.endswith("/depthwise_kernel") and a.ndim == 4:
            assert a.shape[3] == 1, "depth multiplier must be 1 (%s has shape %s)" % (name, a.shape)
            a = a[..., 0]
        sd[key] = a
    return sd


def nested_layers(n_backbone_blocks=6, n_yolo_blocks=8):
    names = []
    for b in range(n_backbone_blocks + 1, n_backbone_blocks + n_yolo_blocks + 1):
        names += [p % b for p in INNER_OF_NESTED]
    return names + ["conv_23"]


def state_to_keras_weights(sd):
    """inverse mapping: {layer group name: [(weight name, array), ...]} with the reference's nesting."""
    inner = set(nested_layers())
    groups = {}
    for key in sorted(sd):
        layer, w = key.split("/")
        a = np.asarray(sd[key], np.float32)
        if w == "depthwise_kernel":
            a = a[..., None]
        groups.setdefault(NESTED_MODEL if layer in inner else layer, []).append(("%s/%s:0" % (layer, w), a))
    return groups


def read_h5(path):
    import h5py                                              # noqa: F401  (user-side dependency)
    out = {}
    with h5py.File(path, "r") as f:
        root = f["model_weights"] if "model_weights" in f else f
        for lname in root.attrs["layer_names"]:
            g = root[lname.decode() if isinstance(lname, bytes) else lname]
            for wname in g.attrs["weight_names"]:
                wname = wname.decode() if isinstance(wname, bytes) else wname
                out[wname] = np.asarray(g[wname])
    return out


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
        sys.exit("h5py is required on the machine doing the conversion (it is not part of the MI355X image)")


if __name__ == "__main__":
    main()
