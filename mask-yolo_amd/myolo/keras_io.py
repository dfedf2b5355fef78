"""Keras weight-file (.h5) <-> myolo state dict.

What ``keras_model.save_weights`` / ``ModelCheckpoint(save_weights_only=True)`` write (reference model.py:1024-1027) and what
``MaskYOLO.load_weights(filepath, by_name, exclude)`` reads back through h5py (reference model.py:1157-1196):
  root.attrs['layer_names'] -> one group per layer; group.attrs['weight_names'] -> datasets named '<layer>/<weight>:0'
  (keras/engine/saving.py, Keras >= 2.0.8 as pinned by model.py:28).  A nested Model -- the reference's 'yolo_model',
  model.py:851-852 -- is ONE layer group whose weight_names are the inner layers' ('conv_dw_7/depthwise_kernel:0', ...).
  A full-model file (model.save) keeps the same tree under 'model_weights'.
Array layouts are Keras' (HWIO conv kernels, [kh,kw,C,1] depthwise kernels, [kh,kw,Cout,Cin] Conv2DTranspose kernels): the engine
uses the same layouts, the only reshape is the depthwise kernel's trailing multiplier axis.

Reading needs no h5py: myolo/h5lite.py parses the HDF5 subset Keras files use.  Writing an .h5 (tools/h5_to_npz.py --to-h5) is
a user-side convenience and does need h5py."""
import numpy as np

from . import h5lite

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


def _names(v):
    if v is None:
        return []
    return [n.decode("utf8") if isinstance(n, bytes) else str(n) for n in np.atleast_1d(v).tolist()]


def read_keras_h5(path, opener=None):
    """-> ({'<weight name>': array}, [top-level layer names in file order], {weight name: top-level layer}).
    `opener` defaults to the built-in pure-Python reader; h5py.File works too (same interface)."""
    opener = opener or h5lite.File
    out, owner = {}, {}
    with opener(path) as f:
        root = f["model_weights"] if "model_weights" in f else f
        layers = _names(root.attrs.get("layer_names"))
        # large name lists are split into layer_names0, layer_names1, ... (saving.py save_attributes_to_hdf5_group)
        if not layers:
            k = 0
            while ("layer_names%d" % k) in root.attrs:
                layers += _names(root.attrs["layer_names%d" % k])
                k += 1
        for lname in layers:
            g = root[lname]
            wnames = _names(g.attrs.get("weight_names"))
            k = 0
            while ("weight_names%d" % k) in g.attrs:
                wnames += _names(g.attrs["weight_names%d" % k])
                k += 1
            for wname in wnames:
                out[wname] = np.asarray(g[wname])
                owner[wname] = lname
    return out, layers, owner


def load_h5_state(path, exclude=None):
    """state dict {'<layer>/<weight>': float32 array} of a Keras weight file; `exclude` drops whole TOP-LEVEL layers by name
    (reference model.py:1181-1183 filters `keras_model.layers`, where the nested 'yolo_model' counts as one layer)."""
    named, _, owner = read_keras_h5(path)
    if exclude:
        named = {k: v for k, v in named.items() if owner[k] not in exclude and k.split("/")[0] not in exclude}
    return keras_weights_to_state(named)
