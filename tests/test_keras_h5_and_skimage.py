"""Fixtures produced by the REAL third-party libraries the reference depends on, generated in this image's Anaconda Python
(/opt/conda/bin/python3.9: h5py 3.3.0 / libhdf5 1.10.6, scikit-image 0.18.3 -- neither is importable from the project
interpreter) by tests/golden/make_h5_fixture.py and tests/golden/make_skimage_fixture.py:

  * keras_weights_small.h5 -- a weight file with exactly the tree keras.engine.saving.save_weights_to_hdf5_group writes
    (reference model.py:1024-1027 / :1157-1196), read back by the product's pure-Python HDF5 reader (myolo/h5lite.py) and
    loaded through MaskYOLO.load_weights(by_name=True);
  * skimage_resize_fixture.npz -- skimage.transform.resize outputs for the exact call of the reference's wrapper
    (myolo_utils.py:433-447, used by unmold_mask :903): pins the order-1 resize of the oracle (oracle/np_post.py), of the host
    code and of the HIP kernel myolo_unmold_masks, including the zero border of mode='constant'."""
import os

import numpy as np
import pytest

from oracle import np_post as Q
from myolo import h5lite, keras_io, myolo_utils as mutils

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H5 = os.path.join(GOLD, "keras_weights_small.h5")


def _expected():
    e = np.load(os.path.join(GOLD, "keras_weights_small_expected.npz"))
    return {k: e[k] for k in e.files}


def test_h5lite_reads_a_real_h5py_file_exactly():
    exp = _expected()
    with h5lite.File(H5) as f:
        assert f.attrs["backend"] == b"tensorflow" and f.attrs["keras_version"] == b"2.2.4"
        layers = [n.decode() for n in f.attrs["layer_names"]]            # fixed-length string array (h5py 2.x / Keras 2.2 style)
        assert layers[:3] == ["conv1", "conv1_bn", "conv_dw_1"] and "yolo_model" in layers and len(layers) == 10
        assert sorted(f.keys()) == sorted(layers)
        n = 0
        for ln in layers:
            g = f[ln]
            wn = [w.decode() for w in np.atleast_1d(g.attrs["weight_names"])] if g.attrs["weight_names"] is not None else []
            for w in wn:                                                  # variable-length strings (h5py 3 style), global heap
                a = np.asarray(g[w])
                assert a.dtype == np.float32 and a.shape == exp[ln + "|" + w].shape and np.array_equal(a, exp[ln + "|" + w]), (ln, w)
                n += 1
        assert n == len(exp) == 29
        assert f["yolo_model"]["conv_pw_7"]["kernel:0"].shape == (1, 1, 128, 128)      # the chunked dataset, nested groups
        assert "model_weights" not in f and "conv1/kernel:0" in f["conv1"]
        with pytest.raises(KeyError):
            f["conv1"]["nope"]
    with pytest.raises(h5lite.H5LiteError):
        h5lite.File(os.path.join(GOLD, "keras_weights_small_expected.npz"))


def test_keras_h5_to_state_dict_mapping_on_the_real_file():
    exp = _expected()
    sd = keras_io.load_h5_state(H5)
    assert sd["conv_dw_1/depthwise_kernel"].shape == (3, 3, 8)            # multiplier axis dropped
    assert np.array_equal(sd["conv_dw_1/depthwise_kernel"], exp["conv_dw_1|conv_dw_1/depthwise_kernel:0"][..., 0])
    assert np.array_equal(sd["conv_23/bias"], exp["yolo_model|conv_23/bias:0"])      # nested model flattened to layer names
    assert np.array_equal(sd["conv1_bn/moving_variance"], exp["conv1_bn|conv1_bn/moving_variance:0"])
    assert len(sd) == 29
    ex = keras_io.load_h5_state(H5, exclude=["yolo_model", "myolo_mask"])  # exclude acts on top-level layers (model.py:1181-1183)
    assert not any(k.startswith(("conv_23", "conv_dw_7", "conv_pw_7", "myolo_mask/")) for k in ex) and "myolo_mask_bn1/gamma" in ex


@pytest.mark.gpu
def test_load_weights_reads_keras_h5_by_name():
    from myolo.config import make_config, ShapesConfig
    from myolo.model import MaskYOLO
    cfg = make_config(ShapesConfig, ALPHA=0.25, IMAGE_SHAPE=[128, 128, 3], BATCH_SIZE=2)
    m = MaskYOLO(mode="inference", config=cfg, seed=3)
    before = m.state_dict()
    m.load_weights(H5, by_name=True)
    after = m.state_dict()
    want = keras_io.load_h5_state(H5)
    for k in after:
        if k in want:
            assert np.array_equal(after[k], want[k]), k
        else:
            assert np.array_equal(after[k], before[k]), k
    with pytest.raises(KeyError):
        MaskYOLO(mode="inference", config=cfg).load_weights(H5)           # by_name=False: every tensor must be in the file
    m2 = MaskYOLO(mode="inference", config=cfg, seed=3)
    m2.load_weights(H5, exclude=["yolo_model"])
    assert np.array_equal(m2.state_dict()["conv_23/kernel"], before["conv_23/kernel"])
    assert np.array_equal(m2.state_dict()["conv1/kernel"], want["conv1/kernel"])


# ------------------------------------------------------------------ skimage.transform.resize
def _fixture():
    z = np.load(os.path.join(GOLD, "skimage_resize_fixture.npz"))
    return [(z["mask_%d" % k], tuple(int(v) for v in z["shapes"][k]), z["resized_%d" % k]) for k in range(len(z["shapes"]))], str(z["skimage_version"])


def test_order1_resize_matches_scikit_image():
    cases, ver = _fixture()
    assert ver.startswith("0.1")
    for m, (oh, ow), ref in cases:
        for fn in (Q.resize_bilinear_f32, mutils._resize_bilinear):
            got = fn(m, oh, ow)
            assert got.shape == ref.shape == (oh, ow)
            assert np.abs(got - ref).max() < 5e-6, (oh, ow, float(np.abs(got - ref).max()))
            clear = np.abs(ref - 0.5) > 1e-5                                # thresholding agrees wherever 0.5 is not within rounding
            assert np.array_equal((got >= 0.5)[clear], (ref >= 0.5)[clear])
        assert np.array_equal(Q.resize_bilinear_f32(m, oh, ow), mutils._resize_bilinear(m, oh, ow))      # oracle == host code, bitwise
    # the zero border is what distinguishes mode='constant' from an edge clamp: an up-scaled all-ones mask fades at its rim
    one = np.ones((28, 28), np.float32)
    one[14, 14] = 0.0                                                       # (a minimum of 0 keeps clip=True out of the way)
    r = Q.resize_bilinear_f32(one, 112, 112)
    assert r[0, 0] < 0.5 and r[1, 1] >= 0.5 and r[20, 20] == 1.0 and abs(r[20, 0] - 0.625) < 1e-6
    # clip=True: the output is clipped to the input's range, so a mask that is >= 0.5 everywhere keeps its rim
    assert Q.resize_bilinear_f32(np.full((28, 28), 0.8, np.float32), 112, 112).min() == np.float32(0.8)


@pytest.mark.gpu
def test_unmold_kernel_matches_scikit_image():
    import torch
    from myolo import _ext as X
    cases, _ = _fixture()
    H = W = 224
    C = 3
    for m, (oh, ow), ref in cases:
        masks = np.zeros((1, 28, 28, C), np.float32)
        masks[0, :, :, 1] = m
        det = np.array([[0.0, 0.0, (ow + 0.5) / W, (oh + 0.5) / H, 0.9, 1.0]], np.float32)     # window [0, ow) x [0, oh)
        full = torch.zeros(H, W, 1, dtype=torch.uint8, device="cuda")
        wsb = torch.empty(4, dtype=torch.int32, device="cuda")
        mt, dtt = torch.as_tensor(masks).cuda(), torch.as_tensor(det).cuda()      # named: a temporary would be freed before the launch
        X.call("myolo_unmold_masks", X.ptr(mt), X.ptr(dtt), X.ptr(full), 1, 28, 28, C, H, W, wsb.data_ptr(), 16, X.stream())
        got = full.cpu().numpy()[:, :, 0].astype(bool)
        assert not got[oh:, :].any() and not got[:, ow:].any()
        clear = np.abs(ref - 0.5) > 1e-5
        assert np.array_equal(got[:oh, :ow][clear], (ref >= 0.5)[clear]), (oh, ow)
        assert np.array_equal(got, Q.unmold_mask(m, det[0, :4], (H, W, 3)))                     # and the oracle bit for bit
