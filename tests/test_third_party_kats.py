"""Known-answer tests quoted from the test suites / sources of the THIRD-PARTY libraries that carry the reference's
arithmetic (TensorFlow 1.x, Keras 2.2.x, keras_applications -- un-vendored, un-pinned, not
installable here).  Each vector is one their own projects publish; it is run against the CPU oracle (always) and against the
HIP kernels through the C-ABI (-m gpu).  They pin the *semantics restated from documentation* that the oracle could
otherwise only agree with itself on:

  1. tf.image.crop_and_resize (model.py:385-387, 581-583)
     -- tensorflow/core/kernels/crop_and_resize_op_test.cc (r1.12): TestCropAndResize2x2To1x1, ...2x2To1x1Flipped,
        ...2x2To3x3, ...2x2To3x3Flipped, ...3x3To2x2, ...3x3To2x2Flipped, ...2x2To3x3Extrapolated.
  2. Keras Adam (model.py:1071-1075) -- tensorflow/python/training/adam_test.py (r1.12) `adam_update_numpy` + testBasic's
     variables / gradients (var0=[1,2], grads0=[0.1,0.1], var1=[3,4], grads1=[0.01,0.01], three steps); Keras'
     optimizers.Adam.get_updates is the same recurrence with epsilon outside the square root.
  3. BatchNormalization moving-variance update (behind model.py:51, 690) -- keras/layers/normalization.py (2.2.4)
     `variance *= sample_size / (sample_size - (1.0 + self.epsilon))` on the value returned by
     tensorflow_backend.normalize_batch_in_training, which for 4-D NHWC / axis=-1 is tf.nn.fused_batch_norm's batch_variance:
     tensorflow/core/kernels/fused_batch_norm_op.cc (r1.12) `batch_var = variance * rest_size / (rest_size - 1)`;
     tensorflow/python/ops/nn_fused_batchnorm_test.py `_training_ref` states the same (`var * factor`, factor = n/(n-1)).
  4. keras_applications/mobilenet.py (1.0.6) `_depthwise_conv_block`: strides != (1,1) -> ZeroPadding2D(((0, 1), (0, 1)))
     then DepthwiseConv2D(..., padding='valid'); `_conv_block`: ZeroPadding2D(((0, 1), (0, 1))) in 1.0.6, but the reference
     writes its own conv_block with ZeroPadding2D(padding=(1, 1)) (model.py:45), which is what is pinned here.
  5. tf.round (model.py:591) -- "Rounds half to even. Also known as bankers rounding." (tf.round docstring, r1.12), with the
     docstring's example x = [0.9, 2.5, 2.3, 1.5, -4.5] -> [1.0, 2.0, 2.0, 2.0, -4.0].
"""
import numpy as np
import pytest

from oracle import np_ops as O

F32 = np.float32

# ------------------------------------------------------------------ 1. crop_and_resize_op_test.cc
CROP_CASES = [
    # name, image (H,W), boxes, crop, expected
    ("2x2To1x1", (2, 2), [[0, 0, 1, 1]], (1, 1), [2.5]),
    ("2x2To1x1Flipped", (2, 2), [[1, 1, 0, 0]], (1, 1), [2.5]),
    ("2x2To3x3", (2, 2), [[0, 0, 1, 1]], (3, 3), [1, 1.5, 2, 2, 2.5, 3, 3, 3.5, 4]),
    ("2x2To3x3Flipped", (2, 2), [[1, 1, 0, 0]], (3, 3), [4, 3.5, 3, 3, 2.5, 2, 2, 1.5, 1]),
    ("3x3To2x2", (3, 3), [[0, 0, 1, 1], [0, 0, .5, .5]], (2, 2), [1, 3, 7, 9, 1, 2, 4, 5]),
    ("3x3To2x2Flipped", (3, 3), [[1, 1, 0, 0], [.5, .5, 0, 0]], (2, 2), [9, 7, 3, 1, 5, 4, 2, 1]),
    # the C++ test sets extrapolation_value = -1; Mask-YOLO calls crop_and_resize with the default 0 (model.py:385-387),
    # the only value the kernels implement: the out-of-image samples read 0 instead of -1, the in-image ones are the test's
    ("2x2To3x3Extrapolated", (2, 2), [[-1, -1, 1, 1]], (3, 3), [0, 0, 0, 0, 1, 2, 0, 3, 4]),
]


def _crop_image(hw):
    h, w = hw
    return np.arange(1, h * w + 1, dtype=F32).reshape(1, h, w, 1)          # test::FillValues<float>(&image, {1, 2, 3, 4, ...})


@pytest.mark.parametrize("name,hw,boxes,crop,expect", CROP_CASES)
def test_tf_crop_and_resize_vectors_oracle(name, hw, boxes, crop, expect):
    boxes = np.asarray(boxes, F32)
    out = O.crop_and_resize(_crop_image(hw), boxes, np.zeros(len(boxes), np.int32), crop)
    np.testing.assert_allclose(out.reshape(-1), np.asarray(expect, F32), rtol=0, atol=1e-6, err_msg=name)


@pytest.mark.gpu
@pytest.mark.parametrize("name,hw,boxes,crop,expect", CROP_CASES)
def test_tf_crop_and_resize_vectors_hip(name, hw, boxes, crop, expect):
    import torch
    from myolo import _ext as X
    C = 4                                   # the kernel moves 4 channels per lane: replicate the test image over 4 channels
    img = torch.as_tensor(np.repeat(_crop_image(hw), C, axis=3)).cuda().contiguous()
    bx = torch.as_tensor(np.asarray(boxes, F32)).cuda().contiguous()
    bi = torch.zeros(len(boxes), dtype=torch.int32, device="cuda")
    out = torch.full((len(boxes), crop[0], crop[1], C), float("nan"), device="cuda")
    X.call("myolo_crop_and_resize_fwd", X.ptr(img), X.ptr(bx), X.ptr(bi), X.ptr(out), 1, hw[0], hw[1], C, len(boxes), crop[0], crop[1], X.stream())
    got = out.cpu().numpy()
    for c in range(C):
        np.testing.assert_allclose(got[..., c].reshape(-1), np.asarray(expect, F32), rtol=0, atol=1e-6, err_msg=name)


# ------------------------------------------------------------------ 2. adam_test.py
def adam_update_numpy(param, g_t, t, m, v, alpha=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8):
    """verbatim recurrence of tensorflow/python/training/adam_test.py::adam_update_numpy (float64)."""
    alpha_t = alpha * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m_t = beta1 * m + (1 - beta1) * g_t
    v_t = beta2 * v + (1 - beta2) * g_t * g_t
    param_t = param - alpha_t * m_t / (np.sqrt(v_t) + epsilon)
    return param_t, m_t, v_t


ADAM_VARS = (np.array([1.0, 2.0]), np.array([0.1, 0.1]), np.array([3.0, 4.0]), np.array([0.01, 0.01]))


def _adam_expected():
    var0, g0, var1, g1 = ADAM_VARS
    m0 = v0 = m1 = v1 = 0.0
    for t in range(1, 4):
        var0, m0, v0 = adam_update_numpy(var0, g0, t, m0, v0)
        var1, m1, v1 = adam_update_numpy(var1, g1, t, m1, v1)
    return np.concatenate([var0, var1])


def test_tf_adam_testbasic_vectors_oracle():
    p = np.concatenate([ADAM_VARS[0], ADAM_VARS[2]]).astype(F32)
    g = np.concatenate([ADAM_VARS[1], ADAM_VARS[3]]).astype(F32)
    m, v = np.zeros_like(p), np.zeros_like(p)
    for t in range(1, 4):
        p, m, v = O.adam_step(p, g, m, v, t, lr=0.001)
    np.testing.assert_allclose(p, _adam_expected(), rtol=1e-6)
    # first step in closed form: m = 0.1 g, v = 0.001 g^2, alpha_1 = lr sqrt(0.001)/0.1 -> every parameter moves by lr (eps aside)
    p1, _, _ = O.adam_step(np.array([1.0], F32), np.array([0.5], F32), np.zeros(1, F32), np.zeros(1, F32), 1, lr=0.001)
    assert abs(float(p1[0]) - 0.999) < 1e-6


@pytest.mark.gpu
def test_tf_adam_testbasic_vectors_hip():
    import torch
    from myolo import _ext as X
    p = torch.tensor(np.concatenate([ADAM_VARS[0], ADAM_VARS[2]]), dtype=torch.float32, device="cuda")
    g = torch.tensor(np.concatenate([ADAM_VARS[1], ADAM_VARS[3]]), dtype=torch.float32, device="cuda")
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t in range(1, 4):
        lr_t = float(0.001 * np.sqrt(1.0 - 0.999 ** t) / (1.0 - 0.9 ** t))
        X.call("myolo_adam_step", X.ptr(p), X.ptr(g), X.ptr(m), X.ptr(v), 4, lr_t, 0.9, 0.999, 1e-8, 1.0, X.stream())
    np.testing.assert_allclose(p.cpu().numpy(), _adam_expected(), rtol=1e-6)


# ------------------------------------------------------------------ 3. BatchNormalization moving statistics
def _bn_case():
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((3, 5, 4, 8)) * 2.0 + 1.0).astype(F32)        # n = 60 samples per channel
    n = 60
    mean = x.reshape(-1, 8).astype(np.float64).mean(0)
    var_b = x.reshape(-1, 8).astype(np.float64).var(0)                      # what y is normalised with (biased)
    eps, mom = 1e-3, 0.99
    # tf.nn.fused_batch_norm (fused_batch_norm_op.cc / nn_fused_batchnorm_test.py::_training_ref): batch_var = var * n/(n-1)
    fused_var = var_b * n / (n - 1.0)
    # keras 2.2.4 normalization.py: variance *= sample_size / (sample_size - (1.0 + epsilon)); K.moving_average_update(..., momentum)
    keras_var = fused_var * n / (n - (1.0 + eps))
    mm0, mv0 = np.full(8, 0.25), np.full(8, 1.5)
    return x, n, mean, var_b, mm0 * mom + mean * (1 - mom), mv0 * mom + keras_var * (1 - mom), mv0 * mom + var_b * n / (n - (1 + eps)) * (1 - mom)


def test_keras_bn_moving_update_on_fused_tf_variance_oracle():
    x, n, mean, var_b, mm_exp, mv_fused, mv_plain = _bn_case()
    g, b = np.ones(8, F32), np.zeros(8, F32)
    y, cache = O.bn_train(x, g, b)
    np.testing.assert_allclose(cache[3], var_b, rtol=1e-5)                  # normalisation itself uses the biased variance
    np.testing.assert_allclose(y.reshape(-1, 8).var(0), var_b / (var_b + 1e-3), rtol=1e-4)
    mm, mv = O.bn_moving_update(np.full(8, 0.25, F32), np.full(8, 1.5, F32), cache[2], cache[3], n)
    np.testing.assert_allclose(mm, mm_exp, rtol=1e-6)
    np.testing.assert_allclose(mv, mv_fused, rtol=1e-6)
    _, mv0 = O.bn_moving_update(np.full(8, 0.25, F32), np.full(8, 1.5, F32), cache[2], cache[3], n, fused_tf=False)
    np.testing.assert_allclose(mv0, mv_plain, rtol=1e-6)
    assert np.all(mv > mv0)                                                 # the two conventions really differ (by n/(n-1))


@pytest.mark.gpu
def test_keras_bn_moving_update_on_fused_tf_variance_hip():
    import torch
    from myolo import _ext as X
    x, n, mean, var_b, mm_exp, mv_fused, mv_plain = _bn_case()
    xt = torch.as_tensor(x.reshape(-1, 8)).cuda().contiguous()
    ws = torch.empty(8 << 20, dtype=torch.uint8, device="cuda")
    for opt, expect in ((1, mv_fused), (0, mv_plain)):
        g, b = torch.ones(8, device="cuda"), torch.zeros(8, device="cuda")
        mean_t, var_t, sc, sh = (torch.empty(8, device="cuda") for _ in range(4))
        mm, mv = torch.full((8,), 0.25, device="cuda"), torch.full((8,), 1.5, device="cuda")
        with X.option("bn_fused_tf_variance", opt):
            X.call("myolo_bn_stats", X.ptr(xt), X.ptr(g), X.ptr(b), X.ptr(mean_t), X.ptr(var_t), X.ptr(sc), X.ptr(sh), X.ptr(mm), X.ptr(mv),
                   n, 8, ws.data_ptr(), ws.numel(), X.stream())
            torch.cuda.synchronize()
        np.testing.assert_allclose(var_t.cpu().numpy(), var_b, rtol=1e-5)
        np.testing.assert_allclose(mm.cpu().numpy(), mm_exp, rtol=1e-6)
        np.testing.assert_allclose(mv.cpu().numpy(), expect, rtol=1e-6)


# ------------------------------------------------------------------ 4. keras_applications / reference padding
def test_keras_applications_depthwise_stride2_pads_bottom_right_oracle():
    # ZeroPadding2D(((0, 1), (0, 1))) + 3x3 'valid' stride 2 on a 4x4 map: output (0,0) reads rows/cols 0..2, output (1,1)
    # reads rows/cols 2..4 where index 4 is the padding -> an impulse at (3,3) reaches output (1,1) through tap (1,1) only.
    x = np.zeros((1, 4, 4, 1), F32)
    x[0, 3, 3, 0] = 1.0
    w = np.arange(1, 10, dtype=F32).reshape(3, 3, 1)
    y = O.dwconv3x3(x, w, 2)
    assert y.shape == (1, 2, 2, 1)
    np.testing.assert_array_equal(y[0, :, :, 0], [[0, 0], [0, w[1, 1, 0]]])
    x[:] = 0
    x[0, 0, 0, 0] = 1.0                                  # symmetric (1,1) padding would put this under tap (1,1) of output (0,0)
    np.testing.assert_array_equal(O.dwconv3x3(x, w, 2)[0, :, :, 0], [[w[0, 0, 0], 0], [0, 0]])


def test_reference_conv_block_pads_symmetric_oracle():
    # model.py:45-46: ZeroPadding2D(padding=(1, 1)) + Conv2D(3x3, strides 2, 'valid'): output (0,0) sees input (0,0) under tap (1,1)
    x = np.zeros((1, 4, 4, 3), F32)
    x[0, 0, 0, :] = 1.0
    w = np.zeros((3, 3, 3, 1), F32)
    w[1, 1, :, 0] = [1, 2, 3]
    y = O.conv2d(x, w, stride=2, pads=O.conv1_pads())
    assert y.shape == (1, 2, 2, 1) and y[0, 0, 0, 0] == 6.0 and np.abs(y).sum() == 6.0


# ------------------------------------------------------------------ 5. tf.round
def test_tf_round_docstring_vector_oracle():
    x = np.array([0.9, 2.5, 2.3, 1.5, -4.5], F32)
    np.testing.assert_array_equal(O.round_half_even(x), [1.0, 2.0, 2.0, 2.0, -4.0])
