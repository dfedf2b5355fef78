"""GPU parity tests, op by op: every C-ABI entry point of libmyolo_hip.so against the CPU oracle
(oracle/np_ops.py) on seeded inputs.  Tolerances: 1e-3 (relative to the tensor's max magnitude)
for floating-point results -- the north-star's fp32 bound; bit-exact (np.array_equal) for
integer / index outputs and for the float outputs that feed integer decisions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import np_ops as O                      # noqa: E402
from myolo import _ext as X                         # noqa: E402
from myolo.config import make_config, ShapesConfig  # noqa: E402

DEV = "cuda:0"
TOL = 1e-3


_KEEP = []   # device tensors must outlive the asynchronous kernel that reads them


@pytest.fixture(autouse=True)
def _keepalive():
    yield
    torch.cuda.synchronize()
    del _KEEP[:]


def dt(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a), device=DEV)
    t = t if dtype is None else t.to(dtype)
    _KEEP.append(t)
    return t


def new(*shape, dtype=torch.float32):
    return torch.full(shape, float("nan") if dtype == torch.float32 else 0, dtype=dtype, device=DEV)


def ws():
    if not hasattr(ws, "buf"):
        ws.buf = torch.empty(512 << 20, dtype=torch.uint8, device=DEV)
    return ws.buf.data_ptr(), ws.buf.numel()


def relerr(got, ref):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    ref = np.asarray(ref, np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    assert np.isfinite(got).all(), "non-finite output"
    return float(np.abs(got - ref).max() / max(1.0, np.abs(ref).max()))


def check(got, ref, tol=TOL, what=""):
    e = relerr(got, ref)
    assert e <= tol, "%s: rel err %.3e > %.1e" % (what, e, tol)
    return e


def rnd(rng, *shape, scale=1.0):
    return (rng.standard_normal(shape) * scale).astype(np.float32)


# ----------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,Cin,Cout,bias", [(300, 32, 64, False), (1568, 1024, 27, True), (130, 16, 16, True), (1568, 1024, 24, True),
                                             (6272, 512, 512, False),                                        # split-K forward
                                             (676, 1024, 35, True), (1571, 1024, 35, False), (10, 16, 63, True),    # pw_skinny_fwd_kernel (conv_23)

                                             (4096, 64, 128, False), (257, 512, 1024, False),
                                             (20003, 32, 64, False), (16391, 64, 128, False), (25088, 64, 64, False),    # thin-layer weight gradient
                                             (25088, 128, 256, False), (100352, 32, 64, False), (8193, 64, 24, False)])   # + pw_bwd_data_thin_kernel
def test_pwconv1x1(M, Cin, Cout, bias):
    rng = np.random.default_rng(1)
    x, w = rnd(rng, M, Cin), rnd(rng, Cin, Cout, scale=0.1)
    b = rnd(rng, Cout) if bias else None
    dy = rnd(rng, M, Cout)
    y = new(M, Cout)
    X.call("myolo_pwconv1x1_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)) if bias else None, X.ptr(y), M, Cin, Cout, *ws(), X.stream())
    ref = x.astype(np.float64) @ w + (b if bias else 0)
    check(y, ref, what="pw fwd")
    dx, dw = new(M, Cin), new(Cin, Cout)
    X.call("myolo_pwconv1x1_bwd_data", X.ptr(dt(dy)), X.ptr(dt(w)), X.ptr(dx), M, Cin, Cout, *ws(), X.stream())
    X.call("myolo_pwconv1x1_bwd_weight", X.ptr(dt(x)), X.ptr(dt(dy)), X.ptr(dw), M, Cin, Cout, *ws(), X.stream())
    check(dx, dy.astype(np.float64) @ w.T, what="pw dx")
    check(dw, x.astype(np.float64).T @ dy, what="pw dw")


@pytest.mark.parametrize("M,Cin,Cout", [(6272, 512, 512), (25088, 256, 512), (1568, 1024, 1024), (1300, 512, 256), (2049, 256, 256), (1100, 256, 1024)])
def test_pwconv1x1_gradients_bf16x6(M, Cin, Cout, request):
    """FP32_MATMUL = "bf16x6" (library switch wino_x6): the data and weight gradients of the pointwise layers with multiples of 256
    channels run on wino_mm_x6_kernel (NT: w itself is the transposed operand) / wino_tn_x6_kernel -- against float64 at the suite's
    bound, and against the fp32-MFMA kernels on the same operands at fp32 level."""
    old = X.set_option("wino_x6", 1)
    request.addfinalizer(lambda: X.set_option("wino_x6", old))
    rng = np.random.default_rng(41)
    x, w, dy = rnd(rng, M, Cin), rnd(rng, Cin, Cout, scale=0.1), rnd(rng, M, Cout)
    xt, wt, dyt = dt(x), dt(w), dt(dy)
    res = {}
    for no in (0, 1):
        with X.option("pw_no_x6", no):
            dx, dw = new(M, Cin), new(Cin, Cout)
            X.call("myolo_pwconv1x1_bwd_data", X.ptr(dyt), X.ptr(wt), X.ptr(dx), M, Cin, Cout, *ws(), X.stream())
            X.call("myolo_pwconv1x1_bwd_weight", X.ptr(xt), X.ptr(dyt), X.ptr(dw), M, Cin, Cout, *ws(), X.stream())
            torch.cuda.synchronize()
            res[no] = (dx, dw)
    check(res[0][0], dy.astype(np.float64) @ w.T, what="pw dx (bf16x6)")
    check(res[0][1], x.astype(np.float64).T @ dy, what="pw dw (bf16x6)")
    for a, b in zip(res[0], res[1]):
        assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max())


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(3, 14, 14, 32, 64), (2, 7, 9, 16, 48), (1, 28, 28, 128, 256), (5, 14, 14, 256, 256)])
def test_conv3x3(N, H, W, Cin, Cout):
    rng = np.random.default_rng(2)
    x, w, b = rnd(rng, N, H, W, Cin), rnd(rng, 3, 3, Cin, Cout, scale=0.05), rnd(rng, Cout)
    dy = rnd(rng, N, H, W, Cout)
    y = new(N, H, W, Cout)
    X.call("myolo_conv3x3_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), X.ptr(y), N, H, W, Cin, Cout, *ws(), X.stream())
    check(y, O.conv2d(x, w, pads=(1, 1, 1, 1), bias=b, acc=np.float64), what="conv3 fwd")
    rdx, rdw, rdb = O.conv2d_bwd(x, w, dy, pads=(1, 1, 1, 1), acc=np.float64)
    dx, dw = new(N, H, W, Cin), new(3, 3, Cin, Cout)
    X.call("myolo_conv3x3_bwd_data", X.ptr(dt(dy)), X.ptr(dt(w)), X.ptr(dx), N, H, W, Cin, Cout, *ws(), X.stream())
    X.call("myolo_conv3x3_bwd_weight", X.ptr(dt(x)), X.ptr(dt(dy)), X.ptr(dw), N, H, W, Cin, Cout, *ws(), X.stream())
    check(dx, rdx, what="conv3 dx")
    check(dw, rdw, what="conv3 dw")
    db = new(Cout)
    X.call("myolo_colsum", X.ptr(dt(dy.reshape(-1, Cout))), X.ptr(db), N * H * W, Cout, *ws(), X.stream())
    check(db, rdb, what="colsum")


@pytest.mark.parametrize("x6", [0, 1])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(3, 14, 14, 32, 64), (2, 7, 9, 16, 48), (1, 28, 28, 128, 256), (5, 14, 14, 256, 256),
                                            (2, 16, 12, 64, 32), (40, 14, 14, 256, 256), (3, 14, 14, 256, 512), (2, 14, 14, 16, 256)])
def test_conv3x3_winograd(N, H, W, Cin, Cout, x6, request):
    """Winograd F(4x4,3x3) form of the same three operators, same oracle, same 1e-3 bound (ragged tiles: 14 = 3.5 tiles,
    7x9, and exact multiples of 4); x6 = 1: the multiply forms its fp32 products from six bf16 piece products (csrc/wino_mm.hip)
    wherever that kernel applies (K % 16 == 0, N % 256 == 0: forward of the 256/512-column cases, data gradient of the 256-row ones)."""
    opt = X.option("wino_x6", x6)
    opt.__enter__()
    request.addfinalizer(lambda: opt.__exit__(None, None, None))
    rng = np.random.default_rng(2)
    x, w, b = rnd(rng, N, H, W, Cin), rnd(rng, 3, 3, Cin, Cout, scale=0.05), rnd(rng, Cout)
    dy = rnd(rng, N, H, W, Cout)
    wsb = torch.empty(max(X.wino_ws_bytes(N, H, W, Cin, Cout, k) for k in (0, 1, 2)), dtype=torch.uint8, device=DEV)
    _KEEP.append(wsb)
    wsa = (wsb.data_ptr(), wsb.numel())
    T = N * ((H + 3) // 4) * ((W + 3) // 4)
    y, vk = new(N, H, W, Cout), new(36, T, Cin)
    X.call("myolo_conv3x3_wino_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), None, None, X.ptr(y), N, H, W, Cin, Cout, 0,
           X.ptr(vk), *wsa, X.stream())
    ref = O.conv2d(x, w, pads=(1, 1, 1, 1), bias=b, acc=np.float64)
    check(y, ref, what="wino fwd")
    # folded-BN affine + ReLU epilogue, V in the workspace
    sc, sh = (1 + 0.1 * rnd(rng, Cout)), rnd(rng, Cout, scale=0.1)
    y2 = new(N, H, W, Cout)
    X.call("myolo_conv3x3_wino_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), X.ptr(dt(sc)), X.ptr(dt(sh)), X.ptr(y2), N, H, W, Cin,
           Cout, 1, None, *wsa, X.stream())
    check(y2, np.maximum(ref * sc + sh, 0), what="wino fwd affine relu")
    rdx, rdw, rdb = O.conv2d_bwd(x, w, dy, pads=(1, 1, 1, 1), acc=np.float64)
    dx, dw, dw2 = new(N, H, W, Cin), new(3, 3, Cin, Cout), new(3, 3, Cin, Cout)
    X.call("myolo_conv3x3_wino_bwd_data", X.ptr(dt(dy)), X.ptr(dt(w)), X.ptr(dx), N, H, W, Cin, Cout, *wsa, X.stream())
    check(dx, rdx, what="wino dx")
    X.call("myolo_conv3x3_wino_bwd_weight", None, X.ptr(vk), X.ptr(dt(dy)), X.ptr(dw), N, H, W, Cin, Cout, *wsa, X.stream())
    check(dw, rdw, what="wino dw (saved V)")
    X.call("myolo_conv3x3_wino_bwd_weight", X.ptr(dt(x)), None, X.ptr(dt(dy)), X.ptr(dw2), N, H, W, Cin, Cout, *wsa, X.stream())
    check(dw2, rdw, what="wino dw (from x)")


@pytest.mark.parametrize("N,H,W,C", [(7, 14, 14, 256), (3, 9, 6, 64), (40, 14, 14, 128)])
def test_winograd_transforms_with_folded_batchnorm(N, H, W, C):
    """(a) output transform that also yields the training-mode BN statistics == plain output transform + myolo_bn_stats;
    (b) input transform with BN apply + ReLU folded into the load == input transform of the separately normalised tensor."""
    rng = np.random.default_rng(11)
    T = N * ((H + 3) // 4) * ((W + 3) // 4)
    Mw, bias = rnd(rng, 36, T, C), rnd(rng, C)
    gamma, beta = 1 + 0.1 * rnd(rng, C), rnd(rng, C, scale=0.1)
    mm0, mv0 = rnd(rng, C, scale=0.1), (1 + 0.1 * rng.random(C)).astype(np.float32)
    y_ref, y = new(N * H * W, C), new(N * H * W, C)
    st = X.stream()
    X.call("myolo_wino_output_transform", X.ptr(dt(Mw)), X.ptr(dt(bias)), None, None, X.ptr(y_ref), N, H, W, C, 0, st)
    outs = []
    for fused in (False, True):
        mean, var, sc, sh = new(C), new(C), new(C), new(C)
        mm, mv = dt(mm0.copy()), dt(mv0.copy())
        if fused:
            X.call("myolo_wino_output_transform_bn_stats", X.ptr(dt(Mw)), X.ptr(dt(bias)), X.ptr(y), N, H, W, C, X.ptr(dt(gamma)),
                   X.ptr(dt(beta)), X.ptr(mean), X.ptr(var), X.ptr(sc), X.ptr(sh), X.ptr(mm), X.ptr(mv), *ws(), st)
        else:
            X.call("myolo_bn_stats", X.ptr(y_ref), X.ptr(dt(gamma)), X.ptr(dt(beta)), X.ptr(mean), X.ptr(var), X.ptr(sc), X.ptr(sh),
                   X.ptr(mm), X.ptr(mv), N * H * W, C, *ws(), st)
        outs.append([t.cpu().numpy() for t in (mean, var, sc, sh, mm, mv)])
    assert torch.equal(y, y_ref)
    for a, b, name in zip(outs[0], outs[1], ("mean", "var", "scale", "shift", "moving_mean", "moving_var")):
        assert np.abs(a - b).max() <= 2e-6 * max(1.0, np.abs(a).max()), name
    ym = y_ref.cpu().numpy().astype(np.float64)
    assert np.abs(outs[1][0] - ym.mean(0)).max() < 1e-5 and np.abs(outs[1][1] - ym.var(0)).max() < 1e-4 * max(1.0, ym.var(0).max())
    # (b)
    sc, sh = dt(outs[1][2]), dt(outs[1][3])
    a = new(N * H * W, C)
    X.call("myolo_bn_apply_act", X.ptr(y_ref), X.ptr(sc), X.ptr(sh), X.ptr(a), N * H * W, C, 1, st)
    V1, V2 = new(36, T, C), new(36, T, C)
    X.call("myolo_wino_input_transform", X.ptr(a), X.ptr(V1), N, H, W, C, st)
    X.call("myolo_wino_input_transform_affine", X.ptr(y_ref), X.ptr(sc), X.ptr(sh), 1, X.ptr(V2), N, H, W, C, st)
    n = X.wino_plane_elems(N, H, W, C)                    # the planes' used prefix (mixed tiling: <= 36*T*C)
    V1, V2 = V1.view(-1)[:n], V2.view(-1)[:n]
    assert not bool(torch.isnan(V1).any()) and float((V1 - V2).abs().max()) <= 1e-5 * float(V1.abs().max())


@pytest.mark.parametrize("B,H,W,C,nb,crop", [(2, 28, 28, 256, 40, 14), (3, 7, 9, 16, 11, 5), (1, 16, 16, 64, 6, 8)])
def test_winograd_input_transform_fused_with_roialign(B, H, W, C, nb, crop):
    """V straight from the feature map == crop_and_resize followed by the input transform (same sampling expressions)."""
    rng = np.random.default_rng(7)
    img = rnd(rng, B, H, W, C)
    boxes = _boxes(rng, nb)
    boxes[0] = [-0.2, 0.1, 0.7, 1.3]                          # extrapolated samples (value 0) on two sides
    bind = rng.integers(0, B, nb).astype(np.int32)
    T = nb * ((crop + 3) // 4) ** 2
    x, V1, V2 = new(nb, crop, crop, C), new(36, T, C), new(36, T, C)
    a = (X.ptr(dt(img)), X.ptr(dt(boxes)), X.ptr(dt(bind)))
    X.call("myolo_crop_and_resize_fwd", *a, X.ptr(x), B, H, W, C, nb, crop, crop, X.stream())
    X.call("myolo_wino_input_transform", X.ptr(x), X.ptr(V1), nb, crop, crop, C, X.stream())
    X.call("myolo_wino_input_transform_roialign", *a, X.ptr(V2), B, H, W, C, nb, crop, crop, X.stream())
    n = X.wino_plane_elems(nb, crop, crop, C)
    V1, V2 = V1.view(-1)[:n], V2.view(-1)[:n]
    assert not bool(torch.isnan(V1).any())
    assert float((V1 - V2).abs().max()) <= 1e-5 * float(V1.abs().max())      # fma contraction differs between the two kernels


def test_winograd_mixed_tiling_counts_and_agrees_with_uniform_tiling():
    """14 = 4+4+4+2: the last tile row / column uses F(2,3) embedded in F(4,3)'s point set, 484 point-tiles per image instead
    of 576; same convolution as the all-F(4,3) tiling (option wino_no_mixed) to fp32 noise, forward and both gradients."""
    assert X.wino_plane_elems(1, 14, 14, 1) == 484 and X.wino_plane_elems(1, 28, 28, 1) == 36 * 49
    assert X.wino_plane_elems(1, 7, 9, 1) == 16 * 6 + 8 * 4 + 8 * 6 + 4 * 4            # 7 = 4+3 (not reduced), 9 = 4+4+1 (reduced)
    with X.option("wino_no_mixed", 1):
        assert X.wino_plane_elems(1, 14, 14, 1) == 576
    rng = np.random.default_rng(3)
    N, H, W, Cin, Cout = 6, 14, 14, 64, 32
    x, w, b, dy = rnd(rng, N, H, W, Cin), rnd(rng, 3, 3, Cin, Cout, scale=0.05), rnd(rng, Cout), rnd(rng, N, H, W, Cout)
    wsb = torch.empty(max(X.wino_ws_bytes(N, H, W, Cin, Cout, k) for k in (0, 1, 2)) * 2, dtype=torch.uint8, device=DEV)
    _KEEP.append(wsb)
    wsa = (wsb.data_ptr(), wsb.numel())
    res = []
    for no_mixed in (0, 1):
        with X.option("wino_no_mixed", no_mixed):
            y, dx, dw = new(N, H, W, Cout), new(N, H, W, Cin), new(3, 3, Cin, Cout)
            X.call("myolo_conv3x3_wino_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), None, None, X.ptr(y), N, H, W, Cin, Cout, 0, None, *wsa, X.stream())
            X.call("myolo_conv3x3_wino_bwd_data", X.ptr(dt(dy)), X.ptr(dt(w)), X.ptr(dx), N, H, W, Cin, Cout, *wsa, X.stream())
            X.call("myolo_conv3x3_wino_bwd_weight", X.ptr(dt(x)), None, X.ptr(dt(dy)), X.ptr(dw), N, H, W, Cin, Cout, *wsa, X.stream())
            torch.cuda.synchronize()
            res.append([t.cpu().numpy() for t in (y, dx, dw)])
    for a, bb, name in zip(res[0], res[1], ("y", "dx", "dw")):
        assert np.isfinite(a).all() and np.abs(a - bb).max() <= 2e-5 * np.abs(bb).max(), name


def test_winograd_error_is_at_fp32_level():
    """The Winograd form is the same fp32 arithmetic with the sums associated differently; its error against a float64
    convolution must stay at fp32 rounding level (a reduced-precision product would sit at 1e-3), recorded here against
    the direct kernel's on the mask-head shape (14x14, 256 -> 256, activations O(1), glorot-scale weights)."""
    rng = np.random.default_rng(5)
    N, H, W, C = 24, 14, 14, 256
    x = np.maximum(rnd(rng, N, H, W, C), 0)                       # post-ReLU activations
    w, b = rnd(rng, 3, 3, C, C, scale=0.03), rnd(rng, C, scale=0.1)
    ref = O.conv2d(x, w, pads=(1, 1, 1, 1), bias=b, acc=np.float64)
    wsb = torch.empty(X.wino_ws_bytes(N, H, W, C, C, 0), dtype=torch.uint8, device=DEV)
    _KEEP.append(wsb)
    yd, yw = new(N, H, W, C), new(N, H, W, C)
    X.call("myolo_conv3x3_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), X.ptr(yd), N, H, W, C, C, *ws(), X.stream())
    X.call("myolo_conv3x3_wino_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), None, None, X.ptr(yw), N, H, W, C, C, 0, None,
           wsb.data_ptr(), wsb.numel(), X.stream())
    ed, ew = relerr(yd, ref), relerr(yw, ref)
    rms = lambda a: float(np.sqrt(np.mean((a.cpu().numpy() - ref) ** 2)) / np.sqrt(np.mean(ref ** 2)))   # noqa: E731
    print("max-norm rel err: direct %.2e  winograd %.2e;  rms rel err: direct %.2e  winograd %.2e" % (ed, ew, rms(yd), rms(yw)))
    assert ed < 5e-6 and ew < 5e-5, (ed, ew)


@pytest.mark.parametrize("x6", [0, 1])
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(3, 14, 14, 256, 256), (2, 5, 7, 32, 64), (23, 14, 14, 256, 256), (25, 14, 13, 64, 256), (21, 14, 14, 512, 256)])
def test_deconv2x2s2(N, H, W, Cin, Cout, x6, request):
    """x6 = 1 (FP32_MATMUL = "bf16x6"): from 4096 input pixels up the forward (scatter epilogue) and the data gradient (the four taps
    gathered into the A operand) run on wino_mm_x6_kernel -- the compacted mask-head backward at realistic positive counts."""
    old = X.set_option("wino_x6", x6)
    request.addfinalizer(lambda: X.set_option("wino_x6", old))
    rng = np.random.default_rng(3)
    x, w, b = rnd(rng, N, H, W, Cin), rnd(rng, 2, 2, Cout, Cin, scale=0.05), rnd(rng, Cout)
    dy = rnd(rng, N, 2 * H, 2 * W, Cout)
    y = new(N, 2 * H, 2 * W, Cout)
    X.call("myolo_deconv2x2s2_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), X.ptr(y), N, H, W, Cin, Cout, 1, *ws(), X.stream())
    ref = O.relu(O.deconv2x2s2(x, w, b))
    check(y, ref, what="deconv fwd")
    rdx, rdw, rdb = O.deconv2x2s2_bwd(x, w, dy)
    dx, dw = new(N, H, W, Cin), new(2, 2, Cout, Cin)
    X.call("myolo_deconv2x2s2_bwd_data", X.ptr(dt(dy)), X.ptr(dt(w)), X.ptr(dx), N, H, W, Cin, Cout, *ws(), X.stream())
    X.call("myolo_deconv2x2s2_bwd_weight", X.ptr(dt(x)), X.ptr(dt(dy)), X.ptr(dw), N, H, W, Cin, Cout, *ws(), X.stream())
    check(dx, rdx, what="deconv dx")
    check(dw, rdw, what="deconv dw")


@pytest.mark.parametrize("N,H,W,Cin,Cout,C", [(3, 14, 14, 256, 256, 4), (5, 14, 14, 256, 256, 2), (2, 5, 7, 32, 128, 1), (37, 14, 14, 64, 128, 3)])
@pytest.mark.parametrize("x6", [0, 1])
def test_deconv_mask_fused(N, H, W, Cin, Cout, C, x6, request):
    """deconv + ReLU + 1x1 + sigmoid in one pass == oracle, and == the two-kernel path up to summation order (x6 = 1: with the
    fp32 products formed from six bf16 piece products where csrc/wino_mm.hip applies, Cout % 256 == 0)."""
    opt = X.option("wino_x6", x6)
    opt.__enter__()
    request.addfinalizer(lambda: opt.__exit__(None, None, None))
    rng = np.random.default_rng(4)
    x, w, b = rnd(rng, N, H, W, Cin), rnd(rng, 2, 2, Cout, Cin, scale=0.05), rnd(rng, Cout)
    w2, b2 = rnd(rng, Cout, C, scale=0.1), rnd(rng, C)
    wsb = torch.empty(X.deconv_mask_ws_bytes(N, H, W, Cin, Cout, C), dtype=torch.uint8, device=DEV)
    _KEEP.append(wsb)
    p = new(N, 2 * H, 2 * W, C)
    X.call("myolo_deconv2x2s2_mask_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), X.ptr(dt(w2)), X.ptr(dt(b2)), X.ptr(p), N, H, W, Cin, Cout,
           C, wsb.data_ptr(), wsb.numel(), X.stream())
    d = O.relu(O.deconv2x2s2(x, w, b)).astype(np.float64)
    ref = 1 / (1 + np.exp(-(d.reshape(-1, Cout) @ w2 + b2))).reshape(N, 2 * H, 2 * W, C)
    check(p, ref, 1e-5, "fused deconv+mask")
    y, p2 = new(N, 2 * H, 2 * W, Cout), new(N, 2 * H, 2 * W, C)
    X.call("myolo_deconv2x2s2_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), X.ptr(y), N, H, W, Cin, Cout, 1, *ws(), X.stream())
    X.call("myolo_mask_head_out_fwd", X.ptr(y), X.ptr(dt(w2)), X.ptr(dt(b2)), X.ptr(p2), N * 4 * H * W, Cout, C, X.stream())
    assert float((p - p2).abs().max()) < 2e-6
    p3 = new(N, 2 * H, 2 * W, C)
    X.call("myolo_deconv2x2s2_mask_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), X.ptr(dt(w2)), X.ptr(dt(b2)), X.ptr(p3), N, H, W, Cin, Cout,
           C, wsb.data_ptr(), wsb.numel(), X.stream())
    assert torch.equal(p, p3), "fused deconv+mask is not bit-reproducible"
    if x6 and Cout % 256 == 0:
        # round 6: the bf16x6 kernel forms the tile transposed (a lane = one pixel: the 1x1 conv's channel sum stays in registers) and, at 256
        # channels, sums the two waves' slabs and stores the sigmoid itself.  deconv_mask_legacy = 2: the same tile with partial logits + the
        # finish launch (same bits); = 1: the untransposed tile with the per-class butterfly of rounds 3-5 (another summation order)
        outs = {}
        for mode in (1, 2):
            with X.option("deconv_mask_legacy", mode):
                q = new(N, 2 * H, 2 * W, C)
                X.call("myolo_deconv2x2s2_mask_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), X.ptr(dt(w2)), X.ptr(dt(b2)), X.ptr(q), N, H, W, Cin, Cout,
                       C, wsb.data_ptr(), wsb.numel(), X.stream())
                torch.cuda.synchronize()
            outs[mode] = q
        assert torch.equal(p, outs[2]), "in-kernel finish differs from partial logits + deconv_mask_finish"
        assert float((p - outs[1]).abs().max()) < 2e-6
        check(outs[1], ref, 1e-5, "fused deconv+mask, legacy epilogue")


@pytest.mark.parametrize("M,C,act", [(32 * 112 * 112, 32, 2), (32 * 14 * 14, 512, 2), (32 * 7 * 7, 1024, 2), (4 * 9 * 9, 96, 1), (37, 8, 0), (5000, 256, 2),
                                     (1, 4, 2), (2051, 1024, 1)])
def test_batchnorm_backward_in_one_launch(M, C, act):
    """myolo_bn_act_bwd_fused (sums, grid-wide barrier per channel group, dx in ONE kernel) against the oracle and against the three-launch
    myolo_bn_act_bwd; the same counters serve many launches in a row (the barrier never resets them), and the result is bit-reproducible."""
    rng = np.random.default_rng(16)
    x = rnd(rng, M, C, scale=2.0) + rnd(rng, 1, C)
    g, b = 1 + rnd(rng, C, scale=0.2), rnd(rng, C, scale=0.3)
    dy = rnd(rng, M, C)
    actf = {0: lambda v: v, 1: O.relu, 2: O.relu6}[act]
    actb = {0: lambda a, d: d, 1: O.relu_bwd, 2: O.relu6_bwd}[act]
    y_ref, cache = O.bn_train(x, g, b)
    rdx, rdg, rdb = O.bn_train_bwd(cache, g, actb(actf(y_ref), dy))
    mean, var, scale, shift = new(C), new(C), new(C), new(C)
    tmm, tmv = new(C), new(C)
    xd, dyd = dt(x), dt(dy)
    X.call("myolo_bn_stats", X.ptr(xd), X.ptr(dt(g)), X.ptr(dt(b)), X.ptr(mean), X.ptr(var), X.ptr(scale), X.ptr(shift), X.ptr(tmm), X.ptr(tmv), M, C,
           *ws(), X.stream())
    assert X.load().myolo_bn_act_bwd_fused_ws_bytes(M, C) > 0
    sync = torch.zeros(64, dtype=torch.int32, device=DEV)
    outs = []
    for rep in range(5):
        dx, dg, db = new(M, C), new(C), new(C)
        X.call("myolo_bn_act_bwd_fused", X.ptr(dyd), X.ptr(xd), X.ptr(mean), X.ptr(var), X.ptr(scale), X.ptr(shift), X.ptr(dx), X.ptr(dg), X.ptr(db),
               M, C, act, X.ptr(sync), *ws(), X.stream())
        outs.append((dx, dg, db))
    torch.cuda.synchronize()
    assert int(sync[0].item()) == 5 * 256
    dx, dg, db = outs[0]
    check(dx, rdx, what="fused bn dx")
    check(dg, rdg, what="fused bn dgamma")
    check(db, rdb, what="fused bn dbeta")
    for o in outs[1:]:
        assert all(torch.equal(a, b2) for a, b2 in zip(o, outs[0])), "not bit-reproducible"
    dx3, dg3, db3 = new(M, C), new(C), new(C)
    X.call("myolo_bn_act_bwd", X.ptr(dyd), X.ptr(xd), X.ptr(dt(g)), X.ptr(mean), X.ptr(var), X.ptr(scale), X.ptr(shift), X.ptr(dx3), X.ptr(dg3), X.ptr(db3),
           M, C, act, 1, *ws(), X.stream())
    tol = 1e-5 * max(1.0, float(dx3.abs().max()))
    assert float((dx - dx3).abs().max()) <= tol and float((dg - dg3).abs().max()) <= 1e-4 * max(1.0, float(dg3.abs().max()))


def test_positive_index_on_device():
    """myolo_positive_index == the host construction the engine used through round 3 (an image's positives are its first n_pos ROIs, model.py:593),
    counts clamped to [0, R]."""
    rng = np.random.default_rng(5)
    for B, R in ((1, 7), (32, 147), (5, 245), (300, 3)):
        npos = rng.integers(0, R + 1, size=B).astype(np.int32)
        npos[rng.integers(0, B)] = 0
        if B > 2:
            npos[1] = R
        flags, idx, inv = (torch.full((B * R,), -7, dtype=torch.int32, device=DEV) for _ in range(3))
        tot = torch.zeros(1, dtype=torch.int32, device=DEV)
        X.call("myolo_positive_index", X.ptr(dt(npos)), B, R, X.ptr(flags), X.ptr(idx), X.ptr(inv), X.ptr(tot), X.stream())
        pos = np.concatenate([np.arange(b * R, b * R + int(npos[b]), dtype=np.int32) for b in range(B)])
        ref_inv = np.full(B * R, -1, np.int32)
        ref_inv[pos] = np.arange(len(pos), dtype=np.int32)
        assert int(tot.item()) == len(pos)
        assert np.array_equal(inv.cpu().numpy(), ref_inv)
        assert np.array_equal(idx.cpu().numpy()[:len(pos)], pos)
        assert np.all(idx.cpu().numpy()[len(pos):] == -7), "entries beyond the total must stay untouched"
        assert np.array_equal(flags.cpu().numpy(), (ref_inv >= 0).astype(np.int32))


@pytest.mark.parametrize("x6", [0, 1])
@pytest.mark.parametrize("N,H,W", [(9, 14, 14), (40, 5, 3), (66, 14, 14)])
def test_deconv_mask_fused_keeps_the_positives_rows(N, H, W, x6, request):
    """myolo_deconv2x2s2_mask_fwd_keep: the probabilities are those of the plain fused pass bit for bit, and the kept rows are bit for bit what
    myolo_deconv2x2s2_fwd(ReLU) gives on the gathered inputs (what the sparse backward used to re-run), slots at / beyond the cap untouched."""
    opt = X.option("wino_x6", x6)
    opt.__enter__()
    request.addfinalizer(lambda: opt.__exit__(None, None, None))
    Cin = Cout = 256
    C = 4
    rng = np.random.default_rng(6)
    x, w, b = rnd(rng, N, H, W, Cin), rnd(rng, 2, 2, Cout, Cin, scale=0.05), rnd(rng, Cout)
    w2, b2 = rnd(rng, Cout, C, scale=0.1), rnd(rng, C)
    wsb = torch.empty(X.deconv_mask_ws_bytes(N, H, W, Cin, Cout, C), dtype=torch.uint8, device=DEV)
    _KEEP.append(wsb)
    chosen = np.sort(rng.choice(N, size=max(2, N // 3), replace=False)).astype(np.int32)
    inv = np.full(N, -1, np.int32)
    inv[chosen] = np.arange(len(chosen), dtype=np.int32)
    cap = len(chosen) - 1                                  # the last chosen image is beyond the cap: not kept
    sentinel = -123.0
    dk = torch.full((len(chosen), 2 * H, 2 * W, Cout), sentinel, dtype=torch.float32, device=DEV)
    p0, p1 = new(N, 2 * H, 2 * W, C), new(N, 2 * H, 2 * W, C)
    args = (X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), X.ptr(dt(w2)), X.ptr(dt(b2)))
    X.call("myolo_deconv2x2s2_mask_fwd", *args, X.ptr(p0), N, H, W, Cin, Cout, C, wsb.data_ptr(), wsb.numel(), X.stream())
    X.call("myolo_deconv2x2s2_mask_fwd_keep", *args, X.ptr(p1), N, H, W, Cin, Cout, C, X.ptr(dt(inv)), X.ptr(dk), cap, wsb.data_ptr(), wsb.numel(), X.stream())
    assert torch.equal(p0, p1)
    xs = np.ascontiguousarray(x[chosen])
    y = new(len(chosen), 2 * H, 2 * W, Cout)
    X.call("myolo_deconv2x2s2_fwd", X.ptr(dt(xs)), X.ptr(dt(w)), X.ptr(dt(b)), X.ptr(y), len(chosen), H, W, Cin, Cout, 1, *ws(), X.stream())
    if x6 and len(chosen) * H * W >= 4096:      # the product's setting at >= 21 positives: both run wino_mm_x6_kernel, rows are independent of the launch's size
        assert torch.equal(dk[:cap], y[:cap]), "kept rows differ from the re-run deconv"
    else:                      # the stand-alone deconv is an fp32-MFMA kernel (below 4096 rows also under "wino_x6"): other summation order
        assert float((dk[:cap] - y[:cap]).abs().max()) < 2e-5
    assert bool((dk[cap:] == sentinel).all()), "a slot at / beyond the cap was written"
    check(dk[:cap], O.relu(O.deconv2x2s2(xs[:cap], w, b)), 1e-5, "kept deconv rows")


@pytest.mark.parametrize("N,H,W,C,stride", [(2, 16, 16, 32, 1), (2, 16, 16, 32, 2), (3, 7, 7, 128, 1), (1, 14, 10, 64, 2),
                                            (2, 9, 13, 16, 1),
                                            # round 4, the row-sliding kernels on ragged shapes: several strips with a partial last one, row chunks
                                            # that do not divide the height, channel counts of 3 / 5 blocks, the 32-quad form on rows of 7 and fewer
                                            (2, 46, 40, 64, 1), (2, 46, 40, 96, 2), (1, 30, 58, 160, 1), (4, 9, 7, 1024, 1), (2, 18, 14, 1024, 2),
                                            (3, 5, 3, 128, 1), (40, 28, 28, 32, 1)])
def test_dwconv3x3(N, H, W, C, stride):
    rng = np.random.default_rng(4)
    x, w = rnd(rng, N, H, W, C), rnd(rng, 3, 3, C)
    ref = O.dwconv3x3(x, w, stride)
    Ho, Wo = ref.shape[1], ref.shape[2]
    y = new(N, Ho, Wo, C)
    X.call("myolo_dwconv3x3_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(y), N, H, W, C, stride, X.stream())
    check(y, ref, what="dw fwd")
    dy = rnd(rng, N, Ho, Wo, C)
    rdx, rdw = O.dwconv3x3_bwd(x, w, dy, stride)
    dx, dw = new(N, H, W, C), new(3, 3, C)
    X.call("myolo_dwconv3x3_bwd_data", X.ptr(dt(dy)), X.ptr(dt(w)), X.ptr(dx), N, H, W, C, stride, X.stream())
    X.call("myolo_dwconv3x3_bwd_weight", X.ptr(dt(x)), X.ptr(dt(dy)), X.ptr(dw), N, H, W, C, stride, *ws(), X.stream())
    check(dx, rdx, what="dw dx")
    check(dw, rdw, what="dw dw")


def test_dw_s2_impulse_pins_pad_side():
    """4x4 impulse at (3,3): with pad bottom/right (keras_applications>=1.0.5) the stride-2 output
    at (1,1) sees it through tap (1,1)."""
    x = np.zeros((1, 4, 4, 4), np.float32)
    x[0, 3, 3, :] = 1
    w = np.arange(36, dtype=np.float32).reshape(3, 3, 4)
    y = new(1, 2, 2, 4)
    X.call("myolo_dwconv3x3_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(y), 1, 4, 4, 4, 2, X.stream())
    got = y.cpu().numpy()
    assert np.array_equal(got[0, 1, 1], w[1, 1]) and got[0, 0, 0].sum() == 0


@pytest.mark.parametrize("N,H,W,Co", [(2, 32, 32, 16), (3, 16, 24, 32), (4, 128, 128, 32), (2, 224, 224, 32), (3, 36, 20, 64), (1, 2, 2, 4), (2, 14, 16, 8), (1, 6, 400, 256), (1, 8, 544, 16), (1, 4, 548, 8)])
def test_conv1(N, H, W, Co):
    rng = np.random.default_rng(5)
    x, w = rng.random((N, H, W, 3), dtype=np.float32), rnd(rng, 3, 3, 3, Co)
    y = new(N, H // 2, W // 2, Co)
    X.call("myolo_conv3x3s2_c3_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(y), N, H, W, Co, X.stream())
    check(y, O.conv2d(x, w, stride=2, pads=(1, 1, 1, 1), acc=np.float64), what="conv1 fwd")
    dy = rnd(rng, N, H // 2, W // 2, Co)
    _, rdw, _ = O.conv2d_bwd(x, w, dy, stride=2, pads=(1, 1, 1, 1), acc=np.float64, need_dx=False)
    dw = new(3, 3, 3, Co)
    X.call("myolo_conv3x3s2_c3_bwd_weight", X.ptr(dt(x)), X.ptr(dt(dy)), X.ptr(dw), N, H, W, Co, *ws(), X.stream())
    check(dw, rdw, what="conv1 dw")


@pytest.mark.parametrize("M,C,act", [(1000, 32, 2), (37, 256, 1), (50000, 64, 2), (200, 1024, 0)])
def test_batchnorm(M, C, act):
    rng = np.random.default_rng(6)
    x = rnd(rng, M, C, scale=2.0) + rnd(rng, 1, C)
    g, b = 1 + rnd(rng, C, scale=0.2), rnd(rng, C, scale=0.3)
    mm, mv = rnd(rng, C, scale=0.1), 1 + np.abs(rnd(rng, C, scale=0.1))
    dy = rnd(rng, M, C)
    actf = {0: lambda v: v, 1: O.relu, 2: O.relu6}[act]
    actb = {0: lambda a, d: d, 1: O.relu_bwd, 2: O.relu6_bwd}[act]
    # training mode
    y_ref, cache = O.bn_train(x, g, b)
    a_ref = actf(y_ref)
    mean, var, scale, shift = new(C), new(C), new(C), new(C)
    tmm, tmv = dt(mm), dt(mv)
    X.call("myolo_bn_stats", X.ptr(dt(x)), X.ptr(dt(g)), X.ptr(dt(b)), X.ptr(mean), X.ptr(var), X.ptr(scale), X.ptr(shift),
           X.ptr(tmm), X.ptr(tmv), M, C, *ws(), X.stream())
    check(mean, cache[2], 1e-5, "bn mean")
    check(var, cache[3], 1e-4, "bn var")
    rmm, rmv = O.bn_moving_update(mm, mv, cache[2], cache[3], M)
    check(tmm, rmm, 1e-5, "moving mean")
    check(tmv, rmv, 1e-5, "moving var")
    a = new(M, C)
    X.call("myolo_bn_apply_act", X.ptr(dt(x)), X.ptr(scale), X.ptr(shift), X.ptr(a), M, C, act, X.stream())
    check(a, a_ref, what="bn apply")
    rdx, rdg, rdb = O.bn_train_bwd(cache, g, actb(a_ref, dy))
    dx, dg, db = new(M, C), new(C), new(C)
    X.call("myolo_bn_act_bwd", X.ptr(dt(dy)), X.ptr(dt(x)), X.ptr(dt(g)), X.ptr(mean), X.ptr(var), X.ptr(scale), X.ptr(shift),
           X.ptr(dx), X.ptr(dg), X.ptr(db), M, C, act, 1, *ws(), X.stream())
    check(dx, rdx, what="bn dx")
    check(dg, rdg, what="bn dgamma")
    check(db, rdb, what="bn dbeta")
    # frozen mode
    y_ref, cache = O.bn_infer(x, g, b, mm, mv)
    a_ref = actf(y_ref)
    X.call("myolo_bn_frozen_coeffs", X.ptr(dt(g)), X.ptr(dt(b)), X.ptr(dt(mm)), X.ptr(dt(mv)), X.ptr(scale), X.ptr(shift), C, X.stream())
    X.call("myolo_bn_apply_act", X.ptr(dt(x)), X.ptr(scale), X.ptr(shift), X.ptr(a), M, C, act, X.stream())
    check(a, a_ref, what="bn frozen apply")
    rdx, rdg, rdb = O.bn_infer_bwd(cache, g, actb(a_ref, dy))
    X.call("myolo_bn_act_bwd", X.ptr(dt(dy)), X.ptr(dt(x)), X.ptr(dt(g)), X.ptr(dt(mm)), X.ptr(dt(mv)), X.ptr(scale), X.ptr(shift),
           X.ptr(dx), X.ptr(dg), X.ptr(db), M, C, act, 0, *ws(), X.stream())
    check(dx, rdx, what="bn frozen dx")
    check(dg, rdg, what="bn frozen dgamma")
    check(db, rdb, what="bn frozen dbeta")


def _boxes(rng, nb):
    c = rng.random((nb, 2)) * 1.2 - 0.1
    s = rng.random((nb, 2)) * 0.6
    b = np.concatenate([c - s / 2, c + s / 2], 1).astype(np.float32)
    b[0] = [0, 0, 0, 0]                      # degenerate box (deprecated/test_mask.py:39)
    b[1] = [0.2, 0.6, 1.3, 0.9]              # partly outside -> extrapolation rows
    return b


@pytest.mark.parametrize("B,H,W,C,nb,crop", [(2, 28, 28, 256, 40, 14), (3, 7, 9, 16, 11, 5), (1, 16, 16, 64, 6, 1)])
def test_crop_and_resize(B, H, W, C, nb, crop):
    rng = np.random.default_rng(7)
    img = rnd(rng, B, H, W, C)
    boxes = _boxes(rng, nb)
    bind = rng.integers(0, B, nb).astype(np.int32)
    out = new(nb, crop, crop, C)
    X.call("myolo_crop_and_resize_fwd", X.ptr(dt(img)), X.ptr(dt(boxes)), X.ptr(dt(bind)), X.ptr(out), B, H, W, C, nb, crop, crop, X.stream())
    check(out, O.crop_and_resize(img, boxes, bind, (crop, crop)), 1e-5, "crop fwd")
    dout = rnd(rng, nb, crop, crop, C)
    dimg = new(B, H, W, C)
    X.call("myolo_crop_and_resize_bwd_image", X.ptr(dt(dout)), X.ptr(dt(boxes)), X.ptr(dt(bind)), X.ptr(dimg), B, H, W, C, nb, crop, crop, X.stream())
    check(dimg, O.crop_and_resize_bwd_image(dout, boxes, bind, img.shape), 1e-4, "crop bwd")


@pytest.mark.parametrize("B,H,W,C,R,crop", [(2, 28, 28, 256, 20, 14), (3, 7, 9, 16, 5, 5), (2, 12, 12, 64, 7, 14), (8, 28, 28, 256, 147, 14), (3, 8, 12, 256, 70, 7), (2, 52, 52, 256, 40, 14), (2, 16, 16, 256, 48, 14),
                                                  (2, 30, 17, 256, 33, 14)])
def test_roialign_bwd_grouped(B, H, W, C, R, crop):
    """gather formulation == scatter formulation (oracle), incl. degenerate all-zero boxes and boxes
    outside the image; and it is bit-reproducible."""
    rng = np.random.default_rng(17)
    boxes = _boxes(rng, B * R)
    boxes[R] = [0.3, 0.3, 0.3, 0.3]                                # zero-size box: all samples on one point
    bind = np.repeat(np.arange(B), R).astype(np.int32)
    dout = rnd(rng, B * R, crop, crop, C)
    ref = O.crop_and_resize_bwd_image(dout, boxes, bind, (B, H, W, C))
    d1, d2 = new(B, H, W, C), new(B, H, W, C)
    X.call("myolo_roialign_bwd_grouped", X.ptr(dt(dout)), X.ptr(dt(boxes)), X.ptr(d1), B, H, W, C, R, crop, crop, X.stream())
    X.call("myolo_roialign_bwd_grouped", X.ptr(dt(dout)), X.ptr(dt(boxes)), X.ptr(d2), B, H, W, C, R, crop, crop, X.stream())
    check(d1, ref, 1e-4, "roialign bwd grouped")
    assert torch.equal(d1, d2)
    if C == 256 and H % 4 == 0 and W % 4 == 0:
        # the quad-per-wave form (a crop sample that touches a 2 x 2 pixel quad is loaded once; option tune0 & 131072): the same sums bit for bit
        d3 = new(B, H, W, C)
        with X.option("tune0", 131072):
            X.call("myolo_roialign_bwd_grouped", X.ptr(dt(dout)), X.ptr(dt(boxes)), X.ptr(d3), B, H, W, C, R, crop, crop, X.stream())
        assert torch.equal(d1, d3), float((d1 - d3).abs().max())


@pytest.mark.parametrize("B,G,A,C", [(4, 7, 3, 4), (2, 13, 5, 2), (3, 4, 3, 4)])
def test_yolo_decode_bit_exact(B, G, A, C):
    rng = np.random.default_rng(8)
    yp = rnd(rng, B, G, G, A, 5 + C, scale=2.0)
    yp[0, 0, 0, 0, :] = 0                                   # zero-logit KAT row
    anchors = (rng.random(2 * A) * 4 + 0.5).astype(np.float32)
    prop, det = new(B, G * G * A, 4), new(B, G * G * A, 6)
    X.call("myolo_yolo_decode", X.ptr(dt(yp)), X.ptr(dt(anchors)), X.ptr(prop), B, G, A, C, X.stream())
    X.call("myolo_yolo_detections", X.ptr(dt(yp)), X.ptr(dt(anchors)), X.ptr(det), B, G, A, C, X.stream())
    assert np.array_equal(prop.cpu().numpy(), O.yolo_decode(yp, anchors, G)), "proposals not bit-exact"
    ref = O.yolo_detections(yp, anchors, G)
    got = det.cpu().numpy()
    assert np.array_equal(got, ref), "detections not bit-exact"
    assert np.array_equal(got[..., 5].astype(np.int32), np.argmax(yp[..., 5:], -1).reshape(B, -1))


def _targets_case(rng, cfg, B):
    H, W = cfg.IMAGE_SHAPE[:2]
    T, R = cfg.TRUE_BOX_BUFFER, cfg.TRAIN_ROIS_PER_IMAGE
    gt_boxes = np.zeros((B, T, 4), np.int32)
    gt_ids = np.zeros((B, T), np.int32)
    gt_masks = np.zeros((B, H, W, T), bool)
    for b in range(B):
        n = [3, 0, 1, T][b % 4]
        for k in range(n):
            x1, y1 = rng.integers(0, W - 30), rng.integers(0, H - 30)
            w, h = rng.integers(12, 30), rng.integers(12, 30)
            gt_boxes[b, k] = [x1, y1, x1 + w, y1 + h]
            gt_ids[b, k] = rng.integers(1, cfg.NUM_CLASSES)
            yy, xx = np.mgrid[0:H, 0:W]
            gt_masks[b, :, :, k] = ((xx - (x1 + w / 2)) ** 2 / (w / 2) ** 2 + (yy - (y1 + h / 2)) ** 2 / (h / 2) ** 2) <= 1
    prop = np.zeros((B, R, 4), np.float32)
    for b in range(B):
        for r in range(R):
            if r % 3 == 0 and gt_boxes[b].any():
                k = rng.integers(0, max(1, (gt_boxes[b].sum(1) > 0).sum()))
                g = gt_boxes[b, k].astype(np.float32) / np.array([W, H, W, H], np.float32)
                prop[b, r] = g + (rng.random(4).astype(np.float32) - 0.5) * 0.08
            else:
                c, s = rng.random(2), rng.random(2) * 0.5 + 0.02
                prop[b, r] = np.concatenate([c - s / 2, c + s / 2])
    return prop, gt_ids, gt_boxes, gt_masks


@pytest.mark.parametrize("size,nbox", [(128, 3), (224, 5)])
def test_mask_targets_bit_exact(size, nbox):
    anchors = [1.0, 1.0] * nbox
    cfg = make_config(ShapesConfig, IMAGE_SHAPE=[size, size, 3], N_BOX=nbox, ANCHORS=anchors)
    rng = np.random.default_rng(9)
    B = 4
    prop, gt_ids, gt_boxes, gt_masks = _targets_case(rng, cfg, B)
    R, T = cfg.TRAIN_ROIS_PER_IMAGE, cfg.TRUE_BOX_BUFFER
    rois, tcls, tmask, npos = new(B, R, 4), new(B, R, dtype=torch.int32), new(B, R, 28, 28), new(B, dtype=torch.int32)
    X.call("myolo_mask_targets", X.ptr(dt(prop)), X.ptr(dt(gt_ids)), X.ptr(dt(gt_boxes)), X.ptr(dt(gt_masks.view(np.uint8))),
           X.ptr(rois), X.ptr(tcls), X.ptr(tmask), X.ptr(npos), B, R, T, size, size, 28, 28, X.stream())
    r_rois, r_cls, r_mask, r_npos = O.mask_targets(prop, gt_ids, gt_boxes, gt_masks, cfg)
    assert r_npos.sum() > 0, "test case has no positives"
    assert np.array_equal(npos.cpu().numpy(), r_npos)
    assert np.array_equal(tcls.cpu().numpy(), r_cls)
    assert np.array_equal(rois.cpu().numpy(), r_rois)
    assert np.array_equal(tmask.cpu().numpy(), r_mask)


def test_yolo_loss():
    cfg = make_config(ShapesConfig, IMAGE_SHAPE=[224, 224, 3])
    rng = np.random.default_rng(10)
    B, G, A, C, T = 4, cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER
    yp = rnd(rng, B, G, G, A, 5 + C)
    yt = np.zeros_like(yp)
    tb = np.zeros((B, 1, 1, 1, T, 4), np.float32)
    for b in range(B):
        for k in range(b + 1):
            gy, gx, a = rng.integers(0, G, 2).tolist() + [int(rng.integers(0, A))]
            box = [gx + rng.random(), gy + rng.random(), 0.5 + 3 * rng.random(), 0.5 + 3 * rng.random()]
            yt[b, gy, gx, a, :4] = box
            yt[b, gy, gx, a, 4] = 1
            yt[b, gy, gx, a, 5 + rng.integers(1, C)] = 1
            tb[b, 0, 0, 0, k] = box
            yp[b, gy, gx, a, :4] = [0.1, -0.2, np.log(box[2] / cfg.ANCHORS[2 * a]) + 0.1, np.log(box[3] / cfg.ANCHORS[2 * a + 1])]
    ref = O.yolo_loss(yt, yp, tb, cfg, want_grad=True)
    terms, grad = new(8), new(*yp.shape)
    X.call("myolo_yolo_loss", X.ptr(dt(yt)), X.ptr(dt(yp)), X.ptr(dt(tb.reshape(B, T, 4))), X.ptr(dt(np.asarray(cfg.ANCHORS, np.float32))),
           X.ptr(dt(cfg.CLASS_WEIGHTS)), cfg.OBJECT_SCALE, cfg.NO_OBJECT_SCALE, cfg.COORD_SCALE, cfg.CLASS_SCALE, 1.0,
           X.ptr(terms), X.ptr(grad), B, G, A, C, T, *ws(), X.stream())
    t = terms.cpu().numpy()
    for i, k in enumerate(["loss", "loss_xy", "loss_wh", "loss_conf", "loss_class", "recall", "n_coord", "n_conf"]):
        assert abs(t[i] - float(ref[k])) <= 1e-4 * max(1.0, abs(float(ref[k]))), (k, t[i], ref[k])
    check(grad, ref["grad"], 1e-4, "yolo loss grad")


def test_yolo_loss_warmup_branch():
    """myolo_yolo_loss_warmup(warmup=1) == the oracle's warm-up branch (model.py:193-207), terms and gradient; warmup=0 == myolo_yolo_loss bit for bit."""
    cfg = make_config(ShapesConfig, IMAGE_SHAPE=[224, 224, 3], COORD_SCALE=3.0, WARM_UP_BATCHES=4)
    rng = np.random.default_rng(12)
    B, G, A, C, T = 4, cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER
    yp = rnd(rng, B, G, G, A, 5 + C)
    yt = np.zeros_like(yp)
    tb = np.zeros((B, 1, 1, 1, T, 4), np.float32)
    for b in range(B):
        for k in range(b + 1):
            gy, gx, a = rng.integers(0, G, 2).tolist() + [int(rng.integers(0, A))]
            box = [gx + rng.random(), gy + rng.random(), 0.5 + 3 * rng.random(), 0.5 + 3 * rng.random()]
            yt[b, gy, gx, a, :4] = box
            yt[b, gy, gx, a, 4] = 1
            yt[b, gy, gx, a, 5 + rng.integers(1, C)] = 1
            tb[b, 0, 0, 0, k] = box
    args = (X.ptr(dt(yt)), X.ptr(dt(yp)), X.ptr(dt(tb.reshape(B, T, 4))), X.ptr(dt(np.asarray(cfg.ANCHORS, np.float32))),
            X.ptr(dt(cfg.CLASS_WEIGHTS)), cfg.OBJECT_SCALE, cfg.NO_OBJECT_SCALE, cfg.COORD_SCALE, cfg.CLASS_SCALE, 1.0)
    out = {}
    for warm in (0, 1):
        terms, grad = new(8), new(*yp.shape)
        X.call("myolo_yolo_loss_warmup", *args, warm, X.ptr(terms), X.ptr(grad), B, G, A, C, T, *ws(), X.stream())
        ref = O.yolo_loss(yt, yp, tb, cfg, want_grad=True, warmup=bool(warm))
        t = terms.cpu().numpy()
        for i, k in enumerate(["loss", "loss_xy", "loss_wh", "loss_conf", "loss_class", "recall", "n_coord", "n_conf"]):
            assert abs(t[i] - float(ref[k])) <= 1e-4 * max(1.0, abs(float(ref[k]))), (warm, k, t[i], ref[k])
        check(grad, ref["grad"], 1e-4, "yolo loss grad, warmup=%d" % warm)
        out[warm] = (terms, grad)
    assert float(out[1][0][6]) == B * G * G * A and float(out[0][0][6]) == sum(range(1, B + 1)) or float(out[0][0][6]) <= sum(range(1, B + 1))
    terms, grad = new(8), new(*yp.shape)
    X.call("myolo_yolo_loss", *args, X.ptr(terms), X.ptr(grad), B, G, A, C, T, *ws(), X.stream())
    assert torch.equal(terms, out[0][0]) and torch.equal(grad, out[0][1])


def test_mask_head_out_and_bce():
    rng = np.random.default_rng(11)
    NR, hw, Cin, C = 6, 28 * 28, 256, 4
    M = NR * hw
    x = np.maximum(rnd(rng, M, Cin), 0)
    w, b = rnd(rng, Cin, C, scale=0.1), rnd(rng, C)
    p = new(M, C)
    X.call("myolo_mask_head_out_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(b)), X.ptr(p), M, Cin, C, X.stream())
    p_ref = O.sigmoid(x.astype(np.float64) @ w + b)
    check(p, p_ref, 1e-5, "mask out fwd")
    ids = np.array([2, 0, 1, 0, 3, 0], np.int32)
    tm = (rng.random((NR, 28, 28)) > 0.5).astype(np.float32)
    pr = p_ref.reshape(1, NR, 28, 28, C).astype(np.float32)
    l_ref, dp_ref = O.mask_bce(tm[None], ids[None], pr, want_grad=True)
    dz_ref = (dp_ref.astype(np.float64) * pr * (1 - pr)).reshape(M, C)
    lo, dz = new(2), new(M, C)
    X.call("myolo_mask_bce", X.ptr(dt(tm)), X.ptr(dt(ids)), X.ptr(dt(pr.reshape(M, C))), 1.0, X.ptr(lo), X.ptr(dz), NR, 28, 28, C, *ws(), X.stream())
    assert abs(lo.cpu().numpy()[0] - float(l_ref)) < 1e-5 and lo.cpu().numpy()[1] == 3
    assert np.abs(dz.cpu().numpy() - dz_ref).max() <= 1e-3 * np.abs(dz_ref).max()
    dx, dw, db = new(M, Cin), new(Cin, C), new(C)
    dzn = dz_ref.astype(np.float32)
    X.call("myolo_mask_head_out_bwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(dt(dzn)), X.ptr(dx), X.ptr(dw), X.ptr(db), M, Cin, C, *ws(), X.stream())
    rdx = (dzn.astype(np.float64) @ w.T) * (x > 0)
    assert np.abs(dx.cpu().numpy() - rdx).max() <= 1e-3 * np.abs(rdx).max()
    rdw = x.astype(np.float64).T @ dzn
    assert np.abs(dw.cpu().numpy() - rdw).max() <= 1e-3 * np.abs(rdw).max()
    assert np.abs(db.cpu().numpy() - dzn.sum(0)).max() <= 1e-3 * np.abs(dzn.sum(0)).max()


def test_mask_bce_no_positives_is_zero():
    NR, C = 3, 4
    tm, ids = np.zeros((NR, 28, 28), np.float32), np.zeros(NR, np.int32)
    pr = np.full((NR * 784, C), 0.3, np.float32)
    lo, dz = new(2), new(NR * 784, C)
    X.call("myolo_mask_bce", X.ptr(dt(tm)), X.ptr(dt(ids)), X.ptr(dt(pr)), 1.0, X.ptr(lo), X.ptr(dz), NR, 28, 28, C, *ws(), X.stream())
    assert lo.cpu().numpy()[0] == 0 and float(dz.abs().max()) == 0


def test_adam_three_steps():
    rng = np.random.default_rng(12)
    n = 10007
    p0, gs = rnd(rng, n), [rnd(rng, n) for _ in range(3)]
    p, m, v = dt(p0.copy()), dt(np.zeros(n, np.float32)), dt(np.zeros(n, np.float32))
    rp, rm, rv = p0.copy(), np.zeros(n, np.float32), np.zeros(n, np.float32)
    for t, g in enumerate(gs, 1):
        lr_t = float(1e-3 * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t))
        X.call("myolo_adam_step", X.ptr(p), X.ptr(dt(g)), X.ptr(m), X.ptr(v), n, lr_t, 0.9, 0.999, 1e-8, 1.0, X.stream())
        rp, rm, rv = O.adam_step(rp, g, rm, rv, t)
    check(p, rp, 1e-6, "adam")


def test_bad_arguments_fail_loudly():
    with pytest.raises(RuntimeError):
        X.call("myolo_dwconv3x3_fwd", None, None, None, 1, 8, 8, 8, 1, X.stream())
    with pytest.raises(RuntimeError):       # workspace too small
        x = dt(np.zeros((64, 64), np.float32))
        X.call("myolo_pwconv1x1_bwd_data", X.ptr(x), X.ptr(x), X.ptr(x), 64, 64, 64, None, 0, X.stream())


def test_frozen_bn_backward_from_post_activation():
    """bn2-4 of the mask head are frozen affine maps + ReLU (model.py:696,702,708): their backward read off the POST-activation
    tensor (mask = a > 0, xhat = (a - beta) / gamma) equals the oracle's backward from the pre-BN tensor."""
    rng = np.random.default_rng(21)
    M, C = 5000, 64
    x = rnd(rng, M, C)
    g, b = (1 + 0.2 * rnd(rng, C)), rnd(rng, C, scale=0.3)
    g[3] = -0.7                                                     # a negative gamma: xhat's sign must follow it
    mm, mv = rnd(rng, C, scale=0.2), (0.5 + rng.random(C)).astype(np.float32)
    da = rnd(rng, M, C)
    y, cache = O.bn_infer(x, g, b, mm, mv)
    a = O.relu(y)
    rdx, rdg, rdb = O.bn_infer_bwd(cache, g, O.relu_bwd(a, da))
    scale = (g / np.sqrt(mv.astype(np.float64) + 1e-3)).astype(np.float32)
    dx, dg, db = new(M, C), new(C), new(C)
    X.call("myolo_bn_act_bwd_frozen_post", X.ptr(dt(da)), X.ptr(dt(a)), X.ptr(dt(g)), X.ptr(dt(b)), X.ptr(dt(scale)), X.ptr(dx), X.ptr(dg),
           X.ptr(db), M, C, 1, *ws(), X.stream())
    check(dx, rdx, 1e-5, "frozen bn dx from post-activation")
    check(dg, rdg, 1e-4, "frozen bn dgamma from post-activation")
    check(db, rdb, 1e-4, "frozen bn dbeta from post-activation")


def test_wino_multiply_bf16x6_accuracy():
    """cfg.FP32_MATMUL='bf16x6' (library switch wino_x6, csrc/wino_mm.hip): the multiply stage V[q] * U[q] with every fp32 operand
    split exactly into three bf16 pieces and six piece products per fp32 product on the bf16 matrix pipe, against an fp64
    reference of the SAME fp32 operands -- beside the native fp32 MFMA kernel.  The split path must not be less accurate."""
    NR, HW, C = 24, 16, 256                      # 16x16 maps: every plane has NR*16 rows (no reduced tiles), one launch of 36 GEMMs
    T = NR * 16
    gen = torch.Generator(device="cpu").manual_seed(11)
    V = (torch.randn(36, T, C, generator=gen) * 3.0).to(DEV)
    V[0, :8, :8] = torch.tensor(float.fromhex("0x1.fffffep+0"))            # all 24 significand bits set: the split must be exact
    w = (torch.randn(3, 3, C, C, generator=gen) * 0.05).to(DEV)
    st = X.stream()
    with X.option("wino_x6", 0), X.option("wino_no_bt", 1):               # [36][K][N] fp32: the operand values themselves
        Unat = torch.empty(X.wino_u_elems(C, C), device=DEV)
        X.call("myolo_wino_weight_transform", X.ptr(w), X.ptr(Unat), C, C, 0, st)
    ref = torch.bmm(V.double(), Unat[:36 * C * C].view(36, C, C).double())
    scale = float(ref.abs().max())
    err = {}
    for x6 in (0, 1):
        with X.option("wino_x6", x6):
            U = torch.empty(X.wino_u_elems(C, C), device=DEV)
            M = torch.full((36, T, C), float("nan"), device=DEV)
            X.call("myolo_wino_weight_transform", X.ptr(w), X.ptr(U), C, C, 0, st)
            X.call("myolo_wino_multiply", X.ptr(V), X.ptr(U), X.ptr(M), NR, HW, HW, C, C, st)
        e = (M.double() - ref).abs()
        err[x6] = (float(e.max()) / scale, float((e ** 2).mean().sqrt()) / scale)
    print("multiply stage vs fp64, relative to max|ref|: native fp32 MFMA max %.3e rms %.3e | bf16x6 max %.3e rms %.3e" % (err[0] + err[1]))
    assert err[0][0] < 2e-6 and err[1][0] < 2e-6                           # both are fp32-level (K = 256 accumulation)
    assert err[1][1] <= 1.05 * err[0][1] and err[1][0] <= 1.25 * err[0][0], err




def _matmul(A, B, products, b_is_nk=1):
    """C = A B^T (b_is_nk=1: B is [N][K]) or A B (b_is_nk=0: [K][N]) through myolo_matmul_f32 with the products formed as asked."""
    M, K = A.shape
    N = B.shape[0] if b_is_nk else B.shape[1]
    C = torch.full((M, N), float("nan"), device=DEV)
    wsb = torch.empty(max(256, X.matmul_ws_bytes(K, N, b_is_nk, products)), dtype=torch.uint8, device=DEV)
    X.call("myolo_matmul_f32", X.ptr(A), X.ptr(B), X.ptr(C), M, K, N, b_is_nk, products, wsb.data_ptr(), wsb.numel(), X.stream())
    torch.cuda.synchronize()
    return C


@pytest.mark.parametrize("K", [256, 512, 2304])
@pytest.mark.parametrize("b_is_nk", [1, 0])
def test_bf16x6_matmul_accuracy_up_to_feature_map_depth(K, b_is_nk):
    """error of the six-piece bf16 product against an fp64 product of the SAME fp32 operands, beside the
    native fp32 MFMA kernel, at K = 256 (mask head), 512 (feature_map's Winograd multiply) and 2304 (its direct form, 9 x 256)."""
    gen = torch.Generator(device="cpu").manual_seed(100 + K)
    M, N = 300, 256
    A = (torch.randn(M, K, generator=gen) * 2.0).to(DEV)
    Bnk = (torch.randn(N, K, generator=gen) * 0.05).to(DEV)
    B = Bnk if b_is_nk else Bnk.t().contiguous()
    ref = A.double() @ Bnk.double().t()
    scale = float(ref.abs().max())
    e = {}
    for prod in (X.PRODUCTS_NATIVE, X.PRODUCTS_BF16X6):
        C = _matmul(A, B, prod, b_is_nk)
        d = (C.double() - ref).abs()
        e[prod] = (float(d.max()) / scale, float((d ** 2).mean().sqrt()) / scale)
    print("K=%d vs fp64, relative to max|ref|: native max %.3e rms %.3e | bf16x6 max %.3e rms %.3e" % ((K,) + e[0] + e[1]))
    bound = 2e-6 * (K / 256.0) ** 0.5                       # fp32 accumulation noise grows like sqrt(K)
    assert e[0][0] < bound and e[1][0] < bound, e
    assert e[1][1] <= 1.05 * e[0][1] and e[1][0] <= 1.25 * e[0][0], e          # the split path is not less accurate than fp32 MFMA


def test_bf16x6_matmul_adversarial_magnitudes_and_cancellation():
    """Operands spanning 2^-20 .. 2^20 inside ONE dot product, rows that cancel exactly, all-zero rows and columns.  The bound is the
    fp32 one: |err| <= c * K * 2^-24 * sum_k |a_k b_k| per output (what any fp32-accumulating kernel satisfies), checked for both paths;
    exact-cancellation rows must come out as the exact 0 when the cancelling terms are adjacent pairs inside one 16-deep chunk... they
    are not guaranteed to (accumulation order differs between the two pipes), so the check is the bound, which is ~0 there."""
    rng = np.random.default_rng(5)
    M, K, N = 256, 512, 256
    mag = 2.0 ** rng.integers(-20, 21, size=(M, K))
    A = (rng.standard_normal((M, K)) * mag).astype(np.float32)
    B = (rng.standard_normal((N, K)) * 2.0 ** rng.integers(-20, 21, size=(N, K))).astype(np.float32)
    A[7] = 0.0                                               # all-zero row
    B[11] = 0.0                                              # all-zero column
    A[9, 1::2] = -A[9, 0::2]                                 # row 9 x column 13 cancels pairwise: sum = 0 exactly in exact arithmetic
    B[13, 1::2] = B[13, 0::2]
    A[21] = np.float32(np.float32(1.0) + np.float32(2.0 ** -23))        # all 24 significand bits matter: 1 + ulp
    B[23] = np.float32(np.float32(1.0) - np.float32(2.0 ** -24))
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    absdot = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    At, Bt = dt(A), dt(B)
    for prod in (X.PRODUCTS_NATIVE, X.PRODUCTS_BF16X6):
        C = _matmul(At, Bt, prod).cpu().numpy().astype(np.float64)
        assert np.isfinite(C).all()
        err = np.abs(C - ref)
        bound = 4.0 * K * 2.0 ** -24 * absdot + 1e-300
        assert (err <= bound).all(), (prod, float((err / bound).max()))
        assert (C[7] == 0).all() and (C[:, 11] == 0).all()                   # zero rows / columns stay exactly zero
        assert abs(C[9, 13]) <= bound[9, 13]
        # (1 + 2^-23) * (1 - 2^-24) summed K times: the third bf16 piece of both operands is needed to get this to fp32 accuracy
        assert abs(C[21, 23] - ref[21, 23]) <= K * 2.0 ** -24 * 4


def test_bf16x6_matmul_special_values():
    """VERDICT r2 item 1(b).  What the six-piece product does with values outside the normal range, against the native kernel:
      * NaN operand  -> NaN in every output that reads it, in both (identical propagation);
      * Inf operand  -> the native kernel gives +-Inf (NaN for Inf * 0); bf16x6 gives NaN: the split forms Inf - Inf for the lower
                        pieces.  Both are non-finite in exactly the same outputs; outputs that do not read the Inf are untouched;
      * operands < 2^-110: the third piece (2^-16 relative) falls below the smallest normal bf16 / fp32 and is flushed -> the product
                        keeps >= 16 significant bits relative to |a b| instead of 24; absolute error < K * 2^-126 * max|b| (nothing a
                        1e-3 activation bound can see).  Measured and asserted below;
      * operands up to 2^120 with finite products -> finite and fp32-accurate (truncated pieces never exceed the operand)."""
    M, K, N = 128, 256, 256
    rng = np.random.default_rng(9)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = rng.standard_normal((N, K)).astype(np.float32) * np.float32(0.1)
    A[3, 17] = np.nan
    A[5, 40] = np.inf
    A[6, 41] = -np.inf
    B[8, 99] = np.nan
    out = {}
    for prod in (X.PRODUCTS_NATIVE, X.PRODUCTS_BF16X6):
        out[prod] = _matmul(dt(A), dt(B), prod).cpu().numpy()
    nat, x6 = out[X.PRODUCTS_NATIVE], out[X.PRODUCTS_BF16X6]
    assert np.isnan(nat[3]).all() and np.isnan(x6[3]).all() and np.isnan(nat[:, 8]).all() and np.isnan(x6[:, 8]).all()
    assert np.isinf(np.delete(nat[5], 8)).all() and np.isinf(np.delete(nat[6], 8)).all()          # native: +-Inf
    assert np.isnan(x6[5]).all() and np.isnan(x6[6]).all()                                          # bf16x6: NaN (documented difference)
    assert np.array_equal(np.isfinite(nat), np.isfinite(x6))                                        # the SAME outputs are non-finite
    fin = np.isfinite(nat)
    ref = np.where(np.isfinite(A), A, 0).astype(np.float64) @ np.where(np.isfinite(B), B, 0).astype(np.float64).T
    assert np.abs(x6[fin] - ref[fin]).max() < 2e-5 and np.abs(nat[fin] - ref[fin]).max() < 2e-5
    # ---- tiny operands (third piece subnormal) against normal-range partners
    At = (rng.standard_normal((M, K)) * 2.0 ** -115).astype(np.float32)
    Bn = rng.standard_normal((N, K)).astype(np.float32)
    ref = At.astype(np.float64) @ Bn.astype(np.float64).T
    for prod, bits in ((X.PRODUCTS_NATIVE, 20), (X.PRODUCTS_BF16X6, 12)):
        C = _matmul(dt(At), dt(Bn), prod).cpu().numpy().astype(np.float64)
        assert np.isfinite(C).all()
        rel = np.abs(C - ref).max() / np.abs(ref).max()
        print("operands ~2^-115: products %s relative error %.3e" % ("native" if prod == 0 else "bf16x6", rel))
        assert rel < 2.0 ** -bits, (prod, rel)
        assert np.abs(C - ref).max() < K * 2.0 ** -126 * 8
    # ---- huge operands with finite products
    Ah = (rng.standard_normal((M, K)) * 2.0 ** 118).astype(np.float32)
    Bs = (rng.standard_normal((N, K)) * 2.0 ** -4).astype(np.float32)
    ref = Ah.astype(np.float64) @ Bs.astype(np.float64).T
    assert np.abs(ref).max() < 3e38
    for prod in (X.PRODUCTS_NATIVE, X.PRODUCTS_BF16X6):
        C = _matmul(dt(Ah), dt(Bs), prod).cpu().numpy().astype(np.float64)
        assert np.isfinite(C).all(), prod
        assert np.abs(C - ref).max() / np.abs(ref).max() < 2e-6, prod


@pytest.mark.parametrize("x6", [0, 1])
@pytest.mark.parametrize("N,Cin,Cout", [(3, 256, 256), (37, 64, 256), (2, 128, 512)])
def test_winograd_f63_tiling(N, Cin, Cout, x6, request):
    """csrc/wino63_kernels.hip: 14 = 6+4+4, one F(6,3) and two F(4,3) tiles per direction sharing one set of 64 transformed filters
    (400 point-tiles per image).  (a) input transform (with a folded affine + ReLU) -> multiply -> output transform == the float64
    oracle convolution at the suite's 1e-3 bound and within 3x the F(4,3) tiling's own error; (b) the one-kernel layer boundary ==
    output transform followed by input transform, and writes the activation for flagged images only."""
    opt = X.option("wino_x6", x6)
    opt.__enter__()
    request.addfinalizer(lambda: opt.__exit__(None, None, None))
    assert X.wino63_ok(14, 14, Cin, Cout) and not X.wino63_ok(16, 16, Cin, Cout) and not X.wino63_ok(14, 14, Cin, 128)
    rng = np.random.default_rng(9)
    H = W = 14
    xin = rnd(rng, N, H, W, Cin)
    sc, sh = (1 + 0.1 * rnd(rng, Cin)), rnd(rng, Cin, scale=0.1)
    w, b = rnd(rng, 3, 3, Cin, Cout, scale=0.05), rnd(rng, Cout)
    st = X.stream()
    V = torch.full((X.wino63_plane_elems(N, Cin),), float("nan"), device=DEV)
    U = torch.empty(X.wino63_u_elems(Cin, Cout), device=DEV)
    M = torch.full((X.wino63_plane_elems(N, Cout),), float("nan"), device=DEV)
    ya = new(N, H, W, Cin)
    xin_t, sc_t, sh_t, w_t, b_t = dt(xin), dt(sc), dt(sh), dt(w), dt(b)
    X.call("myolo_wino63_input_transform", X.ptr(xin_t), X.ptr(sc_t), X.ptr(sh_t), 1, X.ptr(ya), None, X.ptr(V), N, Cin, st)
    X.call("myolo_wino63_weight_transform", X.ptr(w_t), X.ptr(U), Cin, Cout, st)
    X.call("myolo_wino63_multiply", X.ptr(V), X.ptr(U), X.ptr(M), N, Cin, Cout, st)
    y = new(N, H, W, Cout)
    X.call("myolo_wino63_output_transform", X.ptr(M), X.ptr(b_t), None, None, X.ptr(y), N, Cout, 0, st)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(V).any()) and not bool(torch.isnan(M).any()), "every plane row must be written"
    a0 = ya.cpu().numpy()                                                       # what the input transform formed on load (one fma)
    assert np.abs(a0 - np.maximum(xin * sc + sh, 0)).max() < 1e-6
    ref = O.conv2d(a0, w, pads=(1, 1, 1, 1), bias=b, acc=np.float64)
    check(y, ref, what="F(6,3)/F(4,3) conv")
    e63 = float(np.abs(y.cpu().numpy() - ref).max() / np.abs(ref).max())
    wsb = torch.empty(X.wino_ws_bytes(N, H, W, Cin, Cout, 0), dtype=torch.uint8, device=DEV)
    y43 = new(N, H, W, Cout)
    a0_t = dt(a0)
    X.call("myolo_conv3x3_wino_fwd", X.ptr(a0_t), X.ptr(w_t), X.ptr(b_t), None, None, X.ptr(y43), N, H, W, Cin, Cout, 0, None,
           wsb.data_ptr(), wsb.numel(), st)
    e43 = float(np.abs(y43.cpu().numpy() - ref).max() / np.abs(ref).max())
    print("max error / max|y|: F(6,3)/F(4,3) tiling %.3e, F(4,3)/F(2,3) tiling %.3e" % (e63, e43))
    assert e63 < 5e-5 and e63 < 3.0 * e43 + 2e-6
    # (b) layer boundary: M -> relu((A^T m A + bias) * s2 + t2) -> V2, in one kernel == output transform, then input transform
    if Cout % 64 == 0 and X.wino63_ok(14, 14, Cout, 256):
        s2, t2 = dt(1 + 0.1 * rnd(rng, Cout)), dt(rnd(rng, Cout, scale=0.1))
        flags = torch.zeros(N, dtype=torch.int32, device=DEV)
        flags[::2] = 1
        yk = torch.full((N, H, W, Cout), float("nan"), device=DEV)
        V2 = torch.full((X.wino63_plane_elems(N, Cout),), float("nan"), device=DEV)
        X.call("myolo_wino63_output_input_transform", X.ptr(M), X.ptr(b_t), X.ptr(s2), X.ptr(t2), X.ptr(yk), X.ptr(flags), X.ptr(V2), N, Cout, 1, st)
        y2 = new(N, H, W, Cout)
        X.call("myolo_wino63_output_transform", X.ptr(M), X.ptr(b_t), X.ptr(s2), X.ptr(t2), X.ptr(y2), N, Cout, 1, st)
        V3 = torch.full((X.wino63_plane_elems(N, Cout),), float("nan"), device=DEV)
        X.call("myolo_wino63_input_transform", X.ptr(y2), None, None, 0, None, None, X.ptr(V3), N, Cout, st)
        torch.cuda.synchronize()
        assert torch.equal(V2, V3), "fused boundary differs from output transform + input transform"
        assert torch.equal(yk[::2], y2[::2]) and bool(torch.isnan(yk[1::2]).all()), "activation must be written for flagged images only"


@pytest.mark.parametrize("N,C", [(5, 256), (3, 64), (333, 256), (700, 64), (70, 512), (40, 320)])      # the last two: several units per persistent workgroup (prefetch across units, both LDS buffers)
def test_wino63_boundary_packed_equals_legacy(N, C):
    """Round 6's boundary kernel (two columns / rows per v_pk_* instruction, wave-uniform plane addresses, all of a wave's pixels
    requested at once) against the round-5 kernel kept behind option "w63_legacy": every front x back combination the library
    instantiates, every activation, the keep / slot / pre-BatchNorm / statistics outputs -- bit for bit, NaN patterns included."""
    rng = np.random.default_rng(61)
    st = X.stream()
    pe = X.wino63_plane_elems(N, C)
    Mp = dt(rnd(rng, pe))
    x = dt(rnd(rng, N, 14, 14, C))
    b, sc, sh = dt(rnd(rng, C)), dt(1 + 0.1 * rnd(rng, C)), dt(rnd(rng, C, scale=0.1))
    flags = torch.zeros(N, dtype=torch.int32, device=DEV)
    flags[::2] = 1
    slots = torch.full((N,), -1, dtype=torch.int32, device=DEV)
    slots[1], slots[N - 1] = 1, 0
    cap = 2
    npos = 2
    inv = torch.full((N,), -1, dtype=torch.int32, device=DEV)
    inv[0], inv[N - 2] = 1, 0
    dyc = dt(rnd(rng, npos, 196, C))
    ka, kb = dt(rnd(rng, C, scale=0.01)), dt(rnd(rng, C, scale=0.01))
    B, FH, FW, nb = 2, 28, 28, N
    feat, boxes = dt(rnd(rng, B, FH, FW, C)), _boxes(rng, nb)
    boxes[0] = [-0.2, 0.1, 0.7, 1.3]
    boxes_t, bind = dt(boxes), dt(rng.integers(0, B, nb).astype(np.int32))
    gamma, beta = dt(1 + 0.1 * rnd(rng, C)), dt(rnd(rng, C, scale=0.1))
    nan = lambda *s: torch.full(s, float("nan"), device=DEV)       # noqa: E731

    def run():
        out = {}
        for act in (0, 1, 2):
            V, y = nan(pe), nan(N, 14, 14, C)
            X.call("myolo_wino63_output_input_transform", X.ptr(Mp), X.ptr(b), X.ptr(sc), X.ptr(sh), X.ptr(y), X.ptr(flags), X.ptr(V), N, C, act, st)
            out["M->V act%d" % act] = (V, y)
            y2 = nan(N, 14, 14, C)
            X.call("myolo_wino63_output_transform", X.ptr(Mp), X.ptr(b), X.ptr(sc), X.ptr(sh), X.ptr(y2), N, C, act, st)
            out["M->y act%d" % act] = (y2,)
            V3, y3 = nan(pe), nan(N, 14, 14, C)
            X.call("myolo_wino63_input_transform", X.ptr(x), X.ptr(sc), X.ptr(sh), act, X.ptr(y3), X.ptr(flags), X.ptr(V3), N, C, st)
            out["x->V act%d" % act] = (V3, y3)
            V4, Q4 = nan(pe), nan(pe)
            X.call("myolo_wino63_lazybn_transforms", X.ptr(x), X.ptr(dyc), X.ptr(inv), X.ptr(sc), X.ptr(sh), X.ptr(ka), X.ptr(kb), act, X.ptr(V4), X.ptr(Q4), N, C, st)
            out["lazy->VQ act%d" % act] = (V4, Q4)
        V, yp = nan(pe), nan(N, 14, 14, C)
        X.call("myolo_wino63_output_input_transform", X.ptr(Mp), None, None, None, None, None, X.ptr(V), N, C, 0, st)
        out["M->V plain"] = (V,)
        V = nan(pe)
        X.call("myolo_wino63_output_input_transform_keep_pre", X.ptr(Mp), X.ptr(b), X.ptr(sc), X.ptr(sh), X.ptr(yp), X.ptr(flags), X.ptr(V), N, C, 1, st)
        out["M->V keep_pre"] = (V, yp)
        V, ypc = nan(pe), nan(cap, 14, 14, C)
        X.call("myolo_wino63_output_input_transform_keep_pre_slots", X.ptr(Mp), X.ptr(b), X.ptr(sc), X.ptr(sh), X.ptr(ypc), X.ptr(slots), cap, X.ptr(V), N, C, 1, st)
        out["M->V keep_pre slots"] = (V, ypc)
        y, yp = nan(N, 14, 14, C), nan(N, 14, 14, C)
        X.call("myolo_wino63_output_transform_keep_pre", X.ptr(Mp), X.ptr(b), X.ptr(sc), X.ptr(sh), X.ptr(y), X.ptr(yp), X.ptr(flags), N, C, 1, st)
        out["M->y keep_pre"] = (y, yp)
        y, ypc = nan(N, 14, 14, C), nan(cap, 14, 14, C)
        X.call("myolo_wino63_output_transform_keep_pre_slots", X.ptr(Mp), X.ptr(b), X.ptr(sc), X.ptr(sh), X.ptr(y), X.ptr(ypc), X.ptr(slots), cap, N, C, 1, st)
        out["M->y keep_pre slots"] = (y, ypc)
        V, yc = nan(pe), nan(cap, 14, 14, C)
        X.call("myolo_wino63_input_transform_slots", X.ptr(x), X.ptr(sc), X.ptr(sh), 1, X.ptr(yc), X.ptr(slots), cap, X.ptr(V), N, C, st)
        out["x->V slots"] = (V, yc)
        V = nan(pe)
        X.call("myolo_wino63_input_transform_roialign", X.ptr(feat), X.ptr(boxes_t), X.ptr(bind), X.ptr(V), B, FH, FW, C, nb, st)
        out["roialign->V"] = (V,)
        y = nan(N, 14, 14, C)
        mean, var, s2, t2 = new(C), new(C), new(C), new(C)
        mm, mv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
        wsb = torch.empty(X.wino63_out_bn_ws_bytes(N, C), dtype=torch.uint8, device=DEV)
        X.call("myolo_wino63_output_transform_bn_stats", X.ptr(Mp), X.ptr(b), X.ptr(y), N, C, X.ptr(gamma), X.ptr(beta), X.ptr(mean), X.ptr(var), X.ptr(s2), X.ptr(t2),
               X.ptr(mm), X.ptr(mv), wsb.data_ptr(), wsb.numel(), st)
        out["M->y + statistics"] = (y, mean, var, s2, t2, mm, mv)
        if C == 256:
            dy, dw = dt(rnd(rng2, N, 14, 14, C)), new(3, 3, C, C)
            wsw = torch.empty(X.wino63_ws_bytes(N, C, C, 2), dtype=torch.uint8, device=DEV)
            X.call("myolo_conv3x3_wino63_bwd_weight", X.ptr(x), None, X.ptr(dy), X.ptr(dw), N, C, C, wsw.data_ptr(), wsw.numel(), st)   # FROM_ACT -> Q
            out["act->Q (weight gradient)"] = (dw,)
            dwl = new(3, 3, C, C)
            Vs = dt(rnd(rng2, pe))
            wsl = torch.empty(X.wino63_bwd_weight_ws_bytes(N, C, C), dtype=torch.uint8, device=DEV)
            X.call("myolo_wino63_bwd_weight_lazybn", X.ptr(Vs), X.ptr(x), X.ptr(dyc), X.ptr(inv), X.ptr(sc), X.ptr(sh), X.ptr(ka), X.ptr(kb), 1, X.ptr(dwl), N, C, C,
                   wsl.data_ptr(), wsl.numel(), st)                                                                                       # FROM_LAZY -> Q
            out["lazy->Q (weight gradient)"] = (dwl,)
        torch.cuda.synchronize()
        return out

    rng2 = np.random.default_rng(62)
    new_out = run()
    rng2 = np.random.default_rng(62)
    with X.option("w63_legacy", 1):
        old_out = run()
    bad = []
    for k in new_out:
        for i, (a, o) in enumerate(zip(new_out[k], old_out[k])):
            same = torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(o, nan=12345.0)) and torch.equal(torch.isnan(a), torch.isnan(o))
            if not same:
                d = (torch.nan_to_num(a) - torch.nan_to_num(o)).abs()
                bad.append((k, i, float(d.max()), float((d > 0).float().mean()), float(torch.nan_to_num(o).abs().max())))
        assert not bool(torch.isnan(new_out[k][0]).any()), k
    assert not bad, bad


def test_winograd_f63_conv1_pieces():
    """conv1 of the mask head on the F(6,3)/F(4,3) tiling: (a) ROIAlign fused into the input transform == crop_and_resize followed by
    the plain input transform, bit for bit (same sampling expressions, same transform code); (b) the output transform that also
    yields the training-mode BatchNorm statistics == plain output transform + myolo_bn_stats; (c) the weight gradient from the kept V
    planes and a lazily formed output gradient == the F(4,3) kernel's on the same operands; (d) likewise the data gradient."""
    rng = np.random.default_rng(12)
    B, FH, FW, C, nb, Co = 2, 28, 28, 256, 21, 256
    st = X.stream()
    img, boxes = rnd(rng, B, FH, FW, C), _boxes(rng, nb)
    boxes[0] = [-0.2, 0.1, 0.7, 1.3]
    bind = rng.integers(0, B, nb).astype(np.int32)
    img_t, boxes_t, bind_t = dt(img), dt(boxes), dt(bind)
    a = (X.ptr(img_t), X.ptr(boxes_t), X.ptr(bind_t))
    x = new(nb, 14, 14, C)
    n63 = X.wino63_plane_elems(nb, C)
    V1, V2 = torch.full((n63,), float("nan"), device=DEV), torch.full((n63,), float("nan"), device=DEV)
    X.call("myolo_crop_and_resize_fwd", *a, X.ptr(x), B, FH, FW, C, nb, 14, 14, st)
    X.call("myolo_wino63_input_transform", X.ptr(x), None, None, 0, None, None, X.ptr(V1), nb, C, st)
    X.call("myolo_wino63_input_transform_roialign", *a, X.ptr(V2), B, FH, FW, C, nb, st)
    torch.cuda.synchronize()
    assert not bool(torch.isnan(V2).any()) and float((V1 - V2).abs().max()) <= 1e-5 * float(V1.abs().max())
    # (b) conv + BN statistics
    w, b = rnd(rng, 3, 3, C, Co, scale=0.05), rnd(rng, Co)
    w_t, b_t = dt(w), dt(b)
    U = torch.empty(X.wino63_u_elems(C, Co), device=DEV)
    M = torch.empty(X.wino63_plane_elems(nb, Co), device=DEV)
    X.call("myolo_wino63_weight_transform", X.ptr(w_t), X.ptr(U), C, Co, st)
    X.call("myolo_wino63_multiply", X.ptr(V2), X.ptr(U), X.ptr(M), nb, C, Co, st)
    gamma, beta = dt(1 + 0.1 * rnd(rng, Co)), dt(rnd(rng, Co, scale=0.1))
    outs = []
    for fused in (True, False):
        y = new(nb, 14, 14, Co)
        mean, var, sc, sh = new(Co), new(Co), new(Co), new(Co)
        mm, mv = torch.zeros(Co, device=DEV), torch.ones(Co, device=DEV)
        if fused:
            wsb = torch.empty(X.wino63_out_bn_ws_bytes(nb, Co), dtype=torch.uint8, device=DEV)
            X.call("myolo_wino63_output_transform_bn_stats", X.ptr(M), X.ptr(b_t), X.ptr(y), nb, Co, X.ptr(gamma), X.ptr(beta), X.ptr(mean),
                   X.ptr(var), X.ptr(sc), X.ptr(sh), X.ptr(mm), X.ptr(mv), wsb.data_ptr(), wsb.numel(), st)
        else:
            X.call("myolo_wino63_output_transform", X.ptr(M), X.ptr(b_t), None, None, X.ptr(y), nb, Co, 0, st)
            X.call("myolo_bn_stats", X.ptr(y), X.ptr(gamma), X.ptr(beta), X.ptr(mean), X.ptr(var), X.ptr(sc), X.ptr(sh), X.ptr(mm), X.ptr(mv),
                   nb * 196, Co, *ws(), st)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (y, mean, var, sc, sh, mm, mv)])
    assert torch.equal(outs[0][0], outs[1][0])
    for t0, t1 in zip(outs[0][1:], outs[1][1:]):
        assert float((t0 - t1).abs().max()) <= 2e-6 * max(1.0, float(t1.abs().max()))
    y_pre, sc, sh = outs[0][0], outs[0][3], outs[0][4]
    # (c), (d): gradients behind that BN with a row-sparse upstream gradient, against the F(4,3) kernels on the same operands
    npos = 5
    inv = np.full(nb, -1, np.int32)
    inv[[1, 4, 7, 8, 20]] = np.arange(npos)
    inv_t = dt(inv)
    dyc = dt(rnd(rng, npos, 196, Co))
    ka, kb = dt(rnd(rng, Co, scale=0.01)), dt(rnd(rng, Co, scale=0.01))
    lazy = (X.ptr(y_pre), X.ptr(dyc), X.ptr(inv_t), X.ptr(sc), X.ptr(sh), X.ptr(ka), X.ptr(kb), 1)
    T = nb * 16
    V43 = new(36, T, C)
    X.call("myolo_wino_input_transform_roialign", *a, X.ptr(V43), B, FH, FW, C, nb, 14, 14, st)
    wsz = max(X.wino_ws_bytes(nb, 14, 14, C, Co, k) for k in (1, 2)) + X.wino63_bwd_weight_ws_bytes(nb, C, Co) + X.wino63_bwd_data_ws_bytes(nb, C, Co)
    wsb = torch.empty(wsz, dtype=torch.uint8, device=DEV)
    wsa = (wsb.data_ptr(), wsb.numel())
    dw43, dw63, dx43, dx63 = new(3, 3, C, Co), new(3, 3, C, Co), new(nb, 14, 14, C), new(nb, 14, 14, C)
    X.call("myolo_conv3x3_wino_bwd_weight_lazybn", X.ptr(V43), *lazy, X.ptr(dw43), nb, 14, 14, C, Co, *wsa, st)
    X.call("myolo_wino63_bwd_weight_lazybn", X.ptr(V2), *lazy, X.ptr(dw63), nb, C, Co, *wsa, st)
    X.call("myolo_conv3x3_wino_bwd_data_lazybn", *lazy, X.ptr(w_t), X.ptr(dx43), nb, 14, 14, C, Co, *wsa, st)
    X.call("myolo_wino63_bwd_data_lazybn", *lazy, X.ptr(w_t), X.ptr(dx63), nb, C, Co, *wsa, st)
    torch.cuda.synchronize()
    for got, want, what in ((dw63, dw43, "dw"), (dx63, dx43, "dx")):
        err = float((got - want).abs().max()) / float(want.abs().max())
        assert err < 1e-4, (what, err)
    # (e) the merged form the engine uses: ONE kernel sends the lazily formed gradient out as V and as Q, the two gradients finish separately
    pe = X.wino63_plane_elems(nb, Co)
    Vd, Qd = new(pe), new(pe)
    dwm, dxm = new(3, 3, C, Co), new(nb, 14, 14, C)
    X.call("myolo_wino63_lazybn_transforms", *lazy, X.ptr(Vd), X.ptr(Qd), nb, Co, st)
    X.call("myolo_wino63_bwd_weight_from_q", X.ptr(V2), X.ptr(Qd), X.ptr(dwm), nb, C, Co, *wsa, st)
    X.call("myolo_wino63_bwd_data_from_v", X.ptr(Vd), X.ptr(w_t), X.ptr(dxm), nb, C, Co, *wsa, st)
    torch.cuda.synchronize()
    assert X.wino63_bwd_weight_from_q_ws_bytes(nb, C, Co) <= wsb.numel() and X.wino63_bwd_data_from_v_ws_bytes(nb, C, Co) <= wsb.numel()
    assert torch.equal(dwm, dw63) and torch.equal(dxm, dx63)


@pytest.mark.parametrize("x6", [0, 1])
@pytest.mark.parametrize("N,Cin,Cout", [(11, 256, 256), (3, 64, 256), (40, 256, 256), (5, 256, 512)])
def test_winograd_f63_conv_operators(N, Cin, Cout, x6, request):
    """myolo_conv3x3_wino63_{fwd,bwd_data,bwd_weight}: the three operators as single calls on the F(6,3)/F(4,3) tiling, against the
    float64 oracle at the suite's 1e-3 bound (the compacted mask-head backward at realistic positive counts runs through them).
    x6 = 1: six exact bf16 piece products per fp32 product -- the multiply (wino_mm_x6_kernel) and, for channel counts that are
    multiples of 256, the weight gradient's dU = V^T Q (wino_tn_x6_kernel, one launch over the 64 planes + a fixed-order reduce)."""
    old = X.set_option("wino_x6", x6)
    request.addfinalizer(lambda: X.set_option("wino_x6", old))
    rng = np.random.default_rng(21)
    H = W = 14
    x, w, b, dy = rnd(rng, N, H, W, Cin), rnd(rng, 3, 3, Cin, Cout, scale=0.05), rnd(rng, Cout), rnd(rng, N, H, W, Cout)
    wsb = torch.empty(max(X.wino63_ws_bytes(N, Cin, Cout, k) for k in (0, 1, 2)), dtype=torch.uint8, device=DEV)
    wsa = (wsb.data_ptr(), wsb.numel())
    st = X.stream()
    x_t, w_t, b_t, dy_t = dt(x), dt(w), dt(b), dt(dy)
    y, vk = new(N, H, W, Cout), torch.empty(X.wino63_plane_elems(N, Cin), device=DEV)
    X.call("myolo_conv3x3_wino63_fwd", X.ptr(x_t), X.ptr(w_t), X.ptr(b_t), None, None, X.ptr(y), N, Cin, Cout, 0, X.ptr(vk), *wsa, st)
    ref = O.conv2d(x, w, pads=(1, 1, 1, 1), bias=b, acc=np.float64)
    check(y, ref, what="wino63 fwd")
    rdx, rdw, rdb = O.conv2d_bwd(x, w, dy, pads=(1, 1, 1, 1), acc=np.float64)
    dw, dw2 = new(3, 3, Cin, Cout), new(3, 3, Cin, Cout)
    X.call("myolo_conv3x3_wino63_bwd_weight", None, X.ptr(vk), X.ptr(dy_t), X.ptr(dw), N, Cin, Cout, *wsa, st)
    check(dw, rdw, what="wino63 dw (saved V)")
    X.call("myolo_conv3x3_wino63_bwd_weight", X.ptr(x_t), None, X.ptr(dy_t), X.ptr(dw2), N, Cin, Cout, *wsa, st)
    check(dw2, rdw, what="wino63 dw (from x)")
    if x6:          # the split-product weight gradient against the fp32-MFMA one on the same operands: fp32-level agreement, and reproducible
        dw3, dw4 = new(3, 3, Cin, Cout), new(3, 3, Cin, Cout)
        with X.option("tn_no_x6", 1):
            X.call("myolo_conv3x3_wino63_bwd_weight", None, X.ptr(vk), X.ptr(dy_t), X.ptr(dw3), N, Cin, Cout, *wsa, st)
        X.call("myolo_conv3x3_wino63_bwd_weight", None, X.ptr(vk), X.ptr(dy_t), X.ptr(dw4), N, Cin, Cout, *wsa, st)
        torch.cuda.synchronize()
        assert float((dw - dw3).abs().max()) <= 2e-5 * float(dw3.abs().max())
        assert torch.equal(dw, dw4)
        if Cin % 256 == 0:          # option tn_wgs: a fixed number of workgroups walking the work units (the kernel's other instantiation; the split differs)
            dw5, dw6 = new(3, 3, Cin, Cout), new(3, 3, Cin, Cout)
            with X.option("tn_wgs", 24):
                X.call("myolo_conv3x3_wino63_bwd_weight", None, X.ptr(vk), X.ptr(dy_t), X.ptr(dw5), N, Cin, Cout, *wsa, st)
                X.call("myolo_conv3x3_wino63_bwd_weight", None, X.ptr(vk), X.ptr(dy_t), X.ptr(dw6), N, Cin, Cout, *wsa, st)
            torch.cuda.synchronize()
            check(dw5, rdw, what="wino63 dw, walking workgroups")
            assert torch.equal(dw5, dw6)
    if X.wino63_ok(14, 14, Cout, Cin):
        dx = new(N, H, W, Cin)
        X.call("myolo_conv3x3_wino63_bwd_data", X.ptr(dy_t), X.ptr(w_t), X.ptr(dx), N, Cin, Cout, *wsa, st)
        check(dx, rdx, what="wino63 dx")


@pytest.mark.parametrize("M,C,act", [(1000, 32, 2), (37, 1024, 1), (196 * 5, 256, 1), (64, 4, 0)])
def test_bn_frozen_apply_act_equals_coeffs_then_apply(M, C, act):
    """the one-launch frozen BatchNorm + activation == myolo_bn_frozen_coeffs followed by myolo_bn_apply_act, bit for bit"""
    rng = np.random.default_rng(5)
    x = dt(rnd(rng, M, C))
    gamma, beta, mm, mv = dt(1 + 0.1 * rnd(rng, C)), dt(rnd(rng, C, scale=0.1)), dt(rnd(rng, C, scale=0.2)), dt(np.abs(rnd(rng, C)) + 0.1)
    st = X.stream()
    sc1, sh1, y1 = new(C), new(C), new(M, C)
    X.call("myolo_bn_frozen_coeffs", X.ptr(gamma), X.ptr(beta), X.ptr(mm), X.ptr(mv), X.ptr(sc1), X.ptr(sh1), C, st)
    X.call("myolo_bn_apply_act", X.ptr(x), X.ptr(sc1), X.ptr(sh1), X.ptr(y1), M, C, act, st)
    sc2, sh2, y2 = new(C), new(C), new(M, C)
    X.call("myolo_bn_frozen_apply_act", X.ptr(x), X.ptr(gamma), X.ptr(beta), X.ptr(mm), X.ptr(mv), X.ptr(sc2), X.ptr(sh2), X.ptr(y2), M, C, act, st)
    torch.cuda.synchronize()
    assert torch.equal(sc1, sc2) and torch.equal(sh1, sh2) and torch.equal(y1, y2)


def _frozen_bn(rng, C):
    return dt(1 + 0.1 * rnd(rng, C)), dt(rnd(rng, C, scale=0.1)), dt(rnd(rng, C, scale=0.2)), dt(np.abs(rnd(rng, C)) + 0.1)


@pytest.mark.parametrize("M,Cin,Cout,act", [(4 * 26 * 26, 128, 256, 2),     # the MFMA fast path
                                            (2 * 7 * 7, 512, 1024, 2),      # few tiles, long K: split-K with the affine in its epilogue
                                            (300, 24, 40, 1), (77, 32, 64, 0),    # the generic kernel's shapes; ReLU; no activation
                                            (4 * 104 * 104 + 5, 64, 64, 2), (16384, 32, 64, 1)])   # conv_pw_2 / conv_pw_1 shapes: the register-fed thin kernel (ragged last block)
def test_pwconv1x1_affine_act_fwd_equals_conv_then_frozen_bn(M, Cin, Cout, act):
    """inference fold (model.py:68-76 with BatchNormalization in inference mode): the pointwise conv with the frozen BatchNorm's affine and
    the activation in its epilogue == myolo_pwconv1x1_fwd, then myolo_bn_apply_act, bit for bit; the coefficients come from the batched
    kernel (one launch for many layers), which must equal myolo_bn_frozen_coeffs"""
    rng = np.random.default_rng(8)
    x, w = dt(rnd(rng, M, Cin)), dt(rnd(rng, 1, 1, Cin, Cout, scale=0.1))
    gamma, beta, mm, mv = _frozen_bn(rng, Cout)
    ws = torch.empty(8 * M * Cout * 4 + 4096, dtype=torch.uint8, device=DEV)
    st = X.stream()
    y0, y1, y2, sc, sh = new(M, Cout), new(M, Cout), new(M, Cout), new(Cout), new(Cout)
    X.call("myolo_pwconv1x1_fwd", X.ptr(x), X.ptr(w), None, X.ptr(y0), M, Cin, Cout, ws.data_ptr(), ws.numel(), st)
    X.call("myolo_bn_frozen_coeffs", X.ptr(gamma), X.ptr(beta), X.ptr(mm), X.ptr(mv), X.ptr(sc), X.ptr(sh), Cout, st)
    X.call("myolo_bn_apply_act", X.ptr(y0), X.ptr(sc), X.ptr(sh), X.ptr(y1), M, Cout, act, st)
    # two "layers" in one batched launch: this one and a dummy of 8 channels in front of it
    params = torch.cat([torch.ones(8, device=DEV), torch.zeros(8, device=DEV), gamma, beta])
    stats = torch.cat([torch.zeros(8, device=DEV), torch.ones(8, device=DEV), mm, mv])
    table = torch.tensor([[0, 8, 0, 8, 0, 8], [16, 16 + Cout, 16, 16 + Cout, 16, Cout]], dtype=torch.int64, device=DEV)
    co = new(16 + 2 * Cout)
    X.call("myolo_bn_frozen_coeffs_batched", X.ptr(params), X.ptr(stats), table.data_ptr(), 2, X.ptr(co), st)
    X.call("myolo_pwconv1x1_affine_act_fwd", X.ptr(x), X.ptr(w), co.data_ptr() + 64, co.data_ptr() + 64 + 4 * Cout, act, X.ptr(y2), M, Cin, Cout,
           ws.data_ptr(), ws.numel(), st)
    torch.cuda.synchronize()
    assert torch.equal(co[16:16 + Cout], sc) and torch.equal(co[16 + Cout:], sh)
    assert torch.equal(y1, y2)
    g, b_, m_, v_ = (t.cpu().numpy().astype(np.float64) for t in (gamma, beta, mm, mv))
    ref = x.cpu().numpy().astype(np.float64) @ w.cpu().numpy().reshape(Cin, Cout).astype(np.float64)
    ref = ref * (g / np.sqrt(v_ + 1e-3)) + (b_ - m_ * g / np.sqrt(v_ + 1e-3))
    ref = np.clip(ref, 0, 6) if act == 2 else (np.maximum(ref, 0) if act == 1 else ref)
    assert np.abs(y2.cpu().numpy() - ref).max() < 2e-4


@pytest.mark.parametrize("M,Cin,Cout", [(4 * 26 * 26, 512, 512), (4 * 13 * 13, 512, 1024), (4 * 13 * 13, 1024, 1024), (1568, 1024, 1024), (1568, 512, 1024),
                                        (2705, 256, 128), (33, 288, 256), (5, 256, 128), (16384, 256, 128)])
def test_pwconv1x1_small_m_kernel(M, Cin, Cout):
    """pw_smallm_kernel (few rows, K >= 256, N % 128 == 0: a workgroup's four waves split K, partial tiles summed in LDS, one launch) through the
    four entry points that reach it -- plain / bias forward, the inference fold (bias-free, per-column affine + ReLU6), the training forward with the
    producing BatchNorm on its loads, the data gradient -- against float64, and against the split-K pair of launches it replaces (pw_no_smallm=1:
    same fp32 products, another summation order)"""
    rng = np.random.default_rng(31)
    x, w = rnd(rng, M, Cin), rnd(rng, Cin, Cout, scale=0.1)
    b = rnd(rng, Cout)
    sc, sh = 1 + rnd(rng, Cout, scale=0.2), rnd(rng, Cout, scale=0.5)
    isc, ish = 1 + rnd(rng, Cin, scale=0.3), rnd(rng, Cin, scale=0.5) + 1.0
    dy = rnd(rng, M, Cout)
    st = X.stream()
    xd, wd, bd, scd, shd, iscd, ishd, dyd = (dt(t) for t in (x, w, b, sc, sh, isc, ish, dy))
    g, be = dt(1 + rnd(rng, Cout, scale=0.2)), dt(rnd(rng, Cout, scale=0.3))
    wsb = torch.empty(max(X.pw_bnstats_ws_bytes(M, Cin, Cout), 64 << 20), dtype=torch.uint8, device=DEV)

    def run():
        y0, y1, y2, y3, dx = new(M, Cout), new(M, Cout), new(M, Cout), new(M, Cout), new(M, Cin)
        X.call("myolo_pwconv1x1_fwd", X.ptr(xd), X.ptr(wd), None, X.ptr(y0), M, Cin, Cout, wsb.data_ptr(), wsb.numel(), st)
        X.call("myolo_pwconv1x1_fwd", X.ptr(xd), X.ptr(wd), X.ptr(bd), X.ptr(y1), M, Cin, Cout, wsb.data_ptr(), wsb.numel(), st)
        X.call("myolo_pwconv1x1_affine_act_fwd", X.ptr(xd), X.ptr(wd), X.ptr(scd), X.ptr(shd), 2, X.ptr(y2), M, Cin, Cout, wsb.data_ptr(), wsb.numel(), st)
        mean, var, scale, shift, tmm, tmv = new(Cout), new(Cout), new(Cout), new(Cout), dt(np.zeros(Cout, np.float32)), dt(np.ones(Cout, np.float32))
        X.call("myolo_pwconv1x1_bnstats_fwd", X.ptr(xd), X.ptr(iscd), X.ptr(ishd), 2, X.ptr(wd), X.ptr(y3), X.ptr(g), X.ptr(be), X.ptr(mean), X.ptr(var),
               X.ptr(scale), X.ptr(shift), X.ptr(tmm), X.ptr(tmv), M, Cin, Cout, 3, wsb.data_ptr(), wsb.numel(), st)
        X.call("myolo_pwconv1x1_bwd_data", X.ptr(dyd), X.ptr(wd), X.ptr(dx), M, Cin, Cout, wsb.data_ptr(), wsb.numel(), st)
        torch.cuda.synchronize()
        return [t.cpu().numpy() for t in (y0, y1, y2, y3, dx, mean, var)]

    new_ = run()
    with X.option("pw_no_smallm", 1):
        old_ = run()
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    r0 = x64 @ w64
    r3 = np.clip(x64 * isc + ish, 0, 6) @ w64
    refs = [r0, r0 + b, np.clip(r0 * sc + sh, 0, 6), r3, dy.astype(np.float64) @ w64.T, r3.mean(0), r3.var(0)]
    for name, got, was, ref in zip(("fwd", "fwd + bias", "affine + relu6 fwd", "bnstats fwd", "dx", "batch mean", "batch variance"), new_, old_, refs):
        scale_ = np.abs(ref).max() + 1e-6
        assert np.abs(got - ref).max() <= 2e-5 * scale_ + 1e-6, (name, np.abs(got - ref).max(), scale_)
        assert np.abs(got - was).max() <= 2e-5 * scale_ + 1e-6, (name, "against the split-K pair")
    assert np.array_equal(new_[0] + 0, run()[0])            # run to run: the same bits


@pytest.mark.parametrize("N,H,W,Cout,act", [(2, 416, 416, 32, 2), (3, 64, 48, 16, 1), (1, 30, 26, 32, 2)])      # the row kernel (W % 4 == 0) twice, the generic one
def test_conv1_affine_act_fwd_equals_conv_then_frozen_bn(N, H, W, Cout, act):
    """conv_block in inference mode (model.py:42-52): myolo_conv3x3s2_c3_affine_act_fwd == myolo_conv3x3s2_c3_fwd + myolo_bn_apply_act, bit for bit"""
    rng = np.random.default_rng(10)
    x, w = dt(rnd(rng, N, H, W, 3)), dt(rnd(rng, 3, 3, 3, Cout, scale=0.3))
    gamma, beta, mm, mv = _frozen_bn(rng, Cout)
    Ho, Wo = H // 2, W // 2
    st = X.stream()
    y0, y1, y2, sc, sh = new(N, Ho, Wo, Cout), new(N, Ho, Wo, Cout), new(N, Ho, Wo, Cout), new(Cout), new(Cout)
    X.call("myolo_conv3x3s2_c3_fwd", X.ptr(x), X.ptr(w), X.ptr(y0), N, H, W, Cout, st)
    X.call("myolo_bn_frozen_coeffs", X.ptr(gamma), X.ptr(beta), X.ptr(mm), X.ptr(mv), X.ptr(sc), X.ptr(sh), Cout, st)
    X.call("myolo_bn_apply_act", X.ptr(y0), X.ptr(sc), X.ptr(sh), X.ptr(y1), N * Ho * Wo, Cout, act, st)
    X.call("myolo_conv3x3s2_c3_affine_act_fwd", X.ptr(x), X.ptr(w), X.ptr(sc), X.ptr(sh), act, X.ptr(y2), N, H, W, Cout, st)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    assert float(y1.abs().max()) > 0


@pytest.mark.parametrize("N,H,W,C,stride", [(2, 52, 52, 128, 1), (2, 52, 52, 128, 2), (3, 13, 13, 512, 1), (1, 6, 10, 8, 2), (1, 3, 5, 4, 1)])
def test_dwconv3x3_affine_act_fwd_equals_conv_then_frozen_bn(N, H, W, C, stride):
    """the depthwise half of the same fold: myolo_dwconv3x3_affine_act_fwd == myolo_dwconv3x3_fwd + myolo_bn_apply_act, bit for bit"""
    rng = np.random.default_rng(9)
    x, w = dt(rnd(rng, N, H, W, C)), dt(rnd(rng, 3, 3, C, 1, scale=0.3))
    gamma, beta, mm, mv = _frozen_bn(rng, C)
    Ho, Wo = H // stride, W // stride
    st = X.stream()
    y0, y1, y2, sc, sh = new(N, Ho, Wo, C), new(N, Ho, Wo, C), new(N, Ho, Wo, C), new(C), new(C)
    X.call("myolo_dwconv3x3_fwd", X.ptr(x), X.ptr(w), X.ptr(y0), N, H, W, C, stride, st)
    X.call("myolo_bn_frozen_coeffs", X.ptr(gamma), X.ptr(beta), X.ptr(mm), X.ptr(mv), X.ptr(sc), X.ptr(sh), C, st)
    X.call("myolo_bn_apply_act", X.ptr(y0), X.ptr(sc), X.ptr(sh), X.ptr(y1), N * Ho * Wo, C, 2, st)
    X.call("myolo_dwconv3x3_affine_act_fwd", X.ptr(x), X.ptr(w), X.ptr(sc), X.ptr(sh), 2, X.ptr(y2), N, H, W, C, stride, st)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)



# ----------------------------------------------------------------------------------------
# training-mode BatchNorm fusion of the trunk (include/myolo_hip_internal.h "*_bnstats_fwd", "*_bwd_weight_affine_in"):
# model.py:42-79, 249-278 with BatchNormalization on batch statistics, against the oracle's conv -> bn_train -> relu6 chain
# ----------------------------------------------------------------------------------------
def _check_bn_outputs(y_ref2d, g, b, mm, mv, mean, var, scale, shift, tmm, tmv):
    """mean / var / folded scale, shift / Keras moving averages of a fused kernel against oracle.bn_train on the reference output"""
    M = y_ref2d.shape[0]
    _, cache = O.bn_train(y_ref2d, g, b)
    check(mean, cache[2], 1e-5, "fused bn mean")
    check(var, cache[3], 1e-4, "fused bn var")
    sc_ref = g / np.sqrt(cache[3] + 1e-3)
    check(scale, sc_ref, 1e-4, "fused bn scale")
    check(shift, b - cache[2] * sc_ref, 1e-4, "fused bn shift")
    rmm, rmv = O.bn_moving_update(mm, mv, cache[2], cache[3], M)
    check(tmm, rmm, 1e-5, "fused moving mean")
    check(tmv, rmv, 1e-5, "fused moving var")


@pytest.mark.parametrize("act", [2, 1, 0])
@pytest.mark.parametrize("N,H,W,C,stride", [(2, 16, 16, 32, 1), (3, 14, 14, 512, 1), (2, 8, 8, 1024, 1), (32, 28, 28, 256, 1), (2, 46, 40, 96, 1), (4, 9, 7, 1024, 1),
                                            (2, 112, 112, 32, 1), (2, 16, 16, 32, 2), (2, 46, 40, 64, 2), (1, 30, 58, 160, 2), (3, 112, 112, 64, 2), (2, 56, 56, 128, 2),
                                            (2, 28, 28, 512, 2), (2, 12, 12, 24, 1)])
def test_dw_bwd_data_with_batchnorm_backward_sums(N, H, W, C, stride, act):
    """round 5: the depthwise data gradient whose output dx reaches a training-mode BatchNorm leaves that BatchNorm's backward
    sums in its epilogue (myolo_dwconv3x3_bwd_data_bnsums), finished by myolo_bn_act_bwd_from_partials.  Against the two plain calls
    (myolo_dwconv3x3_bwd_data + myolo_bn_act_bwd, themselves checked against the oracle above): dx of the conv bit-identical; dgamma / dbeta / the
    BatchNorm's dx equal up to the order of the fp32 partial sums; sizes the fused kernels do not take report rows == 0.  Repeated calls are
    bit-identical (fixed-order finish)."""
    rng = np.random.default_rng(33)
    Ho, Wo = H // stride, W // stride
    dy, w = dt(rnd(rng, N, Ho, Wo, C)), dt(rnd(rng, 3, 3, C))
    xbn = dt(rnd(rng, N, H, W, C, scale=2.0))                      # the pre-BN tensor of the BatchNorm in front of this conv
    g, b = 1 + rnd(rng, C, scale=0.2), rnd(rng, C, scale=0.3) + (1.0 if act else 0.0)
    M = N * H * W
    mean, var, scale, shift = new(C), new(C), new(C), new(C)
    mm, mv = dt(np.zeros(C, np.float32)), dt(np.ones(C, np.float32))
    st = X.stream()
    X.call("myolo_bn_stats", X.ptr(xbn), X.ptr(dt(g)), X.ptr(dt(b)), X.ptr(mean), X.ptr(var), X.ptr(scale), X.ptr(shift), X.ptr(mm), X.ptr(mv),
           M, C, *ws(), st)
    dx0, dbn0, dg0, db0 = new(N, H, W, C), new(M, C), new(C), new(C)
    X.call("myolo_dwconv3x3_bwd_data", X.ptr(dy), X.ptr(w), X.ptr(dx0), N, H, W, C, stride, st)
    X.call("myolo_bn_act_bwd", X.ptr(dx0), X.ptr(xbn), X.ptr(dt(g)), X.ptr(mean), X.ptr(var), X.ptr(scale), X.ptr(shift), X.ptr(dbn0), X.ptr(dg0),
           X.ptr(db0), M, C, act, 1, *ws(), st)
    rows = X.dw_bwd_data_bnsums_rows(N, H, W, C, stride)
    if C % 32 or (stride == 2 and (C > 256 or 256 % (C // 4))):
        assert rows == 0                   # sizes the fused kernels do not take: the caller falls back to the two plain calls
        return
    assert rows > 0
    outs = []
    for rep in range(2):
        part = torch.full((rows * 2 * C,), float("nan"), dtype=torch.float64, device="cuda")
        dx1, dbn1, dg1, db1 = new(N, H, W, C), new(M, C), new(C), new(C)
        X.call("myolo_dwconv3x3_bwd_data_bnsums", X.ptr(dy), X.ptr(w), X.ptr(dx1), N, H, W, C, stride, X.ptr(xbn), X.ptr(scale), X.ptr(shift),
               X.ptr(mean), X.ptr(var), act, X.ptr(part), rows, st)
        X.call("myolo_bn_act_bwd_from_partials", X.ptr(dx1), X.ptr(xbn), X.ptr(mean), X.ptr(var), X.ptr(scale), X.ptr(shift), X.ptr(dbn1),
               X.ptr(dg1), X.ptr(db1), M, C, act, X.ptr(part), rows, *ws(), st)
        torch.cuda.synchronize()
        assert not torch.isnan(part).any()
        outs.append((dx1, dbn1, dg1, db1))
    dx1, dbn1, dg1, db1 = outs[0]
    assert torch.equal(dx0, dx1), "the conv's data gradient must not change"
    check(dg1, dg0.cpu().numpy(), 2e-5, "fused sums dgamma")
    check(db1, db0.cpu().numpy(), 2e-5, "fused sums dbeta")
    check(dbn1, dbn0.cpu().numpy(), 2e-5, "BatchNorm dx from fused sums")
    assert all(torch.equal(u, v) for u, v in zip(outs[0], outs[1])), "not bit-reproducible"
    lib = X.load()
    assert lib.myolo_dwconv3x3_bwd_data_bnsums(X.ptr(dy), X.ptr(w), X.ptr(dx1), N, H, W, C, stride, X.ptr(xbn), X.ptr(scale), X.ptr(shift),
                                               X.ptr(mean), X.ptr(var), act, X.ptr(part), rows + 1, st) != 0          # wrong row count: refused


@pytest.mark.parametrize("nofuse", [0, 1])
@pytest.mark.parametrize("N,H,W,C,stride,lazy", [(2, 16, 16, 32, 1, True), (2, 16, 16, 32, 2, True), (3, 14, 14, 512, 1, True), (3, 14, 14, 512, 2, False),
                                                 (2, 8, 8, 1024, 1, True), (1, 14, 10, 64, 2, True), (2, 12, 12, 16, 1, True), (2, 6, 6, 24, 1, True),
                                                 (32, 28, 28, 256, 1, True), (2, 46, 40, 96, 1, True), (2, 46, 40, 64, 2, True), (4, 9, 7, 1024, 1, False),
                                                 (1, 30, 58, 160, 2, True)])
def test_dwconv3x3_bnstats_fwd_and_affine_in_weight_gradient(N, H, W, C, stride, lazy, nofuse):
    """depthwise conv whose input is relu6(x * in_scale + in_shift) formed on load, with the batch statistics of its output from the
    conv's own epilogue; and its weight gradient re-normalising x on load.  (C = 24: a channel count the in-kernel reduction does
    not take -- the entry point falls back to a statistics pass.)"""
    rng = np.random.default_rng(21)
    x, w = rnd(rng, N, H, W, C, scale=2.0), rnd(rng, 3, 3, C)
    isc, ish = (1 + rnd(rng, C, scale=0.3)), rnd(rng, C, scale=0.5) + 1.0
    g, b = 1 + rnd(rng, C, scale=0.2), rnd(rng, C, scale=0.3)
    mm, mv = rnd(rng, C, scale=0.1), 1 + np.abs(rnd(rng, C, scale=0.1))
    a_in = O.relu6(x * isc + ish) if lazy else x
    ref = O.dwconv3x3(a_in, w, stride)
    Ho, Wo = ref.shape[1], ref.shape[2]
    y = new(N, Ho, Wo, C)
    mean, var, scale, shift = new(C), new(C), new(C), new(C)
    tmm, tmv = dt(mm), dt(mv)
    wsb = torch.empty(X.dw_bnstats_ws_bytes(N, H, W, C, stride), dtype=torch.uint8, device=DEV)
    with X.option("no_trunk_fusion", nofuse):
        X.call("myolo_dwconv3x3_bnstats_fwd", X.ptr(dt(x)), X.ptr(dt(isc)) if lazy else None, X.ptr(dt(ish)) if lazy else None, 2, X.ptr(dt(w)), X.ptr(y),
               X.ptr(dt(g)), X.ptr(dt(b)), X.ptr(mean), X.ptr(var), X.ptr(scale), X.ptr(shift), X.ptr(tmm), X.ptr(tmv),
               N, H, W, C, stride, 3, wsb.data_ptr(), wsb.numel(), X.stream())
    check(y, ref, what="fused dw fwd")
    _check_bn_outputs(ref.reshape(-1, C), g, b, mm, mv, mean, var, scale, shift, tmm, tmv)
    if lazy:
        dy = rnd(rng, N, Ho, Wo, C)
        _, rdw = O.dwconv3x3_bwd(a_in, w, dy, stride)
        dw = new(3, 3, C)
        X.call("myolo_dwconv3x3_bwd_weight_affine_in", X.ptr(dt(x)), X.ptr(dt(isc)), X.ptr(dt(ish)), 2, X.ptr(dt(dy)), X.ptr(dw), N, H, W, C, stride,
               *ws(), X.stream())
        check(dw, rdw, what="dw dw with the input normalised on load")


@pytest.mark.parametrize("x6", [0, 1])
@pytest.mark.parametrize("nofuse", [0, 1])
@pytest.mark.parametrize("M,Cin,Cout,lazy", [(300, 32, 64, True), (4096, 64, 128, True), (25088, 64, 64, True), (6272, 512, 512, True), (1568, 512, 1024, True),
                                             (1568, 1024, 1024, False), (130, 16, 16, True), (20003, 32, 64, True), (257, 256, 512, True), (100352, 64, 128, True),
                                             (25088, 256, 256, True), (3000, 256, 512, False),
                                             # round 4: the register-fed thin-layer forward (32 / 64 -> 64 / 128 channels from 8192 rows): the fourth instantiation, no prologue, a 1-row tail
                                             (9001, 32, 128, False), (8192, 64, 64, False), (4100, 128, 256, True)])
def test_pwconv1x1_bnstats_fwd_and_affine_in_weight_gradient(M, Cin, Cout, lazy, nofuse, x6, request):
    """pointwise conv whose A operand is relu6(x * in_scale + in_shift) formed on load, with the batch statistics of its output from
    the GEMM epilogue (one pass) or, for the split-K shapes (M = 1568), from a statistics pass; and its weight gradient
    re-normalising x on load (thin-layer kernel, 128x128-tile kernel).  x6 = 1 (FP32_MATMUL = "bf16x6"): the layers with >= 256 input and
    a multiple of 256 output channels run on wino_mm_x6_kernel<PLAIN, PW> / wino_tn_x6_kernel with the same prologue and epilogue."""
    old = X.set_option("wino_x6", x6)
    request.addfinalizer(lambda: X.set_option("wino_x6", old))
    rng = np.random.default_rng(22)
    x, w = rnd(rng, M, Cin, scale=2.0), rnd(rng, Cin, Cout, scale=0.1)
    isc, ish = (1 + rnd(rng, Cin, scale=0.3)), rnd(rng, Cin, scale=0.5) + 1.0
    g, b = 1 + rnd(rng, Cout, scale=0.2), rnd(rng, Cout, scale=0.3)
    mm, mv = rnd(rng, Cout, scale=0.1), 1 + np.abs(rnd(rng, Cout, scale=0.1))
    a_in = (O.relu6(x * isc + ish) if lazy else x).astype(np.float64)
    ref = a_in @ w
    y = new(M, Cout)
    mean, var, scale, shift = new(Cout), new(Cout), new(Cout), new(Cout)
    tmm, tmv = dt(mm), dt(mv)
    wsb = torch.empty(X.pw_bnstats_ws_bytes(M, Cin, Cout), dtype=torch.uint8, device=DEV)
    with X.option("no_trunk_fusion", nofuse), X.option("tune0", 8192):        # (8192: the thin-layer forward also for 128 output channels)
        X.call("myolo_pwconv1x1_bnstats_fwd", X.ptr(dt(x)), X.ptr(dt(isc)) if lazy else None, X.ptr(dt(ish)) if lazy else None, 2, X.ptr(dt(w)), X.ptr(y),
               X.ptr(dt(g)), X.ptr(dt(b)), X.ptr(mean), X.ptr(var), X.ptr(scale), X.ptr(shift), X.ptr(tmm), X.ptr(tmv),
               M, Cin, Cout, 3, wsb.data_ptr(), wsb.numel(), X.stream())
    check(y, ref, what="fused pw fwd")
    _check_bn_outputs(ref.astype(np.float32), g, b, mm, mv, mean, var, scale, shift, tmm, tmv)
    if lazy:
        dy = rnd(rng, M, Cout)
        dw = new(Cin, Cout)
        X.call("myolo_pwconv1x1_bwd_weight_affine_in", X.ptr(dt(x)), X.ptr(dt(isc)), X.ptr(dt(ish)), 2, X.ptr(dt(dy)), X.ptr(dw), M, Cin, Cout, *ws(), X.stream())
        check(dw, a_in.T @ dy, what="pw dw with the input normalised on load")


@pytest.mark.parametrize("nofuse", [0, 1])
@pytest.mark.parametrize("N,H,W,Co", [(2, 32, 32, 16), (3, 16, 24, 32), (8, 64, 64, 32), (2, 14, 16, 8), (5, 224, 224, 32), (2, 30, 18, 32)])
def test_conv1_bnstats_fwd(N, H, W, Co, nofuse):
    rng = np.random.default_rng(23)
    x, w = rng.random((N, H, W, 3), dtype=np.float32), rnd(rng, 3, 3, 3, Co, scale=0.3)
    g, b = 1 + rnd(rng, Co, scale=0.2), rnd(rng, Co, scale=0.3)
    mm, mv = rnd(rng, Co, scale=0.1), 1 + np.abs(rnd(rng, Co, scale=0.1))
    ref = O.conv2d(x, w, stride=2, pads=(1, 1, 1, 1), acc=np.float64)
    y = new(N, H // 2, W // 2, Co)
    mean, var, scale, shift = new(Co), new(Co), new(Co), new(Co)
    tmm, tmv = dt(mm), dt(mv)
    wsb = torch.empty(X.conv1_bnstats_ws_bytes(N, H, W, Co), dtype=torch.uint8, device=DEV)
    with X.option("no_trunk_fusion", nofuse):
        X.call("myolo_conv3x3s2_c3_bnstats_fwd", X.ptr(dt(x)), X.ptr(dt(w)), X.ptr(y), X.ptr(dt(g)), X.ptr(dt(b)), X.ptr(mean), X.ptr(var),
               X.ptr(scale), X.ptr(shift), X.ptr(tmm), X.ptr(tmv), N, H, W, Co, 3, wsb.data_ptr(), wsb.numel(), X.stream())
    check(y, ref, what="fused conv1 fwd")
    _check_bn_outputs(ref.reshape(-1, Co).astype(np.float32), g, b, mm, mv, mean, var, scale, shift, tmm, tmv)


def test_measurement_probes_run_and_are_plausible():
    """the two ceiling probes bench.py prints beside the nominal peaks (myolo_stream_copy, myolo_mfma_probe): they run, reject bad
    arguments, and land in a plausible range on an MI355X (a tenth of the nominal peaks at least -- the point is that they execute)."""
    r = X.measure_mfma_tflops(iters=4000, reps=2, device=DEV)
    assert r["bf16_32x32x16"] > 250.0 and r["f32_32x32x2"] > 15.0, r
    out = torch.zeros(512 * 256, device=DEV)
    with pytest.raises(RuntimeError):
        X.call("myolo_mfma_probe", 7, 10, 512, out.data_ptr(), X.stream())
    with pytest.raises(RuntimeError):
        X.call("myolo_mfma_probe", 0, 0, 512, out.data_ptr(), X.stream())
    for kind in (2, 3):                                    # the dependent-chain variants
        assert X.call("myolo_mfma_probe", kind, 10, 512, out.data_ptr(), X.stream()) in (0, None)
    torch.cuda.synchronize()
    assert float(out.abs().sum()) == 0.0                   # the probe never writes (its store is behind an impossible condition)
    hb = X.measure_hbm_copy_gbs(nbytes=256 << 20, iters=2, device=DEV)
    assert hb["float4"] > 500.0 and hb["read_only_1wg_per_cu"] > 500.0, hb


def test_gather_groups_matches_index_select():
    """myolo_gather_groups (round 4: groups on blockIdx.y, no per-element division) against torch indexing: many small groups (more than
    the grid's y extent), a few big ones, scalar groups."""
    from myolo import _ext as X
    g = torch.Generator(device="cuda:0").manual_seed(3)
    for n_src, n, ge in ((4704, 3000, 49 * 64), (300, 7, 784 * 256), (5000, 4999, 4), (64, 64, 196 * 256)):
        src = torch.randn(n_src, ge, device="cuda:0", generator=g)
        idx = torch.randint(0, n_src, (n,), device="cuda:0", generator=g, dtype=torch.int32)
        dst = torch.empty(n, ge, device="cuda:0")
        X.call("myolo_gather_groups", X.ptr(src), X.ptr(idx), X.ptr(dst), n, ge, X.stream())
        assert torch.equal(dst, src[idx.long()])
    src = torch.arange(1000, device="cuda:0", dtype=torch.float32).view(1000, 1)          # one 4-byte element per group
    idx = torch.randint(0, 1000, (333,), device="cuda:0", generator=g, dtype=torch.int32)
    dst = torch.empty(333, 1, device="cuda:0")
    X.call("myolo_gather_groups", X.ptr(src), X.ptr(idx), X.ptr(dst), 333, 1, X.stream())
    assert torch.equal(dst, src[idx.long()])
