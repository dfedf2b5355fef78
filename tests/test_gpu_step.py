"""GPU parity of the whole hot path: one training step (forward, both losses, backward, Adam) and
the inference forward of myolo.model.MaskYOLO against the CPU oracle (oracle/np_model.py) on the
same seeded Shapes batch and the same weights.  Integer outputs bit-exact, floats within 1e-3."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import np_model, np_ops as O                      # noqa: E402
from myolo.config import make_config, ShapesConfig, ShapesHeadConfig  # noqa: E402
from myolo.model import MaskYOLO                               # noqa: E402
from myolo.shapes import make_shapes_samples                   # noqa: E402
from myolo.myolo_utils import BatchGenerator                   # noqa: E402

TOL = 1e-3


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))


def make_case(base, size, alpha, B, seed=0, need_pos=2):
    cfg = make_config(base, IMAGE_SHAPE=[size, size, 3], ALPHA=alpha, BATCH_SIZE=B)
    P = np_model.init_params(cfg, seed=seed, bias_scale=0.05)
    for start in range(0, 40 * B, B):
        samples = make_shapes_samples(B, cfg, start_index=start)
        batch, _ = BatchGenerator(samples, cfg, 'training', shuffle=False, norm=True)[0]
        ref = np_model.train_step_fwd_bwd(P, batch, cfg)
        if ref["n_pos"].sum() >= need_pos:
            return cfg, P, batch, ref
    raise RuntimeError("no batch with positive ROIs found")


def compare_step(cfg, P, batch, ref, verbose=False):
    model = MaskYOLO(mode="training", config=cfg)
    model.load_state_dict(P)
    out = model.train_on_batch(batch, learning_rate=0.0)
    grads = model.net.grads_dict()
    torch.cuda.synchronize()
    rows = []
    # ---- integer outputs: bit-exact
    ok_int = np.array_equal(out["target_class_ids"], ref["target_class_ids"]) and np.array_equal(out["n_pos"], ref["n_pos"])
    rows.append(("target_class_ids/n_pos equal", 0.0 if ok_int else 1.0))
    rows.append(("target_mask equal", 0.0 if np.array_equal(out["target_mask"], ref["target_mask"]) else 1.0))
    # ---- float outputs
    for k in ("yolo_output", "yolo_proposals", "output_rois", "feature_map", "myolo_mask"):
        rows.append((k, rel(out[k], ref[k])))
    for k in ("yolo_sum_loss", "mask_loss", "loss"):
        rows.append((k, abs(out[k] - float(ref[k])) / max(1.0, abs(float(ref[k])))))
    worst_g = 0.0
    for k, g in ref["grads"].items():
        if np.abs(g).max() < 1e-10:
            continue
        e = rel(grads[k], g)
        worst_g = max(worst_g, e)
        if verbose or e > TOL:
            rows.append(("grad " + k, e))
    rows.append(("worst grad", worst_g))
    if verbose:
        for r in rows:
            print("%-40s %.3e" % r)
    return rows


def test_train_step_config1_matches_oracle():
    """BASELINE.json configs[0]: Shapes 128x128, 3 classes, batch 4, MobileNet alpha 0.5."""
    cfg, P, batch, ref = make_case(ShapesConfig, 128, 0.5, 4)
    rows = compare_step(cfg, P, batch, ref)
    bad = [r for r in rows if r[1] > TOL]
    assert not bad, bad


def test_train_step_head_config_nbox5():
    """repository-HEAD head (N_BOX=5, config.py:28 anchors) at 96x96."""
    cfg, P, batch, ref = make_case(ShapesHeadConfig, 96, 0.5, 3, need_pos=1)
    rows = compare_step(cfg, P, batch, ref)
    bad = [r for r in rows if r[1] > TOL]
    assert not bad, bad


def test_adam_update_and_moving_stats_match_oracle():
    cfg, P, batch, ref = make_case(ShapesConfig, 96, 0.5, 2, need_pos=1)
    model = MaskYOLO(mode="training", config=cfg)
    model.load_state_dict(P)
    model.train_on_batch(batch, learning_rate=1e-3)
    sd = model.state_dict()
    P2, _ = np_model.adam_update({k: v.copy() for k, v in P.items()}, ref["grads"], {}, 1, 1e-3)
    worst = 0.0
    for k in np_model.trainable_names(P):
        # Adam's first step is lr*sign(g): compare only where the oracle gradient is well away from 0
        m = np.abs(ref["grads"][k]) > 1e-6 * max(1e-30, np.abs(ref["grads"][k]).max())
        if m.any():
            worst = max(worst, float(np.abs(sd[k][m] - P2[k][m]).max()))
    assert worst < 2e-5, worst
    for name, (mm, mv) in ref["moving"].items():
        assert rel(sd[name + "/moving_mean"], mm) < 1e-4 or np.abs(mm).max() < 1e-6
        assert rel(sd[name + "/moving_variance"], mv) < 1e-4
    # frozen BN layers of the mask head keep their moving statistics
    for i in (2, 3, 4):
        assert np.array_equal(sd["myolo_mask_bn%d/moving_mean" % i], P["myolo_mask_bn%d/moving_mean" % i])


def test_inference_forward_matches_oracle():
    cfg = make_config(ShapesConfig, IMAGE_SHAPE=[128, 128, 3], ALPHA=0.5, BATCH_SIZE=2)
    P = np_model.init_params(cfg, seed=3, bias_scale=0.05)
    samples = make_shapes_samples(2, cfg)
    images = np.stack([s[0] for s in samples]).astype(np.float32) / 255.
    ref = np_model.inference_fwd(P, images, cfg)
    model = MaskYOLO(mode="inference", config=cfg)
    model.load_state_dict(P)
    yo, det, mask = model.keras_model.predict([images])
    assert rel(yo, ref["yolo_output"]) < TOL
    assert np.array_equal(det[..., 5], ref["detections"][..., 5]), "class ids differ"
    assert rel(det[..., :5], ref["detections"][..., :5]) < TOL
    assert rel(mask, ref["myolo_mask"]) < TOL


def test_two_runs_bit_identical_forward():
    """determinism: everything except the ROIAlign scatter-add (fp32 atomics) is order-fixed."""
    cfg, P, batch, ref = make_case(ShapesConfig, 96, 0.5, 2, need_pos=1)
    outs = []
    for _ in range(2):
        model = MaskYOLO(mode="training", config=cfg)
        model.load_state_dict(P)
        outs.append(model.train_on_batch(batch, learning_rate=0.0))
    assert outs[0]["loss"] == outs[1]["loss"]
    assert np.array_equal(outs[0]["myolo_mask"], outs[1]["myolo_mask"])


def test_detect_surface():
    cfg = make_config(ShapesConfig, IMAGE_SHAPE=[128, 128, 3], ALPHA=0.5, BATCH_SIZE=1)
    model = MaskYOLO(mode="inference", config=cfg)
    img = make_shapes_samples(1, cfg)[0][0]
    res = model.detect(img, cs_threshold=0.0)
    assert set(res[0]) == {"bboxes", "class_ids", "confidence_scores", "full_masks"}
    assert res[0]["full_masks"].shape[:2] == (128, 128)
    assert len(res[0]["class_ids"]) == res[0]["full_masks"].shape[-1] <= 10


if __name__ == "__main__":
    import sys
    sys.path[:0] = [".", "mask-yolo_amd"]
    cfg, P, batch, ref = make_case(ShapesConfig, 128, 0.5, 4)
    print("oracle n_pos", ref["n_pos"], "loss", ref["loss"])
    compare_step(cfg, P, batch, ref, verbose=True)
