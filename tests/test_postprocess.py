"""Inference post-processing: product (host numpy helpers + the GPU unmold kernel)
against the oracle's literal restatement (oracle/np_post.py).  Integer / boolean outputs bit-exact."""
import numpy as np
import pytest

from oracle import np_post as Q
from oracle import np_ops as O
from myolo import myolo_utils as mutils
from myolo.config import ShapesConfig, RiceConfig, make_config


def _rand_dets(rng, N, C):
    c = rng.random((N, 2)) * 1.1 - 0.05
    s = rng.random((N, 2)) * 0.6 + 0.01
    det = np.zeros((N, 6), np.float32)
    det[:, 0:2], det[:, 2:4] = c - s / 2, c + s / 2
    det[:, 4] = rng.random(N)
    det[:, 5] = rng.integers(0, C, N)
    det[0, :4] = [0.2, 0.3, 0.2 + 1e-3, 0.3 + 2e-3]       # sub-pixel box -> 1x1 window
    det[1, :4] = [-0.2, -0.1, 1.3, 1.2]                   # larger than the image
    det[2, :4] = [0.5, 0.5, 0.52, 0.9]                    # narrower than the 28-px mask (down-sampling)
    return det


def test_host_unmold_nmb_decode_match_oracle():
    rng = np.random.default_rng(0)
    shape = [224, 224, 3]
    det = _rand_dets(rng, 40, 4)
    for i in range(40):
        m = rng.random((28, 28)).astype(np.float32)
        assert np.array_equal(mutils.unmold_mask(m, det[i, :4], shape), Q.unmold_mask(m, det[i, :4], shape))
    boxes, ids = det[:10, :4], det[:10, 5].astype(np.int32)
    boxes[3] = boxes[2] + 0.001
    ids[3] = ids[2]
    idx = np.arange(10) + 100
    assert np.array_equal(mutils.NMB(boxes, ids, idx, shape, 0.7), Q.nmb(boxes, ids, idx, shape, 0.7))
    G, A, C = 7, 3, 4
    net = rng.standard_normal((G, G, A, 5 + C)) * 2
    got = mutils.decode_one_yolo_output(net.copy(), ShapesConfig.ANCHORS, C, obj_threshold=0.2, nms_threshold=0.3)
    ref = Q.decode_one_yolo_output(net.copy(), ShapesConfig.ANCHORS, C, obj_threshold=0.2, nms_threshold=0.3)
    assert len(got) == len(ref) > 0
    for g, r in zip(got, ref):
        np.testing.assert_allclose([g.xmin, g.ymin, g.xmax, g.ymax, g.c], r[:5], rtol=1e-12)
        np.testing.assert_allclose(g.classes, r[5], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("base,N", [(ShapesConfig, 147), (RiceConfig, 845)])
def test_gpu_unmold_and_detect_post_match_oracle(base, N):
    import torch
    from myolo.model import MaskYOLO
    cfg = make_config(base, BATCH_SIZE=1)
    C, H = cfg.NUM_CLASSES, cfg.IMAGE_SHAPE[0]
    rng = np.random.default_rng(1)
    det = _rand_dets(rng, N, C)
    masks = rng.random((N, 28, 28, C)).astype(np.float32)
    model = MaskYOLO(mode="inference", config=cfg)
    boxes, ids, scores, full = model.decode_masks(det[None], masks[None], cfg.IMAGE_SHAPE)
    rb, ri, rs, rf = Q.decode_masks(det, masks, cfg.IMAGE_SHAPE)
    assert full.shape == (H, H, N) and full.dtype == bool
    assert np.array_equal(ids, ri) and np.array_equal(boxes, rb) and np.array_equal(scores, rs)
    assert np.array_equal(full, rf), "unmolded masks differ: %d pixels" % int((full != rf).sum())
    # a zero-area detection is dropped (model.py:1373-1380)
    det2 = det.copy()
    det2[5, 2] = det2[5, 0]
    b2, i2, s2, f2 = model.decode_masks(det2[None], masks[None], cfg.IMAGE_SHAPE)
    r2 = Q.decode_masks(det2, masks, cfg.IMAGE_SHAPE)
    assert f2.shape[-1] == N - 1 and np.array_equal(f2, r2[3]) and np.array_equal(i2, r2[1])
    # whole detect(): GPU forward, then post-processing on both sides from the SAME network outputs
    img = (rng.random((H, H, 3)) * 255).astype(np.uint8)
    res = model.detect(img, cs_threshold=0.3)[0]
    x = torch.as_tensor((img[None] / 255.).astype(np.float32), device=model.net.dev)
    yo, d, m = model.net.predict(x)
    ref = Q.detect_post(d[0].cpu().numpy(), m[0].cpu().numpy(), cfg.IMAGE_SHAPE, cs_threshold=0.3)
    for k in ("bboxes", "class_ids", "confidence_scores", "full_masks"):
        assert np.array_equal(res[k], ref[k]), k
    assert res["full_masks"].shape[-1] <= 10


@pytest.mark.gpu
@pytest.mark.parametrize("base,size", [(ShapesConfig, 224), (ShapesConfig, 128)])
def test_gpu_shapes_producer_matches_host_pipeline(base, size):
    """the device producer (rasterise + empty-instance filter + extract_bboxes + target encoding)
    reproduces the host pipeline (ShapesDataset -> load_image_gt -> BatchGenerator) bit for bit, 64 images."""
    import torch
    from myolo.shapes import ShapesProducer, make_shapes_samples
    cfg = make_config(base, IMAGE_SHAPE=[size, size, 3], BATCH_SIZE=64)
    start = 40
    samples = make_shapes_samples(64, cfg, start_index=start)
    host, _ = mutils.BatchGenerator(samples, cfg, "training", shuffle=False, norm=True)[0]
    ref = O.encode_batch(samples, cfg)                      # the oracle's literal restatement of the encoding
    d = ShapesProducer(cfg).batch(list(range(start, start + 64)))
    torch.cuda.synchronize()
    images, true_boxes, y_true, gt_ids, gt_boxes, gt_masks = host
    assert np.array_equal(d["images"].cpu().numpy(), images) and np.array_equal(images, ref[0])
    assert np.array_equal(d["gt_masks"].cpu().numpy().astype(bool), gt_masks)
    assert np.array_equal(d["gt_boxes"].cpu().numpy(), gt_boxes) and np.array_equal(d["gt_ids"].cpu().numpy(), gt_ids)
    assert np.array_equal(d["y_true"].cpu().numpy(), ref[2].astype(np.float32))
    assert np.array_equal(d["true_boxes"].cpu().numpy(), ref[1].astype(np.float32).reshape(64, -1, 4))
    assert gt_ids.max() == 3 and (gt_ids > 0).sum() > 64
