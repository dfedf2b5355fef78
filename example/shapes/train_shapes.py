#!/usr/bin/env python
"""Train Mask-YOLO on the synthetic Shapes dataset on one MI355X -- the working counterpart of the reference's
example/shapes/train_shapes.py (same calls: ShapesConfig, ShapesDataset.load_shapes/prepare, MaskYOLO(mode="training"),
model.train(...)), then run detect() on a held-out image with the trained weights.

  python example/shapes/train_shapes.py [--epochs 5] [--train 500] [--val 50] [--size 224] [--device-stream STEPS]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "mask-yolo_amd")]
import numpy as np                                   # noqa: E402
from myolo.config import ShapesConfig, make_config   # noqa: E402
from myolo.model import MaskYOLO                     # noqa: E402
from myolo.shapes import ShapesDataset               # noqa: E402
from myolo import myolo_utils as mutils              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=5)
    ap.add_argument("--train", type=int, default=500)
    ap.add_argument("--val", type=int, default=50)
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--logs", default="./logs")
    ap.add_argument("--device-stream", type=int, default=0,
                    help="instead of model.train(): this many steps on the endless Shapes stream produced on the GPU")
    a = ap.parse_args()

    config = make_config(ShapesConfig, IMAGE_SHAPE=[a.size, a.size, 3], BATCH_SIZE=a.batch)
    model = MaskYOLO(mode="training", config=config, model_dir=a.logs, yolo_pretrain_dir=None, yolo_trainable=True)
    if a.device_stream:
        losses = model.train_shapes_stream(a.device_stream, learning_rate=config.LEARNING_RATE, verbose=1)
        print("mean loss of the last 10 steps: %.4f" % float(np.mean(losses[-10:])))
    else:
        dataset_train, dataset_val = ShapesDataset(), ShapesDataset(seed=99)
        dataset_train.load_shapes(a.train, a.size, a.size)
        dataset_train.prepare()
        dataset_val.load_shapes(a.val, a.size, a.size)
        dataset_val.prepare()
        hist = model.train(dataset_train, dataset_val, learning_rate=config.LEARNING_RATE, epochs=a.epochs, layers='all')
        print("epoch losses:", ["%.4f" % h for h in hist])
    os.makedirs(a.logs, exist_ok=True)
    weights = os.path.join(a.logs, "mask_yolo_shapes_final.npz")
    model.save_weights(weights)

    # inference on a held-out image (model.py:1238 detect)
    test = ShapesDataset(seed=7)
    test.load_shapes(1, a.size, a.size)
    test.prepare()
    image, gt_class_ids, gt_boxes, gt_masks = mutils.load_image_gt(test, config, image_id=0, augment=None, augmentation=None,
                                                                   use_mini_mask=config.USE_MINI_MASK)
    infer = MaskYOLO(mode="inference", config=config)
    res = infer.detect(image, weights_dir=weights, cs_threshold=0.35)[0]
    print("ground truth: classes %s" % [int(c) for c in gt_class_ids])
    print("detected: %d instances, classes %s, scores %s" % (len(res["class_ids"]), [int(c) for c in res["class_ids"]],
                                                             ["%.2f" % s for s in res["confidence_scores"]]))


if __name__ == "__main__":
    main()
