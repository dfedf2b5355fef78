"""Search the seeded Shapes stream for config-2 batches (224x224, batch 32, alpha 1, N_BOX=3) in which EVERY proposal's best IoU with
the ground truth sits further than `margin` from the 0.5 positive/negative threshold under the oracle's trunk (oracle/torch_ref.py)
-- the batch tests/test_gpu_fullsize.py::test_train_step_config2_matches_torch_ref pins by index (SAFE_BATCH_START).
  python tools/find_safe_config2_batch.py [first_candidate] [n_candidates] [margin]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "mask-yolo_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from myolo.config import make_config, ShapesConfig
from myolo.shapes import make_shapes_samples
from myolo.myolo_utils import BatchGenerator
from oracle import np_model, np_ops as O
from oracle.torch_ref import TorchRef

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
margin_min = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
cfg = make_config(ShapesConfig, IMAGE_SHAPE=[224, 224, 3], ALPHA=1.0, BATCH_SIZE=32)
P = np_model.init_params(cfg, seed=0, bias_scale=0.05)
ref = TorchRef(P, cfg, torch.float32)
H, W = cfg.IMAGE_SHAPE[:2]
for cand in range(first, first + count):
    start = 1000 + 32 * cand
    samples = make_shapes_samples(32, cfg, start_index=start)
    batch, _ = BatchGenerator(samples, cfg, 'training', shuffle=False, norm=True)[0]
    with torch.no_grad():
        _, _, yo = ref.trunk(batch[0], True)
    prop = O.yolo_decode(yo.numpy(), cfg.ANCHORS, cfg.GRID_W)
    gtn = O.norm_boxes(batch[4], H, W)
    ov = np.stack([O.overlaps(prop[b], gtn[b]).max(1) for b in range(32)])
    m = np.abs(ov - 0.5)
    print("start_index %d: min |IoU - 0.5| = %.3e, unsafe(<1e-4) = %d, positives = %d" % (start, m.min(), int((m < 1e-4).sum()), int((ov >= 0.5).sum())), flush=True)
