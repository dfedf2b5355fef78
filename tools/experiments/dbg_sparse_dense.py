import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]; os.chdir(ROOT)
from myolo import _ext as X
import importlib
import numpy as np, torch
import test_gpu_step as T
from myolo.model import MaskYOLO
from myolo.config import ShapesConfig
def run(opts):
    for k, v in opts.items():
        X.set_option(k, v)
    seen = []
    for seed in (0, 1, 2, 3, 4):
        cfg, P, batch, ref = T.make_case(ShapesConfig, 128, 0.5, 4, seed=seed)
        grads, cap = [], []
        for sparse in (False, True):
            model = MaskYOLO(mode="training", config=cfg)
            model.load_state_dict(P)
            model.net.sparse_mask_bwd = sparse
            model.net.tape_hook = T._capture_mask_tape(cap)
            out = model.train_on_batch(batch, learning_rate=0.0)
            grads.append(model.net.grads_dict())
        R = out["myolo_mask"].shape[1]
        pos = np.concatenate([np.arange(b * R, b * R + n) for b, n in enumerate(out["n_pos"])])
        flips = T._relu_flips(P, cap[0], cap[1], pos, len(out["n_pos"]) * R) if len(pos) else 0
        worst, wk = 0.0, None
        for k in grads[0]:
            d = grads[0][k]
            if np.abs(d).max() < 1e-12 or k == "myolo_mask_conv1/bias":
                continue
            r = T.rel(grads[1][k], d)
            if r > worst: worst, wk = r, k
        seen.append((seed, flips, round(worst, 6), wk, len(pos)))
    for k in opts:
        X.set_option(k, 0)
    print(opts, seen, flush=True)
run({})
run({"pw_no_smallm": 1})
run({"pw_skinny_nw4": 1})
run({"pw_no_smallm": 1, "pw_skinny_nw4": 1})
