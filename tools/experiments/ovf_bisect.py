"""The 500-step one-batch overfit of tests/test_gpu_step.py under different kernel selections (tune0 bits) and seeds: prints early / tail mask
loss and the YOLO loss ratio for each, plus the first step's conv1 gradient compared between the selections.
  python tools/experiments/ovf_bisect.py 0,1024 1,2,3"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/mask-yolo_amd'); sys.path.insert(0, '/root/repo/tests')
from myolo import _ext as X
X.load()
from test_gpu_step import make_config, make_shapes_samples, ShapesConfig, BatchGenerator, MaskYOLO

opts = [int(a) for a in sys.argv[1].split(",")]
seeds = [int(a) for a in (sys.argv[2] if len(sys.argv) > 2 else "1").split(",")]
B = 4
cfg = make_config(ShapesConfig, IMAGE_SHAPE=[128, 128, 3], BATCH_SIZE=B)
samples = make_shapes_samples(B, cfg)
batch, _ = BatchGenerator(samples, cfg, 'training', shuffle=False, norm=True)[0]
g0 = {}
for seed in seeds:
    for opt in opts:
        X.set_option("tune0", opt)
        m = MaskYOLO(mode="training", config=cfg, seed=seed)
        m.set_trainable(".*")
        m.compile(1e-3, 0.9)
        db = m.net.to_device_batch(batch)
        early, tail, y0, trace = 0.0, [], None, []
        for i in range(500):
            out = m.net.forward_backward(db)
            if i == 0:
                g = {k: v.detach().float().cpu().numpy().copy() for k, v in m.net.g.items()}
                if seed in g0:
                    worst = max((float(np.abs(g[k] - g0[seed][k]).max() / (np.abs(g0[seed][k]).max() + 1e-30)), k) for k in g)
                    print("  seed %d tune0 %d: first-step gradients vs tune0 %d: worst rel %.2e at %s; conv1/kernel %.2e" % (
                        seed, opt, opts[0], worst[0], worst[1],
                        float(np.abs(g["conv1/kernel"] - g0[seed]["conv1/kernel"]).max() / np.abs(g0[seed]["conv1/kernel"]).max())))
                else:
                    g0[seed] = g
            m.net.adam_step(1e-3 if i < 350 else 3e-4)
            ml = float(out["mask_terms"][0])
            if i == 0:
                y0 = float(out["yolo_terms"][0])
            if i < 150:
                early = max(early, ml)
            if i >= 400:
                tail.append(ml)
            if i % 50 == 49:
                trace.append(round(ml, 3))
        print("seed %d tune0 %d: early %.3f tail-median %.3f yolo %.3f -> %.3f   mask every 50: %s" % (
            seed, opt, early, float(np.median(tail)), y0, float(out["yolo_terms"][0]), trace), flush=True)
        m.net.close() if hasattr(m.net, "close") else None
