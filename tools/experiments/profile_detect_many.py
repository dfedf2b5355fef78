"""cProfile of MaskYOLO.detect_many at BASELINE configs[3] (Rice 416x416, batch 4, bf16 mask head): where the host time of the public inference call goes."""
import cProfile, pstats, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
import numpy as np, torch
from myolo.model import MaskYOLO
from myolo.config import make_config, RiceConfig
cfg = make_config(RiceConfig, BATCH_SIZE=4, INFERENCE_DTYPE="bf16")
m = MaskYOLO(mode="inference", config=cfg, seed=0)
rng = np.random.default_rng(0)
imgs = [(rng.random((416, 416, 3)) * 255).astype(np.uint8) for _ in range(48)]
m.detect_many(imgs[:16])
torch.cuda.synchronize()
t0 = time.perf_counter(); m.detect_many(imgs); torch.cuda.synchronize(); print("detect_many: %.1f img/s" % (48 / (time.perf_counter() - t0)))
pr = cProfile.Profile(); pr.enable(); m.detect_many(imgs); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
