"""Where train_shapes_stream()'s 3 % against Net.train_step goes: host time of the specifications, device time of the producer kernels, and the step rate
with a fixed batch through the same public call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "mask-yolo_amd")]
import numpy as np, torch
from myolo.config import make_config, ShapesConfig
from myolo.model import MaskYOLO
from myolo.shapes import ShapesProducer
cfg = make_config(ShapesConfig, BATCH_SIZE=32)
m = MaskYOLO(mode="training", config=cfg, seed=1)
m.set_trainable(".*"); m.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
prod = ShapesProducer(cfg, seed=1234, device=m._device)
t0 = time.perf_counter()
for i in range(20): prod.specs(list(range(32 * i, 32 * i + 32)))
print("specs(): %.2f ms of host time per batch" % (1e3 * (time.perf_counter() - t0) / 20))
d = prod.batch(list(range(32))); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(10): d2 = prod.batch(list(range(32 * i, 32 * i + 32)))
e1.record(); torch.cuda.synchronize()
print("batch(): %.3f ms per batch on an idle GPU (host + device)" % (e0.elapsed_time(e1) / 10))
def run(n, fixed):
    side = m.net._copy_stream
    produce = lambda i: prod.batch(list(range(32 * i, 32 * i + 32)), stream=side, consumer=torch.cuda.current_stream())
    torch.cuda.synchronize(); t = time.perf_counter()
    nxt = d if fixed else produce(0)
    res = []
    for i in range(n):
        cur, nxt = nxt, (d if fixed else produce(i + 1))
        res.append(m.train_on_batch(cur))
        if i >= 2: res[i - 2] = None
    torch.cuda.synchronize()
    return time.perf_counter() - t
for fixed in (True, False, True, False):
    run(8, fixed)
    a, b = run(16, fixed), run(48, fixed)
    print("%s: %.3f ms per step (marginal, 48 - 16 steps)" % ("fixed batch through train_on_batch" if fixed else "produced on the copy stream", 1e3 * (b - a) / 32))
from myolo.model import _gc_parked
def run2(n, read_loss, parked):
    side = m.net._copy_stream
    produce = lambda i: prod.batch(list(range(32 * i, 32 * i + 32)), stream=side, consumer=torch.cuda.current_stream())
    torch.cuda.synchronize(); t = time.perf_counter()
    import contextlib
    with (_gc_parked() if parked else contextlib.nullcontext()):
        nxt = produce(0)
        res = []
        for i in range(n):
            cur, nxt = nxt, produce(i + 1)
            res.append(m.train_on_batch(cur))
            if i >= 2: res[i - 2] = res[i - 2]["loss"] if read_loss else None
    torch.cuda.synchronize()
    return time.perf_counter() - t
for rl, pk in ((False, False), (True, False), (False, True), (True, True), (False, False), (True, True)):
    run2(8, rl, pk)
    a_, b_ = run2(16, rl, pk), run2(64, rl, pk)
    print("loop: read_loss=%d gc_parked=%d: %.3f ms per step" % (rl, pk, 1e3 * (b_ - a_) / 48))
def api(n):
    torch.cuda.synchronize(); t = time.perf_counter(); m.train_shapes_stream(n); torch.cuda.synchronize(); return time.perf_counter() - t
api(8)
for rep in range(3):
    a, b = api(16), api(64)
    print("train_shapes_stream(): %.3f ms per step (marginal, 64 - 16 steps); 64-step call %.1f ms" % (1e3 * (b - a) / 48, 1e3 * b))
# the public call's body with single pieces switched off
def body(n, new_prod=True, recompile=True):
    torch.cuda.synchronize(); t = time.perf_counter()
    p2 = ShapesProducer(cfg, seed=1234, device=m._device) if new_prod else prod
    if recompile:
        m.set_trainable(".*"); m.compile(cfg.LEARNING_RATE, cfg.LEARNING_MOMENTUM)
    side = m.net._copy_stream
    produce = lambda i: p2.batch(list(range(32 * i, 32 * i + 32)), stream=side, consumer=torch.cuda.current_stream())
    results = []
    with _gc_parked():
        nxt = produce(0)
        for i in range(n):
            cur, nxt = nxt, (produce(i + 1) if i + 1 < n else None)
            results.append(m.train_on_batch(cur))
            if i >= 2: results[i - 2] = results[i - 2]["loss"]
    losses = [r if isinstance(r, float) else r["loss"] for r in results]
    torch.cuda.synchronize()
    return time.perf_counter() - t
for np_, rc in ((True, True), (False, True), (True, False), (False, False), (True, True)):
    body(8, np_, rc)
    a_, b_ = body(16, np_, rc), body(64, np_, rc)
    print("body: new producer=%d recompile=%d: %.3f ms per step (16-step call %.1f ms)" % (np_, rc, 1e3 * (b_ - a_) / 48, 1e3 * a_))
