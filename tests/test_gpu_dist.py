"""Data-parallel step on real kernels: two processes share cuda:0, rendezvous over gloo (it accepts
device tensors), each runs the REAL engine (myolo.engine.Net) on its own shard with the bucketed, overlapped GradReducer.
  * the all-reduced, 1/world-scaled gradient equals the mean of the two single-rank gradients (<= 1e-5 relative);
  * after Adam every rank holds bit-identical weights;
  * with frozen layers (set_trainable / the yolo_trainable=False recipe) the frozen weights stay bit-identical -- the
    reducer is joined before the freeze mask is applied (round-1 advisor finding).
RCCL itself needs one GPU per rank; on this 1-GPU box it is exercised as a 1-rank communicator through the C-ABI
(myolo_comm_*), which checks the dlopen'ed librccl, the stream plumbing and the timing instrumentation."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _case():
    from myolo.config import make_config, ShapesConfig
    from myolo.shapes import make_shapes_samples
    from myolo.myolo_utils import BatchGenerator
    from myolo.engine import init_state_dict
    cfg = make_config(ShapesConfig, IMAGE_SHAPE=[128, 128, 3], ALPHA=0.5, BATCH_SIZE=4)
    P = init_state_dict(cfg, seed=5)
    batches = []
    for r in range(2):
        samples = make_shapes_samples(4, cfg, start_index=40 + 4 * r)
        batches.append(BatchGenerator(samples, cfg, 'training', shuffle=False, norm=True)[0][0])
    return cfg, P, batches


def _worker(rank, world, port, q, freeze):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from myolo.model import MaskYOLO
        from myolo.dist import GradReducer
        cfg, P, batches = _case()
        # ---- single-rank gradients of BOTH shards (no communication) -> their mean is the expected exchanged gradient
        solo = MaskYOLO(mode="training", config=cfg, device="cuda:0")
        local = []
        for b in batches:
            solo.load_state_dict(P)
            solo.net.forward_backward(solo.net.to_device_batch(b))
            local.append(solo.net.flat_g.clone())
        expect = (local[0] + local[1]) * 0.5
        # ---- the data-parallel step: own shard, bucketed overlapped all-reduce, 1/world inside Adam
        model = MaskYOLO(mode="training", config=cfg, device="cuda:0")
        model.load_state_dict(P)
        model._reducer = GradReducer(model.net.flat_g, model.net.bucket_ranges, timing=True).attach(model.net)
        assert model._reducer.world == world and model._reducer.ranks_seen() == world
        if freeze:
            model.set_trainable(r"(myolo_mask.*)|(feature_map)|(conv_23)|(conv_.w_1[0-4].*)")   # the mask side has no gradient when a random-init batch has no positive ROI
        else:
            model.set_trainable(".*")
        model.compile(1e-3, 0.9)
        before = model.net.flat_p.clone()
        model.train_on_batch(batches[rank])
        torch.cuda.synchronize()
        got = model.net.flat_g * model.net.grad_scale
        if freeze:
            m = model._train_mask
            err = float(((got - expect * m).abs().max() / expect.abs().max()).item())
        else:
            err = float(((got - expect).abs().max() / expect.abs().max()).item())
        after = model.net.flat_p.clone()
        gathered = [torch.zeros(after.numel()) for _ in range(world)]
        dist.all_gather(gathered, after.cpu())
        same = all(torch.equal(gathered[0], t) for t in gathered)
        frozen_ok = "ok"
        if freeze:
            fz = model._train_mask == 0
            if not torch.equal(after[fz], before[fz]):
                frozen_ok = "frozen weights moved: %d entries, max %g" % (int((after[fz] != before[fz]).sum()), float((after[fz] - before[fz]).abs().max()))
            elif not bool((after[~fz] != before[~fz]).any()):
                frozen_ok = "trainable weights did not move"
        ms = model._reducer.bucket_ms()
        q.put((rank, err, same, frozen_ok, str(ms) if not (ms is not None and len(ms) == 3 and all(v >= 0 for v in ms)) else "ok"))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("freeze", [False, True])
def test_two_rank_step_real_engine_gloo_on_one_gpu(freeze):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, freeze)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(world))
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, err, same, frozen_ok, timed in res:
        assert err < 1e-5, (rank, err)                # averaged gradient == mean of the single-rank gradients
        assert same, "ranks hold different weights after Adam"
        assert frozen_ok == "ok", frozen_ok
        assert timed == "ok", timed


def test_rccl_through_the_c_abi_one_rank():
    """myolo_comm_unique_id / comm_init / comm_size / allreduce_sum_f32 / comm_destroy on a 1-rank RCCL communicator, driven
    by GradReducer(backend='capi'): sum over one rank is the identity, on the reducer's side stream, with event timing."""
    from myolo.dist import GradReducer
    g = torch.randn(1 << 20, device="cuda:0")
    g0 = g.clone()
    red = GradReducer(g, [(0, 300000), (300000, 700000), (700000, 1 << 20)], always=True, backend="capi", timing=True)
    try:
        assert red.active and red.comm is not None and red.ranks_seen() == 1
        for _ in range(2):
            for i in (2, 1, 0):
                red.bucket_ready(i)
            red.wait()
        torch.cuda.synchronize()
        assert torch.equal(g, g0)
        ms = red.bucket_ms()
        assert len(ms) == 3 and all(v > 0 for v in ms)
    finally:
        red.close()


def test_engine_step_with_capi_reducer_is_bit_identical_to_no_reducer():
    from myolo.model import MaskYOLO
    from myolo.dist import GradReducer
    cfg, P, batches = _case()
    outs = []
    for use in (False, True):
        model = MaskYOLO(mode="training", config=cfg, device="cuda:0")
        model.load_state_dict(P)
        red = None
        if use:
            red = GradReducer(model.net.flat_g, model.net.bucket_ranges, always=True, backend="capi").attach(model.net)
        model.set_trainable(".*")
        model.compile(1e-3, 0.9)
        model.train_on_batch(batches[0])
        torch.cuda.synchronize()
        outs.append(model.net.flat_p.clone())
        if red:
            red.close()
    assert torch.equal(outs[0], outs[1])
