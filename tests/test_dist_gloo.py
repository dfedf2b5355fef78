"""Multi-process CPU tests (gloo, world_size 2) of the data-parallel path: the bucketed, overlapped
gradient all-reduce (myolo/dist.py) and the image sharding.  The GPU path differs only in the
backend ("nccl" = RCCL) and in running the collective on a side stream."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from myolo.dist import GradReducer, shard_range, dp_batch_indices


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n = 1000
        ranges = [(0, 300), (300, 720), (720, 1000)]
        g = torch.arange(n, dtype=torch.float32) * (rank + 1)            # rank r holds (r+1) * [0..n)
        red = GradReducer(g, ranges)
        assert red.world == world and abs(red.grad_scale - 1.0 / world) < 1e-12
        for i in (2, 1, 0):                                              # backward order: mask head, yolo head, backbone
            red.bucket_ready(i)
        red.wait()
        mean = g * red.grad_scale
        expect = torch.arange(n, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        ok = bool(torch.allclose(mean, expect, rtol=1e-6))
        # fused "Adam after all-reduce" on the CPU copy: identical parameters on every rank
        p = torch.ones(n) - 1e-3 * mean
        gathered = [torch.zeros(n) for _ in range(world)]
        dist.all_gather(gathered, p)
        same = all(torch.equal(gathered[0], t) for t in gathered)
        lo, hi = shard_range(64, rank, world)
        out.put((rank, ok, same, lo, hi))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True] and [r[2] for r in res] == [True, True]
    assert [(r[3], r[4]) for r in res] == [(0, 32), (32, 64)]          # rank r trains on images [32r, 32r+32)


def test_single_rank_reducer_is_a_noop():
    g = torch.randn(100)
    g0 = g.clone()
    red = GradReducer(g, [(0, 50), (50, 100)])
    for i in (1, 0):
        red.bucket_ready(i)
    red.wait()
    assert red.world == 1 and red.grad_scale == 1.0 and torch.equal(g, g0)


def test_shard_range_requires_even_split():
    assert shard_range(256, 3, 8) == (96, 128)
    with pytest.raises(AssertionError):
        shard_range(100, 0, 8)


def test_engine_bucket_ranges_cover_flat_buffer_contiguously():
    """the five all-reduce buckets tile Net.flat_g exactly, in table order, each cut where backward finishes a group of layers: backbone | YOLO blocks +
    conv_23 | feature_map | myolo_mask_conv1 + bn1 | the rest of the mask head (layout computed without touching the GPU; same offsets as rounds 1-5)."""
    from myolo.engine import layer_table, N_BUCKETS, BUCKET_BACKBONE, BUCKET_YOLO, BUCKET_FEATURE_MAP, BUCKET_MASK_CONV1, BUCKET_MASK_REST
    from myolo.config import ShapesConfig
    cfg = ShapesConfig()
    sizes = {k: 0 for k in range(N_BUCKETS)}
    names = {k: [] for k in range(N_BUCKETS)}
    order = []
    for name, kind, shp, bk in layer_table(cfg):
        if kind in ("conv", "convb"):
            n = [int(np.prod(shp))] + ([shp[3]] if kind == "convb" else [])
        elif kind == "dw":
            n = [int(np.prod(shp))]
        elif kind == "deconv":
            n = [int(np.prod(shp)), shp[2]]
        else:
            n = [shp, shp]
        sizes[bk] += sum((k + 3) // 4 * 4 for k in n)
        names[bk].append(name)
        order.append(bk)
    assert order == sorted(order), "buckets must be contiguous runs of the table (the flat buffer is laid out bucket by bucket)"
    total = sum(sizes.values())
    assert 7296031 <= total < 7296031 + 4 * 200                        # SURVEY.md Appendix B count + alignment padding
    assert names[BUCKET_BACKBONE][0] == "conv1" and names[BUCKET_BACKBONE][-1] == "conv_pw_6_bn"
    assert names[BUCKET_YOLO][0] == "conv_dw_7" and names[BUCKET_YOLO][-1] == "conv_23"
    assert names[BUCKET_FEATURE_MAP] == ["feature_map"]
    assert names[BUCKET_MASK_CONV1] == ["myolo_mask_conv1", "myolo_mask_bn1"]
    assert names[BUCKET_MASK_REST][0] == "myolo_mask_conv2" and names[BUCKET_MASK_REST][-1] == "myolo_mask"
    # mask head 10.5 MB, feature_map 4.7 MB, YOLO head 13.0 MB, backbone 1.0 MB
    mb = {k: 4 * v / 1e6 for k, v in sizes.items()}
    assert abs(mb[BUCKET_YOLO] - 13.0) < 0.2 and abs(mb[BUCKET_FEATURE_MAP] - 4.72) < 0.05 and abs(mb[BUCKET_BACKBONE] - 1.02) < 0.05
    assert abs(mb[BUCKET_MASK_CONV1] + mb[BUCKET_MASK_REST] - 10.5) < 0.1


@pytest.mark.parametrize("n,batch,world", [(50, 8, 2), (64, 8, 8), (500, 32, 4), (9, 8, 2), (100, 8, 3)])
def test_dp_batch_schedule_is_uniform_and_disjoint(n, batch, world):
    """MaskYOLO.train under torchrun: every rank runs the same number of steps (no rank left waiting in the all-reduce),
    on disjoint batches; one rank runs the reference's ceil(n/batch) steps per epoch (model.py:1048), the wrapped last
    batch included."""
    nb = -(-n // batch)
    per_rank = [dp_batch_indices(n, batch, r, world) for r in range(world)]
    assert len({len(p) for p in per_rank}) == 1
    flat = sorted(i for p in per_rank for i in p)
    assert flat == list(range(len(flat))) and len(flat) == nb - nb % world
    assert all(i < nb for i in flat)
    assert dp_batch_indices(n, batch, 0, 1) == list(range(nb))


def test_dp_batch_schedule_rejects_less_than_one_batch():
    with pytest.raises(ValueError):
        dp_batch_indices(7, 8, 0, 1)
