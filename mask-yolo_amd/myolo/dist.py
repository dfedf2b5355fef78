"""Data-parallel gradient exchange for the Mask-YOLO step: one process per GPU, one
all-reduce(sum) per gradient bucket over RCCL/xGMI, launched on a side stream as soon as the
backward pass has produced the bucket, so the exchange overlaps the rest of backward.

The reference has no distributed code (GPU_COUNT = 0 everywhere, config.py:47); the step
shards naturally over images: BN batch statistics and the loss
normalisers stay local to each replica, gradients are averaged (sum all-reduce, then 1/world
inside the fused Adam kernel).  Buckets are the three contiguous ranges of Net.flat_g in the
order backward completes them: mask head -> YOLO head + feature_map -> backbone.

torch.distributed is plumbing here (backend "nccl" is RCCL on ROCm; "gloo" on CPU for tests).
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_range(global_batch, rank, world):
    """images [rank*per, (rank+1)*per) of the global batch (weak scaling: per-GPU batch fixed)."""
    per = global_batch // world
    assert per * world == global_batch, "global batch must divide evenly over ranks"
    return rank * per, (rank + 1) * per


def dp_batch_indices(n_samples, batch, rank, world, n_batches=None):
    """Batch indices rank `rank` trains on in one epoch: only FULL batches, the same count on every rank (a rank that skipped
    a step would leave the others waiting in the all-reduce), rank r taking batches r, r+world, r+2*world, ..."""
    n_full = n_samples // batch
    if n_batches is not None:
        n_full = min(n_full, n_batches)
    return list(range(rank, n_full - (n_full % world), world))


class GradReducer(object):
    """Bucketed, overlapped sum all-reduce of a flat gradient buffer.

    bucket_ready(i) is called by the backward pass (Net.on_bucket_ready) when bucket i is complete;
    wait() is called before the optimiser (Net.before_optimizer).  With world_size 1 both are no-ops,
    so the single-GPU path is bit-identical to running without a reducer."""

    def __init__(self, flat_grad, bucket_ranges, group=None, always=False):
        self.flat = flat_grad
        self.ranges = list(bucket_ranges)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = flat_grad.is_cuda
        self.handles = []
        # always=True issues the collectives even in a 1-rank group (used to test the RCCL/stream path on one GPU)
        self.active = self.world > 1 or (always and dist.is_initialized())
        if self.cuda and self.active:
            self.comm_stream = torch.cuda.Stream(device=flat_grad.device)
            self.done = [torch.cuda.Event() for _ in self.ranges]

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def bucket_ready(self, i):
        if not self.active:
            return
        lo, hi = self.ranges[i]
        view = self.flat[lo:hi]
        if self.cuda:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                self.done[i].record(self.comm_stream)
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        if not self.active:
            return
        if self.cuda:
            cur = torch.cuda.current_stream()
            for ev in self.done:
                cur.wait_event(ev)
        else:
            for h in self.handles:
                h.wait()
            self.handles = []

    def attach(self, net):
        """hook into a myolo.engine.Net."""
        net.on_bucket_ready = self.bucket_ready
        net.before_optimizer = self.wait
        net.grad_scale = self.grad_scale
        return self
