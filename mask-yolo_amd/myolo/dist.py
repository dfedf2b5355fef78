"""Data-parallel gradient exchange for the Mask-YOLO step: one process per GPU, one
all-reduce(sum) per gradient bucket over RCCL/xGMI, launched on a side stream as soon as the
backward pass has produced the bucket, so the exchange overlaps the rest of backward.

The reference has no distributed code (GPU_COUNT = 0 everywhere, config.py:47); the step
shards naturally over images: BN batch statistics and the loss
normalisers stay local to each replica, gradients are averaged (sum all-reduce, then 1/world
inside the fused Adam kernel).  Buckets are the five contiguous ranges of Net.flat_g (engine.layer_table), each released the moment
backward has produced it: YOLO blocks + conv_23 (under the mask head's forward, about half a step before the end) -> the mask head behind
conv1 (when the compact chain's weight gradients retire) -> myolo_mask_conv1 + bn1 -> feature_map -> backbone.

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
    """Batch indices rank `rank` trains on in one epoch.  The epoch has ceil(n_samples / batch) batches, as the reference's
    steps_per_epoch = len(generator) (model.py:1048; BatchGenerator wraps the last batch back so that it is full-size,
    myolo_utils.py:730-735).  Every rank runs the same number of steps (a rank that skipped one would leave the others waiting
    in the all-reduce): rank r takes batches r, r+world, r+2*world, ... and the n_batches % world trailing batches are dropped
    when world > 1.  Fewer samples than one batch is an error (the reference would train on a short batch; the engine's
    buffers are sized for BATCH_SIZE)."""
    if n_samples < batch:
        raise ValueError("%d samples cannot fill one batch of %d" % (n_samples, batch))
    nb = -(-n_samples // batch)
    if n_batches is not None:
        nb = min(nb, n_batches)
    return list(range(rank, nb - (nb % world), world))


class GradReducer(object):
    """Bucketed, overlapped sum all-reduce of a flat gradient buffer.

    bucket_ready(i) is called by the backward pass (Net.on_bucket_ready) when bucket i is complete;
    wait() is called before the optimiser (Net.before_optimizer).  With world_size 1 both are no-ops,
    so the single-GPU path is bit-identical to running without a reducer."""

    def __init__(self, flat_grad, bucket_ranges, group=None, always=False, backend="torch", timing=False, stream=None):
        """backend "torch": torch.distributed.all_reduce (nccl = RCCL on GPU tensors, gloo otherwise).
        backend "capi": the library's own myolo_comm_* entry points (include/myolo_hip.h) -- RCCL driven through the C-ABI,
        torch.distributed only carries the 128-byte unique id to the other ranks.  GPU tensors only.
        timing=True brackets every bucket's collective with HIP events on the comm stream (bucket_ms())."""
        self.flat = flat_grad
        self.ranges = list(bucket_ranges)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.cuda = flat_grad.is_cuda
        self.handles = []
        self.backend = backend
        self.timing = bool(timing) and self.cuda
        self.comm = None
        if backend not in ("torch", "capi"):
            raise ValueError("GradReducer backend must be 'torch' or 'capi' (got %r)" % (backend,))
        # always=True issues the collectives even in a 1-rank group (used to test the RCCL/stream path on one GPU)
        self.active = self.world > 1 or (always and (dist.is_initialized() or backend == "capi"))
        if self.cuda and self.active:
            # stream: an existing side stream to issue the collectives on (MaskYOLO passes the engine's copy stream, idle during backward).  A NEW
            # HIP stream is not free on this runtime: streams are multiplexed onto a few hardware queues and one more can put a side stream on
            # the compute stream's queue -- measured +3.4 ms per step with a fresh comm stream (bench comm probe 24.7 vs 21.3 ms), round 4.
            self.comm_stream = stream if stream is not None else torch.cuda.Stream(device=flat_grad.device, priority=-1)    # (high priority: engine._shared_stream)
            self.done = [torch.cuda.Event() for _ in self.ranges]
            # timing: one (start, end) event pair per collective, kept until bucket_ms() reads them -- nothing synchronises inside a step
            self._pairs = [[] for _ in self.ranges]
            self._ms = [[] for _ in self.ranges]
            # timing: when was bucket i released, relative to the end of backward (the wait() in front of the optimiser)?
            self._rel_open = [None for _ in self.ranges]
            self._rel_pairs = [[] for _ in self.ranges]
            self._rel_ms = [[] for _ in self.ranges]
        if backend == "capi" and self.active:
            if not self.cuda:
                raise ValueError("backend 'capi' (RCCL through the C-ABI) needs device tensors")
            self._init_capi()

    def _init_capi(self):
        import ctypes
        from . import _ext as X
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        ident = (ctypes.c_char * 128)()
        if rank == 0:
            X.call("myolo_comm_unique_id", ctypes.addressof(ident))
        if dist.is_initialized() and self.world > 1:
            box = [bytes(ident.raw)]
            # `rank` is the GROUP rank; broadcast takes a GLOBAL rank: the group's first member holds the id
            src = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.broadcast_object_list(box, src=src, group=self.group)
            ident = (ctypes.c_char * 128).from_buffer_copy(box[0])
        comm = ctypes.c_void_p()
        with torch.cuda.device(self.flat.device):
            X.call("myolo_comm_init", rank, self.world, ctypes.addressof(ident), ctypes.addressof(comm))
        self.comm = comm

    def ranks_seen(self):
        """number of ranks that actually take part in a collective on this reducer's transport (a sum of ones)."""
        if not self.active:
            return 1
        one = torch.ones(1, dtype=torch.float32, device=self.flat.device)
        if self.backend == "capi":
            from . import _ext as X
            X.call("myolo_allreduce_sum_f32", one.data_ptr(), 1, self.comm, torch.cuda.current_stream().cuda_stream)
        else:
            dist.all_reduce(one, op=dist.ReduceOp.SUM, group=self.group)
        return int(round(float(one.item())))

    def bucket_ms(self):
        """[mean ms of bucket i's all-reduce on the comm stream] since construction (timing=True; synchronises)."""
        if not (self.timing and self.active):
            return None
        self._collect()
        return [float(sum(v) / len(v)) if v else 0.0 for v in self._ms]

    def release_ms_before_wait(self):
        """[mean ms between bucket i's release (the event the collective waits for) and the end of backward (wait())] (timing=True; synchronises):
        how much of the step is left to hide the exchange behind."""
        if not (self.timing and self.active):
            return None
        self._collect()
        return [float(sum(v) / len(v)) if v else 0.0 for v in self._rel_ms]

    def _collect(self):
        torch.cuda.synchronize(self.flat.device)
        for i, pairs in enumerate(self._pairs):
            self._ms[i] += [a.elapsed_time(b) for a, b in pairs]
            del pairs[:]
        for i, pairs in enumerate(self._rel_pairs):
            self._rel_ms[i] += [a.elapsed_time(b) for a, b in pairs]
            del pairs[:]

    def close(self):
        """destroy the C-ABI RCCL communicator (idempotent; also run from __del__ so a dropped reducer does not leak it)."""
        comm, self.comm = getattr(self, "comm", None), None
        if comm is not None:
            from . import _ext as X
            torch.cuda.synchronize(self.flat.device)
            X.call("myolo_comm_destroy", comm)

    def __del__(self):
        try:
            self.close()
        except Exception:          # interpreter teardown: the library or torch may already be gone
            pass

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def bucket_ready(self, i):
        if not self.active:
            return
        lo, hi = self.ranges[i]
        view = self.flat[lo:hi]
        if self.cuda:
            ready = torch.cuda.Event(enable_timing=self.timing)
            ready.record(torch.cuda.current_stream())
            if self.timing:
                self._rel_open[i] = ready
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                if self.timing:
                    if len(self._pairs[i]) >= 512:           # nobody is reading: keep the most recent ones
                        del self._pairs[i][:256]
                    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    t0.record(self.comm_stream)
                if self.backend == "capi":
                    from . import _ext as X
                    X.call("myolo_allreduce_sum_f32", view.data_ptr(), hi - lo, self.comm, self.comm_stream.cuda_stream)
                else:
                    dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
                if self.timing:
                    t1.record(self.comm_stream)
                    self._pairs[i].append((t0, t1))
                self.done[i].record(self.comm_stream)
        else:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def wait(self):
        if not self.active:
            return
        if self.cuda:
            cur = torch.cuda.current_stream()
            if self.timing:
                end = torch.cuda.Event(enable_timing=True)
                end.record(cur)                   # the end of backward on the compute stream, before it waits for the exchange
                for i, ready in enumerate(self._rel_open):
                    if ready is not None:
                        if len(self._rel_pairs[i]) >= 512:
                            del self._rel_pairs[i][:256]
                        self._rel_pairs[i].append((ready, end))
                    self._rel_open[i] = None
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
        net.exchange_active = bool(self.active)       # the engine schedules the YOLO head's backward early when there is an exchange to hide
        return self
