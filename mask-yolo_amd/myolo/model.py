"""
``myolo.model.MaskYOLO`` -- the reference's Python class surface (model.py:761-1391) over the
MI355X-native engine (myolo/engine.py -> libmyolo_hip.so).

Kept: constructor ``MaskYOLO(mode, config, model_dir=None, yolo_pretrain_dir=None,
yolo_trainable=True)`` (model.py:767-785), ``build`` (:787), ``train`` (:943-944), ``compile``
(:1062), ``set_trainable`` (:1120), ``load_weights`` (:1157), ``infer_yolo`` (:1198), ``detect``
(:1238, returns ``[{bboxes, class_ids, confidence_scores, full_masks}]`` :1316-1321),
``decode_masks`` (:1330), and a ``keras_model`` attribute offering ``predict`` / ``summary``.

Deliberate differences: one finalized config object everywhere (the
reference reads the base class from free functions, model.py:25); ``train`` uses every sample of
the dataset unless ``max_samples`` is given (reference hard-codes 50/6, model.py:995,1002);
``detect`` keeps the NMB result (reference overrides it with [109,130], model.py:1306), returns ``bboxes`` in
PIXELS of the input image (x by its width, y by its height; the reference hard-codes ``* 224``, model.py:1307) and
does not mutate ``Config.BATCH_SIZE`` (model.py:1268); ``load_weights`` reads Keras ``.h5`` weight files with a built-in
pure-Python HDF5 reader (no h5py in this image) as well as ``.npz``; ``train`` runs the validation pass Keras' fit_generator runs
(model.py:1053-1054) and keeps the per-epoch numbers in ``self.history``; ``custom_callbacks`` (commented out in the
reference, model.py:1037-1038) are honoured as plain callables ``cb(epoch, logs)``; checkpoints are ``.npz`` keyed by Keras layer names
(and Keras ``.h5`` files are read too); weights are cached by (path, mtime) instead of reloaded per call.
"""
import collections.abc
import contextlib
import datetime
import gc
import os
import queue
import re
import threading
import time

import numpy as np
import torch

from . import myolo_utils as mutils
from .engine import Net, layer_table


class _KerasModelShim(object):
    """The two ``keras_model`` methods the reference's scripts call
    (infer_shapes_yolo_model.py:16, train_rice.py:44)."""

    def __init__(self, owner):
        self._o = owner

    def predict(self, inputs, verbose=0):
        o = self._o
        images = inputs[0] if isinstance(inputs, (list, tuple)) else inputs
        x = torch.as_tensor(np.ascontiguousarray(images, np.float32), device=o.net.dev)
        if o.mode == 'yolo':
            return [o.net.predict_yolo(x).cpu().numpy()]
        if o.mode == 'inference':
            return [t.cpu().numpy() for t in o.net.predict(x)]
        raise RuntimeError("predict() on a training-mode model: use train_on_batch()")

    def summary(self):
        lines = ["%-24s %-8s %s" % (n, k, s) for n, k, s, _ in layer_table(self._o.config)]
        txt = "\n".join(lines) + "\ntrainable parameters: %d" % self._o.net.nparam
        print(txt)
        return txt


class StepResult(collections.abc.Mapping):
    """What one training step hands back, without stopping the pipeline.

    Keras' train_on_batch / fit_generator return loss values, not tensors (model.py:1047-1059); rounds 1-3 copied every
    output of the training graph (model.py:899: ~100 MB at config 2) to the host after every step.  Here a step ends with ONE
    small asynchronous device->pinned copy (the loss terms + the per-image positive counts) and an event; the scalar keys
    ('loss', 'yolo_sum_loss', 'mask_loss', 'loss_xy', 'loss_wh', 'loss_conf', 'loss_class', 'recall') wait for that event
    when first read, and the tensor keys ('yolo_output', 'yolo_proposals', 'output_rois', 'myolo_mask', 'target_class_ids',
    'target_mask', 'n_pos', 'feature_map') are downloaded only when indexed (the device tensors stay referenced here, so
    reading them later still gives this step's values).  A read-only Mapping: out['loss'], out.get(...), dict(out) all work.

    Lifetime: while a StepResult is alive it keeps its step's output tensors on the DEVICE (~100 MB at config 2).  Code that collects results
    over many steps should keep numbers, not StepResults: `out["loss"]`, or `out.release()` once the scalars have been read (drops every device
    reference; tensor keys already fetched stay readable, the others then raise), or `out.as_dict()` for a plain, picklable / json-able dict
    of what has been read plus the scalars.  MaskYOLO.train() and train_shapes_stream() do this themselves (they keep the loss of step i-2)."""

    SCALARS = ("loss", "yolo_sum_loss", "mask_loss", "loss_xy", "loss_wh", "loss_conf", "loss_class", "recall")

    def __init__(self, tensors, yolo_terms, mask_terms, loss_weights):
        self._t = dict(tensors)                       # key -> device tensor | None
        self._host = {}
        self._w = loss_weights
        ny = int(yolo_terms.numel())
        self._ny = ny
        self._pin = torch.empty(ny + 2, dtype=torch.float32, pin_memory=True)
        self._pin[:ny].copy_(yolo_terms, non_blocking=True)
        if mask_terms is not None:
            self._pin[ny:ny + 2].copy_(mask_terms, non_blocking=True)
        self._has_mask = mask_terms is not None
        self._ev = torch.cuda.Event()
        self._ev.record(torch.cuda.current_stream())
        self._sc = None

    def _scalars(self):
        if self._sc is None:
            self._ev.synchronize()
            v = self._pin.numpy()
            yt = v[:self._ny]
            m0 = float(v[self._ny]) if self._has_mask else 0.0
            w1, w2 = self._w
            self._sc = dict(yolo_sum_loss=float(yt[0]), mask_loss=m0, loss=float(yt[0] * w1 + m0 * w2) if self._has_mask
                            else float(yt[0] * w1), loss_xy=float(yt[1]), loss_wh=float(yt[2]), loss_conf=float(yt[3]),
                            loss_class=float(yt[4]), recall=float(yt[5]))
        return self._sc

    def ready(self):
        """True when the loss scalars can be read without waiting for the GPU."""
        return self._sc is not None or self._ev.query()

    def __getitem__(self, k):
        if k in self.SCALARS:
            return self._scalars()[k]
        if k not in self._t:
            raise KeyError(k)
        if k not in self._host:
            t = self._t[k]
            self._host[k] = None if t is None else t.cpu().numpy()
        return self._host[k]

    def device(self, k):
        """the device tensor behind a tensor key (no copy)."""
        return self._t[k]

    def release(self):
        """read the scalars (waits for the step's small copy), then drop every device tensor this result holds; returns self.  Tensor keys that
        were downloaded before stay available, the others raise KeyError afterwards."""
        self._scalars()
        self._t = {k: None for k in self._t if k in self._host}
        return self

    def as_dict(self, tensors=()):
        """a plain dict: the eight scalars, every tensor key already downloaded, and the keys named in `tensors` (downloaded now)."""
        d = dict(self._scalars())
        for k in tensors:
            self[k]
        d.update({k: v for k, v in self._host.items()})
        return d

    def __iter__(self):
        return iter(list(self._t.keys()) + list(self.SCALARS))

    def __len__(self):
        return len(self._t) + len(self.SCALARS)


@contextlib.contextmanager
def _gc_parked():
    """the cyclic collector parked for the duration of a training loop (a gen-2 pass over torch's object graph is a
    multi-ms host stall in the middle of the launch sequence); collected once on the way out.  The interpreter's thread switch
    interval is shortened meanwhile: the launch thread and the batch-prefetch thread share the GIL, and the default 5 ms is a
    quarter of a step."""
    import sys
    was = gc.isenabled()
    gc.collect()
    gc.freeze()                   # everything alive now (the model, the dataset) is long-lived: keep it out of the closing collection,
    gc.disable()                  # which then only walks what the loop itself left behind (a full pass cost ~40 ms per epoch of 16 steps)
    sw = sys.getswitchinterval()
    sys.setswitchinterval(min(sw, 0.0005))
    try:
        yield
    finally:
        sys.setswitchinterval(sw)
        if was:
            gc.enable()
        gc.collect()
        gc.unfreeze()


class _Prefetcher(object):
    """Runs `make(i)` for the items of `schedule` one step ahead on a thread (Keras' fit_generator does the same with its
    generator queue, model.py:1055-1058: max_queue_size 3, one worker).  make = BatchGenerator.fill into the engine's pinned
    staging arrays + the asynchronous upload (Net.stage_batch): numpy's copies and the event waits release the GIL, so this
    overlaps the main thread's kernel launches.  close() (train() calls it on every way out: normal end, exception in a step or
    a callback, KeyboardInterrupt) stops the thread and drops what it had staged."""

    def __init__(self, schedule, make, device, depth=2):
        self._q = queue.Queue(maxsize=depth)
        self._err = None
        self._stop = threading.Event()

        def put(item):
            while not self._stop.is_set():
                try:
                    self._q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        def run():
            try:
                if device is not None:
                    torch.cuda.set_device(device)
                for i in schedule:
                    if self._stop.is_set() or not put((i, make(i))):
                        return
            except BaseException as e:                # surfaced in the consumer
                self._err = e
            finally:
                put(None)
        self._th = threading.Thread(target=run, name="myolo-batch-prefetch", daemon=True)
        self._th.start()

    def __iter__(self):
        while True:
            item = self._q.get()
            if item is None:
                if self._err is not None:
                    raise self._err
                return
            yield item

    def close(self, timeout=10.0):
        """stop the thread (it leaves at its next queue operation), drain the queue so that the staged device batches are released, join."""
        self._stop.set()
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        self._th.join(timeout)
        try:
            while True:
                self._q.get_nowait()
        except queue.Empty:
            pass
        return not self._th.is_alive()


class MaskYOLO(object):
    def __init__(self, mode, config, model_dir=None, yolo_pretrain_dir=None, yolo_trainable=True, device="cuda:0", seed=0):
        assert mode in ['training', 'inference', 'yolo']
        self.mode = mode
        self.config = config.finalize() if hasattr(config, "finalize") else config
        self.model_dir = model_dir
        self.yolo_pretrain_dir = yolo_pretrain_dir
        self.yolo_trainable = yolo_trainable
        self._device, self._seed = device, seed
        self._weights_cache = None
        self._lr = None
        self._trainable_regex = ".*"
        self.net = self.build(mode=mode, config=self.config)
        self.keras_model = _KerasModelShim(self)
        self.epoch = 0
        self.host_times = {"wait_for_batch_s": 0.0, "launch_step_s": 0.0, "steps": 0}    # train(): where the launch thread's wall time goes

    # ------------------------------------------------------------------ build
    def build(self, mode, config):
        assert mode in ['training', 'inference', 'yolo']
        w, h = config.IMAGE_SHAPE[:2]
        if w % 32 != 0 or h % 32 != 0:
            raise Exception("Image size must be dividable by 32 to adapt with YOLO framework. "
                            "For example, use 224, 256, 288, 320, 356, ... etc. ")
        assert config.BACKBONE == "mobilenet"          # model.py:62
        net = Net(config, device=self._device, seed=self._seed)
        if self.yolo_pretrain_dir is not None:          # model.py:854-868
            self._load_npz_into(net, self.yolo_pretrain_dir, by_name=True)
        return net

    # ------------------------------------------------------------------ weights
    def state_dict(self):
        return self.net.state_dict()

    def load_state_dict(self, sd, strict=True):
        self.net.load_state_dict(sd, strict=strict)

    @staticmethod
    def _load_npz_into(net, filepath, by_name=False, exclude=None):
        """.npz keyed '<keras layer name>/<weight name>', or a Keras .h5 / .hdf5 weight file (keras_io.load_h5_state: the
        reference's checkpoint format, model.py:1024-1027, read without h5py)."""
        if str(filepath).lower().endswith((".h5", ".hdf5")):
            from .keras_io import load_h5_state
            sd = load_h5_state(filepath, exclude=exclude)
        else:
            data = np.load(filepath)
            sd = {}
            for k in data.files:
                layer = k.split("/")[0]
                if exclude and layer in exclude:
                    continue
                sd[k] = data[k]
        net.load_state_dict(sd, strict=not (by_name or exclude))

    def load_weights(self, filepath, by_name=False, exclude=None):
        """model.py:1157-1196: Keras weight files (.h5, as the reference's ModelCheckpoint writes them) or the .npz container
        keyed '<keras layer name>/<weight name>'.  by_name=True loads the layers present in the file and leaves the others;
        exclude drops layers by name (and implies by_name, model.py:1181-1187)."""
        if exclude:
            by_name = True
        key = (os.path.abspath(filepath), os.path.getmtime(filepath), by_name, tuple(exclude or ()))
        if self._weights_cache == key:
            return
        self._load_npz_into(self.net, filepath, by_name=by_name, exclude=exclude)
        self._weights_cache = key

    def save_weights(self, filepath):
        np.savez(filepath, **self.net.state_dict())

    # ------------------------------------------------------------------ training
    def set_trainable(self, layer_regex, keras_model=None, indent=0, verbose=1):
        """model.py:1120-1151: layers whose name fully matches the regex are trained; the
        others keep their weights (their gradient is zeroed before Adam)."""
        self._trainable_regex = layer_regex
        mask = torch.zeros_like(self.net.flat_p)
        for k, (off, shp) in self.net.pslots.items():
            layer = k.split("/")[0]
            trainable = bool(re.fullmatch(layer_regex, layer))
            if self.yolo_pretrain_dir is not None and not self.yolo_trainable and not layer.startswith("myolo_mask") \
                    and layer != "feature_map":
                trainable = False                      # model.py:867-868
            if trainable:
                mask[off:off + int(np.prod(shp))] = 1.0
        self._train_mask = None if bool(mask.min() > 0) else mask

    def compile(self, learning_rate, momentum):
        """model.py:1062-1118: Adam(lr, 0.9, 0.999, 1e-8); loss = sum of the batch-mean losses
        times LOSS_WEIGHTS.  (momentum is unused by the reference too.)"""
        self._lr = float(learning_rate)
        # the reference instantiates a new keras.optimizers.Adam on every compile() (model.py:1071): fresh moments and
        # step count.  (This is also what keeps frozen layers still: zero gradient AND zero momentum.)
        self.net.flat_m.zero_()
        self.net.flat_v.zero_()
        self.net.adam_t = 0

    def train_on_batch(self, batch, learning_rate=None):
        """One optimisation step on a host batch (the six arrays of model.py:896-897) or an already staged device batch.
        Returns a StepResult -- a read-only Mapping of the reference's training outputs (model.py:899) and the loss scalars: scalars from one
        small asynchronous copy (first read waits for the step), tensors downloaded when indexed.  It keeps the step's output tensors on the
        device until dropped: see StepResult (release / as_dict) before collecting results over many steps."""
        lr = self._lr if learning_rate is None else learning_rate
        if lr is None:
            lr = self.config.LEARNING_RATE
        net = self.net
        db = batch if isinstance(batch, dict) else net.to_device_batch(batch)
        yolo_only = self.mode == 'yolo' or "gt_masks" not in db
        out = net.forward_backward_yolo(db) if yolo_only else net.forward_backward(db)
        if getattr(self, "_train_mask", None) is not None:
            # data-parallel: the bucketed all-reduces run in place on flat_g on the reducer's stream -- join them BEFORE
            # masking, or RCCL may overwrite zeroed entries / read half-masked data (frozen layers would then move)
            if net.before_optimizer:
                net.before_optimizer()
            net.flat_g.mul_(self._train_mask)          # torch used as a memory op on a flag vector only
        net.adam_step(lr)
        if yolo_only:
            return StepResult(dict(yolo_output=out["yolo_output"]), out["yolo_terms"], None, out["loss_weights"])
        return self._host_outputs(out)

    @staticmethod
    def _host_outputs(out):
        """-> StepResult: the loss scalars from one small async copy, the graph's tensors on demand."""
        return StepResult({k: out[k] for k in ("yolo_output", "yolo_proposals", "output_rois", "myolo_mask", "target_class_ids",
                                               "target_mask", "n_pos", "feature_map")},
                          out["yolo_terms"], out["mask_terms"], out["loss_weights"])

    def _data_parallel(self):
        """(rank, world): when torch.distributed is initialised with more than one rank (myolo.dist.init_from_env under
        torchrun), hook the bucketed gradient all-reduce into the engine once and shard the work by rank -- every rank holds
        the same seeded weights, trains on its own BATCH_SIZE images per step (global batch = world * BATCH_SIZE) and applies
        the same averaged update.  Single process: (0, 1), nothing attached."""
        import torch.distributed as tdist
        if not (tdist.is_available() and tdist.is_initialized()) or tdist.get_world_size() == 1:
            return 0, 1
        if getattr(self, "_reducer", None) is None:
            from .dist import GradReducer
            self._reducer = GradReducer(self.net.flat_g, self.net.bucket_ranges, stream=self.net._copy_stream).attach(self.net)
        return tdist.get_rank(), tdist.get_world_size()

    def evaluate_on_batch(self, batch):
        """Forward-only losses of one host batch in Keras' test phase (every BatchNormalization on its moving statistics):
        what fit_generator's validation pass computes per batch (model.py:1053-1054).  No state is changed."""
        net = self.net
        db = batch if isinstance(batch, dict) else net.to_device_batch(batch)
        out = net.forward_loss(db)
        yt = out["yolo_terms"].cpu().numpy()
        mt = out["mask_terms"].cpu().numpy()
        w1, w2 = out["loss_weights"]
        return dict(loss=float(yt[0] * w1 + mt[0] * w2), yolo_sum_loss=float(yt[0]), mask_loss=float(mt[0]),
                    loss_xy=float(yt[1]), loss_wh=float(yt[2]), loss_conf=float(yt[3]), loss_class=float(yt[4]), recall=float(yt[5]))

    def train(self, train_dataset, val_dataset, learning_rate, epochs, layers,
              augmentation=None, custom_callbacks=None, no_augmentation_sources=None, max_samples=None, verbose=1, shuffle_seed=0):
        """model.py:943-1060.  Returns the list of per-epoch mean training losses; ``self.history`` holds
        {"loss": [...], "val_loss": [...]} like the Keras History the reference's fit_generator produces.
        shuffle_seed: the generators' one-off shuffle (myolo_utils.py:711-712) draws from RandomState(shuffle_seed) -- the
        same permutation on every data-parallel rank, so that dp_batch_indices really hands out disjoint samples; None =
        the global numpy generator, as the reference (single process only)."""
        layer_regex = {"all": ".*"}
        if layers in layer_regex:
            layers = layer_regex[layers]
        cfg = self.config

        def collect(ds):
            ids = list(ds.image_ids)
            if max_samples is not None:
                ids = ids[:max_samples]
            return [list(mutils.load_image_gt(ds, cfg, i, use_mini_mask=cfg.USE_MINI_MASK)) for i in ids]

        train_info = collect(train_dataset)
        val_info = collect(val_dataset) if val_dataset is not None else []
        mode = self.mode if self.mode in ('yolo', 'training') else 'training'
        rank, world = self._data_parallel()
        if world > 1 and shuffle_seed is None:
            raise ValueError("data-parallel train() needs a shuffle_seed: every rank must hold the same permutation")
        rng = None if shuffle_seed is None else np.random.RandomState(shuffle_seed)
        train_gen = mutils.BatchGenerator(train_info, cfg, mode=mode, shuffle=True, jitter=False, norm=True, rng=rng)
        val_gen = mutils.BatchGenerator(val_info, cfg, mode=mode, shuffle=True, jitter=False, norm=True, rng=rng) if val_info else None
        self.set_trainable(layers)
        self.compile(learning_rate, cfg.LEARNING_MOMENTUM)
        from .dist import dp_batch_indices
        schedule = dp_batch_indices(len(train_info), cfg.BATCH_SIZE, rank, world, len(train_gen))   # raises if < one batch
        history = []
        self.history = {"loss": [], "val_loss": []}
        n_steps = len(train_gen)

        bytes_ok = bool(train_info) and all(inst[0].dtype == np.uint8 for inst in train_info)

        def make(i):                                   # runs on the prefetch thread, one batch ahead of the step
            # BatchGenerator.__getitem__'s arrays (the wrapped last batch is full-size, myolo_utils.py:730-735) encoded straight into
            # the engine's pinned staging buffers; images travel as bytes and are normalised on the device
            lo, hi = train_gen.batch_bounds(i)
            return self.net.stage_batch(lambda arrays: train_gen.fill(i, arrays), hi - lo, yolo=(mode == 'yolo'), u8_images=bytes_ok)

        def report(ep, i, out):
            losses.append(out["loss"])
            if verbose:
                print("epoch %d step %d/%d loss %.4f (yolo %.4f mask %.4f recall %.3f)" %
                      (ep + 1, i + 1, n_steps, out["loss"], out["yolo_sum_loss"], out["mask_loss"], out["recall"]))

        # ONE prefetcher for the whole run (Keras' generator queue also keeps running across epoch ends): the first batches of epoch e+1
        # are encoded and uploaded while epoch e finishes
        prefetch = _Prefetcher([(e, i) for e in range(epochs) for i in schedule], lambda it: make(it[1]), self.net.dev)
        stream = iter(prefetch)
        try:
            with _gc_parked():                          # for the whole run: a full collection at every epoch end cost ~50 ms per epoch
                for ep in range(epochs):
                    losses = []
                    pending = None
                    for _ in schedule:
                        t0 = time.perf_counter()
                        (_, i), db = next(stream)
                        t1 = time.perf_counter()
                        out = self.train_on_batch(db)
                        self.host_times["wait_for_batch_s"] += t1 - t0      # the launch thread blocked on the prefetch queue
                        self.host_times["launch_step_s"] += time.perf_counter() - t1
                        self.host_times["steps"] += 1
                        if pending is not None:            # step i-1's numbers are read once step i is queued behind it: the
                            report(ep, *pending)           # host never waits for the step it has just launched
                        pending = (i, out)
                    if pending is not None:
                        report(ep, *pending)
                    gc.collect(0)                          # the young generation only (cheap): what this epoch's steps left behind
                    history.append(float(np.mean(losses)))
                    logs = {"loss": history[-1]}
                    if val_gen is not None and len(val_info) >= cfg.BATCH_SIZE:
                        # validation_data=val_generator, validation_steps=len(val_generator) (model.py:1053-1054): forward only, BN on
                        # moving statistics, batch-mean of the total loss.  Every rank evaluates the same (small) set.
                        vl = [self.evaluate_on_batch(val_gen[j][0])["loss"] for j in range(len(val_gen))]
                        logs["val_loss"] = float(np.mean(vl))
                        if verbose:
                            print("epoch %d val_loss %.4f" % (ep + 1, logs["val_loss"]))
                    self.history["loss"].append(logs["loss"])
                    self.history["val_loss"].append(logs.get("val_loss", float("nan")))
                    if self.model_dir and rank == 0:
                        os.makedirs(self.model_dir, exist_ok=True)
                        stamp = datetime.datetime.now().strftime('%b%d-%H-%M')
                        self.save_weights(os.path.join(self.model_dir, 'saved_model_' + stamp + '.npz'))   # model.py:1026
                    for cb in (custom_callbacks or []):
                        (cb.on_epoch_end if hasattr(cb, "on_epoch_end") else cb)(ep, dict(logs))
            for _ in stream:                               # (drains the prefetch thread: nothing is left when every epoch ran)
                pass
        finally:
            prefetch.close()       # every way out (a failing step or callback, Ctrl-C): no thread is left staging into the engine's ring

        self.epoch = max(self.epoch, epochs)
        return history

    def train_shapes_stream(self, steps, learning_rate=None, seed=1234, start_index=0, verbose=0):
        """Train on the endless synthetic Shapes stream with the inputs produced ON THE DEVICE
        (myolo.shapes.ShapesProducer): image g of the stream is bit-identical to what
        ShapesDataset/load_image_gt/BatchGenerator would feed, without the Python rasteriser in the loop."""
        from .shapes import ShapesProducer
        cfg = self.config
        prod = ShapesProducer(cfg, seed=seed, device=self._device)
        self.set_trainable(".*")
        self.compile(cfg.LEARNING_RATE if learning_rate is None else learning_rate, cfg.LEARNING_MOMENTUM)
        rank, world = self._data_parallel()          # rank r takes images [r*B, (r+1)*B) of each global batch of world*B
        results = []
        import torch
        side = self.net._copy_stream                 # batch i+1 is produced there while step i runs (its kernels are a few hundred microseconds)

        def produce(i):
            lo = start_index + (i * world + rank) * cfg.BATCH_SIZE
            return prod.batch(list(range(lo, lo + cfg.BATCH_SIZE)), stream=side, consumer=torch.cuda.current_stream())
        with _gc_parked():
            nxt = produce(0) if steps else None
            for i in range(steps):
                cur, nxt = nxt, (produce(i + 1) if i + 1 < steps else None)
                results.append(self.train_on_batch(cur))
                if verbose and i:
                    print("step %d loss %.4f" % (i, results[i - 1]["loss"]))      # one step behind: no wait on the step in flight
                if i >= 2:
                    results[i - 2] = results[i - 2]["loss"]                          # long finished: keep the number, drop the tensors
        losses = [r if isinstance(r, float) else r["loss"] for r in results]
        if verbose and steps:
            print("step %d loss %.4f" % (steps, losses[-1]))
        return losses

    # ------------------------------------------------------------------ inference
    def infer_yolo(self, image, weights_dir=None, save_path=None, display=False):
        """model.py:1198-1236 without the plotting: returns the decoded BoundBox list."""
        cfg = self.config
        assert list(image.shape) == list(cfg.IMAGE_SHAPE)
        assert image.dtype == 'uint8'
        assert self.mode == 'yolo'
        if weights_dir is not None:
            self.load_weights(weights_dir)
        normed = np.expand_dims(image / 255., axis=0)
        netout = self.keras_model.predict([normed])[0]
        return mutils.decode_one_yolo_output(netout[0], anchors=cfg.ANCHORS, nms_threshold=0.3, obj_threshold=0.35,
                                             nb_class=cfg.NUM_CLASSES)

    def detect(self, image, weights_dir=None, save_path='./img_results/', cs_threshold=0.35, display=False):
        """model.py:1238-1328.  Returns [ {bboxes, class_ids, confidence_scores, full_masks} ]."""
        cfg = self.config
        assert list(image.shape) == list(cfg.IMAGE_SHAPE)
        assert image.dtype == 'uint8'
        assert self.mode == 'inference'
        if weights_dir is not None:
            self.load_weights(weights_dir)
        # (image / 255.).astype(float32) formed on the device from the uploaded bytes (myolo_u8_to_unit_f32: the float32 of the float64 quotient, as numpy
        # forms it on the host) -- bit-identical, a quarter of the bytes over PCIe and no 4 MB float64 temporary on the host (see detect_many)
        from . import _ext as X
        raw = torch.from_numpy(np.ascontiguousarray(image)).to(self.net.dev)
        x = torch.empty((1,) + tuple(raw.shape), dtype=torch.float32, device=self.net.dev)
        X.call("myolo_u8_to_unit_f32", X.ptr(raw), X.ptr(x), raw.numel(), X.stream())
        selected_only = bool(getattr(cfg, "DETECT_MASKS_FOR_SELECTED_ONLY", False))
        if selected_only:
            yolo_output, det_d, feature = self.net.predict_detections(x)   # the mask head runs below, on the survivors only
            mask_d = None
        else:
            predict = self.net.predict_graphed if getattr(cfg, "INFERENCE_HIP_GRAPH", True) else self.net.predict
            yolo_output, det_d, mask_d = predict(x)                        # device tensors
        return [self._select_and_unmold(det_d[0], None if mask_d is None else mask_d[0], image.shape, cs_threshold,
                                        feature=feature if selected_only else None)]

    def detect_many(self, images, weights_dir=None, cs_threshold=0.35, in_flight=3):
        """detect() for a sequence of images at the throughput of Net.predict_stream: the images go through the inference graph in batches of
        config.BATCH_SIZE with `in_flight` batches running at once (one stream / hipGraph / scratch per lane: the launch-bound trunk of one
        batch under the matrix-pipe-bound mask head of another), and every image gets detect()'s own selection and unmolding
        (model.py:1238-1328).  Returns one result dict per image, equal to detect(image)[0]."""
        cfg = self.config
        assert self.mode == 'inference'
        if weights_dir is not None:
            self.load_weights(weights_dir)
        images = list(images)
        for im in images:
            assert list(im.shape) == list(cfg.IMAGE_SHAPE) and im.dtype == 'uint8'
        B = int(cfg.BATCH_SIZE)
        dev = self.net.dev

        from . import _ext as X
        # the bytes go up as they are (a quarter of the float32 image) from a ring of pinned buffers, and (image / 255.).astype(float32) is formed on the
        # device (myolo_u8_to_unit_f32: the float32 of the float64 quotient, which is what numpy does on the host) -- bit-identical to detect()'s input (the host
        # normalisation + pageable upload of 8 MB per batch of four 416 x 416 images cost 2.3 ms per batch, more than half the forward)
        key = (B, tuple(cfg.IMAGE_SHAPE), 2 * in_flight + 2)
        if getattr(self, "_stage_ring_key", None) != key:            # (pinning costs ~5 ms per buffer: once per model, not once per call)
            self._stage_ring = [torch.empty((B,) + tuple(cfg.IMAGE_SHAPE), dtype=torch.uint8).pin_memory() for _ in range(key[2])]
            self._stage_ring_key = key
        ring = self._stage_ring

        def batches():
            for bi, lo in enumerate(range(0, len(images), B)):
                grp = images[lo:lo + B]
                grp = grp + [grp[-1]] * (B - len(grp))                       # a short last batch is padded (the captured graph has one shape)
                stage = ring[bi % len(ring)]             # (predict_stream runs at most 2 * in_flight batches ahead of the results handed out)
                np.stack(grp, out=stage.numpy())
                raw = stage.to(dev, non_blocking=True)
                x = torch.empty(raw.shape, dtype=torch.float32, device=dev)
                X.call("myolo_u8_to_unit_f32", X.ptr(raw), X.ptr(x), raw.numel(), X.stream())       # one launch (three float64 torch kernels took 0.37 ms per batch)
                yield x
        out = []
        for bi, (_, det_d, mask_d) in enumerate(self.net.predict_stream(batches(), in_flight=in_flight)):
            det_all = det_d.cpu().numpy()
            for k in range(min(B, len(images) - bi * B)):
                out.append(self._select_and_unmold(det_d[k], mask_d[k], images[bi * B + k].shape, cs_threshold, det_host=det_all[k]))
        return out
        # (one unmold launch and one download per BATCH instead of per image was measured: 514 img/s against 865 -- a pageable 6.9 MB download runs at
        # 2 GB/s where four of 1.7 MB do not, and cutting the H x W x 40 block into per-image H x W x n arrays costs the host another 0.5 ms per image)

    def _select_and_unmold(self, det_img, mask_img, image_shape, cs_threshold, feature=None, det_host=None):
        """detect()'s post-processing of ONE image: det_img [R,6], mask_img [R,mh,mw,C] (None: the mask head runs here, on the survivors) device tensors."""
        cfg = self.config
        det_d, mask_d = det_img.unsqueeze(0), None if mask_img is None else mask_img.unsqueeze(0)
        selected_only = mask_img is None
        # decode_masks (model.py:1330-1391) unmolds every box and the caller then keeps <= 10 of them (model.py:1290-1304);
        # the selection needs only boxes / scores / classes, so it runs first and only the survivors are unmolded
        # (same output: full_masks[:, :, nmb] of the all-box result).
        det_h = det_d[0].cpu().numpy() if det_host is None else det_host          # (detect_many: one download per batch)
        boxes, scores, class_ids = det_h[:, :4], det_h[:, 4], det_h[:, 5].astype(np.int32)
        keep = np.where((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) > 0)[0]     # model.py:1373-1380
        boxes, scores, class_ids = boxes[keep], scores[keep], class_ids[keep]
        top10 = np.argsort(scores)[::-1][:10]
        kept = np.array([i for i in top10 if scores[i] >= cs_threshold], dtype=np.int64)
        nmb = mutils.NMB(boxes[kept], class_ids[kept], kept, cfg.IMAGE_SHAPE, nms_threshold=0.7) if len(kept) else kept
        nmb = np.asarray(nmb, dtype=np.int64)
        if len(nmb):
            # (the few indices go up from a pinned buffer without blocking; the download of the pasted masks below ends every image with a
            # synchronisation, so the buffer is free again by the next one)
            pin = getattr(self, "_sel_pin", None)
            if pin is None or pin.numel() < len(nmb):
                pin = self._sel_pin = torch.empty(max(64, len(nmb)), dtype=torch.int64).pin_memory()
            pin[:len(nmb)].copy_(torch.from_numpy(np.ascontiguousarray(keep[nmb], dtype=np.int64)))
            sel = pin[:len(nmb)].to(det_d.device, non_blocking=True)
            det_s = det_d[0].index_select(0, sel).contiguous()
            if selected_only:
                mask_s = self.net.predict_masks(feature, det_s[:, :4].unsqueeze(0))[0]
            else:
                mask_s = mask_d[0].index_select(0, sel).contiguous()
            _, _, _, full_masks = self._decode_masks_device(det_s, mask_s, image_shape, det_host=det_h[keep[nmb]])
        else:
            full_masks = np.empty((int(image_shape[0]), int(image_shape[1]), 0), dtype=bool)
        H_img, W_img = float(image_shape[0]), float(image_shape[1])
        return {
            "bboxes": boxes[nmb] * np.array([W_img, H_img, W_img, H_img], dtype=boxes.dtype),     # pixels (model.py:1307)
            "class_ids": class_ids[nmb],
            "confidence_scores": scores[nmb],
            "full_masks": full_masks,
        }

    def _decode_masks_device(self, det, masks, image_shape, det_host=None):
        """decode_masks (model.py:1330-1391) with the unmold/paste of every detection on the GPU
        (myolo_unmold_masks).  det [N,6], masks [N,mh,mw,C] device tensors of one image (det_host: det's host copy, if the caller has it)."""
        from . import _ext as X
        det_h = det.cpu().numpy() if det_host is None else det_host
        boxes, scores = det_h[:, :4], det_h[:, 4]
        class_ids = det_h[:, 5].astype(np.int32)
        keep = np.where((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) > 0)[0]     # model.py:1373-1380
        if keep.shape[0] != det_h.shape[0]:
            idx = torch.as_tensor(keep, device=det.device)
            det, masks = det.index_select(0, idx).contiguous(), masks.index_select(0, idx).contiguous()
            boxes, scores, class_ids = boxes[keep], scores[keep], class_ids[keep]
        N = int(det.shape[0])
        H, W = int(image_shape[0]), int(image_shape[1])
        if N == 0:
            return boxes, class_ids, scores, np.empty((H, W, 0))
        mh, mw, C = int(masks.shape[1]), int(masks.shape[2]), int(masks.shape[3])
        full = torch.empty(H, W, N, dtype=torch.uint8, device=det.device)
        ws = torch.empty(N, dtype=torch.int32, device=det.device)
        X.call("myolo_unmold_masks", X.ptr(masks.contiguous()), X.ptr(det.contiguous()), X.ptr(full), N, mh, mw, C, H, W,
               ws.data_ptr(), ws.numel() * 4, X.stream())
        # (a pinned staging buffer for this download was measured slower: the host's astype() then reads uncached memory)
        return boxes, class_ids, scores, full.view(torch.bool).cpu().numpy()        # (the kernel writes 0 / 1 bytes: the bool view saves the host's astype pass)

    def decode_masks(self, detections, myolo_mask, image_shape):
        """model.py:1330-1391 (numpy in / numpy out; the unmolding runs on the GPU)."""
        assert len(detections) == 1
        assert len(myolo_mask) == 1
        assert list(image_shape) == list(self.config.IMAGE_SHAPE)
        if torch.cuda.is_available():
            dev = self.net.dev
            return self._decode_masks_device(torch.as_tensor(np.ascontiguousarray(detections[0], np.float32), device=dev),
                                             torch.as_tensor(np.ascontiguousarray(myolo_mask[0], np.float32), device=dev), image_shape)
        detection, masks_all = detections[0], myolo_mask[0]
        N = len(detection)
        boxes = detection[:N, :4]
        scores = detection[:N, 4]
        class_ids = detection[:N, 5].astype(np.int32)
        masks = masks_all[np.arange(N), :, :, class_ids]
        exclude_ix = np.where((boxes[:, 2] - boxes[:, 0]) * (boxes[:, 3] - boxes[:, 1]) <= 0)[0]
        if exclude_ix.shape[0] > 0:
            boxes = np.delete(boxes, exclude_ix, axis=0)
            class_ids = np.delete(class_ids, exclude_ix, axis=0)
            scores = np.delete(scores, exclude_ix, axis=0)
            masks = np.delete(masks, exclude_ix, axis=0)
            N = class_ids.shape[0]
        full_masks = [mutils.unmold_mask(masks[i], boxes[i], image_shape) for i in range(N)]
        full_masks = np.stack(full_masks, axis=-1) if full_masks else np.empty(tuple(image_shape[:2]) + (0,))
        return boxes, class_ids, scores, full_masks
