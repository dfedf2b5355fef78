"""
Execution engine of the Mask-YOLO hot path on one MI355X.

Holds the network state (one flat fp32 parameter buffer + same-shaped gradient / Adam buffers,
laid out so that the three gradient buckets of the data-parallel all-reduce are contiguous) and
runs the training step / inference forward as a fixed sequence of calls into libmyolo_hip.so
(myolo/_ext.py).  torch is used for device memory and streams only -- there is no torch compute
and no fallback: every arithmetic op is a HIP kernel behind the C-ABI of include/myolo_hip.h.

Graph followed (reference file:line): MaskYOLO.build model.py:787-941; mobilenet_graph :55-79;
yolo_branch_graph :249-278; feature_map :848; DecodeYOLOLayer :1442-1473; DetectMaskTargetLayer
:605-661; build_mask_graph :668-715; yolo_custom_loss :86-242; myolo_mask_loss_graph :718-754;
compile :1062-1094 (loss sum + Adam).
"""
import os
import threading
import time

import numpy as np
import torch

from . import _ext as X

ACT_NONE, ACT_RELU, ACT_RELU6 = 0, 1, 2

BACKBONE_BLOCKS = [(64, 1), (64, 2), (128, 1), (256, 2), (256, 1), (512, 1)]                      # model.py:68-77
YOLO_BLOCKS = [(512, 2), (512, 1), (512, 1), (512, 1), (512, 1), (512, 1), (1024, 2), (1024, 1)]   # model.py:256-268
MASK_FILTERS = 256                                                                                 # model.py:688-711


WINO_MIN_ROWS = 8192       # CONV3X3_ALGO='auto': output pixels from which the Winograd form of a 3x3 conv is used (Rice 416 at batch 4: feature_map has 10816)


BUCKET_BACKBONE, BUCKET_YOLO, BUCKET_FEATURE_MAP, BUCKET_MASK_CONV1, BUCKET_MASK_REST = 0, 1, 2, 3, 4
N_BUCKETS = 5


def layer_table(cfg):
    """[(layer name, kind, shape, bucket)] -- names are the reference's Keras layer names (the
    checkpoint schema).  bucket = the gradient all-reduce bucket, cut where backward FINISHES a group of layers (the flat
    buffer is laid out bucket by bucket, so each is one contiguous slice): 0 backbone (complete at the end of the step), 1 YOLO blocks + conv_23
    (complete when the YOLO head's backward retires, under the mask head's forward), 2 feature_map, 3 myolo_mask_conv1 + bn1 (conv1's dense weight
    gradient: late), 4 the rest of the mask head (complete when the compact chain's weight gradients retire).  Same flat layout as rounds 1-5."""
    a, C = cfg.ALPHA, cfg.NUM_CLASSES
    t = [("conv1", "conv", (3, 3, 3, int(32 * a)), 0), ("conv1_bn", "bn", int(32 * a), 0)]
    cin, bid = int(32 * a), 1
    blocks = [(f, s, 0) for f, s in BACKBONE_BLOCKS] + [(f, s, 1) for f, s in YOLO_BLOCKS]
    for f, s, bk in blocks:
        co = int(f * a)
        t += [("conv_dw_%d" % bid, "dw", (3, 3, cin), bk), ("conv_dw_%d_bn" % bid, "bn", cin, bk),
              ("conv_pw_%d" % bid, "conv", (1, 1, cin, co), bk), ("conv_pw_%d_bn" % bid, "bn", co, bk)]
        if bid == len(BACKBONE_BLOCKS):
            c4 = co
        cin = co
        bid += 1
    t += [("conv_23", "convb", (1, 1, cin, cfg.N_BOX * (5 + C)), BUCKET_YOLO),
          ("feature_map", "convb", (3, 3, c4, cfg.TOP_FEATURE_MAP_DEPTH), BUCKET_FEATURE_MAP)]
    cm = cfg.TOP_FEATURE_MAP_DEPTH
    for i in range(1, 5):
        bk = BUCKET_MASK_CONV1 if i == 1 else BUCKET_MASK_REST
        t += [("myolo_mask_conv%d" % i, "convb", (3, 3, cm, MASK_FILTERS), bk), ("myolo_mask_bn%d" % i, "bn", MASK_FILTERS, bk)]
        cm = MASK_FILTERS
    t += [("myolo_mask_deconv", "deconv", (2, 2, MASK_FILTERS, MASK_FILTERS), BUCKET_MASK_REST),
          ("myolo_mask", "convb", (1, 1, MASK_FILTERS, C), BUCKET_MASK_REST)]
    return t


def _glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def init_state_dict(cfg, seed=0):
    """Keras default initialisers: glorot_uniform kernels, zero biases, BN gamma=1 beta=0,
    moving mean 0 / variance 1.  Same draw order on every rank (seeded)."""
    rng = np.random.default_rng(seed)
    sd = {}
    for name, kind, shp, _ in layer_table(cfg):
        if kind in ("conv", "convb"):
            kh, kw, ci, co = shp
            sd[name + "/kernel"] = _glorot(rng, shp, kh * kw * ci, kh * kw * co)
            if kind == "convb":
                sd[name + "/bias"] = np.zeros(co, np.float32)
        elif kind == "dw":
            kh, kw, c = shp
            sd[name + "/depthwise_kernel"] = _glorot(rng, shp, kh * kw * c, kh * kw)
        elif kind == "deconv":
            kh, kw, co, ci = shp
            sd[name + "/kernel"] = _glorot(rng, shp, kh * kw * co, kh * kw * ci)
            sd[name + "/bias"] = np.zeros(co, np.float32)
        else:
            sd[name + "/gamma"] = np.ones(shp, np.float32)
            sd[name + "/beta"] = np.zeros(shp, np.float32)
            sd[name + "/moving_mean"] = np.zeros(shp, np.float32)
            sd[name + "/moving_variance"] = np.ones(shp, np.float32)
    return sd


_SHARED_STREAMS = {}


def _shared_stream(dev, name, **kw):
    """The engine's side streams are created ONCE per device and process, shared by every Net, and are HIGH-PRIORITY streams.
    HIP multiplexes streams onto a few hardware queues, decided by the runtime as streams come into use; a side stream that ends up sharing the
    compute stream's queue / pipe serialises against it.  Measured: 21.0 -> 29.3 ms with the YOLO-head stream there (round 3); round 4: 20.5 ->
    26.3 ms when a torch.distributed NCCL process group had been created BEFORE the Net -- i.e. in every data-parallel run -- and +3.6 ms for a
    fresh stream for the gradient all-reduces; neither GPU_MAX_HW_QUEUES = 4 / 8 / 16 / 24 nor probing candidate streams for concurrency changed
    that (tools/experiments/pg_stream_cost.py).  High-priority streams are served from a queue pool of their own, apart from every
    normal-priority stream torch, RCCL or the user create: 20.63 / 20.67 / 20.68 ms without a process group / with one created before / after the
    Net.  (MYOLO_STREAM_PRIORITY=0 restores normal priority for that experiment.)  Nets of one process do not run concurrently, so sharing costs nothing."""
    key = (str(torch.device(dev)), name)
    st = _SHARED_STREAMS.get(key)
    if st is None:
        kw.setdefault("priority", int(os.environ.get("MYOLO_STREAM_PRIORITY", "-1")))
        st = _SHARED_STREAMS[key] = torch.cuda.Stream(device=dev, **kw)
    return st


class Workspace(object):
    """One growable scratch buffer shared by every call on the compute stream."""

    def __init__(self, device, nbytes=256 << 20, on_realloc=None):
        self.device = device
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.on_realloc = on_realloc      # a captured hipGraph holds raw pointers into buf: the owner drops its graphs

    def ensure(self, nbytes):
        if self.buf.numel() < nbytes:
            torch.cuda.synchronize()
            if self.on_realloc:
                self.on_realloc()
            self.buf = torch.empty(int(nbytes * 1.25), dtype=torch.uint8, device=self.device)

    @property
    def ptr(self):
        return self.buf.data_ptr()

    @property
    def size(self):
        return self.buf.numel()


def _ensure_hw_queues(want=8):
    """HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  A side stream that lands on the COMPUTE stream's queue
    serialises against it in submission order (measured: +4 ... +8 ms per training step, profiles/r3_notes.md "hardware queues"; an extra
    stream cost 6.6 ms in round 4), and inference lanes only overlap with a queue each.  The variable is read when the HIP runtime initialises:
    it is set here -- at the first Net, not at import -- when the user did not set it and HIP is not up yet; otherwise a warning says so."""
    cur = os.environ.get("GPU_MAX_HW_QUEUES")
    if cur is not None:
        return
    if not torch.cuda.is_initialized():
        os.environ["GPU_MAX_HW_QUEUES"] = str(want)
    else:
        import warnings
        warnings.warn("myolo: HIP was initialised before the first Net without GPU_MAX_HW_QUEUES set (default 4 hardware queues): side streams may share "
                      "the compute stream's queue (+4..8 ms per training step measured).  Export GPU_MAX_HW_QUEUES=8 before the first device call.")


class Net(object):
    def __init__(self, cfg, device="cuda:0", seed=0):
        _ensure_hw_queues()
        X.load()
        self.cfg = cfg
        self.dev = torch.device(device)
        self.table = layer_table(cfg)
        # ---- flat parameter / BN-state layout ----
        self.pslots, self.sslots = {}, {}
        self.bucket_ranges = []
        off, soff = 0, 0
        for bucket in range(N_BUCKETS):
            b0 = off
            for name, kind, shp, bk in self.table:
                if bk != bucket:
                    continue
                if kind in ("conv", "convb"):
                    items = [("kernel", shp)] + ([("bias", (shp[3],))] if kind == "convb" else [])
                elif kind == "dw":
                    items = [("depthwise_kernel", shp)]
                elif kind == "deconv":
                    items = [("kernel", shp), ("bias", (shp[2],))]
                else:
                    items = [("gamma", (shp,)), ("beta", (shp,))]
                    for s in ("moving_mean", "moving_variance"):
                        self.sslots[name + "/" + s] = (soff, (shp,))
                        soff += shp
                for s, sh in items:
                    n = int(np.prod(sh))
                    self.pslots[name + "/" + s] = (off, tuple(sh))
                    off += (n + 3) // 4 * 4          # keep every tensor 16-byte aligned
            self.bucket_ranges.append((b0, off))
        self.nparam = off
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.flat_p = torch.zeros(off, **f32)
        self.flat_g = torch.zeros(off, **f32)
        self.flat_m = torch.zeros(off, **f32)
        self.flat_v = torch.zeros(off, **f32)
        self.flat_s = torch.zeros(soff, **f32)
        self.p = {k: self.flat_p[o:o + int(np.prod(sh))].view(sh) for k, (o, sh) in self.pslots.items()}
        self.g = {k: self.flat_g[o:o + int(np.prod(sh))].view(sh) for k, (o, sh) in self.pslots.items()}
        self.s = {k: self.flat_s[o:o + int(np.prod(sh))].view(sh) for k, (o, sh) in self.sslots.items()}
        self.bnbuf = {}
        for name, kind, shp, _ in self.table:
            if kind == "bn":
                self.bnbuf[name] = torch.zeros(4, shp, **f32)        # mean, var, scale, shift
        self._graphs = {}                 # predict_graphed: (input shape, modes, lane) -> (hipGraph, static input, static outputs)
        self._lanes = {}                  # predict_stream: lane -> its stream / scratch / coefficient buffers
        self._cap_stream = None           # graph capture never happens with the default stream current (see _capture_predict)
        self._fm_stream = None            # inference: feature_map's conv beside the YOLO head (trunk_fwd)
        self.infer_fork_feature_map = False   # opt-in, and then only forwards that run alone: measured +0.0-0.5 % alone, -9 % with several batches in flight (forked graphs, 1150 -> 1055 img/s at Rice-416); and the extra stream it brings into a process moved the lanes' hardware-queue mapping in the full bench run (three lanes 1152 -> 1086)
        self._fork_now = None             # the effective value while predict_graphed captures / runs a forward for a given number of lanes
        # a captured graph bakes in pointers to the scratch buffer: growing it (a bigger launch on the same Net) drops the graphs
        self._ws_main = Workspace(self.dev, on_realloc=lambda: self._drop_graphs(0))
        self._ws_side = Workspace(self.dev)    # scratch of the YOLO-head backward running on the side stream
        self._ws_active = self._ws_main
        self._yolo_stream = _shared_stream(self.dev, "yolo_head_bwd")
        self.overlap_yolo_bwd = True      # YOLO-head backward on a side stream, under the mask head (training step)
        # When the YOLO head's backward is launched: 1 = right behind the YOLO loss, under the mask head's FORWARD; 0 = at the start of the mask head's
        # backward (rounds 3-5); -1 (default) = early exactly when a gradient exchange is active (GradReducer.attach sets exchange_active).  Same step
        # time on one GPU either way (19.40 against 19.41 ms, round 6).  Early, its 12.9 MB gradient bucket is complete 9.3 ms before the step ends
        # instead of 2.9 and the weight-gradient stream is free for the compact chain's weight gradients (their bucket: 6.0 instead of 2.4 ms) -- that is
        # what a data-parallel run needs; without an exchange the ~60 small launches only take chip time from the forward's multiplies (the dominant
        # kernel's launches: 1.36 against 1.19 ms) and give it back in the backward, so the late form stays there.
        self.yolo_bwd_early = -1
        self.exchange_active = False      # set by myolo.dist.GradReducer.attach: collectives are really issued (world > 1, or a 1-rank probe)
        # conv1's weight gradient (MFMA-bound, 2.7 ms, nothing downstream but the optimiser) on a third stream with its own
        # scratch, underneath conv1's data gradient -> ROIAlign backward -> backbone backward (launch- / HBM-bound small kernels)
        self._wgrad_stream = _shared_stream(self.dev, "weight_gradients")
        self._ws_wgrad = Workspace(self.dev)
        self.overlap_conv1_wgrad = True
        self.overlap_compact_wgrad = True  # ... and the weight / bias gradients of the compacted conv2-4 / deconv backward in front of it on that stream
        # the trunk's weight-gradient kernels (no consumer but the optimiser) on their own stream and scratch, beside the
        # BatchNorm-backward -> data-gradient chain that is the critical path of the trunk backward
        self._twg_stream_own_ = None           # (created on demand: the two-stream ablation only)
        self._ws_twg_own = Workspace(self.dev)
        # 1 (default): they share conv1's weight-gradient stream and scratch -- ONE weight-gradient stream, its launches strictly in order.
        # Measured (profiles/r3_notes.md, "hardware queues"): with the two streams really concurrent the step is 1.6 ms SLOWER; HIP's default
        # of four hardware queues happened to alias them, a mapping that depends on stream creation order -- so the order is made explicit
        self.single_wgrad_stream = 1
        self.overlap_trunk_wgrad = True
        self._twg_pending = False
        self._twg_used = set()
        self._twg_override = None              # (stream, scratch) the trunk's weight gradients go to instead, see backbone_wgrad_on_yolo_stream
        self.backbone_wgrad_on_yolo_stream = 1 # 1 = the backbone's weight gradients (the last part of the step) run on the YOLO-head stream, idle by then, instead of queueing behind conv1's weight gradient on the weight-gradient stream
        # ... and started only when conv1's data gradient (the other matrix-pipe-bound kernel of that window) has been issued: two MFMA-bound
        # kernels sharing the chip each run at half speed, an MFMA-bound one beside the small HBM- / latency-bound kernels of the trunk
        # backward costs neither much (0 = start it together with the data gradient, as in round 2)
        self.conv1_wgrad_after_dgrad = 0
        self._wgrad_pending = False
        self.mask_keep_pre = True         # F(6,3) chain in training: the positive ROIs' PRE-BatchNorm conv outputs are kept (exact bn2-4 backward); False: post-activation + (a - beta) / gamma
        self.lazy_bn1_bwd = True          # conv1's gradients read bn1's input gradient lazily (never materialised)
        self.fused_frozen_bn = True       # bn_act_fwd on moving statistics: one launch instead of coefficients + apply
        self.fold_frozen_bn = True        # train=False forwards: that BatchNorm + ReLU6 in the epilogue of the depthwise / pointwise conv
        # training forwards of the trunk: every BatchNorm's batch statistics come out of the producing conv's epilogue and its apply +
        # ReLU6 happen on the consumer's load -- the normalised activations are never written (dw_block_fwd / dw_block_bwd)
        self.fuse_trunk_bn = bool(getattr(cfg, "FUSE_TRUNK_BN", True))
        self._fz_table = None
        self.merge_conv1_bwd_transforms = True    # conv1's backward: the V and Q transforms of the lazily formed gradient from one pass over y_pre
        self.anchors = torch.tensor(np.asarray(cfg.ANCHORS, np.float32), device=self.dev)
        self.class_weights = torch.tensor(np.asarray(cfg.CLASS_WEIGHTS, np.float32), device=self.dev)
        self.adam_t = 0
        self.tape = {}
        self.proposals_hook = None        # callable(proposals [B,R,4], device batch): in-place edit before target assignment (bench.py)
        self.tape_hook = None             # callable(net) between forward and backward of forward_backward (tests: teacher-forcing)
        self.on_bucket_ready = None       # callable(bucket_index) -- set by myolo/dist.py
        self.before_optimizer = None      # callable() -- waits for the all-reduce
        self.grad_scale = 1.0
        # exact-sparsity backward of the mask head (see mask_head_bwd_sparse); False = dense reference path
        self.conv3x3_algo = getattr(cfg, "CONV3X3_ALGO", "auto")
        if self.conv3x3_algo not in ("auto", "direct", "winograd"):
            raise ValueError("CONV3X3_ALGO must be 'auto', 'direct' or 'winograd' (got %r)" % (self.conv3x3_algo,))
        self.wino_tiles = getattr(cfg, "WINOGRAD_TILES", "f43")
        if self.wino_tiles not in ("f43", "f63"):
            raise ValueError("WINOGRAD_TILES must be 'f43' or 'f63' (got %r)" % (self.wino_tiles,))
        self.fp32_matmul = getattr(cfg, "FP32_MATMUL", "native")
        if self.fp32_matmul not in ("native", "bf16x6"):
            raise ValueError("FP32_MATMUL must be 'native' or 'bf16x6' (got %r)" % (self.fp32_matmul,))
        self._activate()
        self.sparse_mask_bwd = True
        # exact-sparsity FORWARD of the mask head (see mask_head_fwd_positives): conv2-4 / deconv / myolo_mask only on
        # the positive ROIs.  Same loss, gradients and BN state; the training graph's unused myolo_mask rows of the
        # non-positive ROIs are not produced.  Off by default (cfg.TRAIN_MASK_HEAD_ROIS = "all").
        self.sparse_mask_fwd = getattr(cfg, "TRAIN_MASK_HEAD_ROIS", "all") == "positives"
        self._copy_stream = _shared_stream(self.dev, "n_pos_copy")
        self._npos_ready = torch.cuda.Event()
        self._npos_pinned = None
        self.fuse_bn_bwd_sums = 1         # trunk backward: the depthwise data gradient leaves the sums of the BatchNorm its output reaches (conv_pw_{b-1}_bn / conv1_bn) in its epilogue -- that BatchNorm's backward is finish + dx, its pass over (dy, x) is gone (round 5); 0 = three launches per BatchNorm
        self._bn_sums = {}                # BatchNorm name -> (partials, rows) left by the producer of its output gradient
        self._np_seen = None              # positives of the last step whose counts the host has read (sizes the next step's kept-rows buffer)
        self._keep_cap_hw = 0
        self.fused_bn_bwd = 0             # 1 = training-mode BatchNorm backward in one launch (sums, grid-wide barrier, dx: myolo_bn_act_bwd_fused).  Measured: 28.7 against 20.9 ms per step -- the barrier needs all its workgroups resident, and this step runs its chains BESIDE chip-filling kernels of other streams on purpose (profiles/r4_notes.md section 6)
        self._bn_sync = {}                # stream -> the barrier's counters
        self._bn_fused_bytes = {}
        self.weight_prep = 1              # training step: weight-only re-layouts (transposes, bf16x6 splits, Winograd filter transforms) re-run on a side stream at the step's start instead of inside the chain (X.WeightPrep); 0 = in place (round 3)
        self._wprep = None
        self._wprep_calls, self._wprep_misses, self._wprep_cap = 0, None, 0      # (_wprep_begin: one warning if the arena turns out too small)
        self.wprep_wait_late = 1          # 0: wait for the trunk's prepared weights right behind conv1 (A/B)
        self._wprep_ntrunk = None         # registry entries recorded before the mask head (their consumers are the trunk's first layers)
        self._wprep_ev = None
        self.seen = 0                     # evaluations of the YOLO loss so far (the reference's `seen`, see _yolo_warm)
        self.keep_rows_compact = 1        # the rows the training forward keeps for the sparse backward (pre-BatchNorm outputs of conv2-4, bn1's activation) are written in compact order by the layer boundaries: no gathers in the backward chain; 0 = dense positions + gathers (round 3)
        self.keep_deconv_rows = 48        # training forward keeps the ReLU'd deconv output of up to this many positives PER IMAGE (batch total) for the sparse backward; 0 = re-run the deconv there (round 3); beyond the cap the backward re-runs it
        self.fuse_compact_gather = 1      # compacted mask-head backward: gather + BatchNorm apply in one kernel, each pre-BN tensor gathered once (0: round 3's sequence)
        self.bucket1_on_wgrad_stream = 1  # data-parallel: bucket 1 released on the weight-gradient stream (0: round 3's join of that stream into the compute stream)
        self.pinned_upload = 1            # to_device_batch through pinned staging + the upload stream (0: synchronous torch.as_tensor copies, rounds 1-3)
        self.upload_own_stream = 0        # EXPERIMENT
        self._stage_lock = threading.Lock()
        self._stage = None                # to_device_batch: ring of pinned staging sets + upload stream (created on first use)
        self._bind_cache = {}
        self.timed_tags = set()           # bench.py: kernel tags to bracket with HIP events
        self.timings = {}                 # tag -> [(start_event, end_event), ...]
        self.host_wait_s = 0.0            # wall time the host spent blocked on the n_pos copy (bench.py reports it per step)
        # inference forwards: the weight-only preparations (bf16 packing + BatchNorm folding of the mask head, bf16x6 splits of the pointwise layers,
        # Winograd filter transforms) are made ONCE per weight version and kept -- round 4 re-ran them in every forward (5 packs, ~17 splits, a filter
        # transform: ~0.14 ms of a 3.7 ms Rice-416 batch) although the weights of an inference net are frozen.  `_wver` is bumped by load_state_dict,
        # every optimizer step and every training forward (moving statistics); code that writes flat_p / flat_s through raw pointers calls
        # mark_weights_changed().  In-place torch writes (net.p[k].copy_(...), an EMA, a broadcast into flat_p) are caught by themselves: the tensors'
        # autograd version counters are part of the cached state (_weight_state).
        self._infer_weight_cache = 1      # 0 = prepare inside every forward (round 4); property infer_weight_cache: a change drops the captured graphs
        self._wver = 0
        self._iprep = None                # X.WeightPrep of the inference path (arena owned by it)
        self._iprep_state = None          # _weight_state() + registry entries the arena + packs were last refreshed for
        self._iprep_ev = None
        self._bf16_packs = {}             # layer -> (packed bf16 weights, folded bias) of mask_head_fwd_bf16, refreshed with the arena
        self.load_state_dict(init_state_dict(cfg, seed))

    def _activate(self):
        """The library's kernel-choice switches are process-wide (myolo_set_option); this Net's choices are (re)applied at the start
        of every step / forward, so several Nets with different cfg.FP32_MATMUL can live in one process (an inference MaskYOLO
        beside a trainer).  Captured hipGraphs are keyed by the mode they were captured under (predict_graphed)."""
        X.set_option("wino_x6", 1 if self.fp32_matmul == "bf16x6" else 0)

    # ------------------------------------------------------------------ state
    def trainable_names(self):
        return list(self.pslots)

    def state_dict(self):
        sd = {k: v.detach().cpu().numpy().copy() for k, v in self.p.items()}
        sd.update({k: v.detach().cpu().numpy().copy() for k, v in self.s.items()})
        return sd

    def load_state_dict(self, sd, strict=True):
        for k, v in sd.items():
            dst = self.p.get(k, self.s.get(k))
            if dst is None:
                if strict:
                    raise KeyError("unexpected tensor %s" % k)
                continue
            v = np.asarray(v, np.float32).reshape(dst.shape)
            dst.copy_(torch.from_numpy(np.ascontiguousarray(v)))
        self.mark_weights_changed()
        if strict:
            missing = [k for k in list(self.p) + list(self.s) if k not in sd]
            if missing:
                raise KeyError("missing tensors: %s" % missing[:5])

    def mark_weights_changed(self):
        """tell the engine that parameters or moving statistics were written (load_state_dict, the optimizer and the training forward call this
        themselves): the next inference forward re-makes its cached weight preparations (_infer_prep_sync) before it runs or replays a graph.
        REQUIRED after any write to flat_p / flat_s / p[...] / s[...] that torch cannot see (a kernel launched on their data_ptr()); writes through
        torch's own in-place operators are noticed without it."""
        self._wver += 1

    def _weight_state(self):
        """what the cached weight preparations depend on: the explicit version and the version counters torch bumps on every in-place write to the flat
        parameter / statistics buffers or to any view of them"""
        return (self._wver, int(self.flat_p._version), int(self.flat_s._version))

    @property
    def infer_weight_cache(self):
        return self._infer_weight_cache

    @infer_weight_cache.setter
    def infer_weight_cache(self, v):
        v = int(v)
        if v != self._infer_weight_cache:
            # graphs captured under the other setting hold (or lack) the preparation launches and the arena's pointers: drop them
            if torch.cuda.is_available() and self.dev.type == "cuda":
                torch.cuda.synchronize(self.dev)
            self._graphs.clear()
            self._iprep_state = None
        self._infer_weight_cache = v

    def grads_dict(self):
        self.join_conv1_wgrad()           # conv1's weight gradient may still be running on its side stream
        self.join_trunk_wgrad()
        return {k: v.detach().cpu().numpy().copy() for k, v in self.g.items()}

    # ------------------------------------------------------------------ helpers
    def _new(self, *shape, dtype=torch.float32):
        return torch.empty(*shape, dtype=dtype, device=self.dev)

    @property
    def ws(self):
        return self._ws_active

    def _wsargs(self):
        return self.ws.ptr, self.ws.size

    def _call_timed(self, tag, name, *args):
        """X.call bracketed by HIP events on the launch stream when `tag` is being measured."""
        if tag not in self.timed_tags:
            return X.call(name, *args)
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        X.call(name, *args)
        e1.record(st)
        self.timings.setdefault(tag, []).append((e0, e1))

    def _call_conv_then_finish(self, tag, name, args):
        """a *_bnstats_fwd entry point: both launches in one call, or -- when `tag` is being measured -- the conv launch alone inside the
        event bracket and the statistics finish after it"""
        if tag not in self.timed_tags:
            return X.call(name, *args, 3, *self._wsargs(), X.stream())
        self._call_timed(tag, name, *args, 1, *self._wsargs(), X.stream())
        X.call(name, *args, 2, *self._wsargs(), X.stream())

    def kernel_ms(self, tag):
        """average launch duration (ms) of the timed tag since the last reset (synchronises)."""
        torch.cuda.synchronize()
        ev = self.timings.get(tag, [])
        return sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev)), len(ev)

    # ---- BN + activation --------------------------------------------------
    def bn_act_fwd(self, name, y, act, batch_stats):
        """y [M,C] pre-BN conv output.  Returns act(BN(y)); saves what backward needs."""
        M, C = y.shape
        buf = self.bnbuf[name]
        mean, var, scale, shift = buf[0], buf[1], buf[2], buf[3]
        if batch_stats:
            X.call("myolo_bn_stats", X.ptr(y), X.ptr(self.p[name + "/gamma"]), X.ptr(self.p[name + "/beta"]),
                   X.ptr(mean), X.ptr(var), X.ptr(scale), X.ptr(shift),
                   X.ptr(self.s[name + "/moving_mean"]), X.ptr(self.s[name + "/moving_variance"]),
                   M, C, *self._wsargs(), X.stream())
        elif self.fused_frozen_bn and C % 4 == 0 and 256 % (C // 4) == 0:       # frozen BN: coefficients + apply + activation in one launch
            a = self._new(M, C)
            X.call("myolo_bn_frozen_apply_act", X.ptr(y), X.ptr(self.p[name + "/gamma"]), X.ptr(self.p[name + "/beta"]),
                   X.ptr(self.s[name + "/moving_mean"]), X.ptr(self.s[name + "/moving_variance"]), X.ptr(scale), X.ptr(shift), X.ptr(a),
                   M, C, act, X.stream())
            self.tape[name] = (y, act, batch_stats)
            return a
        else:
            X.call("myolo_bn_frozen_coeffs", X.ptr(self.p[name + "/gamma"]), X.ptr(self.p[name + "/beta"]),
                   X.ptr(self.s[name + "/moving_mean"]), X.ptr(self.s[name + "/moving_variance"]),
                   X.ptr(scale), X.ptr(shift), C, X.stream())
        a = self._new(M, C)
        X.call("myolo_bn_apply_act", X.ptr(y), X.ptr(scale), X.ptr(shift), X.ptr(a), M, C, act, X.stream())
        self.tape[name] = (y, act, batch_stats)
        return a

    def bn_act_bwd(self, name, da, y_override=None):
        y, act, batch_stats = self.tape[name]
        if y_override is not None:
            y = y_override
        M, C = y.shape
        buf = self.bnbuf[name]
        if batch_stats:
            mean, var = buf[0], buf[1]
        else:
            mean, var = self.s[name + "/moving_mean"], self.s[name + "/moving_variance"]
        dx = self._new(M, C)
        pre = self._bn_sums.pop(name, None) if batch_stats else None
        if pre is not None:
            # the producer of `da` (the depthwise data gradient behind this BatchNorm) left the sums: finish + dx
            part, rows = pre
            X.call("myolo_bn_act_bwd_from_partials", X.ptr(da), X.ptr(y), X.ptr(mean), X.ptr(var), X.ptr(buf[2]), X.ptr(buf[3]), X.ptr(dx),
                   X.ptr(self.g[name + "/gamma"]), X.ptr(self.g[name + "/beta"]), M, C, act, X.ptr(part), rows, *self._wsargs(), X.stream())
            return dx
        if batch_stats and self.fused_bn_bwd:
            need = self._bn_fused_bytes.get((M, C))
            if need is None:
                need = self._bn_fused_bytes[(M, C)] = int(X.load().myolo_bn_act_bwd_fused_ws_bytes(M, C))
            if need:
                # one launch (sums, grid-wide barrier, dx) instead of three; the barrier's counters are private to the stream this runs on
                sid = torch.cuda.current_stream().cuda_stream
                sync = self._bn_sync.get(sid)
                if sync is None:
                    sync = self._bn_sync[sid] = torch.zeros(64, dtype=torch.int32, device=self.dev)
                X.call("myolo_bn_act_bwd_fused", X.ptr(da), X.ptr(y), X.ptr(mean), X.ptr(var), X.ptr(buf[2]), X.ptr(buf[3]), X.ptr(dx),
                       X.ptr(self.g[name + "/gamma"]), X.ptr(self.g[name + "/beta"]), M, C, act, X.ptr(sync), *self._wsargs(), X.stream())
                return dx
        X.call("myolo_bn_act_bwd", X.ptr(da), X.ptr(y), X.ptr(self.p[name + "/gamma"]), X.ptr(mean), X.ptr(var),
               X.ptr(buf[2]), X.ptr(buf[3]), X.ptr(dx), X.ptr(self.g[name + "/gamma"]), X.ptr(self.g[name + "/beta"]),
               M, C, act, 1 if batch_stats else 0, *self._wsargs(), X.stream())
        return dx

    def colsum(self, x2d, out):
        M, C = x2d.shape
        X.call("myolo_colsum", X.ptr(x2d), X.ptr(out), M, C, *self._wsargs(), X.stream())

    # ---- depthwise-separable block ------------------------------------------
    # ---- 3x3 / s1 / SAME convolution: direct implicit GEMM or Winograd F(4x4,3x3) -------------------------
    def _wino_ok(self, nimg, h, w, cin, cout):
        """cfg.CONV3X3_ALGO: 'direct' | 'winograd' | 'auto' (Winograd for the big dense launches, where its 3x fewer
        multiplications outweigh two extra streaming passes; the direct kernel for the small / compacted ones)."""
        algo = self.conv3x3_algo
        if algo == "direct":
            return False
        fits = cin % 16 == 0 and cout % 16 == 0 and h >= 4 and w >= 4
        if algo == "winograd":
            return fits
        return fits and nimg * h * w >= WINO_MIN_ROWS

    def _wino63(self, h, w, cin, cout):
        """the F(6,3)/F(4,3) tiling applies to this conv (cfg.WINOGRAD_TILES='f63', 14x14 maps, channel counts the one-launch multiply takes)"""
        return self.wino_tiles == "f63" and X.wino63_ok(h, w, cin, cout)

    def _timed(self, tag):
        """(start, stop) closures bracketing a multi-launch op with HIP events on the launch stream, if `tag` is measured."""
        if tag is None or tag not in self.timed_tags:
            return (lambda: None), (lambda: None)
        st = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def stop():
            e1.record(st)
            self.timings.setdefault(tag, []).append((e0, e1))
        return (lambda: e0.record(st)), stop

    def conv3x3_fwd(self, x, layer, y, nimg, h, w, cin, cout, scale=None, shift=None, act=ACT_NONE, keep_v=False, tag=None):
        """y = act(affine(conv3x3(x) + bias)) with the layer's kernel/bias; affine (scale, shift) optional.
        Returns the Winograd-transformed input when keep_v (for conv3x3_bwd_weight), else None."""
        kern, bias = self.p[layer + "/kernel"], self.p[layer + "/bias"]
        start, stop = self._timed(tag)
        start()
        v = None
        if self._wino_ok(nimg, h, w, cin, cout) and not keep_v and self._wino63(h, w, cin, cout):
            # F(6,3)/F(4,3) tiling of a 14x14 map (csrc/wino63_kernels.hip); a kept V stays in the F(4,3) layout its consumer expects
            U = self._new(X.wino63_u_elems(cin, cout))
            V, M = self._new(X.wino63_plane_elems(nimg, cin)), self._new(X.wino63_plane_elems(nimg, cout))
            X.call("myolo_wino63_input_transform", X.ptr(x), None, None, ACT_NONE, None, None, X.ptr(V), nimg, cin, X.stream())
            self._call_timed("wino_multiply" if tag == "mask_conv3x3_fwd" else None, "myolo_wino63_multiply_w", X.ptr(V), X.ptr(kern), X.ptr(U), X.ptr(M),
                             nimg, cin, cout, X.stream())
            X.call("myolo_wino63_output_transform", X.ptr(M), X.ptr(bias), X.ptr(scale), X.ptr(shift), X.ptr(y), nimg, cout, act, X.stream())
        elif self._wino_ok(nimg, h, w, cin, cout):
            T = nimg * ((h + 3) // 4) * ((w + 3) // 4)
            U, V, M = self._new(X.wino_u_elems(cin, cout)), self._new(36, T, cin), self._new(36, T, cout)
            X.call("myolo_wino_input_transform", X.ptr(x), X.ptr(V), nimg, h, w, cin, X.stream())
            # only the dense mask-head launches (tag given) feed bench.py's roofline; feature_map / compacted ones do not
            self._call_timed("wino_multiply" if tag == "mask_conv3x3_fwd" else None, "myolo_wino_multiply_w", X.ptr(V), X.ptr(kern), X.ptr(U), X.ptr(M),
                             nimg, h, w, cin, cout, X.stream())
            X.call("myolo_wino_output_transform", X.ptr(M), X.ptr(bias), X.ptr(scale), X.ptr(shift), X.ptr(y), nimg, h, w, cout, act,
                   X.stream())
            v = V if keep_v else None
        elif scale is not None or act != ACT_NONE:
            X.call("myolo_conv3x3_affine_act_fwd", X.ptr(x), X.ptr(kern), X.ptr(bias), X.ptr(scale), X.ptr(shift), X.ptr(y), nimg, h, w,
                   cin, cout, act, *self._wsargs(), X.stream())
        else:
            X.call("myolo_conv3x3_fwd", X.ptr(x), X.ptr(kern), X.ptr(bias), X.ptr(y), nimg, h, w, cin, cout, *self._wsargs(), X.stream())
        stop()
        return v

    def conv3x3_bwd_weight(self, x, v_saved, dy, layer, nimg, h, w, cin, cout):
        dw = self.g[layer + "/kernel"]
        if self._wino_ok(nimg, h, w, cin, cout) and v_saved is None and self._wino63(h, w, cin, cout):
            self.ws.ensure(X.wino63_ws_bytes(nimg, cin, cout, 2))
            X.call("myolo_conv3x3_wino63_bwd_weight", X.ptr(x), None, X.ptr(dy), X.ptr(dw), nimg, cin, cout, *self._wsargs(), X.stream())
        elif self._wino_ok(nimg, h, w, cin, cout):
            self.ws.ensure(X.wino_ws_bytes(nimg, h, w, cin, cout, 2))
            X.call("myolo_conv3x3_wino_bwd_weight", None if v_saved is not None else X.ptr(x), X.ptr(v_saved), X.ptr(dy), X.ptr(dw),
                   nimg, h, w, cin, cout, *self._wsargs(), X.stream())
        else:
            X.call("myolo_conv3x3_bwd_weight", X.ptr(x), X.ptr(dy), X.ptr(dw), nimg, h, w, cin, cout, *self._wsargs(), X.stream())

    def conv3x3_bwd_data(self, dy, layer, dx, nimg, h, w, cin, cout):
        if self._wino_ok(nimg, h, w, cout, cin) and self._wino63(h, w, cout, cin):
            self.ws.ensure(X.wino63_ws_bytes(nimg, cin, cout, 1))
            X.call("myolo_conv3x3_wino63_bwd_data", X.ptr(dy), X.ptr(self.p[layer + "/kernel"]), X.ptr(dx), nimg, cin, cout,
                   *self._wsargs(), X.stream())
        elif self._wino_ok(nimg, h, w, cout, cin):
            self.ws.ensure(X.wino_ws_bytes(nimg, h, w, cin, cout, 1))
            X.call("myolo_conv3x3_wino_bwd_data", X.ptr(dy), X.ptr(self.p[layer + "/kernel"]), X.ptr(dx), nimg, h, w, cin, cout,
                   *self._wsargs(), X.stream())
        else:
            X.call("myolo_conv3x3_bwd_data", X.ptr(dy), X.ptr(self.p[layer + "/kernel"]), X.ptr(dx), nimg, h, w, cin, cout,
                   *self._wsargs(), X.stream())

    def _frozen_affine_all(self):
        """scale / shift of every depthwise / pointwise BatchNorm on its moving statistics, ONE launch (the weights may have changed since
        the last forward, so this runs at the start of each train=False trunk forward)."""
        if self._fz_table is None:
            rows, off = [], 0
            self._fz_slot = {}
            for name in sorted(k[:-len("/gamma")] for k in self.pslots if k.endswith("_bn/gamma") and (k.startswith("conv_dw_") or k.startswith("conv_pw_") or k.startswith("conv1_bn/"))):
                C = int(self.p[name + "/gamma"].numel())
                rows.append([self.pslots[name + "/gamma"][0], self.pslots[name + "/beta"][0], self.sslots[name + "/moving_mean"][0],
                             self.sslots[name + "/moving_variance"][0], off, C])
                self._fz_slot[name] = (off, C)
                off += 2 * C
            self._fz_table = torch.tensor(rows, dtype=torch.int64, device=self.dev)
            self._fz_coeffs = torch.empty(off, dtype=torch.float32, device=self.dev)
        X.call("myolo_bn_frozen_coeffs_batched", X.ptr(self.flat_p), X.ptr(self.flat_s), self._fz_table.data_ptr(), int(self._fz_table.shape[0]),
               X.ptr(self._fz_coeffs), X.stream())

    def _frozen_affine(self, name):
        off, C = self._fz_slot[name]
        base = self._fz_coeffs.data_ptr()
        return base + 4 * off, base + 4 * (off + C)

    # A "lazy" activation: ("lazy", y_pre, bn_layer) = act(BN(y_pre)) of a training-mode BatchNorm whose scale / shift sit in
    # self.bnbuf[bn_layer][2:4]; consumers normalise y_pre while loading it.  _in_args gives the four C-ABI arguments (x, in_scale,
    # in_shift, in_act) for either a plain tensor or such a reference.
    @staticmethod
    def _is_lazy(a):
        return isinstance(a, tuple) and a[0] == "lazy"

    def _in_args(self, a):
        if self._is_lazy(a):
            _, y, bn = a
            buf = self.bnbuf[bn]
            return X.ptr(y), X.ptr(buf[2]), X.ptr(buf[3]), ACT_RELU6
        return X.ptr(a), None, None, ACT_NONE

    def _materialize(self, a):
        """the activation of a lazy reference as a tensor (one bn_apply launch); a plain tensor is returned as it is"""
        if not self._is_lazy(a):
            return a
        _, y, bn = a
        buf = self.bnbuf[bn]
        out = self._new(*y.shape)
        X.call("myolo_bn_apply_act", X.ptr(y), X.ptr(buf[2]), X.ptr(buf[3]), X.ptr(out), y.shape[0], y.shape[1], ACT_RELU6, X.stream())
        return out

    def _bn_args(self, name):
        buf = self.bnbuf[name]
        return (X.ptr(self.p[name + "/gamma"]), X.ptr(self.p[name + "/beta"]), X.ptr(buf[0]), X.ptr(buf[1]), X.ptr(buf[2]), X.ptr(buf[3]),
                X.ptr(self.s[name + "/moving_mean"]), X.ptr(self.s[name + "/moving_variance"]))

    def _block_fusable(self, C, Co):
        return self.fuse_trunk_bn and C % 4 == 0 and X.pw_bnstats_ok(C, Co)

    def dw_block_fwd(self, bid, a, shape, stride, train):
        """a [N*H*W, C] activation (or a lazy reference to one, training only), shape=(N,H,W,C).  Returns (activation, new shape);
        in fused training mode the returned activation is a lazy reference."""
        N, H, W, C = shape
        Ho, Wo = H // stride, W // stride
        dwn, pwn = "conv_dw_%d" % bid, "conv_pw_%d" % bid
        if not train and self.fold_frozen_bn:
            # inference: BatchNorm on moving statistics + ReLU6 in the epilogue of the conv that feeds it (two launches per block, not four;
            # bit-identical to the unfolded sequence; nothing is taped -- there is no backward through a train=False forward)
            sc, sh = self._frozen_affine(dwn + "_bn")
            ad = self._new(N * Ho * Wo, C)
            self._call_timed("dw%d_fwd" % bid, "myolo_dwconv3x3_affine_act_fwd", X.ptr(a), X.ptr(self.p[dwn + "/depthwise_kernel"]), sc, sh,
                             ACT_RELU6, X.ptr(ad), N, H, W, C, stride, X.stream())
            Co = self.p[pwn + "/kernel"].shape[3]
            sc, sh = self._frozen_affine(pwn + "_bn")
            ap = self._new(N * Ho * Wo, Co)
            self._call_timed("pw%d_fwd" % bid, "myolo_pwconv1x1_affine_act_fwd", X.ptr(ad), X.ptr(self.p[pwn + "/kernel"]), sc, sh,
                             ACT_RELU6, X.ptr(ap), N * Ho * Wo, C, Co, *self._wsargs(), X.stream())
            return ap, (N, Ho, Wo, Co)
        Co = self.p[pwn + "/kernel"].shape[3]
        M = N * Ho * Wo
        if train and self._block_fusable(C, Co):
            # training-mode fusion (model.py:57-66 on batch statistics): four launches per block -- depthwise conv (input normalised on
            # load, output statistics in its epilogue), finish, pointwise GEMM (A operand normalised on load, column statistics in its
            # epilogue), finish.  Neither normalised tensor is written; the backward re-normalises the pre-BN tensors on load.
            y = self._new(M, C)
            self.ws.ensure(max(X.dw_bnstats_ws_bytes(N, H, W, C, stride), X.pw_bnstats_ws_bytes(M, C, Co)))
            self._call_conv_then_finish("dw%d_fwd" % bid, "myolo_dwconv3x3_bnstats_fwd", (*self._in_args(a), X.ptr(self.p[dwn + "/depthwise_kernel"]), X.ptr(y),
                                        *self._bn_args(dwn + "_bn"), N, H, W, C, stride))
            self.tape[dwn + "_bn"] = (y, ACT_RELU6, True)
            ad = ("lazy", y, dwn + "_bn")
            y2 = self._new(M, Co)
            self._call_conv_then_finish("pw%d_fwd" % bid, "myolo_pwconv1x1_bnstats_fwd", (*self._in_args(ad), X.ptr(self.p[pwn + "/kernel"]), X.ptr(y2),
                                        *self._bn_args(pwn + "_bn"), M, C, Co))
            self.tape[pwn + "_bn"] = (y2, ACT_RELU6, True)
            self.tape["blk%d" % bid] = (a, shape, stride, ad)
            return ("lazy", y2, pwn + "_bn"), (N, Ho, Wo, Co)
        a = self._materialize(a)
        y = self._new(M, C)
        self._call_timed("dw%d_fwd" % bid, "myolo_dwconv3x3_fwd", X.ptr(a), X.ptr(self.p[dwn + "/depthwise_kernel"]), X.ptr(y),
                         N, H, W, C, stride, X.stream())
        ad = self.bn_act_fwd(dwn + "_bn", y, ACT_RELU6, train)
        y2 = self._new(M, Co)
        self._call_timed("pw%d_fwd" % bid, "myolo_pwconv1x1_fwd", X.ptr(ad), X.ptr(self.p[pwn + "/kernel"]), None, X.ptr(y2), M, C, Co,
                         *self._wsargs(), X.stream())
        ap = self.bn_act_fwd(pwn + "_bn", y2, ACT_RELU6, train)
        self.tape["blk%d" % bid] = (a, shape, stride, ad)
        return ap, (N, Ho, Wo, Co)

    @property
    def _twg_stream(self):
        if self._twg_override is not None:
            return self._twg_override[0]
        if self.single_wgrad_stream:
            return self._wgrad_stream
        if self._twg_stream_own_ is None:
            self._twg_stream_own_ = _shared_stream(self.dev, "trunk_weight_gradients")
        return self._twg_stream_own_

    @property
    def _ws_twg(self):
        if self._twg_override is not None:
            return self._twg_override[1]
        return self._ws_wgrad if self.single_wgrad_stream else self._ws_twg_own

    def _on_wgrad_stream(self, fn, tensors):
        """run fn(ws_ptr, ws_size) -- a weight-gradient launch sequence whose inputs are complete on the current stream -- on the trunk's
        weight-gradient stream (own scratch); `tensors` are kept from the allocator until that stream is done with them"""
        if not self.overlap_trunk_wgrad:
            return fn(*self._wsargs())
        cur = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(cur)
        self._twg_stream.wait_event(ev)
        with torch.cuda.stream(self._twg_stream):
            fn(self._ws_twg.ptr, self._ws_twg.size)
        for t in tensors:
            if isinstance(t, tuple) and len(t) == 3 and t[0] == "lazy":       # ("lazy", pre-BN tensor, bn): the tensor inside is what the kernels read
                t = t[1]
            if torch.is_tensor(t):
                t.record_stream(self._twg_stream)
        self._twg_pending = True
        self._twg_used.add(self._twg_stream)

    def join_trunk_wgrad(self):
        """make the current stream wait for the trunk's weight-gradient stream"""
        if self._twg_pending:
            for st in self._twg_used:
                torch.cuda.current_stream().wait_stream(st)
            self._twg_used.clear()
            self._twg_pending = False

    def dw_block_bwd(self, bid, da, next_bn=None):
        """next_bn: the training-mode BatchNorm this block's input gradient reaches next (conv_pw_{bid-1}_bn / conv1_bn); with
        fuse_bn_bwd_sums the depthwise data gradient leaves that BatchNorm's backward sums in its epilogue (bn_act_bwd then finishes them)."""
        a, shape, stride, ad = self.tape["blk%d" % bid]
        N, H, W, C = shape
        Ho, Wo = H // stride, W // stride
        dwn, pwn = "conv_dw_%d" % bid, "conv_pw_%d" % bid
        Co = self.p[pwn + "/kernel"].shape[3]
        M = N * Ho * Wo
        dy2 = self.bn_act_bwd(pwn + "_bn", da)
        # the scratch the weight gradients will really run on (their own stream's, or the main one without the overlap), sized for both of them
        (self._ws_twg if self.overlap_trunk_wgrad else self.ws).ensure(max(X.workspace_bytes(M, C, Co), X.dw_bwd_weight_ws_bytes(N, H, W, C, stride)))

        def pw_wgrad(wsp, wsz):
            if self._is_lazy(ad):       # the forward normalised the depthwise output on load: so does the weight gradient
                X.call("myolo_pwconv1x1_bwd_weight_affine_in", *self._in_args(ad), X.ptr(dy2), X.ptr(self.g[pwn + "/kernel"]), M, C, Co, wsp, wsz, X.stream())
            else:
                X.call("myolo_pwconv1x1_bwd_weight", X.ptr(ad), X.ptr(dy2), X.ptr(self.g[pwn + "/kernel"]), M, C, Co, wsp, wsz, X.stream())
        self._on_wgrad_stream(pw_wgrad, (dy2, ad))
        dad = self._new(M, C)
        X.call("myolo_pwconv1x1_bwd_data", X.ptr(dy2), X.ptr(self.p[pwn + "/kernel"]), X.ptr(dad), M, C, Co, *self._wsargs(), X.stream())
        dy = self.bn_act_bwd(dwn + "_bn", dad)

        def dw_wgrad(wsp, wsz):
            if self._is_lazy(a):
                X.call("myolo_dwconv3x3_bwd_weight_affine_in", *self._in_args(a), X.ptr(dy), X.ptr(self.g[dwn + "/depthwise_kernel"]), N, H, W, C, stride, wsp, wsz, X.stream())
            else:
                X.call("myolo_dwconv3x3_bwd_weight", X.ptr(a), X.ptr(dy), X.ptr(self.g[dwn + "/depthwise_kernel"]), N, H, W, C, stride, wsp, wsz, X.stream())
        self._on_wgrad_stream(dw_wgrad, (dy, a))
        dx = self._new(N * H * W, C)
        rows = 0
        if next_bn is not None and self.fuse_bn_bwd_sums and next_bn in self.tape and self.tape[next_bn][2] and not self.fused_bn_bwd:
            rows = X.dw_bwd_data_bnsums_rows(N, H, W, C, stride)
        if rows:
            yb, actb, _ = self.tape[next_bn]
            bb = self.bnbuf[next_bn]
            part = self._new(rows * 2 * C, dtype=torch.float64)
            X.call("myolo_dwconv3x3_bwd_data_bnsums", X.ptr(dy), X.ptr(self.p[dwn + "/depthwise_kernel"]), X.ptr(dx), N, H, W, C, stride, X.ptr(yb),
                   X.ptr(bb[2]), X.ptr(bb[3]), X.ptr(bb[0]), X.ptr(bb[1]), actb, X.ptr(part), rows, X.stream())
            self._bn_sums[next_bn] = (part, rows)
        else:
            X.call("myolo_dwconv3x3_bwd_data", X.ptr(dy), X.ptr(self.p[dwn + "/depthwise_kernel"]), X.ptr(dx), N, H, W, C, stride, X.stream())
        return dx

    # ---- trunk: backbone, feature_map, YOLO head -----------------------------------
    def trunk_fwd(self, images, train):
        cfg = self.cfg
        N, H, W, _ = images.shape
        C0 = self.p["conv1/kernel"].shape[3]
        if not train and self.fold_frozen_bn:
            self._frozen_affine_all()
        y = self._new(N * (H // 2) * (W // 2), C0)
        if train and self.fuse_trunk_bn and C0 % 4 == 0:
            # conv_block (model.py:42-52): the conv leaves the partial sums of its BatchNorm statistics; the apply + ReLU6 go into the first
            # depthwise conv's load
            self.ws.ensure(X.conv1_bnstats_ws_bytes(N, H, W, C0))
            X.call("myolo_conv3x3s2_c3_bnstats_fwd", X.ptr(images), X.ptr(self.p["conv1/kernel"]), X.ptr(y), *self._bn_args("conv1_bn"),
                   N, H, W, C0, 3, *self._wsargs(), X.stream())
            self.tape["conv1_bn"] = (y, ACT_RELU6, True)
            a = ("lazy", y, "conv1_bn")
        elif not train and self.fold_frozen_bn and C0 % 4 == 0:
            # inference: conv1_bn (frozen) + ReLU6 in the conv's store, like every depthwise / pointwise layer behind it
            sc, sh = self._frozen_affine("conv1_bn")
            X.call("myolo_conv3x3s2_c3_affine_act_fwd", X.ptr(images), X.ptr(self.p["conv1/kernel"]), sc, sh, ACT_RELU6, X.ptr(y), N, H, W, C0, X.stream())
            a = y
        else:
            X.call("myolo_conv3x3s2_c3_fwd", X.ptr(images), X.ptr(self.p["conv1/kernel"]), X.ptr(y), N, H, W, C0, X.stream())
            a = self.bn_act_fwd("conv1_bn", y, ACT_RELU6, train)
        shape = (N, H // 2, W // 2, C0)
        self.tape["images"] = images
        # the trunk's prepared weights (a few small kernels on the side stream, ~0.1 ms from the start of the step): their first consumers are the
        # pointwise layers with >= 256 input channels (bf16x6 splits) and feature_map's filter transform; the thin first layers read w as it is, so
        # the wait sits in front of the first block with >= 128 channels and not behind conv1, where it stalled the stream for ~50 us
        waited = not train
        bid = 1
        for f, s in BACKBONE_BLOCKS:
            if not waited and (shape[3] >= 128 or not self.wprep_wait_late):
                self._wprep_wait(0)
                waited = True
            a, shape = self.dw_block_fwd(bid, a, shape, s, train)
            bid += 1
        if not waited:
            self._wprep_wait(0)
        a = self._materialize(a)          # C4 feeds feature_map's 3x3 conv and the YOLO head: written once (28x28x512)
        if train:
            self._wprep_phase2()
        C4, c4shape = a, shape
        n, h, w, c = c4shape
        Cf = cfg.TOP_FEATURE_MAP_DEPTH
        Fm = self._new(n * h * w, Cf)
        # (running this conv on a side stream underneath the YOLO head's forward blocks was measured in round 3 for the TRAINING step: 21.55 vs
        # 21.55 / 21.67 ms, nothing -- at batch 32 both branches fill the chip.  An inference forward at batch 4 is a chain of ~45 small launches:
        # there the conv's three kernels run beside the YOLO head's sixteen, infer_fork_feature_map.)
        fork = (not train) and (self.infer_fork_feature_map if self._fork_now is None else self._fork_now) and self._wino_ok(n, h, w, c, Cf)
        if fork:
            cur = torch.cuda.current_stream()
            if self._fm_stream is None:
                self._fm_stream = _shared_stream(self.dev, "infer_feature_map")
            self._fm_stream.wait_stream(cur)          # (inside a capture this pulls the side stream into the graph: the fork is a graph edge)
            with torch.cuda.stream(self._fm_stream):  # the planes it allocates live and die in this stream's pool; Fm and C4 belong to `cur`
                self.conv3x3_fwd(C4, "feature_map", Fm, n, h, w, c, Cf)
        else:
            self.conv3x3_fwd(C4, "feature_map", Fm, n, h, w, c, Cf)
        for f, s in YOLO_BLOCKS:
            a, shape = self.dw_block_fwd(bid, a, shape, s, train)
            bid += 1
        a = self._materialize(a)          # input of conv_23 and of its weight gradient (7x7x1024)
        n2, h2, w2, c2 = shape
        D = cfg.N_BOX * (5 + cfg.NUM_CLASSES)
        yo = self._new(n2 * h2 * w2, D)
        X.call("myolo_pwconv1x1_fwd", X.ptr(a), X.ptr(self.p["conv_23/kernel"]), X.ptr(self.p["conv_23/bias"]), X.ptr(yo),
               n2 * h2 * w2, c2, D, *self._wsargs(), X.stream())
        self.tape["trunk"] = (C4, c4shape, a, shape)
        if fork:
            cur.wait_stream(self._fm_stream)         # Fm is complete for whatever the caller does next (detections, ROIAlign)
        if train:
            self._wprep_mark_trunk()
            self._wprep_wait(1)           # everything else that was prepared (before any stream forks off this one)
        return Fm, (n, h, w, Cf), yo

    def yolo_head_bwd(self, dyolo):
        """conv_23 and the YOLO blocks (conv_dw/pw_7..14) backward: returns the gradient reaching C4 through the YOLO head."""
        C4, c4shape, a14, s14 = self.tape["trunk"]
        n2, h2, w2, c2 = s14
        D = dyolo.shape[1]
        M7 = n2 * h2 * w2
        X.call("myolo_pwconv1x1_bwd_weight", X.ptr(a14), X.ptr(dyolo), X.ptr(self.g["conv_23/kernel"]), M7, c2, D, *self._wsargs(), X.stream())
        self.colsum(dyolo, self.g["conv_23/bias"])
        da = self._new(M7, c2)
        X.call("myolo_pwconv1x1_bwd_data", X.ptr(dyolo), X.ptr(self.p["conv_23/kernel"]), X.ptr(da), M7, c2, D, *self._wsargs(), X.stream())
        bid = len(BACKBONE_BLOCKS) + len(YOLO_BLOCKS)
        first = len(BACKBONE_BLOCKS) + 1
        for _ in YOLO_BLOCKS:
            # (the first YOLO block's input gradient is added to feature_map's before it reaches conv_pw_6_bn: no fused sums there)
            da = self.dw_block_bwd(bid, da, next_bn=("conv_pw_%d_bn" % (bid - 1)) if bid > first else None)
            bid -= 1
        return da

    def start_yolo_head_bwd(self, dyolo):
        """Launch yolo_head_bwd on the side stream (its own scratch buffer).  It depends only on the trunk forward and the YOLO
        loss, so it runs underneath the mask head's forward and backward; its small 7x7 / 14x14 kernels fill a fraction of the
        chip on their own.  trunk_bwd joins it before adding the two gradients of C4.  Same kernels, same results."""
        cur = torch.cuda.current_stream()
        self._yolo_stream.wait_stream(cur)
        self._ws_active = self._ws_side
        try:
            with torch.cuda.stream(self._yolo_stream):
                da = self.yolo_head_bwd(dyolo)
                self._release_yolo_bucket()
        finally:
            self._ws_active = self._ws_main
        self.tape["yolo_bwd"] = da

    def _release_on_wgrad_stream(self, bucket, stream, pending):
        """hand `bucket` to the all-reduce hook once the current stream AND `stream` (the one its weight gradients were queued on) have produced it:
        on `stream`, behind an event of the current stream -- the current stream does not wait"""
        if not self.on_bucket_ready:
            return
        if not pending:
            self.on_bucket_ready(bucket)
            return
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        stream.wait_event(ev)
        with torch.cuda.stream(stream):
            self.on_bucket_ready(bucket)

    def _release_yolo_bucket(self):
        """bucket 1 = YOLO blocks + conv_23: complete when the YOLO head's backward retires -- about half a step before the step ends when that
        backward runs on its side stream under the mask head (start_yolo_head_bwd).  Its data gradients and BatchNorm / conv_23 gradients are on the
        current stream, the blocks' weight gradients on the trunk's weight-gradient stream."""
        self._release_on_wgrad_stream(BUCKET_YOLO, self._twg_stream, self.overlap_trunk_wgrad and self._twg_pending)

    def trunk_bwd(self, dF, dyolo):
        C4, c4shape, a14, s14 = self.tape["trunk"]
        da = self.tape.pop("yolo_bwd", None)
        started = da is not None                # launched earlier on the side stream (start_yolo_head_bwd)
        if not started:
            da = self.yolo_head_bwd(dyolo)
            self._release_yolo_bucket()

        def join():
            if started:
                cur = torch.cuda.current_stream()
                cur.wait_stream(self._yolo_stream)
                da.record_stream(cur)
        bid = len(BACKBONE_BLOCKS)
        n, h, w, c = c4shape
        if dF is None:
            # 'yolo' mode (model.py:906-920): feature_map and the mask head are not in the graph, their gradients are zero
            self.g["feature_map/kernel"].zero_()
            self.g["feature_map/bias"].zero_()
            join()
            dC4 = da
        else:
            Cf = dF.shape[1]
            def fm_wgrad(wsp, wsz):           # feature_map's weight and bias gradients: beside its data gradient and the backbone chain
                prev, self._ws_active = self._ws_active, (self._ws_twg if self.overlap_trunk_wgrad else self._ws_active)
                try:
                    self.conv3x3_bwd_weight(C4, None, dF, "feature_map", n, h, w, c, Cf)
                    self.colsum(dF, self.g["feature_map/bias"])
                finally:
                    self._ws_active = prev
            self._on_wgrad_stream(fm_wgrad, (dF, C4))
            dC4 = self._new(n * h * w, c)
            self.conv3x3_bwd_data(dF, "feature_map", dC4, n, h, w, c, Cf)
            join()
            X.call("myolo_add_inplace", X.ptr(dC4), X.ptr(da), dC4.numel(), X.stream())
        if self.on_bucket_ready:
            # bucket 2 = feature_map: its weight gradient sits on the weight-gradient stream (or was zeroed on this one).  Released ON that
            # stream, behind an event of the compute stream -- the compute stream itself does not wait (round 3 joined the whole
            # weight-gradient stream here, conv1's 2.7 ms weight gradient included, in front of the backbone backward: +3 ms per step whenever
            # a reducer was attached).
            if self.bucket1_on_wgrad_stream and self.overlap_trunk_wgrad and self._twg_pending:
                self._release_on_wgrad_stream(BUCKET_FEATURE_MAP, self._twg_stream, True)
            else:
                self.join_trunk_wgrad()
                self.on_bucket_ready(BUCKET_FEATURE_MAP)
        da = dC4
        if self.backbone_wgrad_on_yolo_stream and self.overlap_trunk_wgrad and self.overlap_yolo_bwd:
            self._twg_override = (self._yolo_stream, self._ws_side)
        try:
            for _ in BACKBONE_BLOCKS:
                da = self.dw_block_bwd(bid, da, next_bn=("conv_pw_%d_bn" % (bid - 1)) if bid > 1 else "conv1_bn")
                bid -= 1
        finally:
            self._twg_override = None
        dy = self.bn_act_bwd("conv1_bn", da)
        images = self.tape["images"]
        N, H, W, _ = images.shape
        C0 = self.p["conv1/kernel"].shape[3]
        X.call("myolo_conv3x3s2_c3_bwd_weight", X.ptr(images), X.ptr(dy), X.ptr(self.g["conv1/kernel"]), N, H, W, C0, *self._wsargs(), X.stream())
        self.join_trunk_wgrad()
        if self.on_bucket_ready:
            self.on_bucket_ready(BUCKET_BACKBONE)

    def _box_image_index(self, B, R):
        """box_ind of crop_and_resize for R boxes per image (cached: no host-synchronising op in the step / under graph capture)."""
        t = self._bind_cache.get((B, R))
        if t is None:
            t = torch.arange(B, device=self.dev, dtype=torch.int32).repeat_interleave(R, output_size=B * R).contiguous()
            self._bind_cache[(B, R)] = t
        return t

    # ---- mask head -----------------------------------------------------------
    def mask_head_fwd(self, Fm, fshape, rois, train, pos_flags=None, keep=None):
        """rois [B,R,4] (x1,y1,x2,y2).  Returns pred masks [B*R, mh*mw, C] (post-sigmoid).  keep = (inv_d, cap): the fused deconv + mask pass
        also writes the ReLU'd deconv output of the ROIs with a compact slot < cap (the sparse backward reads it instead of re-running the deconv)."""
        cfg = self.cfg
        B, R = rois.shape[:2]
        n, h, w, cf = fshape
        ps = cfg.MASK_POOL_SIZE
        if cfg.ROI_BOX_ORDER == "xyxy_as_yxyx":
            boxes = rois.reshape(B * R, 4)                    # model.py:385-387: read as (y1,x1,y2,x2)
        else:
            boxes = rois.reshape(B * R, 4)[:, [1, 0, 3, 2]].contiguous()
        bind = self._box_image_index(B, R)
        NR = B * R
        self.tape["roi"] = (boxes, bind, fshape, NR)
        cin = cf
        convs = []
        fuse = self.sparse_mask_bwd or not train     # frozen BN + ReLU folded into the conv epilogue
        q = ps * ps
        # Winograd chain: where conv_i's epilogue is foldable (frozen BN) and conv_{i+1} is a Winograd conv too, the layer
        # boundary is ONE pass per ROI through LDS (M_i -> V_{i+1}); the activation in between is written only for the ROIs
        # the sparse backward will gather (pos_flags), and not at all in inference.
        chain = (fuse and self._wino_ok(NR, ps, ps, cf, MASK_FILTERS) and self._wino_ok(NR, ps, ps, MASK_FILTERS, MASK_FILTERS)
                 and MASK_FILTERS % 32 == 0 and ((ps + 3) // 4) ** 2 <= 32 and q * 128 <= 65536
                 and (not train or pos_flags is not None))
        if chain:
            # ROIAlign is fused into conv1's input transform: the [NR,14,14,256] crops are never written
            x = self._mask_convs_winograd_chain(None, convs, NR, ps, cf, train, pos_flags, roi=(Fm, boxes, bind, n, h, w),
                                                slots=keep[0] if (keep is not None and self.keep_rows_compact) else None)
        else:
            x = self._new(NR * ps * ps, cf)
            self._call_timed("roialign_fwd", "myolo_crop_and_resize_fwd", X.ptr(Fm), X.ptr(boxes), X.ptr(bind), X.ptr(x),
                             n, h, w, cf, NR, ps, ps, X.stream())
            x = self._mask_convs_layerwise(x, convs, NR, ps, cf, train, fuse)
        C = cfg.NUM_CLASSES
        p = self._new(NR * 4 * ps * ps, C)
        if fuse and C <= 4 and MASK_FILTERS % 128 == 0:
            # deconv + ReLU + 1x1 + sigmoid in one pass; the 28x28x256 tensor is never written.  The sparse backward
            # recomputes it for the positive ROIs (mask_head_bwd_sparse); the dense backward needs it whole.
            self.ws.ensure(X.deconv_mask_ws_bytes(NR, ps, ps, MASK_FILTERS, MASK_FILTERS, C))
            d = None
            if keep is not None and MASK_FILTERS % 256 == 0:
                inv_d, cap = keep
                dk = self._new(cap * 4 * q, MASK_FILTERS)
                X.call("myolo_deconv2x2s2_mask_fwd_keep", X.ptr(x), X.ptr(self.p["myolo_mask_deconv/kernel"]),
                       X.ptr(self.p["myolo_mask_deconv/bias"]), X.ptr(self.p["myolo_mask/kernel"]), X.ptr(self.p["myolo_mask/bias"]),
                       X.ptr(p), NR, ps, ps, MASK_FILTERS, MASK_FILTERS, C, X.ptr(inv_d), X.ptr(dk), cap, *self._wsargs(), X.stream())
                d = ("kept", dk, cap)
            else:
                X.call("myolo_deconv2x2s2_mask_fwd", X.ptr(x), X.ptr(self.p["myolo_mask_deconv/kernel"]),
                       X.ptr(self.p["myolo_mask_deconv/bias"]), X.ptr(self.p["myolo_mask/kernel"]), X.ptr(self.p["myolo_mask/bias"]),
                       X.ptr(p), NR, ps, ps, MASK_FILTERS, MASK_FILTERS, C, *self._wsargs(), X.stream())
        else:
            d = self._new(NR * 4 * ps * ps, MASK_FILTERS)
            X.call("myolo_deconv2x2s2_fwd", X.ptr(x), X.ptr(self.p["myolo_mask_deconv/kernel"]), X.ptr(self.p["myolo_mask_deconv/bias"]),
                   X.ptr(d), NR, ps, ps, MASK_FILTERS, MASK_FILTERS, ACT_RELU, *self._wsargs(), X.stream())
            X.call("myolo_mask_head_out_fwd", X.ptr(d), X.ptr(self.p["myolo_mask/kernel"]), X.ptr(self.p["myolo_mask/bias"]), X.ptr(p),
                   NR * 4 * ps * ps, MASK_FILTERS, C, X.stream())
        self.tape["mask"] = (convs, x, d)
        return p

    def _mask_convs_winograd_chain(self, x, convs, NR, ps, cin, train, pos_flags, roi=None, slots=None):
        """myolo_mask_conv1-4 (+bn, ReLU) as a chain of Winograd stages; appends each conv's input to `convs` (an input the
        forward never materialised is recorded as ("lazy_bn", pre-BN tensor, bn layer)).  Returns conv4's activation.
        cfg.WINOGRAD_TILES = "f63": conv2-4 (whose inputs and outputs are MASK_FILTERS wide 14x14 maps) use the F(6,3)/F(4,3)
        tiling of csrc/wino63_kernels.hip (400 instead of 484 point-tiles per ROI); conv1 keeps the F(4,3)/F(2,3) tiling (its input
        transform is fused with ROIAlign and its V planes feed the weight gradient)."""
        q = ps * ps
        Vcur = None
        t63 = self.wino_tiles == "f63" and X.wino63_ok(ps, ps, MASK_FILTERS, MASK_FILTERS)
        # conv1 too, when its backward has the matching kernels (the lazy-BN gradients of the sparse backward) or there is none
        c1_63 = (t63 and X.wino63_ok(ps, ps, cin, MASK_FILTERS) and X.wino63_ok(ps, ps, MASK_FILTERS, cin)
                 and (not train or (self.lazy_bn1_bwd and self.sparse_mask_bwd)))
        for i in range(1, 5):
            cn, bn = "myolo_mask_conv%d" % i, "myolo_mask_bn%d" % i
            batch_stats = train and i == 1
            fold = not batch_stats
            use63 = t63 and (i >= 2 or c1_63)          # this conv's V / M planes are in the F(6,3) layout
            next63 = t63 and i < 4                     # ... and so are the next conv's
            T = NR * ((ps + 3) // 4) ** 2
            start, stop = self._timed("mask_conv3x3_fwd")
            start()
            if Vcur is None:
                if use63:
                    Vcur = self._new(X.wino63_plane_elems(NR, cin))
                    if x is None:                         # conv1: crops sampled from the feature map on the fly (roi)
                        Fm, boxes, bind, fn, fh, fw = roi
                        self._call_timed("roialign_fwd", "myolo_wino63_input_transform_roialign", X.ptr(Fm), X.ptr(boxes), X.ptr(bind),
                                         X.ptr(Vcur), fn, fh, fw, cin, NR, X.stream())
                    else:
                        self._call_timed("wino_in", "myolo_wino63_input_transform", X.ptr(x), None, None, ACT_NONE, None, None, X.ptr(Vcur),
                                         NR, cin, X.stream())
                else:
                    Vcur = self._new(36, T, cin)
                    if x is None:                         # conv1: crops sampled from the feature map on the fly (roi)
                        Fm, boxes, bind, fn, fh, fw = roi
                        self._call_timed("roialign_fwd", "myolo_wino_input_transform_roialign", X.ptr(Fm), X.ptr(boxes), X.ptr(bind),
                                         X.ptr(Vcur), fn, fh, fw, cin, NR, ps, ps, X.stream())
                    else:
                        self._call_timed("wino_in", "myolo_wino_input_transform", X.ptr(x), X.ptr(Vcur), NR, ps, ps, cin, X.stream())
            if use63:
                U, M = self._new(X.wino63_u_elems(cin, MASK_FILTERS)), self._new(X.wino63_plane_elems(NR, MASK_FILTERS))
                # (the transformed filters: prepared at the step's start in training, X.WeightPrep, else formed into U here)
                self._call_timed("wino_multiply", "myolo_wino63_multiply_w", X.ptr(Vcur), X.ptr(self.p[cn + "/kernel"]), X.ptr(U), X.ptr(M), NR, cin,
                                 MASK_FILTERS, X.stream())
            else:
                U, M = self._new(X.wino_u_elems(cin, MASK_FILTERS)), self._new(36, T, MASK_FILTERS)
                X.call("myolo_wino_weight_transform", X.ptr(self.p[cn + "/kernel"]), X.ptr(U), cin, MASK_FILTERS, 0, X.stream())
                self._call_timed("wino_multiply", "myolo_wino_multiply", X.ptr(Vcur), X.ptr(U), X.ptr(M), NR, ps, ps, cin, MASK_FILTERS,
                                 X.stream())
            if i == 1 and train:
                self.tape["conv1_V"] = Vcur          # reused by conv1's weight gradient
                self.tape["conv1_V_fmt"] = "f63" if use63 else "f43"
            convs.append(x)                          # for i >= 3 in training: valid only in the rows of flagged ROIs
            bias = self.p[cn + "/bias"]
            buf = self.bnbuf[bn]
            if fold:
                X.call("myolo_bn_frozen_coeffs", X.ptr(self.p[bn + "/gamma"]), X.ptr(self.p[bn + "/beta"]),
                       X.ptr(self.s[bn + "/moving_mean"]), X.ptr(self.s[bn + "/moving_variance"]),
                       X.ptr(buf[2]), X.ptr(buf[3]), MASK_FILTERS, X.stream())
                self.tape[bn] = (None, ACT_RELU, False)       # pre-BN tensor never materialised
            if fold and i < 4 and use63 == next63:
                ykeep = self._new(NR * q, MASK_FILTERS) if train else None
                keep_pre = bool(train and self.mask_keep_pre)     # what is kept for the positive ROIs is the conv's PRE-BatchNorm output (exact backward, any gamma)
                if use63:
                    Vn = self._new(X.wino63_plane_elems(NR, MASK_FILTERS))
                    if keep_pre and slots is not None:
                        # the kept rows in COMPACT order (slot of each positive ROI, myolo_positive_index): the sparse backward reads them without a gather
                        self._call_timed("wino_out_in", "myolo_wino63_output_input_transform_keep_pre_slots", X.ptr(M), X.ptr(bias), X.ptr(buf[2]),
                                         X.ptr(buf[3]), X.ptr(ykeep), X.ptr(slots), NR, X.ptr(Vn), NR, MASK_FILTERS, ACT_RELU, X.stream())
                        self.tape.setdefault("compact_rows", set()).add(id(ykeep))
                    elif keep_pre:
                        self._call_timed("wino_out_in", "myolo_wino63_output_input_transform_keep_pre", X.ptr(M), X.ptr(bias), X.ptr(buf[2]),
                                         X.ptr(buf[3]), X.ptr(ykeep), X.ptr(pos_flags), X.ptr(Vn), NR, MASK_FILTERS, ACT_RELU, X.stream())
                    else:
                        self._call_timed("wino_out_in", "myolo_wino63_output_input_transform", X.ptr(M), X.ptr(bias), X.ptr(buf[2]),
                                         X.ptr(buf[3]), X.ptr(ykeep), X.ptr(pos_flags) if train else None, X.ptr(Vn), NR, MASK_FILTERS,
                                         ACT_RELU, X.stream())
                else:
                    Vn = self._new(36, T, MASK_FILTERS)
                    self._call_timed("wino_out_in", "myolo_wino_output_input_transform_keep_pre" if keep_pre else "myolo_wino_output_input_transform",
                                     X.ptr(M), X.ptr(bias), X.ptr(buf[2]), X.ptr(buf[3]), X.ptr(ykeep), X.ptr(pos_flags) if train else None, X.ptr(Vn),
                                     NR, ps, ps, MASK_FILTERS, ACT_RELU, X.stream())
                if keep_pre:
                    # the next conv's input and this BatchNorm's backward are both formed from the kept pre-BN rows (mask_head_bwd_sparse: "lazy_bn")
                    self.tape[bn] = (ykeep, ACT_RELU, False)
                    x, Vcur = ("lazy_bn", ykeep, bn), Vn
                else:
                    x, Vcur = ykeep, Vn
            else:
                y = self._new(NR * q, MASK_FILTERS)
                if fold:
                    if use63 and train and pos_flags is not None and self.mask_keep_pre:
                        ypre = self._new(NR * q, MASK_FILTERS)
                        if slots is not None:
                            X.call("myolo_wino63_output_transform_keep_pre_slots", X.ptr(M), X.ptr(bias), X.ptr(buf[2]), X.ptr(buf[3]), X.ptr(y), X.ptr(ypre),
                                   X.ptr(slots), NR, NR, MASK_FILTERS, ACT_RELU, X.stream())
                            self.tape.setdefault("compact_rows", set()).add(id(ypre))
                        else:
                            X.call("myolo_wino63_output_transform_keep_pre", X.ptr(M), X.ptr(bias), X.ptr(buf[2]), X.ptr(buf[3]), X.ptr(y), X.ptr(ypre),
                                   X.ptr(pos_flags), NR, MASK_FILTERS, ACT_RELU, X.stream())
                        self.tape[bn] = (ypre, ACT_RELU, False)
                    elif use63:
                        X.call("myolo_wino63_output_transform", X.ptr(M), X.ptr(bias), X.ptr(buf[2]), X.ptr(buf[3]), X.ptr(y), NR, MASK_FILTERS,
                               ACT_RELU, X.stream())
                    else:
                        X.call("myolo_wino_output_transform", X.ptr(M), X.ptr(bias), X.ptr(buf[2]), X.ptr(buf[3]), X.ptr(y), NR, ps, ps,
                               MASK_FILTERS, ACT_RELU, X.stream())
                        if train and self.mask_keep_pre and self.sparse_mask_bwd:
                            # (the F(4,3)-tiling output transform has no flagged second output: a second pass writes the pre-BatchNorm rows of every ROI.
                            #  Non-default tiling; the F(6,3) boundary above writes the flagged ROIs' rows in the same pass)
                            ypre = self._new(NR * q, MASK_FILTERS)
                            X.call("myolo_wino_output_transform", X.ptr(M), X.ptr(bias), None, None, X.ptr(ypre), NR, ps, ps, MASK_FILTERS, ACT_NONE, X.stream())
                            self.tape[bn] = (ypre, ACT_RELU, False)
                    x = y
                    Vcur = None                      # (a tiling change at this boundary: the next conv transforms y itself)
                elif 256 % (MASK_FILTERS // 4) == 0 and i < 4:
                    # training-mode BN behind this conv (bn1): its statistics come out of the output transform, and its
                    # apply + ReLU go into the next conv's input transform -- the normalised activation is never written
                    # (the sparse backward re-applies it to the positive ROIs' rows)
                    if use63:
                        self.ws.ensure(X.wino63_out_bn_ws_bytes(NR, MASK_FILTERS))
                        X.call("myolo_wino63_output_transform_bn_stats", X.ptr(M), X.ptr(bias), X.ptr(y), NR, MASK_FILTERS,
                               X.ptr(self.p[bn + "/gamma"]), X.ptr(self.p[bn + "/beta"]), X.ptr(buf[0]), X.ptr(buf[1]), X.ptr(buf[2]),
                               X.ptr(buf[3]), X.ptr(self.s[bn + "/moving_mean"]), X.ptr(self.s[bn + "/moving_variance"]),
                               *self._wsargs(), X.stream())
                    else:
                        self.ws.ensure(X.wino_out_bn_ws_bytes(MASK_FILTERS))
                        X.call("myolo_wino_output_transform_bn_stats", X.ptr(M), X.ptr(bias), X.ptr(y), NR, ps, ps, MASK_FILTERS,
                               X.ptr(self.p[bn + "/gamma"]), X.ptr(self.p[bn + "/beta"]), X.ptr(buf[0]), X.ptr(buf[1]), X.ptr(buf[2]),
                               X.ptr(buf[3]), X.ptr(self.s[bn + "/moving_mean"]), X.ptr(self.s[bn + "/moving_variance"]),
                               *self._wsargs(), X.stream())
                    self.tape[bn] = (y, ACT_RELU, True)
                    if next63:
                        Vcur = self._new(X.wino63_plane_elems(NR, MASK_FILTERS))
                        if slots is not None and train:
                            # ... and conv2's input (bn1's activation) of the positive ROIs in compact order: conv2's weight gradient reads it as it is
                            a1k = self._new(NR * q, MASK_FILTERS)
                            X.call("myolo_wino63_input_transform_slots", X.ptr(y), X.ptr(buf[2]), X.ptr(buf[3]), ACT_RELU, X.ptr(a1k), X.ptr(slots), NR,
                                   X.ptr(Vcur), NR, MASK_FILTERS, X.stream())
                            self.tape["conv2_in_rows"] = a1k
                        else:
                            X.call("myolo_wino63_input_transform", X.ptr(y), X.ptr(buf[2]), X.ptr(buf[3]), ACT_RELU, None, None, X.ptr(Vcur),
                                   NR, MASK_FILTERS, X.stream())
                    else:
                        Vcur = self._new(36, T, MASK_FILTERS)
                        X.call("myolo_wino_input_transform_affine", X.ptr(y), X.ptr(buf[2]), X.ptr(buf[3]), ACT_RELU, X.ptr(Vcur),
                               NR, ps, ps, MASK_FILTERS, X.stream())
                    x = ("lazy_bn", y, bn)
                else:
                    X.call("myolo_wino_output_transform", X.ptr(M), X.ptr(bias), None, None, X.ptr(y), NR, ps, ps, MASK_FILTERS,
                           ACT_NONE, X.stream())
                    x = self.bn_act_fwd(bn, y, ACT_RELU, batch_stats)
                    Vcur = None
            stop()
            cin = MASK_FILTERS
        return x

    def _mask_convs_layerwise(self, x, convs, NR, ps, cin, train, fuse):
        """myolo_mask_conv1-4 one self-contained conv op at a time (direct kernels or un-chained Winograd)."""
        for i in range(1, 5):
            cn, bn = "myolo_mask_conv%d" % i, "myolo_mask_bn%d" % i
            y = self._new(NR * ps * ps, MASK_FILTERS)
            convs.append(x)
            # bn1 uses batch statistics in training (model.py:690 has no training= argument);
            # bn2-4 are called with training=False (model.py:696,702,708) -> moving statistics
            batch_stats = train and i == 1
            if fuse and not batch_stats:
                buf = self.bnbuf[bn]
                X.call("myolo_bn_frozen_coeffs", X.ptr(self.p[bn + "/gamma"]), X.ptr(self.p[bn + "/beta"]),
                       X.ptr(self.s[bn + "/moving_mean"]), X.ptr(self.s[bn + "/moving_variance"]),
                       X.ptr(buf[2]), X.ptr(buf[3]), MASK_FILTERS, X.stream())
                self.conv3x3_fwd(x, cn, y, NR, ps, ps, cin, MASK_FILTERS, scale=buf[2], shift=buf[3], act=ACT_RELU,
                                 tag="mask_conv3x3_fwd")
                self.tape[bn] = (None, ACT_RELU, False)       # pre-BN tensor never materialised
                x = y
            else:
                # conv1 in training: its transformed input is kept for the dense weight gradient
                v = self.conv3x3_fwd(x, cn, y, NR, ps, ps, cin, MASK_FILTERS, keep_v=train and i == 1 and self.sparse_mask_bwd,
                                     tag="mask_conv3x3_fwd")
                if v is not None:
                    self.tape["conv1_V"] = v
                x = self.bn_act_fwd(bn, y, ACT_RELU, batch_stats)
            cin = MASK_FILTERS
        return x

    def mask_head_fwd_bf16(self, Fm, fshape, rois):
        """Inference-only mask head with bf16 activations / fp32 accumulation (cfg.INFERENCE_DTYPE == "bf16").
        Same graph as mask_head_fwd(train=False) (model.py:680-714); the frozen BN of each conv is folded
        into bf16 weights -- packed once per weight version into persistent buffers (_bf16_operand / _infer_prep_sync)."""
        cfg = self.cfg
        B, R = rois.shape[:2]
        n, h, w, cf = fshape
        ps = cfg.MASK_POOL_SIZE
        if cfg.ROI_BOX_ORDER == "xyxy_as_yxyx":
            boxes = rois.reshape(B * R, 4)
        else:
            boxes = rois.reshape(B * R, 4)[:, [1, 0, 3, 2]].contiguous()
        bind = self._box_image_index(B, R)
        NR = B * R
        bf = torch.bfloat16
        x = self._new(NR * ps * ps, cf, dtype=bf)
        self._call_timed("roialign_fwd", "myolo_crop_and_resize_bf16_fwd", X.ptr(Fm), X.ptr(boxes), X.ptr(bind), X.ptr(x),
                         n, h, w, cf, NR, ps, ps, X.stream())
        cin = cf
        for i in range(1, 5):
            wt, bfold = self._bf16_operand(i, cin)
            y = self._new(NR * ps * ps, MASK_FILTERS, dtype=bf)
            self._call_timed("mask_conv3x3_fwd", "myolo_conv3x3_bf16_fwd", X.ptr(x), X.ptr(wt), X.ptr(bfold), X.ptr(y),
                             NR, ps, ps, cin, MASK_FILTERS, ACT_RELU, X.stream())
            x, cin = y, MASK_FILTERS
        wt, _ = self._bf16_operand("deconv", MASK_FILTERS)
        C = cfg.NUM_CLASSES
        p = self._new(NR * 4 * ps * ps, C)
        if C <= 4 and MASK_FILTERS % 128 == 0:       # deconv + ReLU + 1x1 + sigmoid fused: the 28x28x256 tensor is never written
            self.ws.ensure((MASK_FILTERS // 128) * 2 * 4 * NR * ps * ps * C * 4)
            self._call_timed("mask_deconv_fwd", "myolo_deconv2x2s2_mask_bf16_fwd", X.ptr(x), X.ptr(wt),
                             X.ptr(self.p["myolo_mask_deconv/bias"]), X.ptr(self.p["myolo_mask/kernel"]), X.ptr(self.p["myolo_mask/bias"]),
                             X.ptr(p), NR, ps, ps, MASK_FILTERS, MASK_FILTERS, C, *self._wsargs(), X.stream())
            return p
        d = self._new(NR * 4 * ps * ps, MASK_FILTERS, dtype=bf)
        self._call_timed("mask_deconv_fwd", "myolo_deconv2x2s2_bf16_fwd", X.ptr(x), X.ptr(wt), X.ptr(self.p["myolo_mask_deconv/bias"]),
                         X.ptr(d), NR, ps, ps, MASK_FILTERS, MASK_FILTERS, ACT_RELU, X.stream())
        X.call("myolo_mask_head_out_bf16_fwd", X.ptr(d), X.ptr(self.p["myolo_mask/kernel"]), X.ptr(self.p["myolo_mask/bias"]), X.ptr(p),
               NR * 4 * ps * ps, MASK_FILTERS, C, X.stream())
        return p

    def mask_head_bwd(self, dz):
        """dz [NR*mh*mw, C] gradient wrt the pre-sigmoid mask logits.  Returns dF."""
        cfg = self.cfg
        convs, a4, d = self.tape["mask"]
        boxes, bind, fshape, NR = self.tape["roi"]
        ps = cfg.MASK_POOL_SIZE
        C = cfg.NUM_CLASSES
        Md = NR * 4 * ps * ps
        dd = self._new(Md, MASK_FILTERS)
        X.call("myolo_mask_head_out_bwd", X.ptr(d), X.ptr(self.p["myolo_mask/kernel"]), X.ptr(dz), X.ptr(dd),
               X.ptr(self.g["myolo_mask/kernel"]), X.ptr(self.g["myolo_mask/bias"]), Md, MASK_FILTERS, C, *self._wsargs(), X.stream())
        X.call("myolo_deconv2x2s2_bwd_weight", X.ptr(a4), X.ptr(dd), X.ptr(self.g["myolo_mask_deconv/kernel"]), NR, ps, ps,
               MASK_FILTERS, MASK_FILTERS, *self._wsargs(), X.stream())
        self.colsum(dd, self.g["myolo_mask_deconv/bias"])
        da = self._new(NR * ps * ps, MASK_FILTERS)
        X.call("myolo_deconv2x2s2_bwd_data", X.ptr(dd), X.ptr(self.p["myolo_mask_deconv/kernel"]), X.ptr(da), NR, ps, ps,
               MASK_FILTERS, MASK_FILTERS, *self._wsargs(), X.stream())
        del dd
        for i in range(4, 0, -1):
            cn = "myolo_mask_conv%d" % i
            dy = self.bn_act_bwd("myolo_mask_bn%d" % i, da)
            xin = convs[i - 1]
            cin = xin.shape[1]
            self.conv3x3_bwd_weight(xin, None, dy, cn, NR, ps, ps, cin, MASK_FILTERS)
            self.colsum(dy, self.g[cn + "/bias"])
            da = self._new(NR * ps * ps, cin)
            self.conv3x3_bwd_data(dy, cn, da, NR, ps, ps, cin, MASK_FILTERS)
        n, h, w, cf = fshape
        dF = self._new(n * h * w, cf)
        X.call("myolo_roialign_bwd_grouped", X.ptr(da), X.ptr(boxes), X.ptr(dF), n, h, w, cf, NR // n, ps, ps, X.stream())
        if self.on_bucket_ready:
            self.on_bucket_ready(BUCKET_MASK_REST)
            self.on_bucket_ready(BUCKET_MASK_CONV1)
        return dF

    def _gather(self, t, idx, n, group_rows):
        C = t.shape[1]
        out = self._new(n * group_rows, C)
        X.call("myolo_gather_groups", X.ptr(t), X.ptr(idx), X.ptr(out), n, group_rows * C, X.stream())
        return out

    def _start_npos_copy(self, npos):
        """async D2H of the per-image positive counts on a side stream (read at the start of backward)."""
        B = npos.shape[0]
        if self._npos_pinned is None or self._npos_pinned.shape[0] != B:
            self._npos_pinned = torch.empty(B, dtype=torch.int32, pin_memory=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(ev)
            self._npos_pinned.copy_(npos, non_blocking=True)
            self._npos_ready.record(self._copy_stream)

    def _positive_index(self, B, R):
        """(NP, idx_d, inv_d): flat ROI indices of the positives (the first n_pos rows of each image,
        model.py:593) and the inverse map, from the async copy started by _start_npos_copy."""
        t0 = time.perf_counter()
        self._npos_ready.synchronize()          # the step's one host wait: the per-image positive counts (32 ints) from mid-forward
        self.host_wait_s += time.perf_counter() - t0
        npos_h = self._npos_pinned.numpy()
        if "pos_index" in self.tape:            # built on the device in front of the forward (myolo_positive_index): only the total is needed here
            NP = int(np.clip(npos_h[:B], 0, R).sum())
            self._np_seen = NP
            return (NP,) + (self.tape["pos_index"] if NP else (None, None))
        pos = np.concatenate([np.arange(b * R, b * R + int(npos_h[b]), dtype=np.int32) for b in range(B)]) if B else np.zeros(0, np.int32)
        NP = int(pos.shape[0])
        if NP == 0:
            return 0, None, None
        inv = np.full(B * R, -1, np.int32)
        inv[pos] = np.arange(NP, dtype=np.int32)
        return NP, torch.from_numpy(pos).to(self.dev, non_blocking=True), torch.from_numpy(inv).to(self.dev, non_blocking=True)

    def mask_head_fwd_positives(self, Fm, fshape, rois, tmask, tcls):
        """Training forward of the mask head that spends conv2-4 / deconv / myolo_mask on the positive ROIs only.
        Exact, not approximate: the mask loss reads only positive ROIs (model.py:739-746) and bn2-4 are frozen
        (model.py:696,702,708: no batch statistics, no moving-average update), so nothing downstream of bn1 in a
        non-positive ROI reaches the loss, a gradient or any state.  ROIAlign, conv1 and bn1's batch statistics
        (model.py:690) still cover every ROI.  Returns (pred_p [NP*mh*mw, C] | None, tmask_p, tcls_p)."""
        cfg = self.cfg
        B, R = rois.shape[:2]
        n, h, w, cf = fshape
        ps = cfg.MASK_POOL_SIZE
        q = ps * ps
        if cfg.ROI_BOX_ORDER == "xyxy_as_yxyx":
            boxes = rois.reshape(B * R, 4)
        else:
            boxes = rois.reshape(B * R, 4)[:, [1, 0, 3, 2]].contiguous()
        bind = self._box_image_index(B, R)
        NR = B * R
        self.tape["roi"] = (boxes, bind, fshape, NR)
        y1 = self._new(NR * q, MASK_FILTERS)
        bn = "myolo_mask_bn1"
        buf = self.bnbuf[bn]
        x = None
        if self._wino_ok(NR, ps, ps, cf, MASK_FILTERS) and 256 % (MASK_FILTERS // 4) == 0:
            T = NR * ((ps + 3) // 4) ** 2
            start, stop = self._timed("mask_conv3x3_fwd")
            start()
            bn_args = (X.ptr(self.p[bn + "/gamma"]), X.ptr(self.p[bn + "/beta"]), X.ptr(buf[0]), X.ptr(buf[1]), X.ptr(buf[2]),
                       X.ptr(buf[3]), X.ptr(self.s[bn + "/moving_mean"]), X.ptr(self.s[bn + "/moving_variance"]))
            c1_63 = (self.wino_tiles == "f63" and X.wino63_ok(ps, ps, cf, MASK_FILTERS) and X.wino63_ok(ps, ps, MASK_FILTERS, cf)
                     and self.lazy_bn1_bwd and self.sparse_mask_bwd)
            if c1_63:          # the F(6,3)/F(4,3) tiling, as in _mask_convs_winograd_chain
                V, U, M = (self._new(X.wino63_plane_elems(NR, cf)), self._new(X.wino63_u_elems(cf, MASK_FILTERS)),
                           self._new(X.wino63_plane_elems(NR, MASK_FILTERS)))
                self._call_timed("roialign_fwd", "myolo_wino63_input_transform_roialign", X.ptr(Fm), X.ptr(boxes), X.ptr(bind), X.ptr(V),
                                 n, h, w, cf, NR, X.stream())
                self._call_timed("wino_multiply", "myolo_wino63_multiply_w", X.ptr(V), X.ptr(self.p["myolo_mask_conv1/kernel"]), X.ptr(U), X.ptr(M), NR, cf,
                                 MASK_FILTERS, X.stream())
                self.ws.ensure(X.wino63_out_bn_ws_bytes(NR, MASK_FILTERS))
                X.call("myolo_wino63_output_transform_bn_stats", X.ptr(M), X.ptr(self.p["myolo_mask_conv1/bias"]), X.ptr(y1), NR, MASK_FILTERS,
                       *bn_args, *self._wsargs(), X.stream())
            else:
                V, U, M = self._new(36, T, cf), self._new(X.wino_u_elems(cf, MASK_FILTERS)), self._new(36, T, MASK_FILTERS)
                self._call_timed("roialign_fwd", "myolo_wino_input_transform_roialign", X.ptr(Fm), X.ptr(boxes), X.ptr(bind), X.ptr(V),
                                 n, h, w, cf, NR, ps, ps, X.stream())        # ROIAlign fused into the input transform
                X.call("myolo_wino_weight_transform", X.ptr(self.p["myolo_mask_conv1/kernel"]), X.ptr(U), cf, MASK_FILTERS, 0, X.stream())
                self._call_timed("wino_multiply", "myolo_wino_multiply", X.ptr(V), X.ptr(U), X.ptr(M), NR, ps, ps, cf, MASK_FILTERS, X.stream())
                self.ws.ensure(X.wino_out_bn_ws_bytes(MASK_FILTERS))
                X.call("myolo_wino_output_transform_bn_stats", X.ptr(M), X.ptr(self.p["myolo_mask_conv1/bias"]), X.ptr(y1), NR, ps, ps,
                       MASK_FILTERS, *bn_args, *self._wsargs(), X.stream())
            self.tape["conv1_V_fmt"] = "f63" if c1_63 else "f43"
            stop()
            self.tape["conv1_V"] = V
        else:
            x = self._new(NR * q, cf)
            self._call_timed("roialign_fwd", "myolo_crop_and_resize_fwd", X.ptr(Fm), X.ptr(boxes), X.ptr(bind), X.ptr(x),
                             n, h, w, cf, NR, ps, ps, X.stream())
            v = self.conv3x3_fwd(x, "myolo_mask_conv1", y1, NR, ps, ps, cf, MASK_FILTERS, keep_v=True, tag="mask_conv3x3_fwd")
            if v is not None:
                self.tape["conv1_V"] = v
            X.call("myolo_bn_stats", X.ptr(y1), X.ptr(self.p[bn + "/gamma"]), X.ptr(self.p[bn + "/beta"]),
                   X.ptr(buf[0]), X.ptr(buf[1]), X.ptr(buf[2]), X.ptr(buf[3]),
                   X.ptr(self.s[bn + "/moving_mean"]), X.ptr(self.s[bn + "/moving_variance"]),
                   NR * q, MASK_FILTERS, *self._wsargs(), X.stream())
        self.tape[bn] = (y1, ACT_RELU, True)
        NP, idx_d, inv_d = self._positive_index(B, R)
        self.tape["compact"] = (NP, idx_d, inv_d)
        if NP == 0:
            self.tape["mask"] = ([x], None, None)
            return None, None, None
        c1_p = self._gather(y1, idx_d, NP, q)
        a = self._new(NP * q, MASK_FILTERS)
        X.call("myolo_bn_apply_act", X.ptr(c1_p), X.ptr(buf[2]), X.ptr(buf[3]), X.ptr(a), NP * q, MASK_FILTERS, ACT_RELU, X.stream())
        convs = [x]
        for i in range(2, 5):
            cn, bn = "myolo_mask_conv%d" % i, "myolo_mask_bn%d" % i
            convs.append(a)
            y = self._new(NP * q, MASK_FILTERS)
            self.conv3x3_fwd(a, cn, y, NP, ps, ps, MASK_FILTERS, MASK_FILTERS)
            a = self.bn_act_fwd(bn, y, ACT_RELU, False)          # keeps the pre-BN tensor for backward
        d = self._new(NP * 4 * q, MASK_FILTERS)
        X.call("myolo_deconv2x2s2_fwd", X.ptr(a), X.ptr(self.p["myolo_mask_deconv/kernel"]), X.ptr(self.p["myolo_mask_deconv/bias"]),
               X.ptr(d), NP, ps, ps, MASK_FILTERS, MASK_FILTERS, ACT_RELU, *self._wsargs(), X.stream())
        C = cfg.NUM_CLASSES
        pred = self._new(NP * 4 * q, C)
        X.call("myolo_mask_head_out_fwd", X.ptr(d), X.ptr(self.p["myolo_mask/kernel"]), X.ptr(self.p["myolo_mask/bias"]), X.ptr(pred),
               NP * 4 * q, MASK_FILTERS, C, X.stream())
        self.tape["mask"] = (convs, a, d)
        tmask_p = self._new(NP, 4 * q)
        X.call("myolo_gather_groups", X.ptr(tmask), X.ptr(idx_d), X.ptr(tmask_p), NP, 4 * q, X.stream())
        tcls_p = self._new(NP, dtype=torch.int32)
        X.call("myolo_gather_groups", X.ptr(tcls), X.ptr(idx_d), X.ptr(tcls_p), NP, 1, X.stream())
        return pred, tmask_p, tcls_p

    def mask_head_bwd_sparse(self, dz, B, R):
        """Same gradients as mask_head_bwd, exploiting a structural zero: bn2-4 are frozen affine maps
        (model.py:696,702,708) and the mask loss only reads positive ROIs (model.py:739-746), so behind bn1
        every non-positive ROI's gradient is exactly 0.  conv2-4 / deconv / myolo_mask backward therefore run
        on the positive ROIs only (compacted); bn1 (batch statistics, model.py:690) and conv1 stay dense,
        because bn1's backward spreads gradient to every ROI.  Positives are the first n_pos rows of each
        image (detect_mask_target_graph puts them first, model.py:593)."""
        cfg = self.cfg
        convs, a4, d = self.tape["mask"]
        boxes, bind, fshape, NR = self.tape["roi"]
        ps = cfg.MASK_POOL_SIZE
        C = cfg.NUM_CLASSES
        n, h, w, cf = fshape
        compact = "compact" in self.tape          # forward already ran on the positives only
        NP, idx_d, inv_d = self.tape["compact"] if compact else self._positive_index(B, R)
        if NP == 0:                       # no positive ROI: mask loss is the constant 0 (model.py:750-752)
            self.flat_g[self.bucket_ranges[BUCKET_MASK_CONV1][0]:self.bucket_ranges[BUCKET_MASK_REST][1]].zero_()
            dF = torch.zeros(n * h * w, cf, dtype=torch.float32, device=self.dev)
            if self.on_bucket_ready:
                self.on_bucket_ready(BUCKET_MASK_REST)
                self.on_bucket_ready(BUCKET_MASK_CONV1)
            return dF
        q = ps * ps
        kept = self.tape.get("compact_rows", ())          # tensors the forward wrote in compact order (rows of positive k at block k)

        def gather(t, rows):
            if compact:
                return t
            if id(t) in kept:
                return t[:NP * rows]
            return self._gather(t, idx_d, NP, rows)
        dz_p = gather(dz, 4 * q)
        # conv4's activation on the positives: re-formed from its kept pre-BatchNorm rows where those are compact (only the deconv's weight gradient reads it
        # then, off the chain), else gathered
        pre4 = self.tape["myolo_mask_bn4"][0]
        a4_lazy = (not compact and torch.is_tensor(pre4) and id(pre4) in kept and isinstance(d, tuple) and NP <= d[2])
        a4_p = None if a4_lazy else gather(a4, q)
        if isinstance(d, tuple):          # ("kept", rows, cap): the fused forward wrote the positives' deconv output in compact order
            d_p = d[1] if NP <= d[2] else None
            d = None
        else:
            d_p = None
        if d_p is not None:
            pass
        elif d is None:          # fused forward (deconv + 1x1 in one pass): rebuild the deconv output of the positives
            d_p = self._new(NP * 4 * q, MASK_FILTERS)
            X.call("myolo_deconv2x2s2_fwd", X.ptr(a4_p), X.ptr(self.p["myolo_mask_deconv/kernel"]),
                   X.ptr(self.p["myolo_mask_deconv/bias"]), X.ptr(d_p), NP, ps, ps, MASK_FILTERS, MASK_FILTERS, ACT_RELU,
                   *self._wsargs(), X.stream())
        else:
            d_p = gather(d, 4 * q)
        Md = NP * 4 * q
        dd = self._new(Md, MASK_FILTERS)
        X.call("myolo_mask_head_out_bwd", X.ptr(d_p), X.ptr(self.p["myolo_mask/kernel"]), X.ptr(dz_p), X.ptr(dd),
               X.ptr(self.g["myolo_mask/kernel"]), X.ptr(self.g["myolo_mask/bias"]), Md, MASK_FILTERS, C, *self._wsargs(), X.stream())
        # The compacted part is a chain of data gradients (deconv -> conv4 -> conv3 -> conv2 -> bn1's coefficients) that conv1's dense backward
        # waits for; the weight / bias gradients hanging off it have no consumer before the bucket's all-reduce.  They go to the stream (and
        # scratch) conv1's weight gradient uses later: queued in front of it, finished before the bucket is released there.
        side_wg = bool(self.overlap_compact_wgrad and self.overlap_conv1_wgrad)

        def off_chain(fn, tensors):
            if not side_wg:
                return fn()
            cur = torch.cuda.current_stream()
            ev = torch.cuda.Event()
            ev.record(cur)
            self._wgrad_stream.wait_event(ev)
            saved = self._ws_active
            self._ws_active = self._ws_wgrad
            try:
                with torch.cuda.stream(self._wgrad_stream):
                    fn()
            finally:
                self._ws_active = saved
            for t in tensors:
                t.record_stream(self._wgrad_stream)
            self._wgrad_pending = True

        a4_w = self._new(NP * q, MASK_FILTERS) if a4_lazy else a4_p

        def deconv_wgrad():
            if a4_lazy:
                b4 = self.bnbuf["myolo_mask_bn4"]
                X.call("myolo_bn_apply_act", X.ptr(pre4), X.ptr(b4[2]), X.ptr(b4[3]), X.ptr(a4_w), NP * q, MASK_FILTERS, ACT_RELU, X.stream())
            X.call("myolo_deconv2x2s2_bwd_weight", X.ptr(a4_w), X.ptr(dd), X.ptr(self.g["myolo_mask_deconv/kernel"]), NP, ps, ps,
                   MASK_FILTERS, MASK_FILTERS, *self._wsargs(), X.stream())
            self.colsum(dd, self.g["myolo_mask_deconv/bias"])
        off_chain(deconv_wgrad, (a4_w, dd) + ((pre4,) if a4_lazy else ()))
        da = self._new(NP * q, MASK_FILTERS)
        X.call("myolo_deconv2x2s2_bwd_data", X.ptr(dd), X.ptr(self.p["myolo_mask_deconv/kernel"]), X.ptr(da), NP, ps, ps,
               MASK_FILTERS, MASK_FILTERS, *self._wsargs(), X.stream())
        pre_rows = {}                         # id(pre-BN tensor) -> its positive rows: layer i's input tensor is layer i-1's BatchNorm input, gathered ONCE
        for i in range(4, 1, -1):
            cn, bn = "myolo_mask_conv%d" % i, "myolo_mask_bn%d" % i
            src = convs[i - 1]
            lazy_xin = None
            if isinstance(src, tuple) and not compact and id(src[1]) in kept:
                # the forward kept this conv's input rows compact, as pre-BatchNorm values: the BatchNorm backward below reads them as they are, and the
                # normalised form -- only this conv's WEIGHT gradient needs it -- is formed where that runs
                _, ypre, bsrc = src
                yp = ypre[:NP * q]
                xin = self._new(NP * q, MASK_FILTERS)
                def lazy_xin(xin=xin, yp=yp, bsrc=bsrc):
                    X.call("myolo_bn_apply_act", X.ptr(yp), X.ptr(self.bnbuf[bsrc][2]), X.ptr(self.bnbuf[bsrc][3]), X.ptr(xin), NP * q, MASK_FILTERS,
                           ACT_RELU, X.stream())
                pre_rows[id(ypre)] = yp
            elif isinstance(src, tuple) and not compact and src[2] == "myolo_mask_bn1" and "conv2_in_rows" in self.tape:
                xin = self.tape["conv2_in_rows"][:NP * q]          # bn1's activation of the positives, written compact by conv2's input transform
            elif isinstance(src, tuple):      # ("lazy_bn", pre-BN tensor, bn layer): the forward normalised on load
                _, ypre, bsrc = src
                xin = self._new(NP * q, MASK_FILTERS)
                if compact or not self.fuse_compact_gather:
                    yp = gather(ypre, q)
                    X.call("myolo_bn_apply_act", X.ptr(yp), X.ptr(self.bnbuf[bsrc][2]), X.ptr(self.bnbuf[bsrc][3]), X.ptr(xin), NP * q,
                           MASK_FILTERS, ACT_RELU, X.stream())
                else:                         # one kernel: the gathered pre-BN rows (kept for bn_{i-1}'s backward below) and their normalised form
                    yp = self._new(NP * q, MASK_FILTERS)
                    X.call("myolo_gather_groups_affine_act", X.ptr(ypre), X.ptr(idx_d), X.ptr(self.bnbuf[bsrc][2]), X.ptr(self.bnbuf[bsrc][3]), ACT_RELU,
                           X.ptr(yp), X.ptr(xin), NP, q, MASK_FILTERS, X.stream())
                pre_rows[id(ypre)] = yp
            else:
                xin = gather(src, q)
            if self.tape[bn][0] is None:
                # the fused forward never wrote the pre-BN tensor.  bn2-4 are frozen affine maps followed by a ReLU, so their
                # backward can be read off the POST-activation tensor, which the forward did keep for the positive ROIs (it is the
                # next layer's input): mask = a > 0, xhat = (a - beta) / gamma there -- no convolution is re-run
                a_post = a4_p if i == 4 else a_next
                buf = self.bnbuf[bn]
                dy = self._new(NP * q, MASK_FILTERS)
                X.call("myolo_bn_act_bwd_frozen_post", X.ptr(da), X.ptr(a_post), X.ptr(self.p[bn + "/gamma"]), X.ptr(self.p[bn + "/beta"]),
                       X.ptr(buf[2]), X.ptr(dy), X.ptr(self.g[bn + "/gamma"]), X.ptr(self.g[bn + "/beta"]), NP * q, MASK_FILTERS,
                       self.tape[bn][1], *self._wsargs(), X.stream())
            else:
                c_p = pre_rows.get(id(self.tape[bn][0])) if self.fuse_compact_gather else None
                if c_p is None:
                    c_p = gather(self.tape[bn][0], q)          # (a view when the forward kept these rows compact)
                dy = self.bn_act_bwd(bn, da, y_override=c_p)
            a_next = xin                  # conv_i's input = post-activation of layer i-1
            def conv_wgrad(xin=xin, dy=dy, cn=cn, lazy_xin=lazy_xin):
                if lazy_xin is not None:
                    lazy_xin()
                self.conv3x3_bwd_weight(xin, None, dy, cn, NP, ps, ps, MASK_FILTERS, MASK_FILTERS)
                self.colsum(dy, self.g[cn + "/bias"])
            off_chain(conv_wgrad, (xin, dy) + ((src[1],) if lazy_xin is not None else ()))
            da = self._new(NP * q, MASK_FILTERS)
            self.conv3x3_bwd_data(dy, cn, da, NP, ps, ps, MASK_FILTERS, MASK_FILTERS)
        # bucket 4 (conv2-4, bn2-4, deconv, myolo_mask) is complete: its data-path gradients on this stream, the weight gradients hanging off the
        # compact chain on the side stream -- in FRONT of conv1's dense weight gradient, i.e. several milliseconds before the step ends
        self._release_on_wgrad_stream(BUCKET_MASK_REST, self._wgrad_stream, side_wg and self._wgrad_pending)
        # bn1: batch statistics -> dense dx from the row-sparse upstream gradient
        released = [False]                      # conv1's bucket handed to the all-reduce hook on the side stream (fork_wgrad)
        c1, act, _ = self.tape["myolo_mask_bn1"]
        buf = self.bnbuf["myolo_mask_bn1"]
        M1 = NR * q
        x0 = convs[0]                           # None when ROIAlign was fused into conv1's input transform (V is kept instead)
        cin = cf
        v1 = self.tape.pop("conv1_V", None)
        dp0 = self._new(M1, cin)
        if self.lazy_bn1_bwd and v1 is not None and self._wino_ok(NR, ps, ps, MASK_FILTERS, cin):
            # dc1 = d loss / d conv1-output is dense (batch statistics spread the gradient over every ROI) but it is an affine
            # function of conv1's output outside the positive ROIs: conv1's two gradients form it while loading that output
            # (Winograd transforms with a lazy operand) and it is never written.  Its column sums -- conv1's bias gradient --
            # cancel exactly through the batch statistics (sum dz - M*(sum dz)/M + 0), so that gradient is set to 0.
            kab = self._new(2, MASK_FILTERS)
            X.call("myolo_bn_bwd_rowsparse_coeffs", X.ptr(da), X.ptr(c1), X.ptr(idx_d), X.ptr(buf[0]), X.ptr(buf[1]), X.ptr(buf[2]),
                   X.ptr(buf[3]), X.ptr(self.g["myolo_mask_bn1/gamma"]), X.ptr(self.g["myolo_mask_bn1/beta"]), X.ptr(kab[0]),
                   X.ptr(kab[1]), M1, MASK_FILTERS, NP, q, act, *self._wsargs(), X.stream())
            lazy = (X.ptr(c1), X.ptr(da), X.ptr(inv_d), X.ptr(buf[2]), X.ptr(buf[3]), X.ptr(kab[0]), X.ptr(kab[1]), act)
            self.g["myolo_mask_conv1/bias"].zero_()
            v63 = self.tape.pop("conv1_V_fmt", "f43") == "f63"      # the layout the forward left conv1's V planes in
            d63 = self.wino_tiles == "f63" and X.wino63_ok(ps, ps, MASK_FILTERS, cin)
            merged = v63 and d63 and self.merge_conv1_bwd_transforms
            if merged:
                # ONE pass over conv1's output forms the lazily built gradient tile and sends it out both ways: V (data gradient) and Q
                # (weight gradient); the two gradients then finish on their own streams
                pe = X.wino63_plane_elems(NR, MASK_FILTERS)
                Vd, Qd = self._new(pe), self._new(pe)
                X.call("myolo_wino63_lazybn_transforms", *lazy, X.ptr(Vd), X.ptr(Qd), NR, MASK_FILTERS, X.stream())

            def conv1_wgrad(wsp, wsz):
                if merged:
                    X.call("myolo_wino63_bwd_weight_from_q", X.ptr(v1), X.ptr(Qd), X.ptr(self.g["myolo_mask_conv1/kernel"]), NR, cin, MASK_FILTERS,
                           wsp, wsz, X.stream())
                elif v63:
                    X.call("myolo_wino63_bwd_weight_lazybn", X.ptr(v1), *lazy, X.ptr(self.g["myolo_mask_conv1/kernel"]), NR, cin, MASK_FILTERS,
                           wsp, wsz, X.stream())
                else:
                    X.call("myolo_conv3x3_wino_bwd_weight_lazybn", X.ptr(v1), *lazy, X.ptr(self.g["myolo_mask_conv1/kernel"]), NR, ps, ps,
                           cin, MASK_FILTERS, wsp, wsz, X.stream())
            wg_bytes = (X.wino63_bwd_weight_from_q_ws_bytes(NR, cin, MASK_FILTERS) if merged else
                        X.wino63_bwd_weight_ws_bytes(NR, cin, MASK_FILTERS) if v63 else X.wino_ws_bytes(NR, ps, ps, cin, MASK_FILTERS, 2))
            def fork_wgrad():
                released[0] = bool(self.on_bucket_ready)
                self._ws_wgrad.ensure(wg_bytes)
                cur = torch.cuda.current_stream()
                self._wgrad_stream.wait_stream(cur)
                with torch.cuda.stream(self._wgrad_stream):
                    conv1_wgrad(self._ws_wgrad.ptr, self._ws_wgrad.size)
                    if self.on_bucket_ready:          # bn1's gradients (this bucket's other members) were complete at the fork
                        self.on_bucket_ready(BUCKET_MASK_CONV1)
                for t in (v1, c1, da, inv_d, kab) + ((Qd,) if merged else ()):
                    t.record_stream(self._wgrad_stream)
                self._wgrad_pending = True
            late = bool(self.conv1_wgrad_after_dgrad) and merged
            if self.overlap_conv1_wgrad:
                if not late:
                    fork_wgrad()
                self.ws.ensure(X.wino_ws_bytes(NR, ps, ps, cin, MASK_FILTERS, 1))
            else:
                self.ws.ensure(max(X.wino_ws_bytes(NR, ps, ps, cin, MASK_FILTERS, 1), wg_bytes))
                conv1_wgrad(*self._wsargs())
            if merged:
                self.ws.ensure(X.wino63_bwd_data_from_v_ws_bytes(NR, cin, MASK_FILTERS))
                X.call("myolo_wino63_bwd_data_from_v", X.ptr(Vd), X.ptr(self.p["myolo_mask_conv1/kernel"]), X.ptr(dp0), NR, cin, MASK_FILTERS,
                       *self._wsargs(), X.stream())
                if self.overlap_conv1_wgrad and late:
                    fork_wgrad()          # behind the data gradient: runs beside ROIAlign's backward and the trunk's backward
            elif d63:
                self.ws.ensure(X.wino63_bwd_data_ws_bytes(NR, cin, MASK_FILTERS))
                X.call("myolo_wino63_bwd_data_lazybn", *lazy, X.ptr(self.p["myolo_mask_conv1/kernel"]), X.ptr(dp0), NR, cin, MASK_FILTERS,
                       *self._wsargs(), X.stream())
            else:
                X.call("myolo_conv3x3_wino_bwd_data_lazybn", *lazy, X.ptr(self.p["myolo_mask_conv1/kernel"]), X.ptr(dp0), NR, ps, ps, cin,
                       MASK_FILTERS, *self._wsargs(), X.stream())
        else:
            dc1 = self._new(M1, MASK_FILTERS)
            X.call("myolo_bn_act_bwd_rowsparse", X.ptr(da), X.ptr(c1), X.ptr(idx_d), X.ptr(inv_d), X.ptr(buf[0]), X.ptr(buf[1]),
                   X.ptr(buf[2]), X.ptr(buf[3]), X.ptr(dc1), X.ptr(self.g["myolo_mask_bn1/gamma"]), X.ptr(self.g["myolo_mask_bn1/beta"]),
                   M1, MASK_FILTERS, NP, q, act, *self._wsargs(), X.stream())
            self.conv3x3_bwd_weight(x0, v1, dc1, "myolo_mask_conv1", NR, ps, ps, cin, MASK_FILTERS)
            self.colsum(dc1, self.g["myolo_mask_conv1/bias"])
            self.conv3x3_bwd_data(dc1, "myolo_mask_conv1", dp0, NR, ps, ps, cin, MASK_FILTERS)
        dF = self._new(n * h * w, cf)
        X.call("myolo_roialign_bwd_grouped", X.ptr(dp0), X.ptr(boxes), X.ptr(dF), n, h, w, cf, NR // n, ps, ps, X.stream())
        if self.on_bucket_ready and not released[0]:
            self.join_conv1_wgrad()       # the compacted weight gradients on the side stream, if any
            self.on_bucket_ready(BUCKET_MASK_CONV1)       # (otherwise the bucket was released on the weight gradient's stream, behind that kernel)
        return dF

    def join_conv1_wgrad(self):
        """make the current stream wait for conv1's weight gradient (side stream)."""
        if self._wgrad_pending:
            torch.cuda.current_stream().wait_stream(self._wgrad_stream)
            self._wgrad_pending = False

    # ------------------------------------------------------------------ steps
    def to_device_batch(self, batch):
        """host batch (the six arrays of model.py:896-897, or the three of 'yolo' mode, myolo_utils.py:849-851) -> device
        tensors in C-ABI dtypes.  Upload path: each array is converted into a PINNED staging buffer (a ring of
        `_STAGE_SETS` sets, so a prefetching caller can stage batch i+1 while batch i's copies are in flight) and copied
        with an asynchronous H2D on the engine's upload stream into FRESH device tensors; the dict carries the event
        (`_ready`) that the consuming step makes its stream wait on (`_await_batch`).  Nothing here blocks the host except
        re-use of a staging set whose previous copy is still running."""
        dev = self.dev
        T = self.cfg.TRUE_BOX_BUFFER
        if not self.pinned_upload:
            keys = ("images", "true_boxes", "y_true", "gt_ids", "gt_boxes", "gt_masks")[:len(batch)]
            dts = dict(images=np.float32, true_boxes=np.float32, y_true=np.float32, gt_ids=np.int32, gt_boxes=np.int32, gt_masks=np.uint8)
            out = {k: torch.as_tensor(np.ascontiguousarray(np.asarray(a).reshape(-1, T, 4) if k == "true_boxes" else a, dts[k]), device=dev)
                   for k, a in zip(keys, batch)}
            return out
        n = int(np.asarray(batch[0]).shape[0])

        def fill(arrays):
            for dst, src in zip(arrays, batch):
                np.copyto(dst, np.asarray(src).reshape(dst.shape), casting="unsafe")     # dtype conversion + the one host copy, into pinned memory
        return self.stage_batch(fill, n, yolo=len(batch) == 3, u8_images=False)    # (the arrays are used as they are: no normalisation here)

    def stage_batch(self, fill, n, yolo=False, u8_images=True):
        """fill(arrays) writes a host batch of n images into the pinned staging arrays -- [images, true_boxes [n,T,4] f32, y_true f32
        (, gt_ids i32, gt_boxes i32, gt_masks u8)]; images uint8 [n,H,W,3] (the raw bytes: `/ 255.` of myolo_utils.py:824 then happens on
        the device, myolo_u8_to_unit_f32, and a quarter of the bytes cross PCIe) or float32 -- then the asynchronous upload is queued.
        MaskYOLO.train() lets BatchGenerator.fill encode straight into these arrays on its prefetch thread.  -> device batch dict."""
        cfg, dev = self.cfg, self.dev
        H, W = cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1]
        G, A, C, T = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER
        spec = [("images", (n, H, W, 3), torch.uint8 if u8_images else torch.float32), ("true_boxes", (n, T, 4), torch.float32),
                ("y_true", (n, cfg.GRID_H, G, A, 5 + C), torch.float32)]
        if not yolo:
            spec += [("gt_ids", (n, T), torch.int32), ("gt_boxes", (n, T, 4), torch.int32), ("gt_masks", (n, H, W, cfg.MAX_GT_INSTANCES), torch.uint8)]
        with self._stage_lock:
            if self._stage is None:
                self._stage = [dict(bufs={}, ev=None, busy=threading.Lock()) for _ in range(self._STAGE_SETS)]
                self._stage_i = 0
                self._upload_stream = _shared_stream(dev, "batch_upload") if self.upload_own_stream else self._copy_stream
            # a staging set belongs to ONE caller from here until its upload has been queued and its event recorded (`busy`): MaskYOLO.train()'s
            # prefetch thread and the main thread's validation batches go through this ring at the same time.  First set in ring order
            # that nobody is filling; if every set is taken, queue for the next one in order.
            st = None
            for k in range(self._STAGE_SETS):
                cand = self._stage[(self._stage_i + k) % self._STAGE_SETS]
                if cand["busy"].acquire(False):
                    st = cand
                    self._stage_i = (self._stage_i + k + 1) % self._STAGE_SETS
                    break
            if st is None:
                wait_for = self._stage[self._stage_i]
                self._stage_i = (self._stage_i + 1) % self._STAGE_SETS
        if st is None:
            wait_for["busy"].acquire()
            st = wait_for
        try:
            if st["ev"] is not None:
                st["ev"].synchronize()                    # the staging set's previous H2D (a few batches ago): long finished
            arrays = []
            for key, shape, dt in spec:
                # keyed by (name, dtype, shape): train() stages byte images, validation / to_device_batch float ones through the same ring --
                # each keeps its own pinned buffer instead of re-pinning on every switch
                pin = st["bufs"].get((key, shape, dt))
                if pin is None:
                    pin = st["bufs"][(key, shape, dt)] = torch.empty(shape, dtype=dt, pin_memory=True)
                arrays.append(pin.numpy())
            fill(arrays)
            out = {}
            with torch.cuda.stream(self._upload_stream):
                for key, shape, dt in spec:
                    t = torch.empty(shape, dtype=dt, device=dev)
                    t.copy_(st["bufs"][(key, shape, dt)], non_blocking=True)
                    out[key] = t
                if u8_images:
                    raw = out["images"]
                    out["images"] = torch.empty((n, H, W, 3), dtype=torch.float32, device=dev)
                    X.call("myolo_u8_to_unit_f32", X.ptr(raw), X.ptr(out["images"]), raw.numel(), X.stream())
                    out["_raw_images"] = raw              # (kept until the conversion has run)
                ev = torch.cuda.Event()
                ev.record(self._upload_stream)
                st["ev"] = ev
            out["_ready"] = ev
        finally:
            st["busy"].release()
        return out

    _STAGE_SETS = 3

    def _await_batch(self, db):
        """make the current stream wait for a staged batch's H2D copies (and tell the allocator the tensors are used here)."""
        ev = db.get("_ready") if isinstance(db, dict) else None
        if ev is not None and not db.get("_awaited"):
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for v in db.values():
                if isinstance(v, torch.Tensor) and v.is_cuda:
                    v.record_stream(cur)
            db["_awaited"] = True

    def _yolo_warm(self):
        """yolo_custom_loss's `seen` counter (model.py:113,194: a graph variable incremented by EVERY evaluation of the loss -- training and
        validation batches alike -- and not part of the saved weights): 1 while the warm-up branch (model.py:196-207) is taken."""
        self.seen += 1
        return 1 if self.seen < int(self.cfg.WARM_UP_BATCHES) else 0

    # ---- prepared weights (csrc/myolo_common.h): valid inside ONE forward_backward call, for the weights as they are at its start
    def _wprep_begin(self):
        if not self.weight_prep:
            return
        if self._wprep is None:
            self._wprep_cap = self._wprep_arena_bytes()
            self._wprep = X.WeightPrep(self.dev, arena_bytes=self._wprep_cap)
        wp = self._wprep
        # an arena that is too small makes its sites re-prepare in place every step, silently.  The registry counts the resolve() calls
        # whose new site found no room (a site that merely appears late -- a shape first seen at step 5 -- misses once and is recorded: no warning);
        # if that count still grows between the fourth and the fifth step, say so once.
        self._wprep_calls += 1
        if self._wprep_calls in (4, 5):
            st = wp.stats()
            if self._wprep_calls == 5 and self._wprep_misses is not None and st["overflows"] > self._wprep_misses:
                import warnings
                warnings.warn("myolo: the prepared-weights arena (%d MiB, %d MiB used, %d entries) is too small: %d preparations per step are re-made in place "
                              "(correct, slower).  Net._wprep_arena_bytes() sizes it." % (self._wprep_cap >> 20, st["bytes_used"] >> 20, st["entries"],
                                                                                       st["overflows"] - self._wprep_misses))
            self._wprep_misses = st["overflows"]
        self._wprep_ev = None
        n = wp.count()
        if n:
            # every preparation recorded so far, in the order the step first asked for them, on the weight-gradient stream (idle here: the optimizer
            # joined it), under the first layers of the forward.  Two marks: the trunk's entries, everything.
            cur = torch.cuda.current_stream()
            side = self._wgrad_stream
            side.wait_stream(cur)                 # behind the optimizer's update of the weights
            nt = min(self._wprep_ntrunk if self._wprep_ntrunk is not None else n, n)
            ev1 = torch.cuda.Event()
            with torch.cuda.stream(side):
                wp.refresh(0, nt)
                ev1.record(side)
            self._wprep_ev = [ev1, None]
            self._wprep_rest = (nt, n)            # launched by _wprep_phase2, once the big early layers of the trunk are through
        wp.activate(True)

    def _wprep_arena_bytes(self):
        """capacity of a prepared-weights arena for this net (was a fixed 768 MiB): every weight matrix may be kept once as a transpose
        (4 bytes per value), once as a bf16x6 split (6) and every 3x3 kernel twice (forward, rotated for the data gradient) as 64 transformed planes of
        six bytes (64 / 9 * 6 per value), + 256-byte alignment per entry.  ~360 MiB at the alpha-1 Shapes net; an entry beyond the capacity is made in
        place by its site, and _wprep_begin warns once when that keeps happening."""
        n = 0
        for k, v in self.p.items():
            if v.dim() == 4 and v.shape[0] == 3 and v.shape[2] > 3:            # 3x3 convs with many input channels: Winograd filter planes,
                n += int(v.numel()) * (2 * (64 * 6 // 9 + 1) + 10)               # once for the forward and once rotated for the data gradient (round 6: the
                                                                                 # one-set budget overflowed at config 2 -- 207 of 211 MiB, two sites re-made in place)
            elif v.dim() >= 2:
                n += int(v.numel()) * 10
            n += 512
        return int(min(max(n, 64 << 20), 768 << 20))

    def _wprep_phase2(self):
        """the preparations the mask head and the backward will ask for (Winograd filter transforms, transposes, splits: ~0.25 ms of small kernels):
        started behind the trunk's 112 / 56 / 28-pixel layers -- beside those they cost the bandwidth-bound first layers 20-30 us each -- and run
        under its 14 x 14 / 7 x 7 layers, which leave the chip mostly idle."""
        rest = getattr(self, "_wprep_rest", None)
        if rest is None or self._wprep_ev is None:
            return
        self._wprep_rest = None
        cur = torch.cuda.current_stream()
        side = self._wgrad_stream
        evm = torch.cuda.Event()
        evm.record(cur)
        side.wait_event(evm)
        ev2 = torch.cuda.Event()
        with torch.cuda.stream(side):
            self._wprep.refresh(rest[0], rest[1])
            ev2.record(side)
        self._wprep_ev[1] = ev2

    def _wprep_wait(self, k):
        """order the current stream behind mark k (0: the trunk's preparations, 1: all) of this step's refresh; once each."""
        if self._wprep_ev is not None and self._wprep_ev[k] is not None:
            for j in range(k + 1):
                if self._wprep_ev[j] is not None:
                    torch.cuda.current_stream().wait_event(self._wprep_ev[j])
                    self._wprep_ev[j] = None

    def _wprep_mark_trunk(self):
        if self._wprep is not None and self._wprep_ntrunk is None and self.weight_prep:
            self._wprep_ntrunk = self._wprep.count()

    def _wprep_end(self):
        if self._wprep is not None:
            self._wprep.activate(False)
            self._wprep.invalidate()              # the optimizer (or anyone) may change the weights next
            self._wprep_ev = None
            self._wprep_rest = None

    def forward_backward(self, db):
        """One training forward + backward on a device batch.  Gradients land in self.flat_g."""
        self._wprep_begin()
        try:
            return self._forward_backward(db)
        finally:
            self._wprep_end()

    def forward_backward_yolo(self, db):
        self._wprep_begin()
        try:
            return self._forward_backward_yolo(db)
        finally:
            self._wprep_end()

    def _forward_backward(self, db):
        self._activate()
        self.mark_weights_changed()           # (the BatchNorm moving statistics move in every training forward)
        self._bn_sums.clear()                 # (partials a previous, interrupted backward may have left)
        self._await_batch(db)
        cfg = self.cfg
        self.tape = {}
        images = db["images"]
        B = images.shape[0]
        G, A, C, T = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER
        R = G * G * A
        mh, mw = cfg.MASK_SHAPE
        Fm, fshape, yo = self.trunk_fwd(images, True)
        proposals = self._new(B, R, 4)
        X.call("myolo_yolo_decode", X.ptr(yo), X.ptr(self.anchors), X.ptr(proposals), B, G, A, C, X.stream())
        if self.proposals_hook:
            self.proposals_hook(proposals, db)
        rois = self._new(B, R, 4)
        tcls = self._new(B, R, dtype=torch.int32)
        tmask = self._new(B, R, mh, mw)
        npos = self._new(B, dtype=torch.int32)
        H, W = cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1]
        X.call("myolo_mask_targets", X.ptr(proposals), X.ptr(db["gt_ids"]), X.ptr(db["gt_boxes"]), X.ptr(db["gt_masks"]),
               X.ptr(rois), X.ptr(tcls), X.ptr(tmask), X.ptr(npos), B, R, T, H, W, mh, mw, X.stream())
        if self.sparse_mask_bwd:
            self._start_npos_copy(npos)
        w1 = float(cfg.LOSS_WEIGHTS.get("yolo_sum_loss", 1.))
        w2 = float(cfg.LOSS_WEIGHTS.get("myolo_mask_loss", 1.))
        yterms = self._new(8)
        dyolo = self._new(yo.shape[0], yo.shape[1])
        warm = self._yolo_warm()
        def yolo_loss():
            X.call("myolo_yolo_loss_warmup", X.ptr(db["y_true"]), X.ptr(yo), X.ptr(db["true_boxes"]), X.ptr(self.anchors),
                   X.ptr(self.class_weights), float(cfg.OBJECT_SCALE), float(cfg.NO_OBJECT_SCALE), float(cfg.COORD_SCALE),
                   float(cfg.CLASS_SCALE), w1, warm, X.ptr(yterms), X.ptr(dyolo), B, G, A, C, T, *self._wsargs(), X.stream())
        if self.overlap_yolo_bwd:
            # the loss kernel (one workgroup row per image, ~0.17 ms) only feeds the YOLO head's backward, which runs on the side stream:
            # launch it there too (side scratch buffer), under the mask head's forward, instead of in front of it
            self._yolo_stream.wait_stream(torch.cuda.current_stream())
            self._ws_active = self._ws_side
            try:
                with torch.cuda.stream(self._yolo_stream):
                    yolo_loss()
            finally:
                self._ws_active = self._ws_main
        else:
            yolo_loss()
        # (a tape_hook rewrites saved tensors between forward and backward -- tests force the oracle's activations there: the YOLO head's backward must
        # not have read them before the hook runs, so it then starts at the old place)
        want_early = self.exchange_active if self.yolo_bwd_early < 0 else bool(self.yolo_bwd_early)
        early = bool(self.overlap_yolo_bwd and want_early and not self.tape_hook)
        if early:
            self.start_yolo_head_bwd(dyolo)
        if self.sparse_mask_fwd:
            if not self.sparse_mask_bwd:
                raise RuntimeError("TRAIN_MASK_HEAD_ROIS='positives' needs the sparse backward (sparse_mask_bwd=True)")
            pred, tmask_l, tcls_l = self.mask_head_fwd_positives(Fm, fshape, rois, tmask, tcls)
        else:
            # the ROIs whose activations the sparse backward will gather: the first n_pos rows of each image (exactly the
            # index set of _positive_index, whatever their class id)
            flags = keep = None
            if self.sparse_mask_bwd:
                # flags / compact slots / slot -> ROI of the positives, built on the device from the counts (no host round trip: the forward
                # below already uses them; the host only learns the TOTAL, to size the compacted backward's launches)
                flags = self._new(B * R, dtype=torch.int32)
                idx_d = self._new(B * R, dtype=torch.int32)
                inv_d = self._new(B * R, dtype=torch.int32)
                X.call("myolo_positive_index", X.ptr(npos), B, R, X.ptr(flags), X.ptr(idx_d), X.ptr(inv_d), None, X.stream())
                self.tape["pos_index"] = (idx_d, inv_d)
                if self.keep_deconv_rows:
                    # capacity of the kept-rows buffer (803 KB per ROI): keep_deconv_rows per image at most, and no more than the high-water mark of
                    # what the steps whose counts the host has read really needed -- twice the positives + 64, rounded up to a power of two, never
                    # shrinking: the size changes a handful of times in a run, so the allocator does not see a new request every step (a fresh
                    # hipMalloc inside a step is a ~35 ms stall).  ADVICE r4: the full cap, 1.2 GB at B = 32, was allocated every step whatever the
                    # positives.  Beyond the capacity the backward re-runs the deconv for the positives: same results.
                    cap = min(B * R, int(self.keep_deconv_rows * B))
                    if self._np_seen is not None:
                        want = 64
                        while want < 2 * self._np_seen + 64:
                            want *= 2
                        self._keep_cap_hw = max(self._keep_cap_hw, want)
                        cap = min(cap, self._keep_cap_hw)
                    keep = (inv_d, max(1, cap))
            pred = self.mask_head_fwd(Fm, fshape, rois, True, pos_flags=flags, keep=keep)
            tmask_l, tcls_l = tmask, tcls
        if pred is None:                  # positives-only forward without a positive ROI (model.py:750-752)
            mterms = torch.zeros(2, dtype=torch.float32, device=self.dev)
            dz = None
        else:
            mterms = self._new(2)
            dz = self._new(pred.shape[0], pred.shape[1])
            X.call("myolo_mask_bce", X.ptr(tmask_l), X.ptr(tcls_l), X.ptr(pred), w2, X.ptr(mterms), X.ptr(dz), tcls_l.numel(), mh, mw, C,
                   *self._wsargs(), X.stream())
        if self.tape_hook:
            self.tape_hook(self)
        if self.overlap_yolo_bwd and not early:
            # under the compacted part of the mask head's backward (small launches on the positive ROIs), not under the big
            # forward GEMMs: two streams of small kernels fill the chip together, and the dense kernels keep it to themselves
            self.start_yolo_head_bwd(dyolo)
        dF = self.mask_head_bwd_sparse(dz, B, R) if self.sparse_mask_bwd else self.mask_head_bwd(dz)
        self.trunk_bwd(dF, dyolo)
        self.join_conv1_wgrad()
        if self.sparse_mask_fwd:          # the positives' masks only, in positive order (see mask_head_fwd_positives)
            mm = None if pred is None else pred.view(-1, mh, mw, C)
        else:
            mm = pred.view(B, R, mh, mw, C)
        return dict(yolo_output=yo.view(B, G, G, A, 5 + C), yolo_proposals=proposals, output_rois=rois,
                    myolo_mask=mm, target_class_ids=tcls, target_mask=tmask, n_pos=npos,
                    yolo_terms=yterms, mask_terms=mterms, feature_map=Fm.view(*fshape), loss_weights=(w1, w2))

    def forward_loss(self, db):
        """Validation forward (Keras evaluate_generator as called by fit_generator, model.py:1053-1054): the TRAINING graph
        -- decode, mask targets, ROIAlign, mask head, both losses (model.py:872-904) -- in Keras' test phase, i.e. every
        BatchNormalization on its moving statistics, no gradient, no state change.  Returns the loss terms as device
        tensors (yolo_terms[8], mask_terms[2]); 'yolo' mode batches (three arrays) give the YOLO loss only."""
        self._activate()
        self._await_batch(db)
        cfg = self.cfg
        self.tape = {}
        images = db["images"]
        B = images.shape[0]
        G, A, C, T = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER
        R = G * G * A
        mh, mw = cfg.MASK_SHAPE
        Fm, fshape, yo = self.trunk_fwd(images, False)
        w1 = float(cfg.LOSS_WEIGHTS.get("yolo_sum_loss", 1.))
        w2 = float(cfg.LOSS_WEIGHTS.get("myolo_mask_loss", 1.))
        yterms = self._new(8)
        dyolo = self._new(yo.shape[0], yo.shape[1])          # the loss kernel always forms the gradient; unused here
        warm = self._yolo_warm()
        X.call("myolo_yolo_loss_warmup", X.ptr(db["y_true"]), X.ptr(yo), X.ptr(db["true_boxes"]), X.ptr(self.anchors),
               X.ptr(self.class_weights), float(cfg.OBJECT_SCALE), float(cfg.NO_OBJECT_SCALE), float(cfg.COORD_SCALE),
               float(cfg.CLASS_SCALE), w1, warm, X.ptr(yterms), X.ptr(dyolo), B, G, A, C, T, *self._wsargs(), X.stream())
        if "gt_masks" not in db:
            self.tape = {}
            return dict(yolo_terms=yterms, mask_terms=torch.zeros(2, dtype=torch.float32, device=self.dev), loss_weights=(w1, 0.0))
        proposals = self._new(B, R, 4)
        X.call("myolo_yolo_decode", X.ptr(yo), X.ptr(self.anchors), X.ptr(proposals), B, G, A, C, X.stream())
        rois = self._new(B, R, 4)
        tcls = self._new(B, R, dtype=torch.int32)
        tmask = self._new(B, R, mh, mw)
        npos = self._new(B, dtype=torch.int32)
        H, W = cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1]
        X.call("myolo_mask_targets", X.ptr(proposals), X.ptr(db["gt_ids"]), X.ptr(db["gt_boxes"]), X.ptr(db["gt_masks"]),
               X.ptr(rois), X.ptr(tcls), X.ptr(tmask), X.ptr(npos), B, R, T, H, W, mh, mw, X.stream())
        pred = self.mask_head_fwd(Fm, fshape, rois, False)
        mterms = self._new(2)
        dz = self._new(pred.shape[0], pred.shape[1])
        X.call("myolo_mask_bce", X.ptr(tmask), X.ptr(tcls), X.ptr(pred), w2, X.ptr(mterms), X.ptr(dz), tcls.numel(), mh, mw, C,
               *self._wsargs(), X.stream())
        self.tape = {}
        return dict(yolo_terms=yterms, mask_terms=mterms, loss_weights=(w1, w2), yolo_output=yo.view(B, G, G, A, 5 + C),
                    output_rois=rois, target_class_ids=tcls, myolo_mask=pred.view(B, R, mh, mw, C), n_pos=npos)

    def _forward_backward_yolo(self, db):
        """'yolo' mode training step (model.py:906-920: outputs [yolo_output, yolo_sum_loss]): backbone + YOLO head
        + yolo_custom_loss, no feature_map / ROIAlign / mask head.  db needs images, true_boxes, y_true."""
        self._activate()
        self.mark_weights_changed()
        self._bn_sums.clear()
        self._await_batch(db)
        cfg = self.cfg
        self.tape = {}
        images = db["images"]
        B = images.shape[0]
        G, A, C, T = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES, cfg.TRUE_BOX_BUFFER
        Fm, fshape, yo = self.trunk_fwd(images, True)
        w1 = float(cfg.LOSS_WEIGHTS.get("yolo_sum_loss", 1.))
        yterms = self._new(8)
        dyolo = self._new(yo.shape[0], yo.shape[1])
        warm = self._yolo_warm()
        X.call("myolo_yolo_loss_warmup", X.ptr(db["y_true"]), X.ptr(yo), X.ptr(db["true_boxes"]), X.ptr(self.anchors),
               X.ptr(self.class_weights), float(cfg.OBJECT_SCALE), float(cfg.NO_OBJECT_SCALE), float(cfg.COORD_SCALE),
               float(cfg.CLASS_SCALE), w1, warm, X.ptr(yterms), X.ptr(dyolo), B, G, A, C, T, *self._wsargs(), X.stream())
        self.flat_g[self.bucket_ranges[BUCKET_MASK_CONV1][0]:self.bucket_ranges[BUCKET_MASK_REST][1]].zero_()
        if self.on_bucket_ready:
            self.on_bucket_ready(BUCKET_MASK_REST)
            self.on_bucket_ready(BUCKET_MASK_CONV1)
        self.trunk_bwd(None, dyolo)
        return dict(yolo_output=yo.view(B, G, G, A, 5 + C), yolo_terms=yterms, loss_weights=(w1, 0.0))

    def adam_step(self, lr, b1=0.9, b2=0.999, eps=1e-8):
        """Keras Adam (model.py:1071-1075) over the whole flat buffer."""
        self.join_conv1_wgrad()
        self.join_trunk_wgrad()
        if self.before_optimizer:
            self.before_optimizer()
        self.adam_t += 1
        t = self.adam_t
        lr_t = float(lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t))
        X.call("myolo_adam_step", X.ptr(self.flat_p), X.ptr(self.flat_g), X.ptr(self.flat_m), X.ptr(self.flat_v),
               self.nparam, lr_t, b1, b2, eps, float(self.grad_scale), X.stream())
        self.mark_weights_changed()

    def train_step(self, db, lr):
        out = self.forward_backward(db)
        self.adam_step(lr)
        return out

    def _pack_bf16(self, key, bufs=None):
        """(re)make one bf16 operand of mask_head_fwd_bf16 -- conv i = kernel with the frozen BatchNorm folded + bias; 'deconv' = the transposed-conv
        kernel in its GEMM layout -- in its persistent cache buffers, or in `bufs`."""
        wt, bfold = bufs if bufs is not None else self._bf16_packs[key]
        if key == "deconv":
            X.call("myolo_pack_weights_bf16", X.ptr(self.p["myolo_mask_deconv/kernel"]), MASK_FILTERS, 4 * MASK_FILTERS, 1, None,
                   None, None, None, None, X.ptr(wt), None, X.stream())
            return
        cn, bn = "myolo_mask_conv%d" % key, "myolo_mask_bn%d" % key
        X.call("myolo_pack_weights_bf16", X.ptr(self.p[cn + "/kernel"]), wt.shape[1], MASK_FILTERS, 0, X.ptr(self.p[cn + "/bias"]),
               X.ptr(self.p[bn + "/gamma"]), X.ptr(self.p[bn + "/beta"]), X.ptr(self.s[bn + "/moving_mean"]),
               X.ptr(self.s[bn + "/moving_variance"]), X.ptr(wt), X.ptr(bfold), X.stream())

    def _bf16_operand(self, key, cin):
        """-> (packed bf16 weights, folded bias or None) for conv `key` (1..4) / 'deconv': from the cache (made once per weight version), or packed
        here into fresh tensors when the cache is off."""
        bf = torch.bfloat16
        shape = (4 * MASK_FILTERS, MASK_FILTERS) if key == "deconv" else (MASK_FILTERS, 9 * cin)
        if not self.infer_weight_cache:
            bufs = (self._new(*shape, dtype=bf), None if key == "deconv" else self._new(MASK_FILTERS))
            self._pack_bf16(key, bufs)
            return bufs
        ent = self._bf16_packs.get(key)
        if ent is None or tuple(ent[0].shape) != shape:
            # first use (never while a graph is being captured: _capture_predict's warm-up forwards come first)
            self._bf16_packs[key] = (self._new(*shape, dtype=bf), None if key == "deconv" else self._new(MASK_FILTERS))
            self._pack_bf16(key)
        return self._bf16_packs[key]

    def _infer_prep_sync(self):
        """Bring the inference path's cached weight preparations up to the current weight version on the current stream (no-op when they are),
        and order the current stream behind the last refresh.  Called by every eager inference forward and before every graph replay; never
        launches anything while a graph is being captured (the capture's warm-up forwards have refreshed everything)."""
        if not self.infer_weight_cache:
            return
        if self._iprep is None:
            self._iprep = X.WeightPrep(self.dev, arena_bytes=min(self._wprep_arena_bytes(), 256 << 20))
        ip = self._iprep
        state = (self._weight_state(), ip.count())
        cur = torch.cuda.current_stream()
        if state != self._iprep_state and not torch.cuda.is_current_stream_capturing():
            if self._iprep_state is None or state[0] != self._iprep_state[0]:
                # the weights changed: forwards of other lanes may still be reading the arena / the packs
                torch.cuda.synchronize(self.dev)
                ip.invalidate()
            ip.refresh(0, 1 << 30, max_idle=0)
            for key in list(self._bf16_packs):
                self._pack_bf16(key)
            self._iprep_state = state
            self._iprep_ev = torch.cuda.Event()
            self._iprep_ev.record(cur)
        elif self._iprep_ev is not None and not torch.cuda.is_current_stream_capturing():
            cur.wait_event(self._iprep_ev)

    class _InferPrep(object):
        """with net._infer_prep(): the inference registry is the library's active one (sites resolve their prepared weights from its arena)."""

        def __init__(self, net):
            self.net = net

        def __enter__(self):
            self.net._infer_prep_sync()
            if self.net._iprep is not None and self.net.infer_weight_cache:
                self.net._iprep.activate(True)

        def __exit__(self, *exc):
            if self.net._iprep is not None:
                self.net._iprep.activate(False)

    def _infer_prep(self):
        return Net._InferPrep(self)

    def predict(self, images):
        """inference graph (model.py:922-936): -> yolo_output, detections [B,R,6], myolo_mask."""
        with self._infer_prep():
            self._activate()
            cfg = self.cfg
            self.tape = {}
            B = images.shape[0]
            G, A, C = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES
            R = G * G * A
            mh, mw = cfg.MASK_SHAPE
            Fm, fshape, yo = self.trunk_fwd(images, False)
            det = self._new(B, R, 6)
            X.call("myolo_yolo_detections", X.ptr(yo), X.ptr(self.anchors), X.ptr(det), B, G, A, C, X.stream())
            rois = det[..., :4].contiguous()
            if cfg.INFERENCE_DTYPE == "bf16":
                pred = self.mask_head_fwd_bf16(Fm, fshape, rois)
            elif cfg.INFERENCE_DTYPE == "fp32":
                pred = self.mask_head_fwd(Fm, fshape, rois, False)
            else:
                raise ValueError("INFERENCE_DTYPE must be 'fp32' or 'bf16' (got %r)" % (cfg.INFERENCE_DTYPE,))
            self.tape = {}
            return yo.view(B, G, G, A, 5 + C), det, pred.view(B, R, mh, mw, C)

    def predict_detections(self, images):
        """first half of the inference graph: -> yolo_output, detections [B,R,6], and the feature map the mask head reads."""
        with self._infer_prep():
            self._activate()
            cfg = self.cfg
            self.tape = {}
            B = images.shape[0]
            G, A, C = cfg.GRID_W, cfg.N_BOX, cfg.NUM_CLASSES
            Fm, fshape, yo = self.trunk_fwd(images, False)
            det = self._new(B, G * G * A, 6)
            X.call("myolo_yolo_detections", X.ptr(yo), X.ptr(self.anchors), X.ptr(det), B, G, A, C, X.stream())
            self.tape = {}
            return yo.view(B, G, G, A, 5 + C), det, (Fm, fshape)

    def predict_masks(self, feature, rois):
        """second half for a chosen subset of boxes: rois [B, n, 4] (the first four detection columns) -> [B, n, mh, mw, C]."""
        with self._infer_prep():
            self._activate()
            cfg = self.cfg
            Fm, fshape = feature
            self.tape = {}
            B, n = rois.shape[:2]
            mh, mw = cfg.MASK_SHAPE
            rois = rois.contiguous()
            pred = self.mask_head_fwd_bf16(Fm, fshape, rois) if cfg.INFERENCE_DTYPE == "bf16" else self.mask_head_fwd(Fm, fshape, rois, False)
            self.tape = {}
            return pred.view(B, n, mh, mw, cfg.NUM_CLASSES)

    def _lane_state(self, lane):
        """Everything a forward WRITES outside the allocator (scratch buffer, frozen-BatchNorm coefficient buffers), once per lane of
        predict_stream, plus the lane's stream: two forwards in flight must not share them.  Lane 0 is the Net's own set."""
        st = self._lanes.get(lane)
        if st is None:
            if self._fz_table is None:
                self._frozen_affine_all()            # builds the (read-only) slot table every lane shares
            st = {"stream": _shared_stream(self.dev, "lane%d" % lane)}
            if lane > 0:
                # a lane's scratch growth invalidates only THAT lane's captured graphs (key[-1] == lane); sized like lane 0's so that it
                # does not regrow during warm-up (every lane's graphs used to be dropped, each recapture a device synchronise)
                st["ws"] = Workspace(self.dev, nbytes=max(self._ws_main.size, 256 << 20), on_realloc=lambda lane=lane: self._drop_graphs(lane))
                st["bnbuf"] = {k: torch.zeros_like(v) for k, v in self.bnbuf.items()}
                st["fz"] = torch.empty_like(self._fz_coeffs)
            self._lanes[lane] = st
        return st

    def _drop_graphs(self, lane):
        for k in [k for k in self._graphs if k[-1] == lane]:
            del self._graphs[k]

    def predict_graphed(self, images, lane=0, solo=True):
        """predict() replayed from a captured hipGraph (one per input shape and lane): the ~150 launches of an inference forward
        cost one graph launch on the host.  Same kernels, same buffers for the weights (updates are seen), static input / output
        buffers: the returned tensors are overwritten by the next call with the same shape on the same lane.  lane > 0: a second
        (third ...) graph with its own scratch and static buffers, so that forwards of different lanes may run concurrently on
        different streams (predict_stream)."""
        fork = bool(self.infer_fork_feature_map and solo)       # solo = no other lane's forward runs beside this one (predict_stream says so)
        key = (tuple(images.shape), self.cfg.INFERENCE_DTYPE, self.conv3x3_algo, self.fp32_matmul, self.wino_tiles, fork, lane)
        ent = self._graphs.get(key)
        if ent is None or ent[0] is None:
            saved = None
            self._fork_now = fork
            if lane > 0:
                st = self._lane_state(lane)
                saved = (self._ws_main, self._ws_active, self.bnbuf, self._fz_coeffs)
                self._ws_main = self._ws_active = st["ws"]
                self.bnbuf, self._fz_coeffs = st["bnbuf"], st["fz"]
            try:
                if ent is None:
                    ent = self._capture_predict(images)
                    self._graphs[key] = ent
                if ent[0] is None:
                    return self.predict(images)
            finally:
                self._fork_now = None
                if saved is not None:
                    self._ws_main, self._ws_active, self.bnbuf, self._fz_coeffs = saved
        graph, static_in, outs = ent
        self._infer_prep_sync()                  # the graph reads the cached weight preparations: re-made here (eagerly) when the weights changed
        static_in.copy_(images)
        graph.replay()
        return outs

    def _capture_predict(self, images):
        cur = torch.cuda.current_stream()
        if cur == torch.cuda.default_stream(self.dev):
            # Measured (tools/experiments/README.md, "graph lanes"): a graph captured while the legacy default stream is current never
            # overlaps with another stream's work when replayed, on whichever stream -- two lanes then run strictly one after the
            # other.  Captured with a side stream current, the same graph overlaps (and may still be replayed on the default stream).
            if self._cap_stream is None:
                self._cap_stream = _shared_stream(self.dev, "capture")
            self._cap_stream.wait_stream(cur)
            with torch.cuda.stream(self._cap_stream):
                ent = self._capture_predict(images)
            cur.wait_stream(self._cap_stream)
            return ent
        static_in = images.clone()
        side = _shared_stream(self.dev, "capture_warmup")
        side.wait_stream(cur)
        with torch.cuda.stream(side):            # warm-up off the capture stream: workspace growth, caches, allocator pools
            for _ in range(2):
                self.predict(static_in)
        cur.wait_stream(side)
        torch.cuda.synchronize()
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = self.predict(static_in)
            return (graph, static_in, outs)
        except Exception as e:                   # capture is a launch mechanism, not a compute path: the same kernels run eagerly
            import warnings
            warnings.warn("hipGraph capture of the inference forward failed (%s: %s); launching eagerly" % (type(e).__name__, e))
            torch.cuda.synchronize()
            return (None, None, None)

    def predict_stream(self, batches, in_flight=2):
        """Throughput form of the inference forward: a generator over `batches` (an iterable of [B,H,W,3] device tensors of one shape)
        that keeps `in_flight` forwards running at once -- lane i % in_flight, each lane its own stream, hipGraph, scratch and static
        buffers -- and yields (yolo_output, detections, masks) per batch in submission order.  The trunk of one batch (small
        launch-bound fp32 kernels) runs underneath the matrix-pipe-bound mask head of the other.  Same kernels and bit-identical
        results as predict().  Submission runs two batches per lane ahead of the results handed out, so a lane never idles between
        its batches (and the lanes drift out of phase instead of running trunk beside trunk); each result is copied out of its lane's
        static buffers on the lane's stream, so it stays valid for as long as the caller keeps it."""
        if in_flight < 1:
            raise ValueError("in_flight must be >= 1")
        cur = torch.cuda.current_stream()
        pending = []

        def hand_out():
            ev, outs = pending.pop(0)
            ev.synchronize()
            for t in outs:
                t.record_stream(cur)
            return outs

        for i, images in enumerate(batches):
            if len(pending) == 2 * in_flight:
                yield hand_out()
            s = self._lane_state(i % in_flight)["stream"]
            s.wait_stream(cur)
            images.record_stream(s)
            with torch.cuda.stream(s):
                outs = tuple(t.clone() for t in self.predict_graphed(images, lane=i % in_flight, solo=in_flight == 1))
                ev = torch.cuda.Event()
                ev.record(s)
            pending.append((ev, outs))
        while pending:
            yield hand_out()

    def predict_yolo(self, images):
        """'yolo' mode forward (model.py:906-920)."""
        with self._infer_prep():
            self._activate()
            self.tape = {}
            cfg = self.cfg
            _, _, yo = self.trunk_fwd(images, False)
            self.tape = {}
            return yo.view(images.shape[0], cfg.GRID_H, cfg.GRID_W, cfg.N_BOX, 5 + cfg.NUM_CLASSES)
