#!/usr/bin/env python
"""bench.py -- Mask-YOLO hot-path throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W          (N > 1 without torchrun env: re-launches itself under
                                                          python -m torch.distributed.run, one rank per GPU)
  python bench.py --config rice416-bf16                  (BASELINE configs[3]: inference throughput, bf16 mask head)

Default workload (configs[1]): a step = one full training step of the hot path on one synthetic Shapes batch per GPU:
forward (backbone, YOLO head, decode, mask targets, ROIAlign, mask head on ALL ROIs, both losses), backward of all of it
(the mask head behind bn1 on the positive ROIs only -- exact), gradient all-reduce (N>1) and the
Adam update.  Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line (contract in the task
statement) with:
  roofline      -- the dominant kernel, timed live with HIP events on its launch stream inside the timed region, plus
                   objects for the kernels north_star names: depthwise (14 layers), ROIAlign (SURVEY 8(d) bytes), pointwise
  cpu_baseline  -- the CPU restatement (oracle/torch_ref.py, torch-CPU) timed on this box's host cores
  variants      -- the same K steps with the dense mask-head backward / the positives-only forward / forced positive counts
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4); the inference lanes (Net.predict_stream) only overlap when each has a
# queue of its own (+3 % at Rice 416), and the training step is laid out so that it does not depend on the mapping (profiles/r3_notes.md,
# "hardware queues").  Read by the HIP runtime when it initialises, i.e. before the first device call of the process.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "mask-yolo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np          # noqa: E402
import torch                # noqa: E402
import torch.distributed as dist   # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md
FP32_MFMA_PEAK = 157.3       # TFLOP/s
BF16_MFMA_PEAK = 2500.0      # TFLOP/s dense


MAX_LINE_BYTES = 6000        # the driver keeps a bounded tail of stdout: the ONE line it parses stays far below it (round 4's 22 KB line was lost)


def _r(x, nd=4):
    """round floats (recursively) so the line stays short; everything else unchanged."""
    if isinstance(x, float):
        return round(x, nd) if abs(x) < 1e6 else float("%.6g" % x)
    if isinstance(x, dict):
        return {k: _r(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, nd) for v in x]
    return x


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(res):
    """The ONE stdout line (contract keys + roofline of the dominant kernel + cpu_baseline + a few scalars), < MAX_LINE_BYTES.  The full
    result (per-layer trunk table, probes, variants, the Rice-416 kernels' own roofline ...) goes to bench_detail.json and to stderr."""
    out = _pick(res, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    if "step_ms" in res:
        out["step_ms"] = _pick(res["step_ms"], ("p10", "p50", "p90", "min", "max", "n"))
    cfg = res.get("config", {})
    c = _pick(cfg, ("workload", "winograd_tiles", "fp32_products", "global_batch", "parallelism", "rois_per_image", "n_pos_mean", "n_pos_sweep_ms",
                    "train_api_images_per_sec", "final_loss", "in_flight", "forced_positives", "share_gpu", "lib_options", "net_attrs"))
    if "workload" in c:
        c["workload"] = c["workload"][:360]
    for k in ("lib_options", "net_attrs", "forced_positives", "share_gpu"):
        if k in c and not c[k]:
            del c[k]
    out["config"] = c
    rf = res.get("roofline") or {}
    r = _pick(rf, ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_composite", "traffic", "algorithmic_bytes", "algorithmic_flop",
                   "avg_launch_ms", "launches_timed"))
    if "kernel" in r:
        r["kernel"] = r["kernel"][:200]
    if rf.get("traffic") and rf.get("algorithmic_bytes"):
        r["traffic_over_algorithmic"] = rf["traffic"] / rf["algorithmic_bytes"]
        r["traffic_measured_in_this_run"] = "traffic_from_profiles" in rf
    pl = res.get("power_limit_probe")
    if isinstance(pl, dict) and "ms_random" in pl:
        # the dominant kernel alone, back to back: ms on random / constant operands, package power and shader clock while it runs (cap, nominal 2400 MHz)
        r["power_limit_probe"] = _pick(pl, ("ms_random", "ms_constant", "power_w", "cap_w", "sclk_mhz"))
        if pl.get("sclk_mhz") and rf.get("frac"):
            # `frac` is priced against the NOMINAL peak (2400 MHz); this is the same figure against the peak at the clock the package power cap grants
            r["frac_at_granted_clock"] = rf["frac"] * 2400.0 / max(float(pl["sclk_mhz"]), 1.0)
    if "depthwise" in rf:
        r["depthwise"] = _pick(rf["depthwise"], ("frac", "frac_net", "avg_ms"))
    if "roialign" in rf and "frac" in rf["roialign"]:
        r["roialign"] = _pick(rf["roialign"], ("frac", "avg_ms"))
    if "pointwise" in rf and "frac_of_fp32_mfma_peak" in rf["pointwise"]:
        pw = rf["pointwise"]
        r["pointwise"] = {"frac": pw["frac_of_fp32_mfma_peak"], "frac_net": pw.get("frac_of_fp32_mfma_peak_net"), "avg_ms": pw["avg_ms"],
                          "frac_mfma_bound_layers": (pw.get("mfma_bound_layers") or {}).get("frac_of_fp32_mfma_peak")}
    if "deconv_mask" in rf:         # --config rice416-bf16
        r["deconv_mask_frac"] = rf["deconv_mask"]["achieved"] / rf["deconv_mask"]["peak"]
        r["roialign_frac"] = rf["roialign"]["achieved"] / rf["roialign"]["peak"]
    out["roofline"] = r
    cb = res.get("cpu_baseline")
    if isinstance(cb, dict):
        cb = dict(cb)
        if "sample" in cb:
            cb["sample"] = cb["sample"][:300]
    out["cpu_baseline"] = cb
    if "comm" in res:
        out["comm"] = _pick(res["comm"], ("backend", "rccl_ranks_seen", "bucket_allreduce_ms", "bucket_bytes", "weights_identical_across_ranks"))
        out["comm"]["backend"] = str(out["comm"].get("backend"))[:80]
    sec = res.get("secondary_nbox5")
    if isinstance(sec, dict):
        out["nbox5_images_per_sec"] = sec.get("value", sec.get("error"))
    inf = res.get("inference_rice416_bf16")
    if isinstance(inf, dict):
        out["rice416_bf16_images_per_sec"] = inf.get("value", inf.get("error"))
        out["rice416_bf16_one_in_flight_images_per_sec"] = (inf.get("one_in_flight") or {}).get("value")
        out["rice416_bf16_detect_many_images_per_sec"] = (inf.get("detect_many") or {}).get("images_per_sec")
    if isinstance(res.get("one_in_flight"), dict):
        out["one_in_flight_images_per_sec"] = res["one_in_flight"].get("value")
    if isinstance(res.get("detect_many"), dict):
        out["detect_many_images_per_sec"] = res["detect_many"].get("images_per_sec")
    hw = res.get("host_wait_on_n_pos_ms_per_step")
    if isinstance(hw, dict):
        out["host_wait_on_n_pos_ms_per_step"] = hw.get("value")
    if isinstance(res.get("variants"), dict) and "dense_mask_backward" in res["variants"]:
        out["dense_mask_backward_ms_per_step"] = res["variants"]["dense_mask_backward"]["ms_per_step"]
    out["detail"] = "bench_detail.json"
    return _r(out)


def write_detail(res, args):
    """the full result object: bench_detail.json at the repo root (and in gpurun_out/ when that exists, so it travels back), and on stderr."""
    txt = json.dumps(res, indent=1)
    for d in (ROOT, os.path.join(ROOT, "gpurun_out")):
        if os.path.isdir(d):
            try:
                with open(os.path.join(d, args.detail_name), "w") as f:
                    f.write(txt + "\n")
            except OSError:
                pass
    sys.stderr.write("bench.py detail:\n" + txt + "\n")
    sys.stderr.flush()


def make_batches(cfg, rank, world, per_gpu, nbatches):
    """nbatches distinct host batches for this rank; image g of the global stream is seeded by g."""
    from myolo.shapes import make_shapes_samples
    from myolo.myolo_utils import BatchGenerator
    out = []
    for i in range(nbatches):
        start = (i * world + rank) * per_gpu
        samples = make_shapes_samples(per_gpu, cfg, start_index=start)
        batch, _ = BatchGenerator(samples, cfg, 'training', shuffle=False, norm=True)[0]
        out.append(batch)
    return out


def usable_cores():
    """host cores this process may really use: min(affinity mask, cgroup cpu quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                n = max(1, min(n, int(float(q) / float(per))))
    except Exception:
        pass
    return n


def host_mem_gb():
    """memory this process may use: min(MemAvailable, cgroup limit), GiB."""
    avail = 1e9
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = int(line.split()[1]) / 2.0 ** 20
    except Exception:
        pass
    try:
        v = open("/sys/fs/cgroup/memory.max").read().strip()
        if v != "max":
            used = int(open("/sys/fs/cgroup/memory.current").read().strip())
            avail = min(avail, (int(v) - used) / 2.0 ** 30)
    except Exception:
        pass
    return avail


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(cfg, want_images, seed=0, budget_s=75.0, infer=False):
    """Time the CPU restatement (torch-CPU fp32, oracle/torch_ref.py: autograd training step + Keras Adam, or the inference
    forward) on this box's host cores: SURVEY 8(d) protocol scaled to a bounded run -- one warm-up step, then up to 3 timed
    steps at the full per-GPU batch (fewer images if host memory is short: the autograd tape of a 32-image step holds ~50 GB)
    inside a wall-clock budget.  oneDNN stops scaling past a few dozen threads at this size: min(usable cores, 32) threads."""
    from myolo.config import make_config
    from myolo.shapes import make_shapes_samples
    from myolo.myolo_utils import BatchGenerator
    from oracle import np_model
    from oracle.torch_ref import TorchRef
    avail = usable_cores()
    threads = max(1, min(avail, 32))
    torch.set_num_threads(threads)
    mem = host_mem_gb()
    n = int(want_images)
    per_img_gb = 1.8 * (cfg.TRAIN_ROIS_PER_IMAGE / 147.0) * (0.35 if infer else 1.0)
    while n > 1 and n * per_img_gb > 0.6 * mem:
        n //= 2
    ccfg = make_config(type(cfg).__mro__[1], IMAGE_SHAPE=list(cfg.IMAGE_SHAPE), ALPHA=cfg.ALPHA, BATCH_SIZE=n,
                       N_BOX=cfg.N_BOX, ANCHORS=list(cfg.ANCHORS))
    ref = TorchRef(np_model.init_params(ccfg, seed=seed), ccfg, torch.float32)
    if infer:
        images = np.random.default_rng(0).random((n,) + tuple(ccfg.IMAGE_SHAPE), dtype=np.float32)

        def step(i):
            with torch.no_grad():
                _, Fm, yo = ref.trunk(images, False)
                det = __import__("oracle.np_ops", fromlist=["x"]).yolo_detections(yo.numpy(), ccfg.ANCHORS, ccfg.GRID_W)
                ref.mask_head(Fm, det[..., :4], False)
    else:
        samples = make_shapes_samples(n, ccfg)
        batch, _ = BatchGenerator(samples, ccfg, 'training', shuffle=False, norm=True)[0]
        state = {}

        def step(i):
            ref.train_step(batch)
            ref.adam(state, i + 1, 1e-3)
    t0 = time.time()
    step(0)                                        # warm-up (first touch, oneDNN primitive creation)
    t_warm = time.time() - t0
    times = []
    for i in range(3):
        if times and (time.time() - t0) + times[-1] > budget_s:
            break
        t1 = time.time()
        step(i + 1)
        times.append(time.time() - t1)
    t = float(np.median(times))
    what = "inference forward" if infer else "training step (fwd+bwd+Adam)"
    return dict(value=n / t, unit="images/sec", cores=threads, kind="port", cpu_model=cpu_model(),
                sample="CPU restatement (torch-CPU fp32, oracle/torch_ref.py), NOT Keras: %d timed %s(s) after 1 warm-up on %d image(s) of "
                       "%dx%d, N_BOX=%d (R=%d); %d threads of %d usable host cores (os.cpu_count()=%d, %s); median %.1f s/step "
                       "(all: %s), warm-up %.1f s; host memory available %.0f GiB"
                       % (len(times), what, n, cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1], cfg.N_BOX, cfg.TRAIN_ROIS_PER_IMAGE, threads, avail,
                          os.cpu_count() or 0, cpu_model(), t, ["%.1f" % x for x in times], t_warm, mem))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def relaunch_under_torchrun(args_gpus):
    """plain `python bench.py --gpus N`: become `python -m torch.distributed.run ... bench.py --gpus N ...` (one rank per GPU)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


class KernelTimer(object):
    """aggregate of per-launch HIP-event timings collected by Net._call_timed for a family of tags."""

    def __init__(self, net):
        self.net = net

    def total_ms_per_step(self, prefix, steps):
        tot, n = 0.0, 0
        for tag in list(self.net.timings):
            if tag.startswith(prefix):
                ms, k = self.net.kernel_ms(tag)
                tot += ms * k
                n += k
        return tot / max(1, steps), n // max(1, steps)


def timed_steps(net, dbs, steps, lr, barrier, per_step=None, host_ms=None):
    """K steps between two barriers (host wall clock = the reported time).  per_step (a list): filled with the K step durations in ms
    from HIP events recorded on the compute stream at the step boundaries (no synchronisation inside the region)."""
    import gc
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)] if per_step is not None else None
    # the host runs about one step (~20 ms) ahead of the GPU and cannot run further ahead (it reads n_pos every step): a collector pause inside
    # the region comes straight out of that lead, so the cyclic collector is parked for the K steps (reference counting still frees everything)
    # (a caller that parked it BEFORE its warm-up steps -- bench_train -- keeps the GPU busy right up to the first barrier: a full collection here is
    # 10-40 ms of idle GPU in front of step 0, which then ran 2-14 ms long in most runs' step_ms)
    gc_was = gc.isenabled()
    if gc_was:
        gc.collect()
        gc.disable()
    try:
        barrier()
        t0 = time.perf_counter()
        out = None
        th = [t0]
        diag = []
        for i in range(steps):
            if evs:
                evs[i].record()
            out = net.train_step(dbs[i % len(dbs)], lr)
            th.append(time.perf_counter())
            if host_ms is not None:
                diag.append((int(getattr(net, "_np_seen", -1)), int(torch.cuda.memory_reserved())))
        if evs:
            evs[steps].record()
        barrier()
        el = time.perf_counter() - t0
        if host_ms is not None:                  # host wall time per train_step call (issue + the wait for n_pos): a step that is long HERE and on the GPU is a host stall
            host_ms.extend(1e3 * (th[i + 1] - th[i]) for i in range(steps))
            timed_steps.last_diag = diag         # (positive ROIs the host read, bytes reserved by the caching allocator) after every step
    finally:
        if gc_was:
            gc.enable()
    if evs:
        per_step.extend(evs[i].elapsed_time(evs[i + 1]) for i in range(steps))
    return el, out


def percentiles(ms):
    raw = np.asarray(ms, np.float64)
    a = np.sort(raw)
    return {"p10": float(np.percentile(a, 10)), "p50": float(np.percentile(a, 50)), "p90": float(np.percentile(a, 90)),
            "min": float(a[0]), "max": float(a[-1]), "max_at_step": int(np.argmax(raw)) if raw.size else -1, "n": int(a.size),
            "source": "HIP events on the compute stream at the step boundaries, inside the timed region"}


def bench_train_api(model, cfg, args, dev, n_images=512, epochs=4, stream_steps=48):
    """img/s of the reference's public training calls on this build (world 1):
      train():               MaskYOLO.train(ShapesDataset of n_images, ...) -- host BatchGenerator.__getitem__ on a prefetch thread, pinned
                             staging, async H2D on the upload stream, lazy StepResult; timed between the on_epoch_end callbacks of epochs
                             2..E (epoch 1 warms the staging buffers), each callback after a device synchronise
      train_shapes_stream(): inputs rasterised / encoded on the device (ShapesProducer); whole call timed, losses read at the end"""
    from myolo.shapes import ShapesDataset
    B = cfg.BATCH_SIZE
    ds = ShapesDataset(1234)
    ds.load_shapes(n_images, cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1])
    ds.prepare()
    stamps = []

    def cb(ep, logs):
        torch.cuda.synchronize()
        stamps.append(time.perf_counter())
    t0 = time.perf_counter()
    model.net.host_wait_s = 0.0
    hist = model.train(ds, None, args.lr, epochs=epochs, layers="all", verbose=0, custom_callbacks=[cb])
    t_total = time.perf_counter() - t0
    ht = dict(model.host_times)
    nst = max(1, ht["steps"])
    host_ms = {"wait_for_batch_ms_per_step": 1e3 * ht["wait_for_batch_s"] / nst, "launch_step_ms_per_step": 1e3 * ht["launch_step_s"] / nst,
               "of_which_wait_on_n_pos_ms_per_step": 1e3 * model.net.host_wait_s / nst}
    steps_per_epoch = (n_images + B - 1) // B
    el = stamps[-1] - stamps[0]
    train_ips = B * steps_per_epoch * (len(stamps) - 1) / el
    assert all(np.isfinite(h) for h in hist)
    model.train_shapes_stream(4, learning_rate=args.lr)          # warm-up (producer buffers)

    def timed_stream(n, start):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ls = model.train_shapes_stream(n, learning_rate=args.lr, start_index=start)
        torch.cuda.synchronize()
        assert all(np.isfinite(l) for l in ls)
        return time.perf_counter() - t0
    # the call has a fixed cost (compile(), the collector parked and released, pipeline fill): a short and a long call, the rate from the difference
    short = 16
    el_s = timed_stream(short, 4 * B)
    el_l = timed_stream(short + stream_steps, (4 + short) * B)
    el2 = el_l - el_s
    # the same engine step on resident batches IN THE STATE THE CALLS ABOVE LEFT: by now the net has trained for ~150 steps, its proposals hit ground-truth
    # boxes more often, and every positive ROI per image adds ~0.2 ms of exact-sparsity backward -- the headline `value` is measured on the random-init net
    # (0.2-0.35 positives per image).  This is the figure the public calls are to be compared with like for like.
    ref = None
    try:
        from myolo.shapes import ShapesProducer
        prod = ShapesProducer(cfg, seed=1234, device=dev)
        lo0 = (4 + 2 * short + stream_steps) * B
        dbs_ref = [prod.batch(list(range(lo0 + k * B, lo0 + (k + 1) * B))) for k in range(4)]
        for k in range(4):
            model.net.train_step(dbs_ref[k], args.lr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nref, npos_acc = 24, 0.0
        outs = []
        for k in range(nref):
            outs.append(model.net.train_step(dbs_ref[k % 4], args.lr)["n_pos"])
        torch.cuda.synchronize()
        el_ref = time.perf_counter() - t0
        npos_acc = float(torch.stack([o.float().mean() for o in outs]).mean())
        ref = {"ms_per_step": 1e3 * el_ref / nref, "images_per_sec": B * nref / el_ref, "steps": nref, "n_pos_mean": npos_acc,
               "note": "Net.train_step on four resident Shapes batches right after the calls above, same weights: the like-for-like reference of the public calls "
                       "(the headline runs on the random-init net, whose proposals rarely match a ground-truth box)"}
    except Exception as e:
        ref = {"error": "%s: %s" % (type(e).__name__, e)}
    return {"reference_same_state": ref, "train": {"images_per_sec": train_ips, "ms_per_step": 1e3 * el / (steps_per_epoch * (len(stamps) - 1)),
                      "images": n_images, "epochs_timed": len(stamps) - 1, "steps_per_epoch": steps_per_epoch,
                      "setup_and_first_epoch_s": t_total - el, "epoch_mean_loss": hist, "launch_thread": host_ms},
            "train_shapes_stream": {"images_per_sec": B * stream_steps / el2, "ms_per_step": 1e3 * el2 / stream_steps, "steps": stream_steps,
                                    "how": "time of a %d-step call minus time of a %d-step call (the call's fixed cost -- compile(), parking the collector, "
                                           "pipeline fill -- cancels)" % (short + stream_steps, short),
                                    "whole_call_images_per_sec": B * (short + stream_steps) / el_l, "call_fixed_cost_ms": 1e3 * (el_s - short * el2 / stream_steps)},
            "note": "public calls of myolo.model.MaskYOLO (reference surface model.py:943-1060), single GPU; compare with `value` "
                    "(Net.train_step on device-resident batches)"}


def bench_train(args, rank, world, local):
    from myolo import dist as mdist
    from myolo.config import make_config, ShapesConfig, ShapesHeadConfig
    from myolo.model import MaskYOLO
    dev = "cuda:%d" % local
    base = ShapesConfig if args.nbox == 3 else ShapesHeadConfig
    cfg = make_config(base, IMAGE_SHAPE=[args.size, args.size, 3], ALPHA=args.alpha, BATCH_SIZE=args.batch,
                      TRAIN_MASK_HEAD_ROIS=args.mask_head_rois, CONV3X3_ALGO=args.conv3x3, FP32_MATMUL=args.fp32_matmul,
                      **({"WINOGRAD_TILES": args.wino_tiles} if args.wino_tiles else {}))
    model = MaskYOLO(mode="training", config=cfg, device=dev, seed=0)      # same seed -> same weights on every rank
    net = model.net
    for kv in args.net_attr:
        name, _, val = kv.partition("=")
        assert hasattr(net, name), "unknown engine attribute %s" % name
        setattr(net, name, type(getattr(net, name))(int(val)))
    comm_stream = None if args.comm_own_stream else net._copy_stream        # (see GradReducer: a fresh HIP stream is not free)
    reducer = mdist.GradReducer(net.flat_g, net.bucket_ranges, backend=args.comm, timing=world > 1, stream=comm_stream)
    reducer.attach(net)
    ranks_seen = reducer.ranks_seen() if world > 1 else 1

    nb = 2
    host = make_batches(cfg, rank, world, args.batch, nb)
    dbs = [net.to_device_batch(b) for b in host]
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def maxr(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    if args.force_pos > 0:
        def force_hook(proposals, db, k=args.force_pos):
            gt = db["gt_boxes"].to(torch.float32)                       # [B,T,4] px, x1 y1 x2 y2
            H, W = float(cfg.IMAGE_SHAPE[0]), float(cfg.IMAGE_SHAPE[1])
            norm = (gt - torch.tensor([0., 0., 1., 1.], device=gt.device)) / torch.tensor([W - 1, H - 1, W - 1, H - 1], device=gt.device)
            proposals[:, :k, :] = norm[:, :1, :].expand(-1, k, -1)
        net.proposals_hook = force_hook
    if args.force_pos <= 0:
        # allocator / scratch pre-sizing, NOT a training step (no optimizer update): one forward + backward with 24 positives forced per image, so that no
        # step of the warm-up or of the timed region is the first to need a larger scratch buffer or allocator block when n_pos drifts with the weights
        # (seen once on a fresh box: a 54 ms step inside the timed region, the other 19 at 20.5 ms)
        def presize_hook(proposals, db, k=24):
            gt = db["gt_boxes"].to(torch.float32)
            H, W = float(cfg.IMAGE_SHAPE[0]), float(cfg.IMAGE_SHAPE[1])
            norm = (gt - torch.tensor([0., 0., 1., 1.], device=gt.device)) / torch.tensor([W - 1, H - 1, W - 1, H - 1], device=gt.device)
            proposals[:, :k, :] = norm[:, :1, :].expand(-1, k, -1)
        net.proposals_hook = presize_hook
        # ... repeated for ~0.2 s: a cold GPU's first stretch of sustained 1400 W load should lie in front of the warm-up steps, not ~150 ms into
        # the run where the timed region begins (two first-on-a-fresh-box runs of this session carried one 56 ms step at index 2, host and device
        # alike; profiles/r6_notes.md).  No optimizer update: the weights the warm-up steps start from are the same.
        for _ in range(1 + max(0, args.device_warm)):
            net.forward_backward(dbs[0])
            net.join_conv1_wgrad()
            net.join_trunk_wgrad()
            if net.before_optimizer:                 # data-parallel: the buckets' all-reduces of this pass are joined like an optimizer step would
                net.before_optimizer()
        net.proposals_hook = None
        net.seen = 0
        torch.cuda.synchronize()
    import gc
    gc_was = gc.isenabled()
    gc.collect()                                 # host housekeeping in front of the warm-up, not between it and the timed region (see timed_steps)
    gc.disable()
    for i in range(args.warmup):
        net.train_step(dbs[i % nb], args.lr)
    # ---- the timed region: only the dominant kernel (and the conv op it belongs to) is bracketed with events
    dom_tags = {"mask_conv3x3_fwd", "wino_multiply"}
    net.timed_tags = set(dom_tags)
    net.timings = {}
    step_ms = []
    net.host_wait_s = 0.0
    host_ms = []
    elapsed, out = timed_steps(net, dbs, args.steps, args.lr, barrier, per_step=step_ms, host_ms=host_ms)
    if gc_was:
        gc.enable()
    npos_wait_ms = 1e3 * net.host_wait_s / max(1, args.steps)
    elapsed = maxr(elapsed)
    loss = float(out["yolo_terms"][0]) + float(out["mask_terms"][0])
    assert np.isfinite(loss), "non-finite loss in the timed region"
    npos_mean = float(out["n_pos"].float().mean())          # positives per image in the last timed batch
    conv_ms, conv_n = net.kernel_ms("mask_conv3x3_fwd")
    mul_ms, mul_n = net.kernel_ms("wino_multiply")
    bucket_ms = reducer.bucket_ms() if world > 1 else None
    release_ms = reducer.release_ms_before_wait() if world > 1 else None
    weights_same = None
    if world > 1:
        # every rank must hold bit-identical weights after the K averaged updates: an integer digest of the flat parameter buffer, MIN == MAX over ranks
        net.join_conv1_wgrad()
        net.join_trunk_wgrad()
        dig = net.flat_p.view(torch.int32).to(torch.int64).sum().reshape(1)
        dig2 = (net.flat_p.view(torch.int32).to(torch.int64) * torch.arange(1, net.flat_p.numel() + 1, device=dev, dtype=torch.int64).remainder_(8191)).sum().reshape(1)
        d = torch.cat([dig, dig2])
        lo_, hi_ = d.clone(), d.clone()
        if args.share_gpu:
            lo_, hi_ = lo_.cpu(), hi_.cpu()
        dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        weights_same = bool(torch.equal(lo_, hi_))

    # ---- second pass (not part of `value`): per-launch timings of the kernels north_star names
    net.timed_tags = {"roialign_fwd", "wino_in", "wino_out_in"} | {"dw%d_fwd" % i for i in range(1, 15)} | {"pw%d_fwd" % i for i in range(1, 15)}
    net.timings = {}
    ksteps = 3
    for i in range(ksteps):
        net.train_step(dbs[i % nb], args.lr)
    torch.cuda.synchronize()
    kt = KernelTimer(net)
    dw_ms, _ = kt.total_ms_per_step("dw", ksteps)
    pw_ms, _ = kt.total_ms_per_step("pw", ksteps)
    layer_ms = {tag: net.kernel_ms(tag)[0] for tag in list(net.timings) if tag[:2] in ("dw", "pw")}
    roi_ms, _ = net.kernel_ms("roialign_fwd")
    win_ms, win_n = net.kernel_ms("wino_in")
    woi_ms, woi_n = net.kernel_ms("wino_out_in")
    # what such an event bracket costs by itself: the same pair of timing events around a 4-byte fill (a ~2 us kernel).  The brackets of the
    # trunk's 5-50 us launches carry this much on top of the kernel (rocprofv3's kernel times of the same launches are in profiles/)
    from myolo import _ext as Xb
    net.timed_tags = {"empty_bracket"}
    net.timings = {}
    probe_buf = torch.zeros(4, device=dev)
    for _ in range(50):
        net._call_timed("empty_bracket", "myolo_fill", Xb.ptr(probe_buf), 0.0, 1, Xb.stream())
    torch.cuda.synchronize()
    br = sorted(a.elapsed_time(b) for a, b in net.timings["empty_bracket"])
    bracket_ms = br[len(br) // 2]
    net.timed_tags = set()

    # ---- variants (reported beside `value`, never as `value`)
    variants = {}
    if not args.no_variant:                   # every rank runs them in lockstep (barriers / all-reduces inside)
        def run_variant(setup, restore, note):
            setup()
            for i in range(2):
                net.train_step(dbs[i % nb], args.lr)
            el, o = timed_steps(net, dbs, args.steps, args.lr, barrier)
            el = maxr(el)
            restore()
            assert np.isfinite(float(o["yolo_terms"][0]) + float(o["mask_terms"][0])), "non-finite loss in a variant run"
            return {"value": args.batch * world * args.steps / el, "unit": "images/sec", "ms_per_step": 1e3 * el / args.steps, "note": note}

        variants["dense_mask_backward"] = run_variant(
            lambda: setattr(net, "sparse_mask_bwd", False), lambda: setattr(net, "sparse_mask_bwd", True),
            "mask-head backward on ALL ROIs (conv2-4 / deconv / myolo_mask dense): the structural zeros behind bn1 are not "
            "exploited; same gradients (tests/test_gpu_step.py::test_sparse_mask_backward_equals_dense)")
        mine = net.fp32_matmul
        other = "native" if mine == "bf16x6" else "bf16x6"
        variants["fp32_products_%s" % other] = run_variant(
            lambda: setattr(net, "fp32_matmul", other), lambda: setattr(net, "fp32_matmul", mine),
            "cfg.FP32_MATMUL='%s' instead of '%s': the Winograd multiply's fp32 products formed %s; identical fp32 inputs, outputs and "
            "accumulation, error against fp64 not larger than the native path's (tests/test_gpu_ops.py::test_wino_multiply_bf16x6_accuracy)"
            % (other, net.fp32_matmul, "from six exact bf16 piece products on the bf16 matrix pipe (csrc/wino_mm.hip)" if other == "bf16x6"
               else "by v_mfma_f32_32x32x2_f32"))
        if world == 1:
            def flip_fwd():
                net.sparse_mask_fwd = not net.sparse_mask_fwd
            variants["mask_head_forward_on_positives_only"] = run_variant(
                flip_fwd, flip_fwd,
                "TRAIN_MASK_HEAD_ROIS='%s': conv2-4/deconv/myolo_mask forward on the positive ROIs only, whose outputs are the "
                "only ones the training graph reads; same loss, gradients, weights and BN state" % ("all" if net.sparse_mask_fwd else "positives"))
            # n_pos sweep: the first k proposals of every image are replaced by a ground-truth box (IoU 1 -> positive), so the
            # sparse backward works on ~k positives per image whatever the random-init net predicts
            sweep = {}
            for k in (5, 10, 20):
                def hook(proposals, db, k=k):
                    gt = db["gt_boxes"].to(torch.float32)                       # [B,T,4] px, x1 y1 x2 y2
                    H, W = float(cfg.IMAGE_SHAPE[0]), float(cfg.IMAGE_SHAPE[1])
                    norm = (gt - torch.tensor([0., 0., 1., 1.], device=gt.device)) / torch.tensor([W - 1, H - 1, W - 1, H - 1], device=gt.device)
                    first = norm[:, :1, :].expand(-1, k, -1)                    # every image has >= 1 instance
                    proposals[:, :k, :] = first
                v = run_variant(lambda: setattr(net, "proposals_hook", hook), lambda: setattr(net, "proposals_hook", None),
                                "first %d proposals of every image forced onto a ground-truth box" % k)
                sweep["n_pos_%d" % k] = {"images_per_sec": v["value"], "ms_per_step": v["ms_per_step"]}
            variants["n_pos_sweep"] = sweep

    extras = {}
    if world == 1 and not args.no_extras:
        from myolo import _ext as Xe
        # (a) measured HBM copy bandwidth of a hand-written float4 kernel, beside the nominal peak
        try:
            extras["hbm_copy_measured_gbs"] = Xe.measure_hbm_copy_gbs(device=dev)
        except Exception as e:
            extras["hbm_copy_measured_gbs"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # (a2) what the matrix pipes sustain with nothing else going on (register operands, no memory traffic), beside the nominal peaks:
        #      the clock the power management grants under MFMA load is part of the roofline
        try:
            extras["mfma_measured_tflops"] = dict(Xe.measure_mfma_tflops(device=dev), nominal_bf16=BF16_MFMA_PEAK, nominal_f32=FP32_MFMA_PEAK)
        except Exception as e:
            extras["mfma_measured_tflops"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # (a3) is the dominant kernel power-limited?  The same multiply on random and on constant operands, socket power and shader clock meanwhile
        try:
            pl = Xe.measure_power_limit(n_rois=args.batch * cfg.TRAIN_ROIS_PER_IMAGE, device=dev)
            pl["note"] = ("wino_mm_x6_kernel at the launch shape of the mask-head convs, 500 launches back to back: same instructions and traffic on normally "
                          "distributed and on constant operands; power / clock sampled with rocm-smi while the random-data launches run (nominal 2400 MHz)")
            extras["power_limit_probe"] = pl
        except Exception as e:
            extras["power_limit_probe"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # (b) what the three real gradient buckets cost as RCCL collectives on the comm stream while backward runs: a 1-rank
        #     communicator through the C-ABI (launch + stream + kernel cost of the exchange, not the wire), and the step with it
        try:
            probe = mdist.GradReducer(net.flat_g, net.bucket_ranges, always=True, backend="capi", timing=True, stream=comm_stream)
            probe.attach(net)
            for i in range(2):
                net.train_step(dbs[i % nb], args.lr)
            probe.bucket_ms()                                    # drop the warm-up pairs
            probe._ms = [[] for _ in probe._ms]
            probe._rel_ms = [[] for _ in probe._rel_ms]
            el, _ = timed_steps(net, dbs, args.steps, args.lr, barrier)
            rel = probe.release_ms_before_wait()
            bb = [4 * (hi - lo) for lo, hi in net.bucket_ranges]
            extras["comm_overlap_probe_ms"] = {
                "bucket_allreduce_ms": probe.bucket_ms(), "bucket_bytes": bb,
                "release_ms_before_step_end": rel,
                "bytes_released_3ms_or_more_before_step_end_frac": (sum(b for b, r in zip(bb, rel) if r >= 3.0) / float(sum(bb))) if rel else None,
                "ms_per_step_with_probe": 1e3 * el / args.steps,
                "note": "1-rank RCCL communicator (myolo_comm_* through the C-ABI): the five real buckets all-reduced on the comm stream as backward "
                        "completes them [backbone, YOLO blocks + conv_23, feature_map, mask conv1 + bn1, rest of the mask head]; cost of issuing the "
                        "exchange, not of the wire.  release_ms_before_step_end: from the event a bucket's collective waits for to the end of backward "
                        "(the join in front of Adam) -- the time available to hide that bucket's exchange"}
            probe.close()
        except Exception as e:
            extras["comm_overlap_probe_ms"] = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            reducer.attach(net)
        # (c) the DROP-IN surface (model.py:943-1060): MaskYOLO.train() on a ShapesDataset through BatchGenerator + the pinned
        #     prefetch/upload path, and train_shapes_stream() with the inputs produced on the device -- img/s of the public calls, to be
        #     read against `value` (Net.train_step on batches already resident in HBM)
        try:
            extras["train_api"] = bench_train_api(model, cfg, args, dev)
        except Exception as e:
            extras["train_api"] = {"error": "%s: %s" % (type(e).__name__, e)}
        extras["host_wait_on_n_pos_ms_per_step"] = {
            "value": npos_wait_ms, "note": "wall time the HOST thread is blocked on the per-image positive counts (the step's one device-to-host "
            "read, issued mid-forward on a copy stream): the host runs a whole step ahead of the GPU, so this is host idle time while the GPU "
            "drains its queue -- the GPU itself never waits for it (kernel time == wall time, profiles/r3_timeline.txt)"}
    if rank != 0:
        return None
    R = cfg.TRAIN_ROIS_PER_IMAGE
    ps = cfg.MASK_POOL_SIZE
    M = args.batch * R * ps * ps
    flop_direct = 2.0 * M * (9 * 256) * 256                   # one mask-head 3x3 conv as a direct convolution
    from myolo import _ext as X
    tiles_w = args.batch * R * ((ps + 3) // 4) ** 2
    ptiles = X.wino_plane_elems(args.batch * R, ps, ps, 1)     # point-tiles: 36 per tile, fewer where the ragged edge uses F(2,3)
    wflop = 2.0 * ptiles * 256 * 256                          # one 256x256 product per point-tile
    traffic, traffic_src = None, None
    t63 = bool(mul_n) and net.wino_tiles == "f63" and X.wino63_ok(ps, ps, 256, 256)
    c1_63 = t63 and net.lazy_bn1_bwd and net.sparse_mask_bwd          # engine._mask_convs_winograd_chain: conv1 on that tiling too
    if t63:          # 400 point-tiles per ROI (484 for conv1 where it stays on the F(4,3)/F(2,3) tiling): the four timed launches average to
        ptiles = (400 * args.batch * R if c1_63 else (ptiles + 3 * 400 * args.batch * R) / 4.0)
        wflop = 2.0 * ptiles * 256 * 256
    if mul_n:
        kflop, kms, kn = wflop, mul_ms, mul_n
        kname = ("wino_mm_kernel: ONE launch of the per-point GEMMs V[q] * U[q] (multiply stage of the mask-head 3x3 convs; %s: "
                 "%d point-tiles = %.1f per ROI instead of 576, K=256 N=256)" % (
                     ("F(6,3)/F(4,3) tiling, 14 = 6+4+4: 64 planes" if c1_63 else
                      "conv1 on the F(4,3)/F(2,3) tiling (484 per ROI, 36 planes), conv2-4 on the F(6,3)/F(4,3) tiling (400 per ROI, 64 planes); "
                      "average over the four launches of a step") if t63 else "mixed F(4,3)/F(2,3) tiling", ptiles, ptiles / float(args.batch * R)))
        kbytes = float(ptiles) * (256 + 256) * 4 + ((4 * 64 if c1_63 else 36 + 3 * 64) if t63 else 4 * 36) / 4.0 * 256 * 256 * 4
        pmc = "r2_pmc_wino_multiply.json"
    else:
        kflop, kms, kn = flop_direct, conv_ms, conv_n
        kname = "gemm_nn_fast<CONV3> (mask-head 3x3 conv fwd, M=%d K=2304 N=256)" % M
        kbytes = 2.0 * M * 256 * 4 + 9 * 256 * 256 * 4
        pmc = "r1_pmc_conv3x3_fwd.json"
    try:
        pj = json.load(open(os.path.join(ROOT, "profiles", pmc)))
        if args.batch * R == 32 * 147:        # the counters were collected at exactly this shape
            traffic = pj["traffic_bytes_per_launch_corrected"]
            traffic_src = "profiles/%s (separate rocprofv3 --pmc passes, FETCH_SIZE x2 + WRITE_SIZE; not re-measured in this run)" % pmc
            if t63:
                p63 = json.load(open(os.path.join(ROOT, "profiles", "r2_pmc_wino63_multiply.json")))
                if c1_63:
                    traffic = p63["traffic_bytes_per_launch_corrected"]
                    traffic_src = "profiles/r2_pmc_wino63_multiply.json (separate rocprofv3 --pmc passes, FETCH_SIZE x2 + WRITE_SIZE; not re-measured in this run)"
                else:
                    traffic = (traffic + 3 * p63["traffic_bytes_per_launch_corrected"]) / 4.0
                    traffic_src = ("average over a step's four launches of profiles/%s (conv1) and 3 x profiles/r2_pmc_wino63_multiply.json (conv2-4); "
                                   "separate rocprofv3 --pmc passes, FETCH_SIZE x2 + WRITE_SIZE; not re-measured in this run" % pmc)
    except Exception:
        pass
    peak = FP32_MFMA_PEAK
    x6 = bool(mul_n) and net.fp32_matmul == "bf16x6"
    if x6:          # six bf16 piece products per fp32 product: price the launch in bf16 MFMA flops against the dense bf16 peak
        kname = kname.replace("wino_mm_kernel", "wino_mm_x6_kernel (FP32_MATMUL='bf16x6': flops counted are the 6 bf16 piece products per fp32 product)")
        kflop, peak, pmc = 6.0 * kflop, BF16_MFMA_PEAK, "r2_pmc_wino_multiply_x6.json"
        traffic, traffic_src = None, None
        try:
            if args.batch * R == 32 * 147 and c1_63:            # the default path: every dense launch on the F(6,3)/F(4,3) tiling
                pj = json.load(open(os.path.join(ROOT, "profiles", "r3_pmc_x6.json")))["multiply_x6"]
                traffic = pj["traffic_bytes_per_launch_corrected"]
                traffic_src = ("profiles/r3_pmc_x6.json (tools/collect_pmc.sh x6: separate rocprofv3 --pmc passes on tools/kbench.py wino63_mm, FETCH_SIZE x2 + "
                               "WRITE_SIZE = %.2fx the algorithmic bytes; not re-measured in this run)" % pj["traffic_over_algorithmic"])
            elif args.batch * R == 32 * 147 and not t63:
                pj = json.load(open(os.path.join(ROOT, "profiles", pmc)))
                traffic = pj["traffic_bytes_per_launch_corrected"]
                traffic_src = "profiles/%s (separate rocprofv3 --pmc passes, FETCH_SIZE x2 + WRITE_SIZE; not re-measured in this run)" % pmc
        except Exception:
            pass
    achieved = kflop / (kms * 1e-3) / 1e12 if kms > 0 else 0.0

    # SURVEY 8(d) bytes
    fm = args.size // 8
    roi_bytes = float(M) * 256 * 4 + args.batch * fm * fm * 256 * 4
    dwb = dw_bytes(args.size, args.alpha, args.batch)
    pwf, pwb = pw_flops_bytes(args.size, args.alpha, args.batch)

    def hbm_obj(kernel, nbytes, ms, **extra):
        gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        d = {"kernel": kernel, "bound": "hbm", "algorithmic_bytes": nbytes, "avg_ms": ms, "achieved": gbs, "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS}
        d.update(extra)
        return d
    roofline = {"kernel": kname, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes": kbytes, "algorithmic_flop": kflop, "launches_timed": kn, "avg_launch_ms": kms,
                "fp32_equivalent_tflops": wflop / (kms * 1e-3) / 1e12 if (mul_n and kms > 0) else None,
                "conv_op": {"algo": (("winograd F(6,3)/F(4,3) tiling" if c1_63 else "winograd: conv1 F(4,3)/F(2,3) tiling, conv2-4 F(6,3)/F(4,3) tiling (average of the four ops)") if t63 else "winograd_f4x4_3x3") if mul_n else "direct",
                            "avg_ms": conv_ms, "ops_timed": conv_n, "direct_conv_flop": flop_direct,
                            "direct_equivalent_tflops": flop_direct / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0,
                            "winograd_flop_frac_of_peak": wflop / (conv_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK if conv_ms > 0 else 0.0},
                "depthwise": hbm_obj("dw_rows_kernel (row-sliding LDS-staged, round 4), the 14 depthwise 3x3 layers of backbone + YOLO head (sum over the layers, one step)",
                                     dwb, dw_ms, note="SURVEY 8(d): activation in + out once, 20.97 MB/img at 224^2 alpha 1"),
                "roialign": hbm_obj("ROIAlign forward (fused into conv1's Winograd input transform when CONV3X3_ALGO != direct)", roi_bytes, roi_ms,
                                    note="SURVEY 8(d) bytes: the [B*R,14,14,256] crops + one read of the feature map; the fused kernel of the default "
                                         "path writes conv1's Winograd image instead: %s" % (
                                             "64 planes / 400 point-tiles per ROI = 2.04x those bytes" if c1_63 else "36 planes / 484 point-tiles per ROI = 2.47x those bytes"),
                                    written_bytes=float(ptiles if not t63 else 400 * args.batch * R) * 256 * 4 if mul_n else roi_bytes,
                                    achieved_on_written_bytes=(float(ptiles if not t63 else 400 * args.batch * R) * 256 * 4 / (roi_ms * 1e-3) / 1e9) if (mul_n and roi_ms > 0) else None),
                "pointwise": {"kernel": "gemm_nn<PLAIN> 1x1 convs, the 14 pointwise layers (sum over the layers, one step)", "bound": "mfma+hbm",
                              "algorithmic_flop": pwf, "algorithmic_bytes": pwb, "avg_ms": pw_ms,
                              "achieved_tflops": pwf / (pw_ms * 1e-3) / 1e12 if pw_ms > 0 else 0.0,
                              "frac_of_fp32_mfma_peak": pwf / (pw_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK if pw_ms > 0 else 0.0,
                              "achieved_gbs": pwb / (pw_ms * 1e-3) / 1e9 if pw_ms > 0 else 0.0,
                              "frac_of_hbm_peak": pwb / (pw_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if pw_ms > 0 else 0.0}}
    roofline["trunk_layers"] = trunk_layer_table(args.size, args.alpha, args.batch, layer_ms, bracket_ms)
    roofline["event_bracket_ms"] = bracket_ms
    roofline["event_bracket_note"] = ("median HIP-event bracket around a 4-byte fill kernel on the compute stream: what every per-launch figure of the second pass "
                                      "(depthwise, pointwise, trunk_layers, roialign, hbm_stages) carries on top of its kernel; '*_net' fields subtract it once per "
                                      "bracket (they over-correct by the fill kernel's own ~2 us)")
    n_dw = sum(1 for t in layer_ms if t.startswith("dw"))
    n_pw = sum(1 for t in layer_ms if t.startswith("pw"))
    for key, nbr in (("depthwise", n_dw), ("pointwise", n_pw)):
        o = roofline[key]
        net_ms = max(o["avg_ms"] - nbr * bracket_ms, 1e-6)
        o["avg_ms_net"] = net_ms
        if key == "depthwise":
            o["frac_net"] = o["algorithmic_bytes"] / (net_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        else:
            o["frac_of_fp32_mfma_peak_net"] = o["algorithmic_flop"] / (net_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK
    mf = [l for l in roofline["trunk_layers"] if l["layer"].startswith("pw") and l["roof"] == "mfma" and l["ms"] > 0]
    if mf:
        fl, ms_ = sum(l["flop"] for l in mf), sum(l["ms"] for l in mf)
        roofline["pointwise"]["mfma_bound_layers"] = {"layers": [l["layer"] for l in mf], "flop": fl, "ms": ms_, "achieved_tflops": fl / (ms_ * 1e-3) / 1e12,
                                                      "frac_of_fp32_mfma_peak": fl / (ms_ * 1e-3) / 1e12 / FP32_MFMA_PEAK}
    if mul_n and woi_n:
        vbytes = float(400 * args.batch * R if t63 else ptiles) * 256 * 4          # the timed boundary launches are all on the conv2-4 tiling
        roofline["hbm_stages"] = ([hbm_obj("wino_in_kernel (input transform: activation -> V)", float(M) * 256 * 4 + vbytes, win_ms)] if win_n else []) + [
            hbm_obj("%s (layer boundary M_i -> V_{i+1} through LDS)" % ("wino63_boundary_kernel<FROM_M, TO_V>" if t63 else "wino_out_in_kernel"), 2 * vbytes, woi_ms)]
    wtxt = ("mixed F(6,3)/F(4,3) tiling of the 14x14 maps, 400 point-tiles per ROI; F(4,3)/F(2,3) elsewhere" if net.wino_tiles == "f63"
            else "mixed F(4,3)/F(2,3) tiling, 484 point-tiles per 14x14 ROI")
    res = {
        "metric": "images/sec fwd+bwd, %dx%d Shapes batch %d, at %d MI355X" % (args.size, args.size, args.batch, world),
        "value": args.batch * world * args.steps / elapsed,
        "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "step_ms": percentiles(step_ms),
        "host_step_ms": dict(percentiles(host_ms), source="host wall clock around each train_step call inside the timed region (issue + the wait for n_pos)"),
        "step_trace": {"gpu_ms": [round(v, 3) for v in step_ms], "host_ms": [round(v, 3) for v in host_ms],
                       "n_pos_total": [d[0] for d in getattr(timed_steps, "last_diag", [])],
                       "allocator_reserved_mb": [d[1] >> 20 for d in getattr(timed_steps, "last_diag", [])]},
        "config": {"winograd_tiles": net.wino_tiles, "fp32_products": net.fp32_matmul,
                   "workload": "Shapes %dx%d, batch %d/GPU, MobileNet alpha %.1f, N_BOX=%d (R=%d ROIs/img), fp32 training step (fwd+bwd+Adam%s), mask head forward on %s ROIs" % (
                       args.size, args.size, args.batch, args.alpha, cfg.N_BOX, R, "+RCCL all-reduce" if world > 1 else "", cfg.TRAIN_MASK_HEAD_ROIS),
                   "workload_detail": "Shapes %dx%d, batch %d/GPU, MobileNet alpha %.1f, N_BOX=%d (R=%d ROIs/img), fp32 training step "
                               "(fwd+bwd+Adam%s); mask head FORWARD on %s ROIs; mask-head BACKWARD behind bn1 (conv2-4, deconv, myolo_mask) on "
                               "the positive ROIs only -- exact: bn2-4 are frozen and the loss reads positives only, so the other ROIs' "
                               "gradients are structural zeros (dense-backward time in variants.dense_mask_backward); " % (
                                   args.size, args.size, args.batch, args.alpha, cfg.N_BOX, R,
                                   "+RCCL all-reduce" if world > 1 else "", cfg.TRAIN_MASK_HEAD_ROIS) + "3x3 convs: " +
                               {"auto": "fp32 Winograd for launches >= 16384 pixels (%s), direct implicit GEMM below" % wtxt,
                                "winograd": "fp32 Winograd (%s)" % wtxt, "direct": "direct implicit GEMM"}[cfg.CONV3X3_ALGO] +
                               "; Winograd multiply products: " + ("native fp32 MFMA" if net.fp32_matmul == "native" else
                               "FP32_MATMUL='bf16x6' (each fp32 product = six exact bf16 piece products, fp32 accumulation)"),
                   "global_batch": args.batch * world, "parallelism": "dp%d" % world, "final_loss": loss,
                   "n_pos_mean": npos_mean, "rois_per_image": R, "lib_options": list(args.lib_option), "net_attrs": list(args.net_attr), "comm_own_stream": bool(args.comm_own_stream), "share_gpu": bool(args.share_gpu), "forced_positives": args.force_pos,
                   "peak_hbm_allocated_gb": torch.cuda.max_memory_allocated() / 2.0 ** 30},
        "roofline": None,
    }
    res["roofline"] = roofline
    if "train_api" in extras and "train" in extras["train_api"]:
        ta = extras["train_api"]
        res["config"]["train_api_images_per_sec"] = {"MaskYOLO.train": ta["train"]["images_per_sec"],
                                                     "MaskYOLO.train_shapes_stream": ta["train_shapes_stream"]["images_per_sec"],
                                                     "Net.train_step(value)": res["value"],
                                                     "Net.train_step(same weights as the public calls)": (ta.get("reference_same_state") or {}).get("images_per_sec")}
    if variants.get("n_pos_sweep"):
        sw = variants["n_pos_sweep"]
        res["config"]["n_pos_sweep_ms"] = dict([("%.2f" % npos_mean, res["ms_per_step"])] +
                                               [(k.replace("n_pos_", ""), v["ms_per_step"]) for k, v in sw.items()])
        res["config"]["n_pos_sweep_note"] = ("ms per step at <key> positive ROIs per image: the first key is this run's own batch (random-init net), the others "
                                             "force the first k proposals of every image onto a ground-truth box; a trained Shapes net sits at 5-20")
    if world > 1:
        res["comm"] = {"backend": ("gloo (--share-gpu test mode: all ranks on one GPU)" if args.share_gpu else
                                   "RCCL via %s" % ("the C-ABI (myolo_comm_*)" if args.comm == "capi" else "torch.distributed (nccl)")),
                       "rccl_ranks_seen": ranks_seen, "bucket_allreduce_ms": bucket_ms, "weights_identical_across_ranks": weights_same,
                       "bucket_bytes": [4 * (hi - lo) for lo, hi in net.bucket_ranges],
                       "release_ms_before_step_end": release_ms,
                       "note": "buckets in flat-buffer order [backbone, YOLO blocks + conv_23, feature_map, mask conv1 + bn1, rest of the mask head]; each "
                               "launched on the comm stream as soon as backward completes it (YOLO head first, about half a step before the end)"}
    res.update(extras)
    # the dominant kernel is co-bound: at K = 256 with fp32 in / out its algorithmic bytes at the measured copy rate take as long as its products on the
    # matrix pipe -- both fractions are reported: frac_mfma = flop / peak / time, frac_composite = max(flop / peak, bytes / copy rate) / time
    cp = extras.get("hbm_copy_measured_gbs") if isinstance(extras.get("hbm_copy_measured_gbs"), dict) else None
    copy_gbs = max([v for k, v in cp.items() if k.startswith("float4") and isinstance(v, (int, float))] or [5600.0]) if cp else 5600.0
    if kms > 0:
        t_mfma, t_hbm = kflop / (peak * 1e12), kbytes / (copy_gbs * 1e9)
        res["roofline"]["frac_mfma"] = t_mfma / (kms * 1e-3)
        res["roofline"]["frac_composite"] = max(t_mfma, t_hbm) / (kms * 1e-3)
        res["roofline"]["composite_note"] = ("bound = max(flop / %.0f TFLOP/s, algorithmic bytes / %.0f GB/s [the read+write copy rate %s]) = %.3f ms against %.3f ms measured"
                                             % (peak, copy_gbs, "measured in this run" if cp else "of profiles/r3_notes.md", 1e3 * max(t_mfma, t_hbm), kms))
    if world == 1 and not args.no_extras and x6 and traffic is not None and not args.no_live_pmc:
        # HBM traffic of the dominant kernel measured IN THIS RUN: two separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE) on the same launch
        # shape in a child process (tools/pmc_traffic.py; this process keeps its memory and is idle meanwhile).  Falls back to the committed
        # profiles/r3_pmc_x6.json figure (kept as roofline.traffic_from_profiles) when rocprofv3 is missing or fails.
        try:
            cp = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), "wino63_mm", "wino_mm_x6_kernel", "wino_x6=1"],
                                stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=300)
            live = json.loads(cp.stdout.decode().strip().splitlines()[-1])
        except Exception as e:
            live = {"error": "%s: %s" % (type(e).__name__, e)}
        res["roofline"]["traffic_live"] = live
        if "traffic_bytes_per_launch_corrected" in live:
            res["roofline"]["traffic_from_profiles"] = res["roofline"]["traffic"]
            res["roofline"]["traffic"] = live["traffic_bytes_per_launch_corrected"]
            res["roofline"]["traffic_source"] = ("measured in this run: tools/pmc_traffic.py (two separate rocprofv3 --pmc passes, FETCH_SIZE x2 + WRITE_SIZE, on "
                                                 "tools/kbench.py wino63_mm at the launch shape of the timed kernel, last five of 36 launches) = %.2fx the algorithmic bytes"
                                                 % (live["traffic_bytes_per_launch_corrected"] / res["roofline"]["algorithmic_bytes"]))
    if variants:
        res["variants"] = variants
    if args.cpu_images > 0 and world == 1:
        try:
            res["cpu_baseline"] = cpu_baseline(cfg, args.cpu_images)
        except Exception as e:            # the GPU line must not be lost to a host-side problem
            res["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    else:
        res["cpu_baseline"] = None
    return res


def dw_layers(size, alpha):
    """(H_in, C, stride) of the 14 depthwise layers (model.py:68-77, 256-268)."""
    a = alpha
    chans = [int(32 * a)] + [int(f * a) for f in (64, 64, 128, 256, 256, 512, 512, 512, 512, 512, 512, 512, 1024)]
    strides = [1, 2, 1, 2, 1, 1, 2, 1, 1, 1, 1, 1, 2, 1]
    h = size // 2
    out = []
    for c, s in zip(chans, strides):
        out.append((h, c, s))
        h //= s
    return out


def dw_bytes(size, alpha, batch):
    return float(sum(batch * (h * h + (h // s) * (h // s)) * c * 4 for h, c, s in dw_layers(size, alpha)))


def trunk_layer_table(size, alpha, batch, layer_ms, bracket_ms=0.0):
    """one row per depthwise / pointwise layer: SURVEY 8(d) bytes (in + out once, + weights) and flops, the HIP-event time of its
    forward launch in the step (second pass), and the roof that bounds it: arithmetic intensity against the machine balance
    157.3 TFLOP/s / 8 TB/s = 19.7 flop/B."""
    outs = [int(f * alpha) for f in (64, 64, 128, 256, 256, 512, 512, 512, 512, 512, 512, 512, 1024, 1024)]
    rows = []
    for i, ((h, c, s), co) in enumerate(zip(dw_layers(size, alpha), outs), 1):
        ho = h // s
        for kind in ("dw", "pw"):
            if kind == "dw":
                by = batch * (h * h + ho * ho) * c * 4.0 + 9 * c * 4.0
                fl = 2.0 * 9 * batch * ho * ho * c
                shape = "%dx%dx%d s%d" % (h, h, c, s)
            else:
                m = batch * ho * ho
                by = 4.0 * (m * c + m * co + c * co)
                fl = 2.0 * m * c * co
                shape = "M=%d %d->%d" % (m, c, co)
            ms = float(layer_ms.get("%s%d_fwd" % (kind, i), 0.0))
            roof = "mfma" if fl / by > FP32_MFMA_PEAK * 1e12 / (HBM_PEAK_GBS * 1e9) else "hbm"
            gbs = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
            net_ms = max(ms - bracket_ms, 1e-6) if ms > 0 else 0.0
            frac_net = 0.0
            if net_ms > 0:
                frac_net = (fl / (net_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK) if roof == "mfma" else (by / (net_ms * 1e-3) / 1e9 / HBM_PEAK_GBS)
            rows.append({"layer": "%s%d" % (kind, i), "shape": shape, "bytes": by, "flop": fl, "ms": ms, "gbs": gbs, "tflops": tf, "roof": roof,
                         "frac": (tf / FP32_MFMA_PEAK) if roof == "mfma" else (gbs / HBM_PEAK_GBS), "ms_net": net_ms, "frac_net": frac_net})
    return rows


def pw_flops_bytes(size, alpha, batch):
    outs = [int(f * alpha) for f in (64, 64, 128, 256, 256, 512, 512, 512, 512, 512, 512, 512, 1024, 1024)]
    fl, by = 0.0, 0.0
    for (h, c, s), co in zip(dw_layers(size, alpha), outs):
        m = batch * (h // s) * (h // s)
        fl += 2.0 * m * c * co
        by += 4.0 * (m * c + m * co + c * co)
    return fl, by


def bench_infer(args):
    """BASELINE configs[3]: Rice 416x416, 5 anchors (R = 845), 28x28 mask head, inference forward (trunk + detections + ROIAlign
    + mask head on all R boxes, model.py:922-936) with the bf16 mask head.  A step = one forward of `--batch` images."""
    from myolo.config import make_config, RiceConfig
    from myolo.engine import Net
    dev = "cuda:0"
    bsz = args.batch if args.batch_given else 4
    cfg = make_config(RiceConfig, BATCH_SIZE=bsz, INFERENCE_DTYPE="bf16")
    net = Net(cfg, device=dev, seed=0)
    for kv in args.net_attr:
        name, _, val = kv.partition("=")
        assert hasattr(net, name), "unknown engine attribute %s" % name
        setattr(net, name, type(getattr(net, name))(int(val)))
    x = torch.rand(bsz, 416, 416, 3, device=dev)
    # the timed region runs the forward the way MaskYOLO.detect() does (cfg.INFERENCE_HIP_GRAPH): replayed from a captured hipGraph,
    # one graph launch per step on the host instead of ~150 kernel launches
    run = net.predict_graphed if cfg.INFERENCE_HIP_GRAPH else net.predict
    import gc
    gc_was = gc.isenabled()
    gc.collect()                                 # the cyclic collector is parked for the two timed regions below (as in bench_train)
    gc.disable()
    for _ in range(max(2, args.warmup)):
        run(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run(x)
    torch.cuda.synchronize()
    el1 = time.perf_counter() - t0
    # throughput form (Net.predict_stream): `--in-flight` forwards at once, each on its own stream / hipGraph / scratch -- the launch-bound
    # fp32 trunk of one batch runs underneath the matrix-pipe-bound mask head of the other.  Same kernels, bit-identical results
    # (tests/test_gpu_step.py::test_predict_stream_matches_predict); every batch is a full forward of `--batch` images.
    nfl = max(1, args.in_flight) if cfg.INFERENCE_HIP_GRAPH else 1
    el = el1
    if nfl > 1:
        xs = [x] + [torch.rand(bsz, 416, 416, 3, device=dev) for _ in range(nfl - 1)]
        def feed(n):
            for i in range(n):
                yield xs[i % nfl]
        for _ in net.predict_stream(feed(max(2, args.warmup) * nfl), in_flight=nfl):
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in net.predict_stream(feed(args.steps), in_flight=nfl):
            pass
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    # per-kernel timings (HIP events around eager launches) in a second pass, outside the timed region
    net.timed_tags = {"mask_conv3x3_fwd", "roialign_fwd", "mask_deconv_fwd"}
    net.timings = {}
    for _ in range(5):
        net.predict(x)
    conv_ms, conv_n = net.kernel_ms("mask_conv3x3_fwd")
    roi_ms, _ = net.kernel_ms("roialign_fwd")
    dec_ms, _ = net.kernel_ms("mask_deconv_fwd")
    R = cfg.TRAIN_ROIS_PER_IMAGE
    M = bsz * R * 14 * 14
    flop = 2.0 * M * 9 * 256 * 256
    ach = flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    traffic, traffic_src = None, None
    try:        # HBM bytes per launch from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE at M = 921984), scaled to this launch's rows
        pm = json.load(open(os.path.join(ROOT, "profiles", "r2_pmc_bf16.json")))["conv3x3_bf16_fwd_default"]
        traffic = pm["hbm_traffic_bytes_corrected"] * M / 921984.0
        traffic_src = ("profiles/r2_pmc_bf16.json (separate rocprofv3 --pmc passes on tools/kbench.py conv3x3_bf16_fwd, M = 921984: FETCH_SIZE x2 + "
                       "WRITE_SIZE), scaled by M; not re-measured in this run")
    except Exception:
        pass
    # the public call behind that throughput: MaskYOLO.detect_many on uint8 images (upload, the same graphs, then detect()'s selection / unmolding per image on
    # the host + GPU) -- reported beside the forward-only figures, never as `value`
    api = None
    try:
        from myolo.model import MaskYOLO
        m = MaskYOLO(mode="inference", config=cfg, seed=0)
        rng = np.random.default_rng(0)
        imgs = [(rng.random((416, 416, 3)) * 255).astype(np.uint8) for _ in range(12 * bsz)]
        m.detect_many(imgs[:4 * bsz], in_flight=nfl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        outs = m.detect_many(imgs, in_flight=nfl)
        torch.cuda.synchronize()
        ela = time.perf_counter() - t0
        api = {"images_per_sec": len(imgs) / ela, "images": len(imgs), "in_flight": nfl,
               "detections_kept_per_image": float(np.mean([len(o["class_ids"]) for o in outs])),
               "note": "MaskYOLO.detect_many (uint8 images in, detect() result dicts out): host normalisation + upload + forward + selection + unmolding"}
        del m
    except Exception as e:          # never lets the line fail
        api = {"error": repr(e)[:200]}
    res = {"metric": "images/sec inference, Rice 416x416, 5 anchors, 28x28 mask head, bf16 mask head, 1 MI355X",
           "value": bsz * args.steps / el, "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": max(2, args.warmup),
           "ms_per_step": 1e3 * el / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "bf16", "data": "synthetic",
           "detect_many": api,
           "one_in_flight": {"value": bsz * args.steps / el1, "ms_per_step": 1e3 * el1 / args.steps,
                             "note": "the same forwards strictly one after the other on one stream (Net.predict_graphed), as rounds 1-2 reported"},
           "config": {"workload": "Rice 416x416 inference forward, batch %d, N_BOX=5 (R=845 boxes/img, all through the mask head as the "
                                  "reference graph does, model.py:926-931), fp32 trunk (frozen BatchNorm + ReLU6 folded into the depthwise / pointwise conv epilogues) + bf16 ROIAlign / 3x3 convs / deconv with fp32 accumulation; the timed forwards are replays of captured hipGraphs "
                                  "(cfg.INFERENCE_HIP_GRAPH, as detect() runs them), %d batch(es) in flight (Net.predict_stream: one stream, graph and scratch per lane; "
                                  "ms_per_step = wall time / batches); kernel timings from a separate eager pass" % (bsz, nfl),
                      "global_batch": bsz, "in_flight": nfl, "parallelism": "dp1"},
           "roofline": {"kernel": "conv3_bf16_256 (mask-head 3x3 conv as an implicit GEMM with the activation block resident in LDS across the nine taps, "
                                  "bf16 operands, fp32 accumulate, M=%d K=2304 N=256)" % M,
                        "bound": "mfma", "achieved": ach, "peak": BF16_MFMA_PEAK, "unit": "TFLOP/s", "frac": ach / BF16_MFMA_PEAK,
                        "traffic": traffic, "traffic_source": traffic_src, "algorithmic_flop": flop, "algorithmic_bytes": 2.0 * M * 256 * 2 + 9 * 256 * 256 * 2,
                        "launches_timed": conv_n, "avg_launch_ms": conv_ms,
                        "deconv_mask": {"kernel": "gemm_bf16_256<PLAIN, DECONV_MASK, LOOPN, MEP, FIN> (2x2/s2 transposed conv + ReLU + 1x1 mask conv on the matrix pipe + sigmoid in ONE launch, "
                                                  "the 28x28x256 tensor never written)", "bound": "mfma", "avg_ms": dec_ms,
                                        "algorithmic_flop": 2.0 * M * 256 * 4 * 256,
                                        "achieved": 2.0 * M * 256 * 4 * 256 / (dec_ms * 1e-3) / 1e12 if dec_ms > 0 else 0.0, "peak": BF16_MFMA_PEAK, "unit": "TFLOP/s"},
                        "roialign": {"kernel": "crop_fwd_bf16_walk_kernel (a feature-map column fetched once per output row)", "bound": "hbm", "avg_ms": roi_ms,
                                     "algorithmic_bytes": M * 256 * 2.0 + bsz * 52 * 52 * 256 * 4.0,
                                     "achieved": (M * 256 * 2.0 + bsz * 52 * 52 * 256 * 4.0) / (roi_ms * 1e-3) / 1e9 if roi_ms > 0 else 0.0,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s"}}}
    if args.cpu_images > 0:
        try:
            res["cpu_baseline"] = cpu_baseline(cfg, 1, infer=True, budget_s=60.0)
        except Exception as e:
            res["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 32; 4 for --config rice416-bf16)")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--nbox", type=int, default=3, choices=[3, 5],
                    help="3 = self-consistent Shapes head (R=147, primary); 5 = repository-HEAD head (R=245)")
    ap.add_argument("--in-flight", type=int, default=3, help="--config rice416-bf16: inference batches in flight (Net.predict_stream); 1 = strictly serial")
    ap.add_argument("--config", choices=["shapes224-train", "rice416-bf16"], default="shapes224-train",
                    help="shapes224-train = BASELINE configs[1] (the metric); rice416-bf16 = configs[3] inference throughput")
    ap.add_argument("--cpu-images", type=int, default=32, help="images in the CPU-baseline step (0 = skip; halved while host memory is short)")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--device-warm", type=int, default=8, help="untimed forward+backward passes (no optimizer update) in front of the warm-up steps: allocator pre-sizing + a cold GPU's first load transient")
    ap.add_argument("--mask-head-rois", choices=["all", "positives"], default="all",
                    help="cfg.TRAIN_MASK_HEAD_ROIS of the run that produces `value` (default: all ROIs, as the reference graph)")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not measure the dominant kernel's HBM traffic with rocprofv3 --pmc in a child process (default line only)")
    ap.add_argument("--no-extras", action="store_true", help="skip the extra objects of the default line (HBM copy bandwidth, comm probe, N_BOX=5 and Rice-416 bf16 runs)")
    ap.add_argument("--no-variant", action="store_true", help="skip the extra timed runs (dense backward, positives-only forward, n_pos sweep)")
    ap.add_argument("--conv3x3", choices=["auto", "direct", "winograd"], default="auto", help="cfg.CONV3X3_ALGO")
    ap.add_argument("--comm", choices=["torch", "capi"], default="torch",
                    help="N>1: gradient all-reduce through torch.distributed (nccl = RCCL) or through the library's own myolo_comm_* entry points")
    ap.add_argument("--fp32-matmul", choices=["native", "bf16x6"], default="bf16x6", help="cfg.FP32_MATMUL (how the Winograd multiply forms its fp32 products)")
    ap.add_argument("--wino-tiles", choices=["f43", "f63"], default=None, help="cfg.WINOGRAD_TILES (default: the config's)")
    ap.add_argument("--force-pos", type=int, default=0, metavar="K",
                    help="replace the first K proposals of every image by a ground-truth box for the WHOLE run (the n_pos sweep's hook): the step "
                         "at the positive counts a trained net produces; recorded in config.forced_positives")
    ap.add_argument("--comm-own-stream", action="store_true", help="ablation: the gradient all-reduces on a fresh HIP stream of their own (rounds 1-3) instead of the engine's copy stream")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST ONLY: every rank uses cuda:0 and the ranks rendezvous over gloo (RCCL needs one GPU per rank) -- exercises the whole "
                         "N > 1 code path of this script on a one-GPU box; recorded in config.share_gpu, never a throughput claim")
    ap.add_argument("--net-attr", action="append", default=[], metavar="NAME=VALUE",
                    help="engine (myolo.engine.Net) scheduling switches for this run, e.g. overlap_conv1_wgrad=0; recorded in config.net_attrs")
    ap.add_argument("--lib-option", action="append", default=[], metavar="NAME=VALUE",
                    help="myolo_set_option switches for this run (kernel A/B comparisons, e.g. wino_x6=1); recorded in config.lib_options")
    ap.add_argument("--detail-name", default="bench_detail.json", help="file name of the full result object (written at the repo root and in gpurun_out/)")
    args = ap.parse_args()
    args.batch_given = args.batch is not None
    if args.batch is None:
        args.batch = 32

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args.gpus)

    # stdout carries exactly ONE line (the JSON): everything else any library prints to file descriptor 1 (RCCL's version banner at
    # communicator creation, for one) goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from myolo import dist as mdist
    rank, world, local = mdist.init_from_env("gloo" if args.share_gpu else None)
    if args.share_gpu:
        local = 0
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torchrun --nproc-per-node %d, or plain `python bench.py --gpus %d`)"
                         % (args.gpus, world, args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    if args.lib_option:
        from myolo import _ext as X
        for kv in args.lib_option:
            name, _, val = kv.partition("=")
            X.set_option(name, int(val))
    if args.config == "rice416-bf16":
        res = bench_infer(args) if rank == 0 else None
    else:
        res = bench_train(args, rank, world, local)
        if world == 1 and not args.no_extras and args.nbox == 3 and args.size == 224:
            # ONE driver command, all single-GPU configurations: after the timed region of the headline,
            # SURVEY 8(d)'s secondary head (N_BOX=5, R=245) and BASELINE configs[3] (Rice 416 bf16 inference), each its own object
            import copy
            torch.cuda.empty_cache()
            try:
                a5 = copy.copy(args)
                a5.nbox, a5.no_variant, a5.no_extras, a5.cpu_images, a5.warmup = 5, True, True, 0, max(5, args.warmup)
                r5 = bench_train(a5, rank, world, local)
                res["secondary_nbox5"] = {k: r5[k] for k in ("metric", "value", "unit", "ms_per_step", "step_ms")}
                res["secondary_nbox5"].update(workload="the same step with the repository-HEAD head: N_BOX=5, config.py:28 anchors, R=245 ROIs/img (SURVEY 8(d) 'report both')",
                                              n_pos_mean=r5["config"]["n_pos_mean"], dominant_kernel_frac=r5["roofline"]["frac"],
                                              dominant_kernel_ms=r5["roofline"]["avg_launch_ms"])
            except Exception as e:
                res["secondary_nbox5"] = {"error": "%s: %s" % (type(e).__name__, e)}
            torch.cuda.empty_cache()
            try:
                ai = copy.copy(args)
                ai.batch, ai.batch_given, ai.cpu_images, ai.steps, ai.warmup = 4, True, 0, max(20, args.steps), 3
                ri = bench_infer(ai)
                res["inference_rice416_bf16"] = {k: ri[k] for k in ("metric", "value", "unit", "ms_per_step", "dtype", "one_in_flight", "detect_many", "roofline")}
                res["inference_rice416_bf16"]["workload"] = ri["config"]["workload"]
            except Exception as e:
                res["inference_rice416_bf16"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if rank == 0:
        sys.stdout.flush()
        write_detail(res, args)                       # the full object first: nothing below can lose it
        obj = compact_line(res)
        line = json.dumps(obj)
        # the driver parses ONE line: if it ever outgrows the cap (a long option list, an error string), optional keys go before the result does
        for path in (("config", "lib_options"), ("config", "net_attrs"), ("comm", "note"), ("cpu_baseline", "sample"), ("comm",), ("roofline", "note")):
            if len(line) < MAX_LINE_BYTES:
                break
            d = obj
            for k in path[:-1]:
                d = d.get(k) if isinstance(d, dict) else None
            if isinstance(d, dict) and path[-1] in d:
                d[path[-1]] = "(dropped: line over %d bytes; see bench_detail.json)" % MAX_LINE_BYTES
                line = json.dumps(obj)
        os.write(json_fd, (line + "\n").encode())
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
