#!/usr/bin/env python
"""bench.py -- Mask-YOLO training-step throughput on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

A step = one full training step of the hot path on one synthetic Shapes batch per GPU:
forward (backbone, YOLO head, decode, mask targets, ROIAlign, mask head, both losses), backward of all of
it, gradient all-reduce (N>1) and the Adam update.  Inputs are resident in HBM before the timed region.
Rank 0 prints ONE JSON line (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (mask-head 3x3 implicit-GEMM, fp32 MFMA), timed live with HIP events
  cpu_baseline -- the CPU restatement (oracle/torch_ref.py, torch-CPU, all host cores) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "mask-yolo_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np          # noqa: E402
import torch                # noqa: E402
import torch.distributed as dist   # noqa: E402


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


def cpu_baseline(cfg, sample_images, seed=0, budget_s=25.0):
    """Time the CPU restatement (torch-CPU fp32 + autograd, oracle/torch_ref.py) on a bounded sample of the
    same workload.  oneDNN scales poorly past a few dozen threads at this batch size, so the thread count is
    min(usable cores, 32); that number is what 'cores' reports."""
    from myolo.config import make_config
    from myolo.shapes import make_shapes_samples
    from myolo.myolo_utils import BatchGenerator
    from oracle import np_model
    from oracle.torch_ref import TorchRef
    avail = usable_cores()
    threads = max(1, min(avail, 32))
    torch.set_num_threads(threads)

    def build(n):
        ccfg = make_config(type(cfg).__mro__[1], IMAGE_SHAPE=list(cfg.IMAGE_SHAPE), ALPHA=cfg.ALPHA, BATCH_SIZE=n,
                           N_BOX=cfg.N_BOX, ANCHORS=list(cfg.ANCHORS))
        samples = make_shapes_samples(n, ccfg)
        batch, _ = BatchGenerator(samples, ccfg, 'training', shuffle=False, norm=True)[0]
        return TorchRef(np_model.init_params(ccfg, seed=seed), ccfg, torch.float32), batch

    ref, batch = build(1)
    t0 = time.time()
    ref.train_step(batch)
    ref.adam({}, 1, 1e-3)
    t_probe = time.time() - t0                     # 1-image probe (includes first-touch / oneDNN primitive creation)
    n = int(max(1, min(sample_images, budget_s / max(t_probe, 1e-3))))
    if n > 1:
        ref, batch = build(n)
        ref.train_step(batch)                      # warm-up at the timed batch size
    state = {}
    t0 = time.time()
    ref.train_step(batch)
    ref.adam(state, 1, 1e-3)
    t = time.time() - t0
    return dict(value=n / t, unit="images/sec", cores=threads, kind="port",
                sample="1 timed training step (fwd+bwd+Adam) of the torch-CPU fp32 restatement (oracle/torch_ref.py) on %d Shapes "
                       "%dx%d image(s), N_BOX=%d, after a warm-up step; %d threads of %d usable host cores (os.cpu_count()=%d); "
                       "%.1f s timed, 1-image probe %.1f s"
                       % (n, cfg.IMAGE_SHAPE[0], cfg.IMAGE_SHAPE[1], cfg.N_BOX, threads, avail, os.cpu_count() or 0, t, t_probe))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--size", type=int, default=224)
    ap.add_argument("--alpha", type=float, default=1.0)
    ap.add_argument("--nbox", type=int, default=3, choices=[3, 5],
                    help="3 = self-consistent Shapes head (R=147, primary); 5 = repository-HEAD head (R=245)")
    ap.add_argument("--cpu-images", type=int, default=8, help="images in the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--mask-head-rois", choices=["all", "positives"], default="all",
                    help="cfg.TRAIN_MASK_HEAD_ROIS of the run that produces `value` (default: all ROIs, as the reference graph)")
    ap.add_argument("--no-variant", action="store_true", help="skip the extra timed run of the other TRAIN_MASK_HEAD_ROIS setting")
    ap.add_argument("--conv3x3", choices=["auto", "direct", "winograd"], default="auto", help="cfg.CONV3X3_ALGO")
    args = ap.parse_args()

    from myolo import dist as mdist
    from myolo.config import make_config, ShapesConfig, ShapesHeadConfig
    from myolo.model import MaskYOLO

    rank, world, local = mdist.init_from_env()
    assert world == args.gpus, "launch with torchrun --nproc-per-node %d (WORLD_SIZE=%d)" % (args.gpus, world)
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    torch.cuda.set_device(local)
    dev = "cuda:%d" % local
    base = ShapesConfig if args.nbox == 3 else ShapesHeadConfig
    cfg = make_config(base, IMAGE_SHAPE=[args.size, args.size, 3], ALPHA=args.alpha, BATCH_SIZE=args.batch,
                      TRAIN_MASK_HEAD_ROIS=args.mask_head_rois, CONV3X3_ALGO=args.conv3x3)
    model = MaskYOLO(mode="training", config=cfg, device=dev, seed=0)      # same seed -> same weights on every rank
    net = model.net
    reducer = mdist.GradReducer(net.flat_g, [net.bucket_ranges[i] for i in (0, 1, 2)])
    reducer.attach(net)

    nb = 2
    host = make_batches(cfg, rank, world, args.batch, nb)
    dbs = [net.to_device_batch(b) for b in host]
    torch.cuda.synchronize()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        net.train_step(dbs[i % nb], args.lr)
    barrier()
    net.timed_tags = {"mask_conv3x3_fwd", "roialign_fwd", "wino_multiply", "wino_in", "wino_out_in"}
    net.timings = {}
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = net.train_step(dbs[i % nb], args.lr)
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss = float(out["yolo_terms"][0]) + float(out["mask_terms"][0])
    assert np.isfinite(loss), "non-finite loss in the timed region"
    conv_ms, conv_n = net.kernel_ms("mask_conv3x3_fwd")
    mul_ms, mul_n = net.kernel_ms("wino_multiply")
    win_ms, win_n = net.kernel_ms("wino_in")
    woi_ms, woi_n = net.kernel_ms("wino_out_in")
    roi_ms, _ = net.kernel_ms("roialign_fwd")

    # the same K steps with the other TRAIN_MASK_HEAD_ROIS setting (reported beside `value`, never as `value`)
    variant = None
    if not args.no_variant:
        net.timed_tags = set()
        net.sparse_mask_fwd = not net.sparse_mask_fwd
        for i in range(2):
            net.train_step(dbs[i % nb], args.lr)
        barrier()
        t1 = time.perf_counter()
        for i in range(args.steps):
            out2 = net.train_step(dbs[i % nb], args.lr)
        barrier()
        el2 = time.perf_counter() - t1
        assert np.isfinite(float(out2["yolo_terms"][0]) + float(out2["mask_terms"][0])), "non-finite loss in the variant run"
        if world > 1:
            t = torch.tensor([el2], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el2 = float(t.item())
        variant = {"TRAIN_MASK_HEAD_ROIS": "positives" if net.sparse_mask_fwd else "all",
                   "value": args.batch * world * args.steps / el2, "unit": "images/sec", "ms_per_step": 1e3 * el2 / args.steps,
                   "note": "same loss, gradients, weights and BN state as the all-ROI forward (tests/test_gpu_step.py::"
                           "test_positives_only_forward_equals_full_forward); conv2-4/deconv/myolo_mask forward run on the positive "
                           "ROIs only, whose outputs are the only ones the training graph reads"}
        net.sparse_mask_fwd = not net.sparse_mask_fwd

    if rank == 0:
        R = cfg.TRAIN_ROIS_PER_IMAGE
        ps = cfg.MASK_POOL_SIZE
        M = args.batch * R * ps * ps
        flop = 2.0 * M * (9 * 256) * 256                      # algorithmic FLOPs of one mask-head 3x3 conv as a direct convolution
        wino = mul_n > 0
        if wino:
            # CONV3X3_ALGO auto/winograd: the dominant kernel is the batched GEMM of the 36 Winograd points,
            # M_t = tiles, K = N = 256; its algorithmic FLOPs are what the Winograd form needs, not the direct conv's
            tiles_w = args.batch * R * ((ps + 3) // 4) ** 2
            kflop = 2.0 * 36 * tiles_w * 256 * 256
            kms, kn = mul_ms, mul_n
            kname = "gemm_nn_fast<PLAIN> x36 batched (Winograd F(4x4,3x3) multiply stage of the mask-head 3x3 convs, M=%d K=256 N=256 per point)" % tiles_w
            kbytes = 36.0 * tiles_w * (256 + 256) * 4 + 36 * 256 * 256 * 4
        else:
            kflop, kms, kn = flop, conv_ms, conv_n
            kname = "gemm_nn_fast<CONV3> (mask-head 3x3 conv fwd, M=%d K=2304 N=256)" % M
            kbytes = 2.0 * M * 256 * 4 + 9 * 256 * 256 * 4
        achieved = kflop / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
        roi_bytes = args.batch * R * ps * ps * 256 * 4 + args.batch * (args.size // 8) ** 2 * 256 * 4
        roi_kernel = "crop_fwd_kernel (ROIAlign fwd)"
        if wino:          # ROIAlign is fused into conv1's input transform: it writes V (36 planes of tiles) instead of the crops
            roi_bytes = 36.0 * args.batch * R * ((ps + 3) // 4) ** 2 * 256 * 4 + args.batch * (args.size // 8) ** 2 * 256 * 4
            roi_kernel = "wino_in_crop_kernel (ROIAlign fused into conv1's Winograd input transform: feature map -> V)"
        traffic = None      # HBM-side bytes per launch of the dominant kernel, from the separate --pmc passes (tools/collect_profiles.sh)
        try:
            pj = json.load(open(os.path.join(ROOT, "profiles", "r1_pmc_wino_multiply.json" if wino else "r1_pmc_conv3x3_fwd.json")))
            if args.batch * R == 32 * 147:        # the counters were collected at exactly this shape
                traffic = pj["traffic_bytes_per_launch_corrected"]
        except Exception:
            pass
        res = {
            "metric": "images/sec fwd+bwd, %dx%d Shapes batch %d, at %d MI355X" % (args.size, args.size, args.batch, world),
            "value": args.batch * world * args.steps / elapsed,
            "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Shapes %dx%d, batch %d/GPU, MobileNet alpha %.1f, N_BOX=%d (R=%d ROIs/img), fp32 training step "
                                   "(fwd+bwd+Adam%s), mask head forward on %s ROIs" % (
                                       args.size, args.size, args.batch, args.alpha, cfg.N_BOX, R,
                                       "+RCCL all-reduce" if world > 1 else "", cfg.TRAIN_MASK_HEAD_ROIS) + ", 3x3 convs: " + {"auto": "fp32 Winograd F(4x4,3x3) for launches >= 16384 pixels, direct implicit GEMM below",
                                                          "winograd": "fp32 Winograd F(4x4,3x3)", "direct": "direct implicit GEMM"}[cfg.CONV3X3_ALGO],
                       "global_batch": args.batch * world, "parallelism": "dp%d" % world, "final_loss": loss,
                       "peak_hbm_allocated_gb": torch.cuda.max_memory_allocated() / 2.0 ** 30},
            "roofline": {"kernel": kname,
                         "bound": "mfma", "achieved": achieved, "peak": 157.3, "unit": "TFLOP/s",
                         "frac": achieved / 157.3, "traffic": traffic,
                         "algorithmic_bytes": kbytes, "algorithmic_flop": kflop, "launches_timed": kn, "avg_launch_ms": kms,
                         "conv_op": {"algo": "winograd_f4x4_3x3" if wino else "direct", "avg_ms": conv_ms, "ops_timed": conv_n,
                                     "direct_conv_flop": flop,
                                     "direct_equivalent_tflops": flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0},
                         "secondary": {"kernel": roi_kernel, "bound": "hbm", "algorithmic_bytes": roi_bytes,
                                       "achieved": roi_bytes / (roi_ms * 1e-3) / 1e9 if roi_ms > 0 else 0.0,
                                       "peak": 8000.0, "unit": "GB/s",
                                       "frac": (roi_bytes / (roi_ms * 1e-3) / 1e9 / 8000.0) if roi_ms > 0 else 0.0,
                                       "avg_launch_ms": roi_ms}},
        }
        if wino and woi_n:
            # the HBM-bound stages of the Winograd op (18 % of the step), against 8 TB/s: algorithmic bytes / measured time
            vbytes = 36.0 * tiles_w * 256 * 4
            xbytes = float(M) * 256 * 4
            res["roofline"]["hbm_stages"] = ([
                {"kernel": "wino_in_kernel (input transform: activation -> V)", "bound": "hbm", "algorithmic_bytes": xbytes + vbytes,
                 "avg_launch_ms": win_ms, "achieved": (xbytes + vbytes) / (win_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                 "frac": (xbytes + vbytes) / (win_ms * 1e-3) / 1e9 / 8000.0}] if win_n else []) + [
                {"kernel": "wino_out_in_kernel (layer boundary M_i -> V_{i+1} through LDS)", "bound": "hbm", "algorithmic_bytes": 2 * vbytes,
                 "avg_launch_ms": woi_ms, "achieved": 2 * vbytes / (woi_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                 "frac": 2 * vbytes / (woi_ms * 1e-3) / 1e9 / 8000.0}]
        if variant is not None:
            res["variant"] = variant
        if args.cpu_images > 0 and world == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(cfg, args.cpu_images)
            except Exception as e:            # the GPU line must not be lost to a host-side problem
                res["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
        elif args.cpu_images > 0:
            res["cpu_baseline"] = None
        print(json.dumps(res))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
