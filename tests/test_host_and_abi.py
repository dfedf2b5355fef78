"""CPU tests of the host logic and of the C-ABI boundary (no GPU compute):
config finalisation, Shapes generator, BatchGenerator vs the oracle's literal loop restatement,
post-processing helpers, and that libmyolo_hip.so loads and exports every symbol include/myolo_hip.h declares."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import np_ops as O
from myolo import myolo_utils as mutils
from myolo.config import Config, ShapesConfig, ShapesHeadConfig, RiceConfig, make_config
from myolo.shapes import ShapesDataset, make_shapes_samples

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---------------------------------------------------------------- config (config.py:15-257)
def test_config_defaults_match_reference():
    """values read from tests/golden/ref_config.json = the reference's myolo/config.py imported (make_ref_pipeline_fixtures.py), not typed here;
    the attribute-by-attribute comparison is tests/test_ref_host_pins.py::test_config_equals_the_imported_reference_config."""
    import json
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_config.json")))["Config_instance"]
    c = Config()
    for k in ("N_BOX", "GRID_H", "GRID_W", "TRUE_BOX_BUFFER", "MAX_GT_INSTANCES", "OBJECT_SCALE", "NO_OBJECT_SCALE", "COORD_SCALE", "CLASS_SCALE",
              "WARM_UP_BATCHES", "ANCHORS", "MASK_POOL_SIZE", "MASK_SHAPE", "TOP_FEATURE_MAP_DEPTH", "LEARNING_RATE", "LEARNING_MOMENTUM",
              "TRAIN_ROIS_PER_IMAGE", "IMAGE_SHAPE", "BATCH_SIZE", "NUM_CLASSES", "LABELS", "SECOND_PHASE_YOLO_DEPTH", "GRADIENT_CLIP_NORM"):
        assert getattr(c, k) == ref[k], k
    assert c.TRAIN_ROIS_PER_IMAGE == 245 and c.IMAGE_SHAPE == [224, 224, 3]


def test_finalize_propagates_overrides():
    c = ShapesConfig()
    assert c.N_BOX == 3 and c.TRAIN_ROIS_PER_IMAGE == 147 and len(c.CLASS_WEIGHTS) == 4
    c2 = make_config(ShapesConfig, IMAGE_SHAPE=[128, 128, 3], ALPHA=0.5)
    assert (c2.GRID_H, c2.GRID_W, c2.TRAIN_ROIS_PER_IMAGE, c2.SECOND_PHASE_YOLO_DEPTH) == (4, 4, 48, 256)
    assert ShapesHeadConfig().TRAIN_ROIS_PER_IMAGE == 245
    r = RiceConfig()
    assert r.GRID_W == 13 and r.TRAIN_ROIS_PER_IMAGE == 845
    with pytest.raises(Exception):
        make_config(ShapesConfig, IMAGE_SHAPE=[100, 100, 3])            # model.py:792-794
    with pytest.raises(AssertionError):
        make_config(ShapesConfig, N_BOX=5)                              # 3 anchors but N_BOX 5


# ---------------------------------------------------------------- Shapes dataset (dataset_shapes.py:53-180)
def test_shapes_dataset_is_deterministic_and_well_formed():
    cfg = ShapesConfig()
    a = make_shapes_samples(3, cfg, start_index=10)
    b = make_shapes_samples(3, cfg, start_index=10)
    for (i1, c1, b1, m1), (i2, c2, b2, m2) in zip(a, b):
        assert np.array_equal(i1, i2) and np.array_equal(b1, b2) and np.array_equal(m1, m2)
        assert i1.dtype == np.uint8 and i1.shape == (224, 224, 3) and m1.dtype == bool
        assert 1 <= len(c1) <= 4 and set(c1) <= {1, 2, 3}
        assert np.array_equal(b1, O.extract_bboxes(m1)) and np.array_equal(b1, mutils.extract_bboxes(m1))
        assert m1.sum(axis=2).max() <= 1                                 # occlusion handling: masks are disjoint
    assert not np.array_equal(a[0][0], make_shapes_samples(1, cfg, start_index=11)[0][0])


def test_draw_shape_primitives():
    img = np.zeros((40, 40, 1), np.uint8)
    sq = ShapesDataset.draw_shape(img, "square", (20, 20, 5), 1)[..., 0]
    assert sq.sum() == 11 * 11 and sq[15, 15] == 1 and sq[14, 15] == 0
    ci = ShapesDataset.draw_shape(img, "circle", (20, 20, 5), 1)[..., 0]
    assert ci[20, 25] == 1 and ci[24, 24] == 0 and ci.sum() == 81
    tr = ShapesDataset.draw_shape(img, "triangle", (20, 20, 6), 1)[..., 0]
    assert tr[14, 20] == 1 and tr[26, 14] == 1 and tr[14, 14] == 0


# ---------------------------------------------------------------- BatchGenerator (myolo_utils.py:689-860)
@pytest.mark.parametrize("base,size", [(ShapesConfig, 224), (ShapesHeadConfig, 224), (ShapesConfig, 128)])
def test_batch_generator_matches_oracle_encoding(base, size):
    cfg = make_config(base, IMAGE_SHAPE=[size, size, 3], BATCH_SIZE=6)
    samples = make_shapes_samples(6, cfg, start_index=3)
    inputs, outputs = mutils.BatchGenerator(samples, cfg, "training", shuffle=False, norm=True)[0]
    ref = O.encode_batch(samples, cfg)
    assert outputs == [] and len(inputs) == 6
    for got, exp in zip(inputs, ref):
        assert got.dtype == exp.dtype and np.array_equal(got, exp)
    images, true_boxes, y_true, gt_ids, gt_boxes, gt_masks = inputs
    assert images.dtype == np.float32 and images.max() <= 1.0
    assert true_boxes.shape == (6, 1, 1, 1, cfg.TRUE_BOX_BUFFER, 4)
    assert y_true[..., 4].sum() == sum(len(s[1]) for s in samples)      # every box centre lies inside the grid
    assert np.all(y_true[..., 5:].sum(-1) == y_true[..., 4])            # one-hot class where an object is
    yolo_inputs, _ = mutils.BatchGenerator(samples, cfg, "yolo", shuffle=False, norm=True)[0]
    assert len(yolo_inputs) == 3


def test_batch_generator_last_partial_batch_and_len():
    cfg = make_config(ShapesConfig, BATCH_SIZE=4)
    gen = mutils.BatchGenerator(make_shapes_samples(6, cfg), cfg, "training", shuffle=False, norm=True)
    assert len(gen) == 2 and gen.size() == 6 and gen.num_classes() == 4
    assert gen[1][0][0].shape[0] == 4                                    # reference re-uses the tail (myolo_utils.py:731-733)


# ---------------------------------------------------------------- post-processing ("next" rows)
def test_bbox_iou_and_nmb():
    a, b = mutils.BoundBox(0, 0, 2, 2), mutils.BoundBox(1, 1, 3, 3)
    assert abs(mutils.bbox_iou(a, b) - 1 / 7) < 1e-12 and mutils.bbox_iou(a, mutils.BoundBox(5, 5, 6, 6)) == 0
    boxes = np.array([[0, 0, .5, .5], [0.01, 0, .5, .5], [.6, .6, .9, .9]])
    keep = mutils.NMB(boxes, np.array([1, 1, 1]), np.array([7, 8, 9]), [224, 224, 3], nms_threshold=0.7)
    assert list(keep) == [7, 9]
    keep = mutils.NMB(boxes, np.array([1, 2, 1]), np.array([7, 8, 9]), [224, 224, 3], nms_threshold=0.7)
    assert list(keep) == [7, 8, 9]                                       # different classes are not suppressed


def test_decode_one_yolo_output_and_unmold_mask():
    G, A, C = 7, 3, 4
    net = np.full((G, G, A, 5 + C), -10.0)
    net[3, 2, 1, :4] = 0
    net[3, 2, 1, 4] = 10
    net[3, 2, 1, 5 + 2] = 10
    boxes = mutils.decode_one_yolo_output(net, ShapesConfig.ANCHORS, C, obj_threshold=0.35, nms_threshold=0.3)
    assert len(boxes) == 1 and boxes[0].get_label() == 2
    assert abs((boxes[0].xmin + boxes[0].xmax) / 2 - 2.5 / 7) < 1e-9
    m = mutils.unmold_mask(np.ones((28, 28)), [0.25, 0.5, 0.75, 1.0], [224, 224, 3])
    assert m.shape == (224, 224) and m[112:224, 56:168].all() and m.sum() == 112 * 112


# ---------------------------------------------------------------- C-ABI boundary
def _header_functions(which=("myolo_hip.h", "myolo_hip_internal.h")):
    out = []
    for h in which:
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        out += re.findall(r"\b(?:int|size_t|const char\*)\s+(myolo_\w+)\s*\(([^;]*?)\)\s*;", txt, flags=re.S)
    return out


def test_library_exports_every_declared_symbol():
    from myolo import _ext
    assert os.path.exists(_ext.LIB_PATH), "run `python __graft_entry__.py` (the driver's build()) first"
    lib = _ext.load()
    decl = _header_functions()
    names = [n for n, _ in decl]
    assert len(names) >= 35 and len(set(names)) == len(names)
    for n in names:
        assert hasattr(lib, n), "symbol %s declared in include/myolo_hip*.h is not exported" % n
    assert set(_ext.exported_symbols()) == set(names), set(_ext.exported_symbols()) ^ set(names)
    assert lib.myolo_version() >= 100
    assert isinstance(lib.myolo_last_error_string(), bytes)
    # the operator API (what INTEGRATION.md documents) stays small and free of stage-level names
    ops = [n for n, _ in _header_functions(("myolo_hip.h",))]
    assert len(ops) <= 70, len(ops)
    assert not [n for n in ops if "lazybn" in n or "_transform" in n or "rowsparse" in n or "_from_" in n], ops


def test_ctypes_signatures_match_header_arity():
    from myolo import _ext
    for name, args in _header_functions():
        if name in _ext.SIGS:
            nargs = 0 if args.strip() in ("", "void") else args.count(",") + 1
            assert nargs == len(_ext.SIGS[name]), (name, nargs, len(_ext.SIGS[name]))


def test_workspace_query_and_error_path_without_gpu():
    from myolo import _ext
    lib = _ext.load()
    assert _ext.workspace_bytes(921984, 256, 256) > 9 * 256 * 256 * 4
    rc = lib.myolo_fill(None, ctypes.c_float(0.0), 0, None)             # argument check happens before any launch
    assert rc == -1 and b"fill" in lib.myolo_last_error_string()


def test_host_side_of_the_abi_under_asan_ubsan():
    """SURVEY section 5, VERDICT r2 item 8: the host side of libmyolo_hip.so built with -fsanitize=address,undefined
    (__graft_entry__.build_sanitized) and driven, in a subprocess with the sanitizer runtime preloaded, through every path that
    needs no GPU: symbol binding, option table (unknown names, round trips), error strings of rejected arguments (null pointers,
    bad shapes, short workspaces), workspace / buffer-size queries, the split planner of the bf16x6 weight gradient, the RCCL
    wrappers' argument checks.  Any heap / stack / UB report makes the child exit non-zero."""
    import subprocess
    import sys
    import textwrap
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    import __graft_entry__
    lib, rt = __graft_entry__.build_sanitized()
    assert os.path.exists(lib) and os.path.exists(rt)
    child = textwrap.dedent("""
        import ctypes, sys
        sys.path[:0] = [%r, %r]
        from myolo import _ext as X
        lib = X.load()
        assert X.LIB_PATH.endswith("libmyolo_hip_asan.so")
        P = ctypes.c_void_p
        # option table
        assert X.set_option("wino_x6", 1) in (0, 1) and X.set_option("wino_x6", 0) == 1
        assert lib.myolo_set_option(b"no_such_switch", 1) < 0 and b"no_such_switch" in lib.myolo_last_error_string()
        v = ctypes.c_int(7)
        assert lib.myolo_get_option(b"bn_fused_tf_variance", ctypes.byref(v)) == 0 and v.value == 1
        # rejected arguments of every operator family: null pointers / bad shapes come back as MYOLO_EINVAL with a message, nothing is launched
        bad = 0
        for name, sig in X.SIGS.items():
            if name in ("myolo_set_option", "myolo_get_option") or name.startswith("myolo_comm") or name == "myolo_allreduce_sum_f32":
                continue
            args = []
            for t in sig:
                args.append(None if t is ctypes.c_void_p else (0.0 if t is ctypes.c_float else 0))
            rc = getattr(lib, name)(*args)
            assert rc < 0, (name, rc)
            assert len(lib.myolo_last_error_string()) > 0
            bad += 1
        assert bad > 80, bad
        # size queries (host arithmetic only)
        assert X.workspace_bytes(921984, 2304, 256) > 0 and X.wino_ws_bytes(4704, 14, 14, 256, 256, 2) > 0
        assert X.wino63_plane_elems(4704, 256) == 400 * 4704 * 256 and X.wino63_bwd_weight_from_q_ws_bytes(4704, 256, 256) > 64 * 256 * 256 * 4
        assert X.wino63_bwd_weight_from_q_ws_bytes(7, 256, 256) > 0 and X.wino63_bwd_weight_from_q_ws_bytes(4704, 256, 512) > 0     # bf16x6 split planner, small and wide
        assert X.matmul_ws_bytes(2304, 256, 1, 1) == 2304 * 256 * 6 and X.matmul_ws_bytes(256, 256, 1, 0) == 0
        assert X.pw_bnstats_ws_bytes(401408, 32, 64) > 0 and X.pw_bnstats_ws_bytes(1568, 1024, 1024) > 8 * 1568 * 1024 * 4
        assert X.dw_bnstats_ws_bytes(32, 112, 112, 32, 1) > 0 and X.conv1_bnstats_ws_bytes(32, 224, 224, 32) > 0
        assert X.deconv_mask_ws_bytes(4704, 14, 14, 256, 256, 4) > 0 and X.wino_plane_elems(4704, 14, 14, 256) == 484 * 4704 * 256
        # RCCL wrappers: argument checks before anything touches a device
        assert lib.myolo_comm_init(0, 0, None, None) < 0 and lib.myolo_allreduce_sum_f32(None, 0, None, None) < 0
        assert lib.myolo_comm_size(None, None) < 0 and lib.myolo_comm_destroy(None) in (0, -1, -4)
        print("asan-child-ok", bad)
        """ % (ROOT, os.path.join(ROOT, "mask-yolo_amd")))
    env = dict(os.environ, LD_PRELOAD=rt, MYOLO_LIB=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", child], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "asan-child-ok" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under mask-yolo_amd/ may import it."""
    pkg = os.path.join(ROOT, "mask-yolo_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), os.path.join(dp, f)


def test_keras_h5_name_mapping_roundtrip():
    """tools/h5_to_npz.py mapping (model.py:1157-1196 checkpoint schema): Keras weight names ->
    state_dict keys and back, including the nested 'yolo_model' group and the depthwise multiplier axis."""
    import importlib.util
    import os
    from myolo.config import make_config, ShapesConfig
    from myolo.engine import init_state_dict
    spec = importlib.util.spec_from_file_location("h5_to_npz", os.path.join(os.path.dirname(__file__), "..", "tools", "h5_to_npz.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    cfg = make_config(ShapesConfig, ALPHA=0.25)
    sd = init_state_dict(cfg, seed=1)
    groups = m.state_to_keras_weights(sd)
    assert "yolo_model" in groups and "conv_dw_7" not in groups and "conv_dw_6" in groups and "myolo_mask_deconv" in groups
    inner = {n.split("/")[0] for n, _ in groups["yolo_model"]}
    assert inner == set(m.nested_layers()) and "conv_23" in inner
    dw = dict(groups["conv_dw_1"])["conv_dw_1/depthwise_kernel:0"]
    assert dw.shape == (3, 3, 8, 1)
    flat = {n: a for items in groups.values() for n, a in items}
    back = m.keras_weights_to_state(flat)
    assert set(back) == set(sd)
    assert all(np.array_equal(back[k], sd[k]) for k in sd)
    # tf.keras-style prefixed names map to the same keys
    assert set(m.keras_weights_to_state({"yolo_model/conv_23/kernel:0": sd["conv_23/kernel"]})) == {"conv_23/kernel"}


def test_set_option_roundtrip_and_unknown_name():
    """tuning switches are plain ints behind myolo_set_option / myolo_get_option -- no environment reads in the library."""
    from myolo import _ext
    lib = _ext.load()
    assert _ext.set_option("bf16_no256", 1) == 0
    v = ctypes.c_int(-1)
    assert lib.myolo_get_option(b"bf16_no256", ctypes.byref(v)) == 0 and v.value == 1
    with _ext.option("bf16_no256", 0):
        lib.myolo_get_option(b"bf16_no256", ctypes.byref(v))
        assert v.value == 0
    lib.myolo_get_option(b"bf16_no256", ctypes.byref(v))
    assert v.value == 1
    _ext.set_option("bf16_no256", 0)
    assert lib.myolo_set_option(b"no_such_switch", 1) == -1 and b"unknown option" in lib.myolo_last_error_string()
    for src in ("gemm_kernels.hip", "mem_kernels.hip", "bf16_kernels.hip", "wino_kernels.hip", "exact_kernels.hip", "comm_rccl.hip"):
        assert "getenv" not in open(os.path.join(ROOT, "mask-yolo_amd", "csrc", src)).read(), src


def test_comm_entry_points_validate_arguments_without_gpu():
    from myolo import _ext
    lib = _ext.load()
    assert lib.myolo_comm_init(2, 2, None, None) == -1                  # rank out of range / null id: rejected before RCCL is touched
    assert lib.myolo_allreduce_sum_f32(None, 4, None, None) == -1
    assert lib.myolo_comm_destroy(None) == 0


def test_bench_self_launches_under_torchrun(monkeypatch):
    """plain `python bench.py --gpus 2` (no torchrun env) re-executes itself as one rank per GPU over 127.0.0.1."""
    import importlib
    import sys
    import subprocess
    bench = importlib.import_module("bench")
    seen = {}
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: seen.update(cmd=cmd, env=env) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_bench_line_is_compact():
    """the ONE stdout line of bench.py stays far below what the driver's capture keeps (round 4's 22 KB line was lost: BENCH_r04 parsed = null).
    compact_line() is fed round 4's full result object (profiles/r4_bench.json, 22 KB) and the Rice-416 one: < 6000 bytes, every contract key,
    roofline and cpu_baseline present, main() asserts the same bound before it prints."""
    import importlib
    import json
    bench = importlib.import_module("bench")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert bench.MAX_LINE_BYTES <= 6000
    full = json.load(open(os.path.join(root, "profiles", "r4_bench.json")))
    assert len(json.dumps(full)) > 20000
    line = json.dumps(bench.compact_line(full))
    assert len(line) < bench.MAX_LINE_BYTES, len(line)
    d = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "step_ms", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] == round(full["value"], 4) and d["ms_per_step"] == round(full["ms_per_step"], 4)
    for k in ("workload", "fp32_products", "n_pos_mean", "n_pos_sweep_ms", "train_api_images_per_sec", "global_batch", "parallelism"):
        assert k in d["config"], k
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "frac_composite", "traffic", "algorithmic_bytes", "algorithmic_flop", "avg_launch_ms"):
        assert k in d["roofline"], k
    assert abs(d["roofline"]["frac"] - d["roofline"]["achieved"] / d["roofline"]["peak"]) < 1e-3
    for fam in ("depthwise", "roialign", "pointwise"):
        assert 0 < d["roofline"][fam]["frac"] < 1
    assert "trunk_layers" not in d["roofline"] and "variants" not in d and "train_api" not in d
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1 and d["cpu_baseline"]["value"] > 0
    for k in ("nbox5_images_per_sec", "rice416_bf16_images_per_sec", "rice416_bf16_detect_many_images_per_sec"):
        assert d[k] > 0
    # a line with a comm object (N > 1) and long strings everywhere still fits
    full["comm"] = {"backend": "RCCL via torch.distributed (nccl)" * 4, "rccl_ranks_seen": 8, "bucket_allreduce_ms": [0.123456789] * 3,
                    "weights_identical_across_ranks": True, "bucket_bytes": [1 << 20] * 3, "note": "x" * 4000}
    full["config"]["workload"] = "w" * 5000
    full["roofline"]["kernel"] = "k" * 5000
    full["cpu_baseline"]["sample"] = "s" * 5000
    assert len(json.dumps(bench.compact_line(full))) < bench.MAX_LINE_BYTES
    ri = json.load(open(os.path.join(root, "profiles", "r3c_bench_rice416_bf16.json")))
    di = bench.compact_line(ri)
    assert len(json.dumps(di)) < bench.MAX_LINE_BYTES and di["dtype"] == "bf16" and di["roofline"]["deconv_mask_frac"] > 0


def test_bench_algorithmic_work_matches_survey():
    import importlib
    bench = importlib.import_module("bench")
    assert abs(bench.dw_bytes(224, 1.0, 1) / 1e6 - 20.97) < 0.01          # 20.97 MB / image over the 14 dw layers
    fl, _ = bench.pw_flops_bytes(224, 1.0, 1)
    assert abs(fl / 1e6 - (488.2 + 770.8)) < 1.0                          # backbone + YOLO-head pointwise MFLOP / image


def test_batch_generator_fill_equals_getitem_and_byte_images():
    """BatchGenerator.fill (round 4: what MaskYOLO.train() hands the engine's pinned staging arrays to) writes __getitem__'s arrays in place: float32
    images bit-identical (the 256-entry table = image / 255. stored into float32), byte images = the raw pixels (normalised later by
    myolo_u8_to_unit_f32), the other arrays equal after the cast the upload applies anyway; the wrapped last batch and 'yolo' mode included."""
    cfg = make_config(ShapesConfig, IMAGE_SHAPE=[128, 128, 3], BATCH_SIZE=4)
    samples = make_shapes_samples(6, cfg)
    gen = mutils.BatchGenerator(samples, cfg, "training", shuffle=False, norm=True)
    assert np.array_equal(mutils._U8_OVER_255, (np.arange(256, dtype=np.uint8) / 255.).astype(np.float32))
    T = cfg.TRUE_BOX_BUFFER
    for idx in (0, 1):
        ref, _ = gen[idx]
        for img_dt in (np.float32, np.uint8):
            out = [np.full(ref[0].shape, 7, img_dt), np.full((4, T, 4), 7, np.float32), np.full(ref[2].shape, 7, np.float32),
                   np.full(ref[3].shape, 7, np.int32), np.full(ref[4].shape, 7, np.int32), np.full(ref[5].shape, 7, np.uint8)]
            gen.fill(idx, out)
            if img_dt == np.uint8:
                assert np.array_equal(mutils._U8_OVER_255[out[0]], ref[0])
            else:
                assert np.array_equal(out[0], ref[0])
            assert np.array_equal(out[1].reshape(-1), ref[1].astype(np.float32).reshape(-1))
            assert np.array_equal(out[2], ref[2].astype(np.float32))
            assert np.array_equal(out[3], ref[3]) and np.array_equal(out[4], ref[4]) and np.array_equal(out[5].astype(bool), ref[5])
    assert gen.batch_bounds(1) == (2, 6)                               # wrapped back to a full batch (myolo_utils.py:730-735)
    ygen = mutils.BatchGenerator(samples, cfg, "yolo", shuffle=False, norm=True)
    ref, _ = ygen[0]
    out = [np.zeros(ref[0].shape, np.float32), np.ones((4, T, 4), np.float32), np.ones(ref[2].shape, np.float32)]
    ygen.fill(0, out)
    assert np.array_equal(out[0], ref[0]) and np.array_equal(out[2], ref[2].astype(np.float32))


def test_prefetcher_close_stops_a_blocked_thread_and_surfaces_errors():
    """MaskYOLO.train()'s prefetch thread must not outlive a training loop that ends early (exception in a step or callback,
    KeyboardInterrupt): close() unblocks a thread stuck in q.put(), drains what it had staged and joins it; errors of make() still reach
    the consumer; a full run still delivers every item in order."""
    import threading
    import time
    from myolo.model import _Prefetcher
    made = []

    def make(i):
        made.append(i)
        return i * i
    p = _Prefetcher(range(100), make, None, depth=2)
    it = iter(p)
    assert next(it) == (0, 0) and next(it) == (1, 1)
    time.sleep(0.2)                                  # the thread is now blocked on a full queue, far from the schedule's end
    assert p._th.is_alive() and len(made) < 10
    assert p.close() is True and not p._th.is_alive()
    assert p._q.empty() and len(made) < 10
    assert [x for x in _Prefetcher(range(7), make, None)] == [(i, i * i) for i in range(7)]

    def bad(i):
        if i == 3:
            raise ValueError("boom")
        return i
    q = _Prefetcher(range(10), bad, None)
    got = []
    with pytest.raises(ValueError):
        for item in q:
            got.append(item)
    assert got == [(0, 0), (1, 1), (2, 2)] and q.close()
    assert not [t for t in threading.enumerate() if t.name == "myolo-batch-prefetch" and t.is_alive()]
