"""ctypes binding of libmyolo_hip.so (the C-ABI declared in include/myolo_hip.h and include/myolo_hip_internal.h).

PyTorch is used only as the owner of device memory and streams: every function here takes
torch tensors, checks dtype/contiguity/device, and passes raw device pointers plus the current
HIP stream to the library.  There is no CPU or torch fallback: if the library is missing, or a
call returns a negative status, a RuntimeError is raised."""
import ctypes
import os

# (GPU_MAX_HW_QUEUES is handled by myolo.engine._ensure_hw_queues at the first Net, not at import: ADVICE r3)
import torch      # noqa: E402

_LIB = None
_HERE = os.path.dirname(os.path.abspath(__file__))
# MYOLO_LIB: an alternative build of the same C-ABI (the sanitizer build of __graft_entry__.build_sanitized(), used by the host-side tests)
LIB_PATH = os.environ.get("MYOLO_LIB") or os.path.join(_HERE, "_lib", "libmyolo_hip.so")

P, I, L, F, Z = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_size_t

# name -> argtypes (all return int).  Order/meaning: include/myolo_hip.h
SIGS = {
    "myolo_conv3x3s2_c3_fwd": [P, P, P, I, I, I, I, P],
    "myolo_conv3x3s2_c3_affine_act_fwd": [P, P, P, P, I, P, I, I, I, I, P],
    "myolo_conv3x3s2_c3_bwd_weight": [P, P, P, I, I, I, I, P, Z, P],
    "myolo_dwconv3x3_fwd": [P, P, P, I, I, I, I, I, P],
    "myolo_dwconv3x3_affine_act_fwd": [P, P, P, P, I, P, I, I, I, I, I, P],
    "myolo_dwconv3x3_bwd_data": [P, P, P, I, I, I, I, I, P],
    "myolo_dwconv3x3_bwd_data_bnsums": [P, P, P, I, I, I, I, I, P, P, P, P, P, I, P, I, P],
    "myolo_bn_act_bwd_from_partials": [P, P, P, P, P, P, P, P, P, L, I, I, P, I, P, Z, P],
    "myolo_dwconv3x3_bwd_weight": [P, P, P, I, I, I, I, I, P, Z, P],
    "myolo_pwconv1x1_fwd": [P, P, P, P, L, I, I, P, Z, P],
    "myolo_pwconv1x1_affine_act_fwd": [P, P, P, P, I, P, L, I, I, P, Z, P],
    "myolo_bn_frozen_coeffs_batched": [P, P, P, I, P, P],
    "myolo_pwconv1x1_bwd_data": [P, P, P, L, I, I, P, Z, P],
    "myolo_pwconv1x1_bwd_weight": [P, P, P, L, I, I, P, Z, P],
    "myolo_conv3x3_fwd": [P, P, P, P, I, I, I, I, I, P, Z, P],
    "myolo_conv3x3_affine_act_fwd": [P, P, P, P, P, P, I, I, I, I, I, I, P, Z, P],
    "myolo_conv3x3_bwd_data": [P, P, P, I, I, I, I, I, P, Z, P],
    "myolo_conv3x3_bwd_weight": [P, P, P, I, I, I, I, I, P, Z, P],
    "myolo_deconv2x2s2_fwd": [P, P, P, P, I, I, I, I, I, I, P, Z, P],
    "myolo_deconv2x2s2_bwd_data": [P, P, P, I, I, I, I, I, P, Z, P],
    "myolo_deconv2x2s2_bwd_weight": [P, P, P, I, I, I, I, I, P, Z, P],
    "myolo_colsum": [P, P, L, I, P, Z, P],
    "myolo_bn_stats": [P, P, P, P, P, P, P, P, P, L, I, P, Z, P],
    "myolo_bn_frozen_coeffs": [P, P, P, P, P, P, I, P],
    "myolo_bn_frozen_apply_act": [P, P, P, P, P, P, P, P, L, I, I, P],
    "myolo_bn_apply_act": [P, P, P, P, L, I, I, P],
    "myolo_bn_act_bwd": [P, P, P, P, P, P, P, P, P, P, L, I, I, I, P, Z, P],
    "myolo_bn_act_bwd_frozen_post": [P, P, P, P, P, P, P, P, L, I, I, P, Z, P],
    "myolo_crop_and_resize_fwd": [P, P, P, P, I, I, I, I, I, I, I, P],
    "myolo_crop_and_resize_bwd_image": [P, P, P, P, I, I, I, I, I, I, I, P],
    "myolo_gather_groups": [P, P, P, I, L, P],
    "myolo_bn_act_bwd_rowsparse": [P, P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, P, Z, P],
    "myolo_roialign_bwd_grouped": [P, P, P, I, I, I, I, I, I, I, P],
    "myolo_yolo_decode": [P, P, P, I, I, I, I, P],
    "myolo_yolo_detections": [P, P, P, I, I, I, I, P],
    "myolo_yolo_loss": [P, P, P, P, P, F, F, F, F, F, P, P, I, I, I, I, I, P, Z, P],
    "myolo_yolo_loss_warmup": [P, P, P, P, P, F, F, F, F, F, I, P, P, I, I, I, I, I, P, Z, P],
    "myolo_shapes_batch": [P, I, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, P, Z, P],
    "myolo_unmold_masks": [P, P, P, I, I, I, I, I, I, P, Z, P],
    "myolo_mask_targets": [P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "myolo_mask_head_out_fwd": [P, P, P, P, L, I, I, P],
    "myolo_conv3x3_wino_fwd": [P, P, P, P, P, P, I, I, I, I, I, I, P, P, Z, P],
    "myolo_wino63_weight_transform": [P, P, I, I, P],
    "myolo_wino63_multiply": [P, P, P, I, I, I, P],
    "myolo_wino63_multiply_w": [P, P, P, P, I, I, I, P],
    "myolo_wino63_input_transform_slots": [P, P, P, I, P, P, I, P, I, I, P],
    "myolo_wino63_output_input_transform_keep_pre_slots": [P, P, P, P, P, P, I, P, I, I, I, P],
    "myolo_wino63_output_transform_keep_pre_slots": [P, P, P, P, P, P, P, I, I, I, I, P],
    "myolo_wprep_stats": [P, P, P, P],
    "myolo_wprep_overflows": [P, P],
    "myolo_bn_act_bwd_fused": [P, P, P, P, P, P, P, P, P, L, I, I, P, P, Z, P],
    "myolo_wino63_input_transform": [P, P, P, I, P, P, P, I, I, P],
    "myolo_wino63_output_input_transform": [P, P, P, P, P, P, P, I, I, I, P],
    "myolo_wino63_output_transform": [P, P, P, P, P, I, I, I, P],
    "myolo_wino63_output_input_transform_keep_pre": [P, P, P, P, P, P, P, I, I, I, P],
    "myolo_wino63_output_transform_keep_pre": [P, P, P, P, P, P, P, I, I, I, P],
    "myolo_wino63_input_transform_roialign": [P, P, P, P, I, I, I, I, I, P],
    "myolo_wino63_output_transform_bn_stats": [P, P, P, I, I, P, P, P, P, P, P, P, P, P, Z, P],
    "myolo_wino63_bwd_weight_lazybn": [P, P, P, P, P, P, P, P, I, P, I, I, I, P, Z, P],
    "myolo_conv3x3_wino63_fwd": [P, P, P, P, P, P, I, I, I, I, P, P, Z, P],
    "myolo_conv3x3_wino63_bwd_data": [P, P, P, I, I, I, P, Z, P],
    "myolo_conv3x3_wino63_bwd_weight": [P, P, P, P, I, I, I, P, Z, P],
    "myolo_wino63_bwd_data_lazybn": [P, P, P, P, P, P, P, I, P, P, I, I, I, P, Z, P],
    "myolo_wino63_lazybn_transforms": [P, P, P, P, P, P, P, I, P, P, I, I, P],
    "myolo_wino63_bwd_data_from_v": [P, P, P, I, I, I, P, Z, P],
    "myolo_wino63_bwd_weight_from_q": [P, P, P, I, I, I, P, Z, P],
    "myolo_conv3x3_wino_bwd_data": [P, P, P, I, I, I, I, I, P, Z, P],
    "myolo_conv3x3_wino_bwd_weight": [P, P, P, P, I, I, I, I, I, P, Z, P],
    "myolo_deconv2x2s2_mask_fwd": [P, P, P, P, P, P, I, I, I, I, I, I, P, Z, P],
    "myolo_deconv2x2s2_mask_fwd_keep": [P, P, P, P, P, P, I, I, I, I, I, I, P, P, I, P, Z, P],
    "myolo_positive_index": [P, I, I, P, P, P, P, P],
    "myolo_wino_weight_transform": [P, P, I, I, I, P],
    "myolo_wino_input_transform": [P, P, I, I, I, I, P],
    "myolo_wino_multiply": [P, P, P, I, I, I, I, I, P],
    "myolo_wino_multiply_w": [P, P, P, P, I, I, I, I, I, P],
    "myolo_wino_output_transform": [P, P, P, P, P, I, I, I, I, I, P],
    "myolo_wino_output_input_transform": [P, P, P, P, P, P, P, I, I, I, I, I, P],
    "myolo_wino_output_input_transform_keep_pre": [P, P, P, P, P, P, P, I, I, I, I, I, P],
    "myolo_wino_input_transform_affine": [P, P, P, I, P, I, I, I, I, P],
    "myolo_wino_output_transform_bn_stats": [P, P, P, I, I, I, I, P, P, P, P, P, P, P, P, P, Z, P],
    "myolo_bn_bwd_rowsparse_coeffs": [P, P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, P, Z, P],
    "myolo_conv3x3_wino_bwd_data_lazybn": [P, P, P, P, P, P, P, I, P, P, I, I, I, I, I, P, Z, P],
    "myolo_conv3x3_wino_bwd_weight_lazybn": [P, P, P, P, P, P, P, P, I, P, I, I, I, I, I, P, Z, P],
    "myolo_wino_input_transform_roialign": [P, P, P, P, I, I, I, I, I, I, I, P],
    "myolo_pack_weights_bf16": [P, I, I, I, P, P, P, P, P, P, P, P],
    "myolo_crop_and_resize_bf16_fwd": [P, P, P, P, I, I, I, I, I, I, I, P],
    "myolo_conv3x3_bf16_fwd": [P, P, P, P, I, I, I, I, I, I, P],
    "myolo_deconv2x2s2_bf16_fwd": [P, P, P, P, I, I, I, I, I, I, P],
    "myolo_deconv2x2s2_mask_bf16_fwd": [P, P, P, P, P, P, I, I, I, I, I, I, P, Z, P],
    "myolo_mask_head_out_bf16_fwd": [P, P, P, P, L, I, I, P],
    "myolo_mask_head_out_bwd": [P, P, P, P, P, P, L, I, I, P, Z, P],
    "myolo_mask_bce": [P, P, P, F, P, P, I, I, I, I, P, Z, P],
    "myolo_adam_step": [P, P, P, P, L, F, F, F, F, F, P],
    "myolo_matmul_f32": [P, P, P, L, I, I, I, I, P, Z, P],
    "myolo_stream_copy": [P, P, Z, I, I, P],
    "myolo_mfma_probe": [I, I, I, P, P],
    "myolo_conv3x3s2_c3_bnstats_fwd": [P, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, Z, P],
    "myolo_dwconv3x3_bnstats_fwd": [P, P, P, I, P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, I, P, Z, P],
    "myolo_dwconv3x3_bwd_weight_affine_in": [P, P, P, I, P, P, I, I, I, I, I, P, Z, P],
    "myolo_pwconv1x1_bnstats_fwd": [P, P, P, I, P, P, P, P, P, P, P, P, P, P, L, I, I, I, P, Z, P],
    "myolo_pwconv1x1_bwd_weight_affine_in": [P, P, P, I, P, P, L, I, I, P, Z, P],
    "myolo_gather_groups_affine_act": [P, P, P, P, I, P, P, I, L, I, P],
    "myolo_add_inplace": [P, P, L, P],
    "myolo_fill": [P, F, L, P],
    "myolo_u8_to_unit_f32": [P, P, L, P],
    "myolo_set_option": [ctypes.c_char_p, I],
    "myolo_get_option": [ctypes.c_char_p, P],
    "myolo_comm_unique_id": [P],
    "myolo_comm_init": [I, I, P, P],
    "myolo_comm_size": [P, P],
    "myolo_allreduce_sum_f32": [P, L, P, P],
    "myolo_comm_destroy": [P],
}


def load():
    """dlopen the library and bind every symbol; raises if anything is missing."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libmyolo_hip.so not built (%s): run `python __graft_entry__.py` "
                           "(hipcc --offload-arch=gfx950); there is no fallback path" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGS.items():
        fn = getattr(lib, name)            # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    lib.myolo_version.restype = ctypes.c_int
    lib.myolo_last_error_string.restype = ctypes.c_char_p
    lib.myolo_workspace_bytes.argtypes = [L, I, I]
    lib.myolo_workspace_bytes.restype = Z
    lib.myolo_conv3x3_wino_ws_bytes.argtypes = [I, I, I, I, I, I]
    lib.myolo_conv3x3_wino_ws_bytes.restype = Z
    lib.myolo_matmul_f32_ws_bytes.argtypes = [I, I, I, I]
    lib.myolo_matmul_f32_ws_bytes.restype = Z
    lib.myolo_conv3x3s2_c3_bnstats_ws_bytes.argtypes = [I, I, I, I]
    lib.myolo_conv3x3s2_c3_bnstats_ws_bytes.restype = Z
    lib.myolo_dwconv3x3_bnstats_ws_bytes.argtypes = [I, I, I, I, I]
    lib.myolo_dwconv3x3_bnstats_ws_bytes.restype = Z
    lib.myolo_dwconv3x3_bwd_weight_ws_bytes.argtypes = [I, I, I, I, I]
    lib.myolo_dwconv3x3_bwd_weight_ws_bytes.restype = Z
    lib.myolo_pwconv1x1_bnstats_ws_bytes.argtypes = [L, I, I]
    lib.myolo_pwconv1x1_bnstats_ws_bytes.restype = Z
    lib.myolo_pwconv1x1_bnstats_ok.argtypes = [I, I]
    lib.myolo_pwconv1x1_bnstats_ok.restype = I
    lib.myolo_wino_plane_elems.argtypes = [I, I, I, I]
    lib.myolo_wino_plane_elems.restype = Z
    lib.myolo_wino_u_elems.argtypes = [I, I]
    lib.myolo_wino_u_elems.restype = Z
    lib.myolo_wino63_u_elems.argtypes = [I, I]
    lib.myolo_wino63_u_elems.restype = Z
    lib.myolo_wino63_plane_elems.argtypes = [I, I]
    lib.myolo_wino63_plane_elems.restype = Z
    lib.myolo_wino63_bwd_data_ws_bytes.argtypes = [I, I, I]
    lib.myolo_wino63_bwd_data_ws_bytes.restype = Z
    lib.myolo_conv3x3_wino63_ws_bytes.argtypes = [I, I, I, I]
    lib.myolo_conv3x3_wino63_ws_bytes.restype = Z
    lib.myolo_wino63_bwd_weight_ws_bytes.argtypes = [I, I, I]
    lib.myolo_wino63_bwd_weight_ws_bytes.restype = Z
    for fn in (lib.myolo_wino63_bwd_data_from_v_ws_bytes, lib.myolo_wino63_bwd_weight_from_q_ws_bytes):
        fn.argtypes = [I, I, I]
        fn.restype = Z
    lib.myolo_wino63_output_transform_bn_ws_bytes.argtypes = [I, I]
    lib.myolo_wino63_output_transform_bn_ws_bytes.restype = Z
    lib.myolo_wino63_ok.argtypes = [I, I, I, I]
    lib.myolo_wino63_ok.restype = I
    lib.myolo_deconv2x2s2_mask_ws_bytes.argtypes = [I, I, I, I, I, I]
    lib.myolo_deconv2x2s2_mask_ws_bytes.restype = Z
    lib.myolo_wino_output_transform_bn_ws_bytes.argtypes = [I]
    lib.myolo_wino_output_transform_bn_ws_bytes.restype = Z
    lib.myolo_bn_act_bwd_fused_ws_bytes.argtypes = [L, I]
    lib.myolo_bn_act_bwd_fused_ws_bytes.restype = Z
    lib.myolo_wprep_create.argtypes = [P, Z, P]
    lib.myolo_wprep_create.restype = I
    for fn in (lib.myolo_wprep_destroy, lib.myolo_wprep_activate, lib.myolo_wprep_invalidate):
        fn.argtypes = [P]
        fn.restype = I
    lib.myolo_wprep_count.argtypes = [P]
    lib.myolo_wprep_count.restype = I
    lib.myolo_dwconv3x3_bwd_data_bnsums_rows.argtypes = [I, I, I, I, I]
    lib.myolo_dwconv3x3_bwd_data_bnsums_rows.restype = I
    lib.myolo_wprep_refresh.argtypes = [P, I, I, I, P]
    lib.myolo_wprep_refresh.restype = I
    _LIB = lib
    return lib


def exported_symbols():
    return list(SIGS) + ["myolo_version", "myolo_last_error_string", "myolo_workspace_bytes", "myolo_conv3x3_wino_ws_bytes", "myolo_wino_plane_elems", "myolo_wino_u_elems", "myolo_wino63_u_elems", "myolo_wino63_plane_elems", "myolo_wino63_ok", "myolo_wino63_bwd_data_ws_bytes", "myolo_wino63_bwd_weight_ws_bytes", "myolo_wino63_bwd_data_from_v_ws_bytes", "myolo_wino63_bwd_weight_from_q_ws_bytes", "myolo_wino63_output_transform_bn_ws_bytes", "myolo_conv3x3_wino63_ws_bytes", "myolo_matmul_f32_ws_bytes", "myolo_conv3x3s2_c3_bnstats_ws_bytes", "myolo_dwconv3x3_bnstats_ws_bytes", "myolo_dwconv3x3_bwd_weight_ws_bytes",
                              "myolo_pwconv1x1_bnstats_ws_bytes", "myolo_pwconv1x1_bnstats_ok",
                              "myolo_deconv2x2s2_mask_ws_bytes", "myolo_wino_output_transform_bn_ws_bytes",
                              "myolo_bn_act_bwd_fused_ws_bytes", "myolo_dwconv3x3_bwd_data_bnsums_rows", "myolo_wprep_create", "myolo_wprep_destroy", "myolo_wprep_activate", "myolo_wprep_invalidate", "myolo_wprep_count", "myolo_wprep_refresh"]


def ptr(t):
    """raw device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_cuda and t.is_contiguous(), "C-ABI needs contiguous device tensors"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise RuntimeError("%s failed (%d): %s" % (name, rc, lib.myolo_last_error_string().decode()))


class WeightPrep(object):
    """Prepared-weights registry of the library (include/myolo_hip_internal.h: myolo_wprep_*): the weight-only re-layouts that entry points run in front of
    their kernels, recorded while the registry is active and re-run by refresh() on the caller's current stream.  Owns its arena (a device tensor)."""

    def __init__(self, device, arena_bytes=768 << 20):
        # a recorded preparation that does not fit the arena is simply made in place by its site, as without a registry (myolo_wprep_resolve):
        # the size is a capacity, not a requirement.  Net sizes it from its layer table (engine.Net._wprep_arena_bytes).
        self.arena = torch.empty(int(arena_bytes), dtype=torch.uint8, device=device)
        h = ctypes.c_void_p(0)
        call("myolo_wprep_create", self.arena.data_ptr(), self.arena.numel(), ctypes.byref(h))
        self.h = h.value

    def activate(self, on=True):
        load().myolo_wprep_activate(self.h if on else None)

    def count(self):
        return int(load().myolo_wprep_count(self.h))

    def invalidate(self):
        load().myolo_wprep_invalidate(self.h)

    def refresh(self, first=0, last=1 << 30, max_idle=4):
        n = load().myolo_wprep_refresh(self.h, int(first), int(last), int(max_idle), stream())
        if n < 0:
            raise RuntimeError("myolo_wprep_refresh failed: %s" % load().myolo_last_error_string().decode())
        return n

    def stats(self):
        a, b, c = ctypes.c_longlong(0), ctypes.c_longlong(0), ctypes.c_longlong(0)
        call("myolo_wprep_stats", self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c))
        o = ctypes.c_longlong(0)
        call("myolo_wprep_overflows", self.h, ctypes.byref(o))
        return dict(hits=a.value, misses=b.value, bytes_used=c.value, entries=self.count(), overflows=o.value)

    def close(self):
        if getattr(self, "h", None):
            lib = _LIB
            if lib is not None:
                lib.myolo_wprep_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def set_option(name, value):
    """process-wide tuning switch of the library (include/myolo_hip.h: myolo_set_option); returns the previous value."""
    lib = load()
    old = ctypes.c_int(0)
    if lib.myolo_get_option(name.encode(), ctypes.byref(old)) != 0:
        raise RuntimeError("myolo_get_option failed: %s" % lib.myolo_last_error_string().decode())
    call("myolo_set_option", name.encode(), int(value))
    return old.value


class option(object):
    """with X.option("bf16_force256", 1): ...   (restores the previous value on exit)"""

    def __init__(self, name, value):
        self.name, self.value = name, value

    def __enter__(self):
        self.old = set_option(self.name, self.value)
        return self

    def __exit__(self, *exc):
        set_option(self.name, self.old)
        return False


def measure_hbm_copy_gbs(nbytes=2 << 30, iters=5, device="cuda:0"):
    """{variant name: read+write GB/s} of the hand-written float4 stream-copy kernel (myolo_stream_copy) on buffers far larger than
    the 256 MiB Infinity Cache, timed with HIP events on the current stream.  'read_only' / 'write_only' count one direction."""
    src = torch.empty(nbytes, dtype=torch.uint8, device=device)
    dst = torch.empty(nbytes, dtype=torch.uint8, device=device)
    src.zero_()
    dst.zero_()
    out = {}
    # the rate depends on how the grid is shaped: ONE workgroup of 256 threads per CU (each thread streams many MB) is ~20 % faster than the
    # 8-per-CU grid an ordinary elementwise launch uses -- both are reported ("_1wg_per_cu" / plain)
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    for name, variant, factor in (("float4", 0, 2.0), ("float4_nt_store", 1, 2.0), ("float4_nt_load_store", 2, 2.0), ("read_only", 3, 1.0), ("write_only", 4, 1.0)):
        for suffix, blocks in (("", 8 * cus), ("_1wg_per_cu", cus)):
            call("myolo_stream_copy", src.data_ptr(), dst.data_ptr(), nbytes, variant, blocks, stream())
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                call("myolo_stream_copy", src.data_ptr(), dst.data_ptr(), nbytes, variant, blocks, stream())
            e1.record()
            torch.cuda.synchronize()
            out[name + suffix] = factor * nbytes * iters / (e0.elapsed_time(e1) * 1e-3) / 1e9
    return out


def _smi_power_clock():
    """(socket power W, shader clock MHz) from rocm-smi, or (None, None)."""
    import subprocess
    try:
        t = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
        pw = [float(l.split(":")[-1]) for l in t.splitlines() if "Socket Graphics Package Power" in l]
        sc = [float(l.split("(")[-1].split("Mhz")[0]) for l in t.splitlines() if "sclk" in l]
        return (pw[0] if pw else None), (sc[0] if sc else None)
    except Exception:
        return None, None


def measure_power_limit(n_rois=32 * 147, launches=500, device="cuda:0"):
    """Is the dominant kernel power-limited?  The F(6,3) Winograd multiply (bf16x6 products, the launch shape of the step's mask-head convs) run
    back to back on (a) normally distributed operands and (b) constant operands -- the same instruction stream, the same memory traffic, fewer
    bits toggling -- with the socket power and shader clock rocm-smi reports while (a) runs.  On MI355X (1400 W package cap, 2400 MHz) the kernel
    reaches the cap and is clocked down on real data (profiles/r5_notes.md section 8): its distance from the nominal matrix-pipe peak is then
    an energy budget, not a schedule.  -> dict(ms_random, ms_constant, power_w, sclk_mhz, cap_w)"""
    import subprocess
    old = set_option("wino_x6", 1)
    try:
        C = 256
        pe = wino63_plane_elems(n_rois, C)
        g = torch.Generator(device=device).manual_seed(0)
        V = torch.randn(pe, device=device, generator=g)
        w = torch.randn(3, 3, C, C, device=device, generator=g) * 0.02
        M = torch.empty(pe, device=device)
        U = torch.empty(wino63_u_elems(C, C), device=device)
        out = {}
        for name in ("random", "constant"):
            if name == "constant":
                V.fill_(1.0)
                w.fill_(1.0)
            call("myolo_wino63_weight_transform", ptr(w), ptr(U), C, C, stream())
            for _ in range(40):                                   # through the clock transient after idle
                call("myolo_wino63_multiply", ptr(V), ptr(U), ptr(M), n_rois, C, C, stream())
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(launches):
                call("myolo_wino63_multiply", ptr(V), ptr(U), ptr(M), n_rois, C, C, stream())
            e1.record()
            if name == "random":                                  # the queue holds ~0.65 s of launches: two samples while they run (the power
                import time                                       # figure is a moving average: the later sample is the settled one)
                time.sleep(0.2)
                p1, c1 = _smi_power_clock()
                p2, c2 = _smi_power_clock()
                ps_, cs_ = [v for v in (p1, p2) if v is not None], [v for v in (c1, c2) if v is not None]
                out["power_w"], out["sclk_mhz"] = (max(ps_) if ps_ else None), (min(cs_) if cs_ else None)
            torch.cuda.synchronize()
            out["ms_" + name] = e0.elapsed_time(e1) / launches
        try:
            t = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showmaxpower"], capture_output=True, text=True, timeout=10).stdout
            cap = [float(l.split(":")[-1]) for l in t.splitlines() if "Max Graphics Package Power" in l]
            out["cap_w"] = cap[0] if cap else None
        except Exception:
            out["cap_w"] = None
        return out
    finally:
        set_option("wino_x6", old)


def measure_mfma_tflops(iters=20000, reps=3, device="cuda:0"):
    """{"bf16_32x32x16": TFLOP/s, "f32_32x32x2": TFLOP/s} the matrix pipes sustain with nothing else going on (myolo_mfma_probe: two
    workgroups of four waves per CU, eight independent accumulator blocks per wave, register operands), timed with HIP events."""
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    blocks = 2 * cus
    out_buf = torch.zeros(blocks * 256, device=device)
    res = {}
    for name, kind, flop, it in (("bf16_32x32x16", 0, 32768.0, iters), ("f32_32x32x2", 1, 4096.0, iters // 2)):
        for _ in range(3):                                  # the clock ramps up over the first ~10 ms of MFMA work after idle
            call("myolo_mfma_probe", kind, it, blocks, out_buf.data_ptr(), stream())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call("myolo_mfma_probe", kind, it, blocks, out_buf.data_ptr(), stream())
        e1.record()
        torch.cuda.synchronize()
        res[name] = blocks * 4.0 * it * 8 * flop * reps / (e0.elapsed_time(e1) * 1e-3) / 1e12
    return res


def workspace_bytes(rows, cin, cout):
    return int(load().myolo_workspace_bytes(int(rows), int(cin), int(cout)))


def wino_ws_bytes(n, h, w, cin, cout, which):
    """scratch bytes of myolo_conv3x3_wino_{fwd,bwd_data,bwd_weight} (which = 0, 1, 2)."""
    return int(load().myolo_conv3x3_wino_ws_bytes(int(n), int(h), int(w), int(cin), int(cout), int(which)))


PRODUCTS_NATIVE, PRODUCTS_BF16X6 = 0, 1


def conv1_bnstats_ws_bytes(n, h, w, cout):
    return int(load().myolo_conv3x3s2_c3_bnstats_ws_bytes(int(n), int(h), int(w), int(cout)))


def dw_bnstats_ws_bytes(n, h, w, c, stride):
    return int(load().myolo_dwconv3x3_bnstats_ws_bytes(int(n), int(h), int(w), int(c), int(stride)))


def dw_bwd_weight_ws_bytes(n, h, w, c, stride):
    return int(load().myolo_dwconv3x3_bwd_weight_ws_bytes(int(n), int(h), int(w), int(c), int(stride)))


def pw_bnstats_ws_bytes(m, cin, cout):
    return int(load().myolo_pwconv1x1_bnstats_ws_bytes(int(m), int(cin), int(cout)))


def pw_bnstats_ok(cin, cout):
    return bool(load().myolo_pwconv1x1_bnstats_ok(int(cin), int(cout)))


def matmul_ws_bytes(k, n, b_is_nk, products):
    return int(load().myolo_matmul_f32_ws_bytes(int(k), int(n), int(b_is_nk), int(products)))


def wino_u_elems(cin, cout):
    """floats to allocate for the transformed filters U of myolo_wino_weight_transform"""
    return int(load().myolo_wino_u_elems(int(cin), int(cout)))


def dw_bwd_data_bnsums_rows(n, h, w, c, stride):
    """rows of partial sums myolo_dwconv3x3_bwd_data_bnsums leaves (0: not available for these sizes)."""
    return int(load().myolo_dwconv3x3_bwd_data_bnsums_rows(int(n), int(h), int(w), int(c), int(stride)))


def wino63_ok(h, w, cin, cout):
    return bool(load().myolo_wino63_ok(int(h), int(w), int(cin), int(cout)))


def wino63_ws_bytes(n, cin, cout, which):
    """scratch bytes of myolo_conv3x3_wino63_{fwd,bwd_data,bwd_weight} (which = 0, 1, 2)."""
    return int(load().myolo_conv3x3_wino63_ws_bytes(int(n), int(cin), int(cout), int(which)))


def wino63_bwd_weight_ws_bytes(n, cin, cout):
    return int(load().myolo_wino63_bwd_weight_ws_bytes(int(n), int(cin), int(cout)))


def wino63_bwd_data_from_v_ws_bytes(n, cin, cout):
    return int(load().myolo_wino63_bwd_data_from_v_ws_bytes(int(n), int(cin), int(cout)))


def wino63_bwd_weight_from_q_ws_bytes(n, cin, cout):
    return int(load().myolo_wino63_bwd_weight_from_q_ws_bytes(int(n), int(cin), int(cout)))


def wino63_out_bn_ws_bytes(n, c):
    return int(load().myolo_wino63_output_transform_bn_ws_bytes(int(n), int(c)))


def wino63_bwd_data_ws_bytes(n, cin, cout):
    return int(load().myolo_wino63_bwd_data_ws_bytes(int(n), int(cin), int(cout)))


def wino63_u_elems(cin, cout):
    return int(load().myolo_wino63_u_elems(int(cin), int(cout)))


def wino63_plane_elems(n, c):
    return int(load().myolo_wino63_plane_elems(int(n), int(c)))


def wino_plane_elems(n, h, w, c):
    """floats in the 36 Winograd planes V (or M) of an [n,h,w,c] tensor."""
    return int(load().myolo_wino_plane_elems(int(n), int(h), int(w), int(c)))


def deconv_mask_ws_bytes(n, h, w, cin, cout, ncls):
    return int(load().myolo_deconv2x2s2_mask_ws_bytes(int(n), int(h), int(w), int(cin), int(cout), int(ncls)))


def wino_out_bn_ws_bytes(c):
    return int(load().myolo_wino_output_transform_bn_ws_bytes(int(c)))
