"""ctypes binding of libsy11.so (the C-ABI declared in include/sy11.h).

The product path has NO fallback: if the shared library is missing, or a call returns an error code, this
module raises.  Nothing here imports the oracle.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("SY11_LIB", _HERE / "libsy11.so"))

F32, F16, BF16, U8 = 0, 1, 2, 3
EPI_SILU, EPI_ACCUM, EPI_OUT_F32 = 1, 2, 4


class Sy11Error(RuntimeError):
    pass


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dtype", "B", "IH", "IW", "C", "x_ld", "OH", "OW", "N", "y_ld",
        "KH", "KW", "SH", "SW", "PH", "PW", "DH", "DW", "groups")] + [("flags", C.c_uint32), ("stat_slots", C.c_int32)]


class BnTail(C.Structure):
    """sy11_bn_tail: BatchNorm finalisation folded into the producing conv's tail (include/sy11.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("gamma", "beta", "running_mean", "running_var", "mean", "rstd", "scale", "shift", "ticket")] \
        + [("eps", C.c_float), ("momentum", C.c_float), ("count", C.c_double)]


class OptDesc(C.Structure):
    """sy11_opt_desc: one optimizer step over the flat buffers (include/sy11.h)."""
    _fields_ = [("n", C.c_int64), ("n_buf", C.c_int64), ("group_end", C.c_int64 * 3), ("lr", C.c_float * 3), ("momentum", C.c_float * 3),
                ("weight_decay", C.c_float * 3), ("kind", C.c_int32), ("beta2", C.c_float), ("eps", C.c_float), ("max_norm", C.c_float),
                ("ema_decay", C.c_float), ("amp", C.c_int32), ("growth_factor", C.c_float), ("backoff_factor", C.c_float),
                ("growth_interval", C.c_int32), ("nparts", C.c_int32)]


_vp, _i32, _i64, _f32, _f64, _u32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_uint32
_dp = C.POINTER(ConvDesc)
_bp = C.POINTER(BnTail)
_op = C.POINTER(OptDesc)

# name -> argtypes (restype is int unless noted); the single source the symbol-export test checks against sy11.h
SIGNATURES = {
    "sy11_conv2d_fwd": [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sy11_conv2d_fwd_bn": [_dp, _vp, _vp, _vp, _vp, _vp, _bp, _vp],
    "sy11_stem_conv_fwd_bn": [_dp, _vp, _vp, _vp, _vp, _vp, _bp, _vp],
    "sy11_conv2d_dgrad": [_dp, _vp, _i32, _vp, _vp, _vp],
    "sy11_conv2d_wgrad": [_dp, _vp, _vp, _i32, _vp, _vp],
    "sy11_weight_transpose": [_i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "sy11_weight_transpose_multi": [_i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "sy11_stem_conv_fwd": [_dp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sy11_stem_conv_wgrad": [_dp, _vp, _vp, _i32, _vp, _vp],
    "sy11_bn_finalize": [_i32, _i32, _f64, _vp, _vp, _vp, _vp, _f32, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sy11_bn_act_fwd": [_i32, _i64, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp],
    "sy11_bn_act_bwd_reduce": [_i32, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32, _vp],
    "sy11_bn_act_bwd_apply": [_i32, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32,
                              _vp, _i32, _vp, _vp, _vp],
    "sy11_bn_act_bwd_apply_res": [_i32, _i64, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32,
                                  _vp, _i32, _vp, _vp, _vp, _i32, _i32, _vp],
    "sy11_copy2d": [_i32, _i64, _i32, _vp, _i32, _vp, _i32, _i32, _vp],
    "sy11_upsample2x_fwd": [_i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp],
    "sy11_upsample2x_bwd": [_i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _i32, _vp],
    "sy11_box_iou": [_i32, _i32, _vp, _vp, _f32, _vp, _vp],
    "sy11_fusion_stats": [_i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp],
    "sy11_sab_map_fwd": [_i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "sy11_sab_map_bwd": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sy11_gct_gate_fwd": [_i32, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp],
    "sy11_gct_gate_bwd": [_i32, _i32, _vp, _vp, _vp, _vp, _f32, _vp, _vp, _vp, _vp, _vp, _vp],
    "sy11_fusion_combine": [_i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp],
    "sy11_fusion_bwd_reduce": [_i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp],
    "sy11_fusion_bwd_apply": [_i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, _i32, _vp],
    "sy11_maxpool5_fwd": [_i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp],
    "sy11_maxpool5_bwd": [_i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _vp],
    "sy11_cast": [_i32, _i32, _i64, _vp, _vp, _vp],
    "sy11_bias_grad_cast": [_i32, _i64, _i32, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp],
    "sy11_attention_fwd": [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp],
    "sy11_attention_bwd": [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp],
    "sy11_attention_bwd_o": [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp],
    "sy11_detect_decode": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp],
    "sy11_nms_sorted": [_i32, _vp, _f32, _vp, _vp, _vp],
    "sy11_nms_sorted_batched": [_i32, _vp, _vp, _f32, _i32, _vp, _vp, _vp],
    "sy11_nms_candidates": [_i32, _i32, _i32, _i32, _f32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sy11_nms_sorted_segments": [_i32, _vp, _vp, _i32, _vp, _f32, _i32, _vp, _vp, _vp],
    "sy11_det_loss_assign": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "sy11_det_loss_terms": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp],
    "sy11_det_loss_bwd": [_i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _f32, _f32, _f32, _vp],
    "sy11_det_loss_pack_targets": [_i32, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _f32, _f32, _vp, _vp],
    "sy11_det_loss_finish": [_vp, _i32, _f32, _f32, _f32, _vp, _vp],
    "sy11_stft_logmel": [_i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "sy11_stft_minmax_init": [_i32, _vp, _vp],
    "sy11_stft_normalize": [_i32, _i32, _i32, _vp, _vp, _vp, _vp],
    "sy11_image_u8_to_float": [_i32, C.c_int64, _vp, _vp, _vp],
    "sy11_image_resize_bilinear": [_i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp],
    "sy11_image_letterbox": [_i32] * 12 + [_vp, _vp, _vp],
    "sy11_image_mosaic_warp": [_i32, _i32, _vp, _vp, _i32, _i32, _vp, _i32, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp],
}
SIGNATURES.update({
    "sy11_set_option": [C.c_char_p, _i32],
    "sy11_get_option": [C.c_char_p, C.POINTER(C.c_int32)],
    "sy11_tune_import": [_vp, _i64],
    "sy11_tune_clear": [],
    "sy11_peak_mfma_f16": [_i32, _i32, _vp, _vp],
    "sy11_debug_stamps": [_vp],
    "sy11_debug_stamps_persistent": [_vp],
    "sy11_opt_workspace_floats": [_i32],
    "sy11_opt_grad_norm": [_i64, _vp, _vp, _vp, _vp, _i32, _vp],
    "sy11_opt_step": [_op, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
})
OTHER = {"sy11_version": ([], C.c_int), "sy11_last_error": ([], C.c_char_p),
         "sy11_nms_workspace_bytes": ([_i32], C.c_size_t),
         "sy11_nms_batched_workspace_bytes": ([_i32, _vp], C.c_size_t),
         "sy11_attention_workspace_bytes": ([_i32, _i32, _i32], C.c_size_t),
         "sy11_tune_export": ([_vp, _i64], C.c_int64)}

_lib = None


def load():
    """Load libsy11.so once; raise loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise Sy11Error(f"libsy11.so not found at {LIB_PATH}: build it with "
                        f"`make -C spectrogram-yolov11_amd/csrc` (or __graft_entry__.build()). "
                        f"There is no CPU / PyTorch fallback for the hot path.")
    # torch ships its own libamdhip64 (same SONAME as /opt/rocm's): import it FIRST so that libsy11 binds to the HIP
    # runtime torch uses — two runtimes in one process do not share devices, streams or allocations.
    import torch  # noqa: F401
    lib = C.CDLL(str(LIB_PATH))
    for name, args in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    for name, (args, res) in OTHER.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = res
    _lib = lib
    # tile-pick tables across processes (profiling passes that must run the kernels a normal run picked): SY11_TUNE_LOAD=<file>
    # imports at load, SY11_TUNE_SAVE=<file> exports at interpreter exit
    src, dst = os.environ.get("SY11_TUNE_LOAD"), os.environ.get("SY11_TUNE_SAVE")
    if src and Path(src).exists():
        tune_import(Path(src).read_bytes())
    if dst:
        import atexit
        atexit.register(lambda: Path(dst).write_bytes(tune_export()))
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().sy11_last_error()
        raise Sy11Error(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")


# Optional per-launch timing (bench.py's roofline leg): when PROFILE is a list, every C-ABI call is bracketed by
# events recorded on the CURRENT stream (the stream the kernels are launched on) and appended as
# (name, start_event, end_event, meta).  None (default) = zero overhead.
PROFILE = None
PROFILE_META = None


def call(name: str, *args):
    if PROFILE is None:
        check(getattr(load(), name)(*args), name)
        return
    import torch
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(getattr(load(), name)(*args), name)
    e1.record()
    global PROFILE_META
    PROFILE.append((name, e0, e1, PROFILE_META))
    PROFILE_META = None                                   # meta belongs to exactly one call


def set_option(name: str, value: int):
    """sy11_set_option: "tune", "tune_log", "igemm_cfg", "wgrad_cfg", "igemm_korder", "igemm_deep", "igemm_bpol"."""
    check(load().sy11_set_option(name.encode(), int(value)), "sy11_set_option")


def get_option(name: str) -> int:
    v = C.c_int32(0)
    check(load().sy11_get_option(name.encode(), C.byref(v)), "sy11_get_option")
    return int(v.value)


def tune_export() -> bytes:
    """The autotuner's pick tables as bytes (16-byte records), for broadcasting to the other ranks."""
    lib = load()
    n = int(lib.sy11_tune_export(None, 0))
    buf = C.create_string_buffer(max(n, 1))
    n2 = int(lib.sy11_tune_export(buf, n))
    return buf.raw[:min(n, n2)]


def tune_import(blob: bytes):
    buf = C.create_string_buffer(bytes(blob), len(blob)) if blob else None
    check(load().sy11_tune_import(buf, len(blob)), "sy11_tune_import")
