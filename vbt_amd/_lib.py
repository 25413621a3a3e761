"""ctypes binding of libvbt_hip.so (include/vbt_hip.h).  There is no CPU fallback: if the
HIP library is missing this module raises, loudly."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VBT_LIB_PATH") or os.path.join(_HERE, "libvbt_hip.so")   # (VBT_LIB_PATH: developer builds with in-kernel stamps, tools/*_prof.py)
_lib = None

c_void_p, c_int, c_char_p, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_double


class VbtError(RuntimeError):
    pass


class TrackerParams(ctypes.Structure):
    _fields_ = [("max_age", ctypes.c_int32), ("min_hits", ctypes.c_int32), ("delta_t", ctypes.c_int32),
                ("asso", ctypes.c_int32), ("iou_threshold", c_double), ("inertia", c_double), ("det_thresh", c_double)]


class Run(ctypes.Structure):
    """vbt_run (include/vbt_hip.h): a run of consecutive frames of one clip inside a detector batch."""
    _fields_ = [("clip", ctypes.c_int32), ("slot0", ctypes.c_int32), ("slot_stride", ctypes.c_int32), ("n_frames", ctypes.c_int32),
                ("frame0", ctypes.c_int32), ("frame_step", ctypes.c_int32), ("fps", c_double)]


class KernelStat(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char * 32), ("launches", c_int), ("algorithmic_bytes", c_double), ("macs", c_double)]


class StepTime(ctypes.Structure):
    _fields_ = [("family", ctypes.c_char * 32), ("op", c_int), ("first_op", c_int), ("variant", c_int), ("ms", c_double),
                ("algorithmic_bytes", c_double), ("macs", c_double)]


_SIGS = {
    "vbt_last_error": (c_char_p, []),
    "vbt_device_count": (c_int, []),
    "vbt_model_create": (c_int, [c_char_p, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "vbt_model_create_ex": (c_int, [c_char_p, c_int, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "vbt_model_destroy": (None, [c_void_p]),
    "vbt_model_tensor_materialized": (c_int, [c_void_p, c_int]),
    "vbt_model_num_launches": (c_int, [c_void_p]),
    "vbt_model_input_shape": (c_int, [c_void_p, ctypes.POINTER(c_int)]),
    "vbt_model_num_tensors": (c_int, [c_void_p]),
    "vbt_model_num_ops": (c_int, [c_void_p]),
    "vbt_model_tensor_shape": (c_int, [c_void_p, c_int, ctypes.POINTER(c_int)]),
    "vbt_detect": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "vbt_detect_async": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "vbt_stream_create": (c_int, [c_int, c_void_p]),
    "vbt_stream_destroy": (c_int, [c_void_p]),
    "vbt_streams_share_queue": (c_int, [c_void_p, c_void_p, c_int, c_void_p]),
    "vbt_model_read_tensor": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "vbt_resize_frames": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "vbt_model_kernel_stats": (c_int, [c_void_p, c_int, ctypes.POINTER(KernelStat), c_int, ctypes.POINTER(c_int)]),
    "vbt_model_profile": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, ctypes.POINTER(c_double), c_int]),
    "vbt_model_profile_families": (c_int, [c_void_p, c_int, c_int, c_void_p, ctypes.POINTER(c_double), c_int]),
    "vbt_model_profile_steps": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, ctypes.POINTER(StepTime), c_int, ctypes.POINTER(c_int)]),
    "vbt_model_profile_overlap": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    "vbt_tracker_create": (c_int, [c_int, c_int, ctypes.POINTER(TrackerParams), c_int, ctypes.POINTER(c_void_p)]),
    "vbt_tracker_destroy": (None, [c_void_p]),
    "vbt_tracker_reset": (c_int, [c_void_p]),
    "vbt_tracker_update": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "vbt_tracker_update_from_detections": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, ctypes.c_float, c_void_p]),
    "vbt_tracker_last_output": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, ctypes.POINTER(c_int)]),
    "vbt_tracker_get_trackers": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, ctypes.POINTER(c_int)]),
    "vbt_tracker_status": (c_int, [c_void_p, c_int] + [ctypes.POINTER(ctypes.c_int32)] * 5),
    "vbt_tracker_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, ctypes.POINTER(c_int)]),
    "vbt_tracker_finish": (c_int, [c_void_p, c_double, c_double, c_double, c_void_p]),
    "vbt_tracker_phases": (c_int, [c_void_p, c_int, ctypes.POINTER(ctypes.c_int32), c_void_p, c_int, ctypes.POINTER(c_int)]),
    "vbt_tracker_update_from_slots": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, ctypes.c_float, c_void_p]),
    "vbt_tracker_update_from_detections_seq": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, ctypes.c_float, c_void_p]),
    "vbt_gather_frames": (c_int, [c_void_p, c_void_p, c_int, ctypes.c_size_t, c_void_p]),
    "vbt_tracker_summary": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "vbt_tracker_rows_all": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vbt_analyze": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_double, c_double, c_void_p, c_int, ctypes.POINTER(c_int), c_int]),
    "vbt_window_means": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VbtError(f"{LIB_PATH} is missing: build it with `python -m vbt_amd.build` "
                           "(__graft_entry__.build()). vbt_amd has no CPU fallback.")
        # One HIP runtime per process: the torch wheel bundles its own libamdhip64/libhsa-runtime64. If this
        # library were loaded first it would bind to /opt/rocm's copy and torch's later initialisation would
        # find "no HIP GPUs".  Importing torch first puts its runtime in the global symbol scope, and the
        # hip* symbols of libvbt_hip.so resolve to that same runtime.
        # (GPU_MAX_HW_QUEUES: see vbt_amd/__init__.py.)
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)          # AttributeError if the library does not export the symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def declared_symbols():
    return list(_SIGS)


def check(rc):
    if rc != 0:
        raise VbtError(f"libvbt_hip error {rc}: {lib().vbt_last_error().decode()}")
