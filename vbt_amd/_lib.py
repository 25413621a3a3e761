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


class VbtArgError(VbtError, ValueError):
    """VBT_ERR_ARG: the library refused an argument (also a ValueError, like the host-side checks of the wrappers)."""


class TrackerParams(ctypes.Structure):
    _fields_ = [("max_age", ctypes.c_int32), ("min_hits", ctypes.c_int32), ("delta_t", ctypes.c_int32),
                ("asso", ctypes.c_int32), ("iou_threshold", c_double), ("inertia", c_double), ("det_thresh", c_double)]


class Run(ctypes.Structure):
    """vbt_run (include/vbt_hip.h): a run of consecutive frames of one clip inside a detector batch."""
    _fields_ = [("clip", ctypes.c_int32), ("slot0", ctypes.c_int32), ("slot_stride", ctypes.c_int32), ("n_frames", ctypes.c_int32),
                ("frame0", ctypes.c_int32), ("frame_step", ctypes.c_int32), ("fps", c_double)]


class PipelineParams(ctypes.Structure):
    """vbt_pipeline_params (include/vbt_hip.h)"""
    _fields_ = [("n_slots", ctypes.c_int32), ("n_clips", ctypes.c_int32), ("rows_cap", ctypes.c_int32), ("device", ctypes.c_int32),
                ("depth", ctypes.c_int32), ("tracker_stream", ctypes.c_int32), ("defer", ctypes.c_int32), ("selfcheck", ctypes.c_int32),
                ("strict_placement", ctypes.c_int32), ("model_flags", ctypes.c_int32), ("detection_threshold", ctypes.c_float),
                ("reserved0", ctypes.c_float), ("plate_diameter", c_double), ("diff_threshold", c_double), ("min_distance", c_double),
                ("tracker", TrackerParams)]


class PipelineInfo(ctypes.Structure):
    """vbt_pipeline_info (include/vbt_hip.h)"""
    _fields_ = [(k, ctypes.c_int32) for k in ("n_slots", "n_clips", "rows_cap", "device", "depth", "ring", "defer", "tracker_inline", "image_size",
                                               "frame_count", "steps_enqueued", "placement_ok", "queue_groups_seen")] + \
               [("reserved", ctypes.c_int32 * 3), ("det_streams", c_void_p * 8), ("copy_stream", c_void_p), ("tracker_stream", c_void_p),
                ("h2d_bytes", ctypes.c_uint64), ("step_host_ns", ctypes.c_uint64), ("step_calls", ctypes.c_uint64)]


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
    "vbt_pipeline_default_params": (None, [ctypes.POINTER(PipelineParams)]),
    "vbt_pipeline_create": (c_int, [c_char_p, ctypes.POINTER(PipelineParams), c_void_p, ctypes.POINTER(c_void_p)]),
    "vbt_pipeline_destroy": (None, [c_void_p]),
    "vbt_pipeline_step": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "vbt_pipeline_step_runs": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    "vbt_pipeline_skip_frames": (c_int, [c_void_p, c_int]),
    "vbt_pipeline_set_frame_count": (c_int, [c_void_p, c_int]),
    "vbt_pipeline_reset": (c_int, [c_void_p]),
    "vbt_pipeline_join_detectors": (c_int, [c_void_p, c_void_p]),
    "vbt_pipeline_close": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]),
    "vbt_pipeline_finish": (c_int, [c_void_p]),
    "vbt_pipeline_drain": (c_int, [c_void_p]),
    "vbt_pipeline_rows_all": (c_int, [c_void_p, c_void_p, c_void_p, c_int]),
    "vbt_pipeline_rows": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, ctypes.POINTER(c_int)]),
    "vbt_pipeline_detections": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, ctypes.POINTER(c_int)]),
    "vbt_pipeline_tracker_only_steps": (c_int, [c_void_p, c_int, c_int]),
    "vbt_pipeline_get_info": (c_int, [c_void_p, ctypes.POINTER(PipelineInfo)]),
    "vbt_pipeline_model": (c_void_p, [c_void_p, c_int]),
    "vbt_pipeline_tracker": (c_void_p, [c_void_p]),
    "vbt_track_clip": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int, ctypes.POINTER(c_int)]),
    "vbt_host_alloc": (c_int, [ctypes.c_size_t, ctypes.POINTER(c_void_p)]),
    "vbt_host_free": (c_int, [c_void_p]),
    "vbt_device_alloc": (c_int, [c_int, ctypes.c_size_t, ctypes.POINTER(c_void_p)]),
    "vbt_device_free": (c_int, [c_void_p]),
    "vbt_memcpy": (c_int, [c_void_p, c_void_p, ctypes.c_size_t, c_int]),
    "vbt_stream_synchronize": (c_int, [c_void_p]),
    "vbt_device_synchronize": (c_int, [c_int]),
    "vbt_analyze": (c_int, [c_void_p, c_int, c_int, c_int, c_double, c_double, c_double, c_void_p, c_int, ctypes.POINTER(c_int), c_int]),
    "vbt_window_means": (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_int]),
}


def _preload_hip_runtime():
    """One HIP runtime per process.  A torch wheel bundles its own libamdhip64 / libhsa-runtime64 (same soname as /opt/rocm's): if this
    library bound to the system copy and torch were imported afterwards, torch's runtime would find "no HIP GPUs".  So when a torch
    wheel is installed its libamdhip64 is loaded first - by path, WITHOUT importing torch - and libvbt_hip.so's DT_NEEDED entry then
    resolves to that already-loaded object by soname; `import torch`, before or after, shares it.  No import-order requirement is left.
    Without a torch wheel (or with VBT_HIP_RUNTIME=system) the system runtime under /opt/rocm is used."""
    import sys
    if "torch" in sys.modules or os.environ.get("VBT_HIP_RUNTIME") == "system":
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    so = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(so):
        ctypes.CDLL(so, mode=ctypes.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VbtError(f"{LIB_PATH} is missing: build it with `python -m vbt_amd.build` "
                           "(__graft_entry__.build()). vbt_amd has no CPU fallback.")
        _preload_hip_runtime()
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
        raise (VbtArgError if rc == -1 else VbtError)(f"libvbt_hip error {rc}: {lib().vbt_last_error().decode()}")
