"""vbt_amd: the MI355X-native hot path of simonkosina/vbt (reference track.py:159-247, plot.py:33-47)."""
import os as _os

# Pipeline keeps several HIP streams busy (one per forward in flight + the tracker).  With HIP's default of 4 hardware
# queues two of them can share a queue and serialise (measured: depth 2 39.6 k -> 54 k frames/s with 8).  The setting is
# read when the HIP runtime initialises, so it has to be in the environment before anything (torch included) touches
# the GPU: set here, at package import, never overriding the user's value.
import sys as _sys

if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "8"
    _t = _sys.modules.get("torch")
    if _t is not None and _t.cuda.is_initialized():
        import warnings as _w
        _w.warn("HIP was initialised before vbt_amd was imported: GPU_MAX_HW_QUEUES=8 is not in effect for this process "
                "(import vbt_amd, or export GPU_MAX_HW_QUEUES=8, before the first torch.cuda call)")
