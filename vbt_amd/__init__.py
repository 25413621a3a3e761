"""vbt_amd: the MI355X-native hot path of simonkosina/vbt (reference track.py:159-247, plot.py:33-47)."""
import os as _os

# Pipeline keeps four HIP streams busy (three forwards in flight + the copy stream of the host-fed mode).  The GPU serves four
# hardware queues side by side, one per compute pipe; with more queues than pipes two busy streams can land on queues of one
# pipe and the forwards thrash (measured with 3 + 1 busy streams: GPU_MAX_HW_QUEUES=4 97.5 k frames/s, 8 94.3 k, 6 94.1 k but
# 50 k host-fed, 5 58 k).  Four - HIP's own default - with Pipeline checking at creation that its busy streams sit on distinct
# queues (track.py: _place_streams) is the robust setting.  The variable is read when the HIP runtime initialises, so it has to
# be in the environment before anything (torch included) touches the GPU: set here, at package import, never overriding the
# user's value.
import sys as _sys

if "GPU_MAX_HW_QUEUES" not in _os.environ:
    _os.environ["GPU_MAX_HW_QUEUES"] = "4"
    _t = _sys.modules.get("torch")
    if _t is not None and _t.cuda.is_initialized():
        import warnings as _w
        _w.warn("HIP was initialised before vbt_amd was imported: GPU_MAX_HW_QUEUES=4 is not in effect for this process "
                "(import vbt_amd, or export GPU_MAX_HW_QUEUES=4, before the first torch.cuda call)")
