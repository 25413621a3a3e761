"""`OCSort`: the tracker object of the reference hot loop, backed by libvbt_hip.so.

Mirrors what reference track.py uses of the (unpinned) `ocsort` package:
  OCSort(max_age=30, asso_func="diou", iou_threshold=0.1)          track.py:157
  .update(dets float64[N,6] = x1,y1,x2,y2,score,cls, _unused) -> float64[M,7] = x1,y1,x2,y2,id,cls,score   track.py:186-190
  .trackers: list of objects with .id (0-based) and .kf.x (7x1)    track.py:194-199
plus `MultiClipTracker`, the batched form used by the fused pipeline (n clips in one launch).
"""
import ctypes
from types import SimpleNamespace

import numpy as np

from . import _lib

MAXD = 25
_ASSO = {"iou": 0, "diou": 1}


def _params(det_thresh, max_age, min_hits, iou_threshold, delta_t, asso_func, inertia):
    if asso_func not in _ASSO:
        raise ValueError(f"asso_func must be one of {sorted(_ASSO)} (got {asso_func!r})")
    return _lib.TrackerParams(int(max_age), int(min_hits), int(delta_t), _ASSO[asso_func], float(iou_threshold),
                              float(inertia), float(det_thresh))


class MultiClipTracker:
    def __init__(self, n_clips, rows_cap, det_thresh=0.2, max_age=30, min_hits=3, iou_threshold=0.3, delta_t=3,
                 asso_func="iou", inertia=0.2, device=0):
        self.n_clips, self.rows_cap = int(n_clips), int(rows_cap)
        self._h = ctypes.c_void_p()
        p = _params(det_thresh, max_age, min_hits, iou_threshold, delta_t, asso_func, inertia)
        _lib.check(_lib.lib().vbt_tracker_create(self.n_clips, self.rows_cap, ctypes.byref(p), device, ctypes.byref(self._h)))

    @classmethod
    def _borrowed(cls, handle, n_clips, rows_cap, owner):
        """A view of the tracker owned by a vbt_pipeline (never destroyed from here; `owner` is kept alive)."""
        self = cls.__new__(cls)
        self.n_clips, self.rows_cap = int(n_clips), int(rows_cap)
        self._h, self._owner = ctypes.c_void_p(handle), owner
        return self

    def __del__(self):
        h = getattr(self, "_h", None)
        if h and getattr(self, "_owner", None) is None and _lib is not None and _lib._lib is not None:
            _lib._lib.vbt_tracker_destroy(h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def reset(self):
        _lib.check(_lib.lib().vbt_tracker_reset(self._h))

    def update_frames(self, dets, counts, times):
        """dets [F,n_clips,25,6] float64, counts [F,n_clips] int32, times [F,n_clips] float64."""
        dets = np.ascontiguousarray(dets, np.float64)
        counts = np.ascontiguousarray(counts, np.int32)
        times = np.ascontiguousarray(times, np.float64)
        F = counts.shape[0]
        assert dets.shape == (F, self.n_clips, MAXD, 6) and counts.shape == (F, self.n_clips) and times.shape == (F, self.n_clips)
        _lib.check(_lib.lib().vbt_tracker_update(self._h, dets.ctypes.data, counts.ctypes.data, times.ctypes.data, F))

    def last_output(self, clip=0):
        out = np.empty((MAXD, 7), np.float64)
        vel = np.empty((MAXD, 2), np.float64)
        m = ctypes.c_int()
        _lib.check(_lib.lib().vbt_tracker_last_output(self._h, clip, out.ctypes.data, vel.ctypes.data, MAXD, ctypes.byref(m)))
        return out[:m.value].copy(), vel[:m.value].copy()

    def trackers(self, clip=0):
        ids = np.empty(64, np.int32)
        kfx = np.empty((64, 7), np.float64)
        n = ctypes.c_int()
        _lib.check(_lib.lib().vbt_tracker_get_trackers(self._h, clip, ids.ctypes.data, kfx.ctypes.data, 64, ctypes.byref(n)))
        return [SimpleNamespace(id=int(ids[i]), kf=SimpleNamespace(x=kfx[i].reshape(7, 1).copy())) for i in range(n.value)]

    def status(self, clip=0):
        v = [ctypes.c_int32() for _ in range(5)]
        _lib.check(_lib.lib().vbt_tracker_status(self._h, clip, *[ctypes.byref(x) for x in v]))
        return dict(zip(("rows", "trackers", "overflow", "rows_overflow", "frame_count"), [x.value for x in v]))

    def rows(self, clip=0):
        """The dict of reference track.py:144-145 for one clip, in emission order."""
        ids = np.empty(self.rows_cap, np.int64)
        cols = np.empty((self.rows_cap, 7), np.float64)
        n = ctypes.c_int()
        _lib.check(_lib.lib().vbt_tracker_rows(self._h, clip, ids.ctypes.data, cols.ctypes.data, self.rows_cap, ctypes.byref(n)))
        k = n.value
        names = ("time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width")
        d = {"id": ids[:k].tolist()}
        for j, nm in enumerate(names):
            d[nm] = cols[:k, j].tolist()
        return d

    def finish(self, plate_diameter=0.45, diff_threshold=0.6, min_distance=0.1, stream=None):
        _lib.check(_lib.lib().vbt_tracker_finish(self._h, plate_diameter, diff_threshold, min_distance, stream))

    def phases(self, clip=0):
        ph = np.empty((512, 6), np.float64)
        best = ctypes.c_int32()
        n = ctypes.c_int()
        _lib.check(_lib.lib().vbt_tracker_phases(self._h, clip, ctypes.byref(best), ph.ctypes.data, 512, ctypes.byref(n)))
        return best.value, ph[:n.value].copy()


def _summary(self, cap=64):
    """(best_ids[n], n_rows[n], n_phases[n], overflow[n], phases[n, cap, 6]) of every clip after finish(), in one call."""
    n = self.n_clips
    best, rows, nph, ovf = (np.zeros(n, np.int32) for _ in range(4))
    ph = np.zeros((n, cap, 6), np.float64)
    _lib.check(_lib.lib().vbt_tracker_summary(self._h, best.ctypes.data, rows.ctypes.data, nph.ctypes.data, ovf.ctypes.data, ph.ctypes.data, cap))
    return best, rows, nph, ovf, ph


MultiClipTracker.summary = _summary

ROW_DTYPE = np.dtype([("id", "<i8"), ("time", "<f8"), ("x", "<f8"), ("y", "<f8"), ("dx", "<f8"), ("dy", "<f8"),
                      ("norm_plate_height", "<f8"), ("norm_plate_width", "<f8")])       # the 8 columns of reference track.py:227-234


def _rows_all(self, cap=None, out=None, stream=None):
    """Rows of every clip in one strided copy: (counts[n], rows[n, cap] of ROW_DTYPE).  `out`: optional preallocated
    (pinned) buffer of at least n * cap * 64 bytes exposing .data_ptr() (torch) or the numpy buffer protocol."""
    n = self.n_clips
    cap = int(cap or self.rows_cap)
    counts = np.zeros(n, np.int32)
    if out is None:
        rows = np.zeros((n, cap), ROW_DTYPE)
        ptr = rows.ctypes.data
    elif hasattr(out, "data_ptr"):
        rows = out.numpy().view(np.uint8).reshape(-1)[:n * cap * 64].view(ROW_DTYPE).reshape(n, cap)
        ptr = out.data_ptr()
    else:
        rows = np.frombuffer(out, np.uint8)[:n * cap * 64].view(ROW_DTYPE).reshape(n, cap)
        ptr = rows.ctypes.data
    _lib.check(_lib.lib().vbt_tracker_rows_all(self._h, counts.ctypes.data, ptr, cap, stream))
    return counts, rows


MultiClipTracker.rows_all = _rows_all


class OCSort:
    """Single-clip tracker with the reference's call shape (reference track.py:157,186-199)."""

    def __init__(self, det_thresh=0.2, max_age=30, min_hits=3, iou_threshold=0.3, delta_t=3, asso_func="iou", inertia=0.2,
                 device=0, rows_cap=65536):
        self._mc = MultiClipTracker(1, rows_cap, det_thresh, max_age, min_hits, iou_threshold, delta_t, asso_func, inertia, device)
        self.frame_count = 0
        self._time = 0.0

    def update(self, dets, _=None, time=None):
        dets = np.asarray(dets, dtype=np.float64).reshape(-1, 6)
        n = dets.shape[0]
        if n > MAXD:
            raise ValueError(f"at most {MAXD} detections per frame (got {n})")
        self.frame_count += 1
        if n == 0:                      # the reference never calls update with N = 0 (track.py:180-181)
            return np.empty((0, 7))
        buf = np.zeros((1, 1, MAXD, 6), np.float64)
        buf[0, 0, :n] = dets
        t = float(self.frame_count if time is None else time)
        self._mc.update_frames(buf, np.array([[n]], np.int32), np.array([[t]], np.float64))
        out, self._last_vel = self._mc.last_output(0)
        return out

    @property
    def trackers(self):
        return self._mc.trackers(0)

    @property
    def multi(self):
        return self._mc
