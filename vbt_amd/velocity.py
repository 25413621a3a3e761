"""`VelocityTracker` / `Phase`: the rep analyser of the reference, computed on the GPU.

Mirrors reference VelocityTracker.py:15-230 and Phase.py:6-40 as used by plot.py:33-47:
  VelocityTracker(plate_diameter, diff_threshold=0.6, min_distance=0.1)
  .process_measurements(time, x, y, dx, dy, norm_plate_height, norm_plate_width)
  .end_processing()
  .phases -> objects with time_start,time_end,y_start,y_end,rom,type,y_diff,duration
The state machine is strictly sequential per track, so samples are buffered on the host and the
scan runs as one device launch (vbt_analyze) when `.phases` is read / `end_processing()` is called.
"""
import ctypes

import numpy as np

from . import _lib


class Phase:
    CONCENTRIC = 0      # reference Phase.py:12-14
    ECCENTRIC = 1
    HOLD = 2

    def __init__(self, time_start, time_end, y_start, y_end, rom, phase_type):
        self.time_start, self.time_end = time_start, time_end
        self.y_start, self.y_end = y_start, y_end
        self.type = phase_type
        self.rom = rom

    @property
    def y_diff(self):
        return abs(self.y_start - self.y_end)

    @property
    def duration(self):
        return self.time_end - self.time_start

    def __str__(self):
        name = {0: "concentric", 1: "eccentric"}.get(self.type, "hold")
        return f"{name}, t_start: {self.time_start}, t_end: {self.time_end}, y_start: {self.y_start}, y_end: {self.y_end}"


def analyze_rows(rows7, plate_diameter=0.45, diff_threshold=0.6, min_distance=0.1, preprocess=True, flush=True, device=0):
    """rows7 [T,7] = time,x,y,dx,dy,h,w of one track -> list[Phase].  preprocess=True applies the
    rolling(5)/expanding means of reference plot.py:90-95 on the device first (analyze_df input)."""
    rows7 = np.ascontiguousarray(rows7, np.float64).reshape(-1, 7)
    ph = np.empty((512, 6), np.float64)
    n = ctypes.c_int()
    _lib.check(_lib.lib().vbt_analyze(rows7.ctypes.data if len(rows7) else None, len(rows7), int(preprocess), int(flush),
                                      float(plate_diameter), float(diff_threshold), float(min_distance), ph.ctypes.data, 512,
                                      ctypes.byref(n), device))
    return [Phase(r[0], r[1], r[2], r[3], r[4], int(r[5])) for r in ph[:n.value]]


class VelocityTracker:
    def __init__(self, plate_diameter, diff_threshold=0.6, min_distance=0.1, device=0):
        self.plate_diameter = plate_diameter
        self.min_distance = min_distance
        self.diff_threshold = diff_threshold
        self.device = device
        self._rows = []
        self._flushed = False
        self._cache = None

    def process_measurements(self, time, x, y, dx, dy, norm_plate_height, norm_plate_width):
        if self._flushed:
            raise RuntimeError("process_measurements after end_processing")
        self._rows.append((time, x, y, dx, dy, norm_plate_height, norm_plate_width))
        self._cache = None

    def end_processing(self):
        self._flushed = True
        self._cache = None

    @property
    def phases(self):
        if self._cache is None:
            self._cache = analyze_rows(np.asarray(self._rows, np.float64).reshape(-1, 7), self.plate_diameter, self.diff_threshold,
                                       self.min_distance, preprocess=False, flush=self._flushed, device=self.device)
        return self._cache


def analyze_df(df, plate_diameter):
    """reference plot.py:33-47: df has columns time,x,y,dx,dy,norm_plate_height,norm_plate_width (already preprocessed)."""
    cols = ["time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width"]
    return analyze_rows(np.stack([np.asarray(df[c], np.float64) for c in cols], axis=1), plate_diameter, preprocess=False)
