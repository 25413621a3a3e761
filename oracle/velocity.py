"""ORACLE - TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

CPU restatement of the rep-analysis half of the reference hot path:
  RunningAverage           reference RunningAverage.py:9-27
  Phase                    reference Phase.py:6-40
  VelocityTracker          reference VelocityTracker.py:15-230
  plot.py preprocessing    reference plot.py:87-95 (rolling(5)/expanding means; the arithmetic is
                           pandas' window aggregation `roll_mean` [EXTERNAL, pandas 2.x
                           _libs/window/aggregations.pyx: Kahan-compensated add/remove])
  analyze_df               reference plot.py:33-47
Pinned: tests/test_oracle_velocity.py checks it against the phases the reference's own
VelocityTracker produced over all 34 reference dfs_ocsort clips (tests/golden/phases_ocsort.json,
exact float equality), the pandas-preprocessed columns (tests/golden/pre_ocsort.npz) and the
RunningAverage known answers.  Bug-compatible on purpose:
one RunningAverage(30) fed width then height; incoming dx ignored; dy replaced by y - y_prev.
"""
from __future__ import annotations

import math
from collections import deque

CONCENTRIC, ECCENTRIC, HOLD = 0, 1, 2          # reference Phase.py:12-14
START_THRESHOLD = 3                            # reference VelocityTracker.py:11
END_THRESHOLD = 1                              # reference VelocityTracker.py:12


class RunningAverage:
    def __init__(self, window_size):
        self.window_size = window_size
        self.window = deque()
        self.total = 0.0
        self.count = 0

    def update(self, value):
        self.window.append(value)
        self.total += value
        self.count += 1
        if self.count >= self.window_size:
            avg = self.total / self.window_size
            self.total -= self.window.popleft()
            self.count -= 1
            return avg
        return self.total / self.count


class Phase:
    __slots__ = ("time_start", "time_end", "y_start", "y_end", "rom", "type")

    def __init__(self, time_start, time_end, y_start, y_end, rom, phase_type):
        self.time_start, self.time_end = time_start, time_end
        self.y_start, self.y_end = y_start, y_end
        self.rom, self.type = rom, phase_type

    @property
    def y_diff(self):
        return abs(self.y_start - self.y_end)

    @property
    def duration(self):
        return self.time_end - self.time_start

    def as_row(self):
        return [self.time_start, self.time_end, self.y_start, self.y_end, self.rom, float(self.type)]


class VelocityTracker:
    def __init__(self, plate_diameter, diff_threshold=0.6, min_distance=0.1):
        self.plate_diameter = plate_diameter
        self.min_distance = min_distance
        self.diff_threshold = diff_threshold
        self.current_phase = HOLD
        self.phases = []
        self.max_y_diff = None
        self.y_prev = None
        self.xs, self.ys, self.widths, self.heights, self.times = [], [], [], [], []
        self.avg = RunningAverage(30)          # the reference creates width_avg twice; only one exists
        self.neg = 0
        self.pos = 0

    def _filter(self):
        thr = self.max_y_diff / 2
        self.phases = [p for p in self.phases if not (p.y_diff < thr)]

    def _append(self, x, y, w, h, t):
        self.xs.append(x); self.ys.append(y); self.widths.append(w); self.heights.append(h); self.times.append(t)

    def _reset(self):
        self.xs, self.ys, self.widths, self.heights, self.times = [], [], [], [], []

    def process_measurements(self, time, x, y, dx, dy, norm_plate_height, norm_plate_width):
        width = self.avg.update(norm_plate_width)
        height = self.avg.update(norm_plate_height)
        if self.y_prev is not None:
            dy = y - self.y_prev
        if self.current_phase != HOLD:
            self._append(x, y, width, height, time)
        if self.current_phase == CONCENTRIC:
            if dy > 0:
                self.pos += 1
                self.neg = 0
                if self.pos >= END_THRESHOLD:
                    self._end_phase()
            else:
                self.pos = 0
        if self.current_phase == ECCENTRIC:
            if dy < 0:
                self.neg += 1
                self.pos = 0
                if self.neg >= END_THRESHOLD:
                    self._end_phase()
            else:
                self.neg = 0
                self.pos += 1
        if dy < 0 and self.current_phase == HOLD:
            self.neg += 1
            self.pos = 0
            if self.neg == 1:
                self._reset()
            else:
                self._append(x, y, width, height, time)
            if self.neg >= START_THRESHOLD:
                self.current_phase, self.pos, self.neg = CONCENTRIC, 0, 0
        if dy > 0 and self.current_phase == HOLD:
            self.pos += 1
            self.neg = 0
            if self.pos == 1:
                self._reset()
            else:
                self._append(x, y, width, height, time)
            if self.pos >= START_THRESHOLD:
                self.current_phase, self.pos, self.neg = ECCENTRIC, 0, 0
        self.y_prev = y

    @staticmethod
    def _argmax(v):
        best = 0
        for i in range(1, len(v)):
            if v[i] > v[best]:
                best = i
        return best

    @staticmethod
    def _argmin(v):
        best = 0
        for i in range(1, len(v)):
            if v[i] < v[best]:
                best = i
        return best

    def _end_phase(self):
        if self.current_phase == CONCENTRIC:
            s, e = self._argmax(self.ys), self._argmin(self.ys)
        else:
            s, e = self._argmin(self.ys), self._argmax(self.ys)
        y_diff = abs(self.ys[s] - self.ys[e])
        if self.max_y_diff is None or y_diff > self.max_y_diff:
            self.max_y_diff = y_diff
            self._filter()
        if y_diff > self.max_y_diff * self.diff_threshold:
            distance = 0
            for i in range(s + 1, e + 1):
                ddx = abs(self.xs[i] - self.xs[i - 1]) / ((self.widths[i] + self.widths[i - 1]) / 2) * self.plate_diameter
                ddy = abs(self.ys[i] - self.ys[i - 1]) / ((self.heights[i] + self.heights[i - 1]) / 2) * self.plate_diameter
                distance += ddx + ddy
            if distance < self.min_distance:
                self.neg = self.pos = 0
                self.current_phase = HOLD
                return
            self.phases.append(Phase(self.times[s], self.times[e], self.ys[s], self.ys[e], distance, self.current_phase))
            self._filter()
        self.current_phase = HOLD
        self.pos = self.neg = 0

    def end_processing(self):
        if self.current_phase != HOLD:
            self._end_phase()


# ---- pandas window means, restated (Kahan add/remove, exactly pandas' op order) ----------------
class _RollMean:
    def __init__(self):
        self.nobs = 0
        self.sum = 0.0
        self.neg = 0
        self.c_add = 0.0
        self.c_rem = 0.0
        self.same = 0
        self.prev = math.nan

    def add(self, v):
        self.nobs += 1
        y = v - self.c_add
        t = self.sum + y
        self.c_add = t - self.sum - y
        self.sum = t
        if math.copysign(1.0, v) < 0:
            self.neg += 1
        if v == self.prev:
            self.same += 1
        else:
            self.same = 1
        self.prev = v

    def remove(self, v):
        self.nobs -= 1
        y = -v - self.c_rem
        t = self.sum + y
        self.c_rem = t - self.sum - y
        self.sum = t
        if math.copysign(1.0, v) < 0:
            self.neg -= 1

    def mean(self):
        r = self.sum / float(self.nobs)
        if self.same >= self.nobs:
            r = self.prev
        elif self.neg == 0 and r < 0:
            r = 0.0
        elif self.neg == self.nobs and r > 0:
            r = 0.0
        return r


def rolling_mean(values, window):
    """Series.rolling(window, center=False, min_periods=1).mean() for NaN-free input."""
    st, out = _RollMean(), []
    for i, v in enumerate(values):
        if i >= window:
            st.remove(values[i - window])
        st.add(v)
        out.append(st.mean())
    return out


def expanding_mean(values):
    """Series.expanding(min_periods=1).mean() for NaN-free input."""
    st, out = _RollMean(), []
    for v in values:
        st.add(v)
        out.append(st.mean())
    return out


def preprocess(time, x, y, dx, dy, h, w):
    """reference plot.py:90-95 on the rows of one track id."""
    return (list(time), rolling_mean(x, 5), rolling_mean(y, 5), rolling_mean(dx, 5), rolling_mean(dy, 5),
            expanding_mean(h), expanding_mean(w))


def analyze(time, x, y, dx, dy, h, w, plate_diameter=0.45):
    """reference plot.py:33-47 (analyze_df) on preprocessed columns -> list[Phase]."""
    vt = VelocityTracker(plate_diameter)
    for i in range(len(time)):
        vt.process_measurements(time[i], x[i], y[i], dx[i], dy[i], h[i], w[i])
    vt.end_processing()
    return vt.phases


def analyze_track(time, x, y, dx, dy, h, w, plate_diameter=0.45):
    return analyze(*preprocess(time, x, y, dx, dy, h, w), plate_diameter=plate_diameter)
