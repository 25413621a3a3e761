"""Trajectory validation against Kinovea / Qualisys exports.

Mirrors the numeric part of the reference's accuracy study, without the figures:
  reference kinovea.py:73-172   Kinovea text export vs a tracked DataFrame
  reference qualysis.py:79-187  Qualisys .tsv export (marker "Osa L", X and Z) vs a tracked DataFrame
Steps: parse the export; smooth the tracked rows (window means on the GPU, `vbt_window_means`); convert normalised
image coordinates to metres with the plate diameter; shift both axes so the means agree; resample both trajectories
linearly onto a 30 Hz grid over the common time span; report MSE and Pearson r per axis.
"""
from __future__ import annotations

import math

import numpy as np

from . import _lib


def read_kinovea(path):
    """Kinovea 'trajectory data export': `#` comment lines, then `time x y` separated by blanks with decimal commas in
    x / y (centimetres).  Returns [N,3] float64 time [s], x [m], y [m]   (reference kinovea.py:73-88)."""
    rows = []
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            p = line.split(" ")
            rows.append((float(p[0]), float(p[1].replace(",", ".")) / 100.0, float(p[2].replace(",", ".")) / 100.0))
    return np.asarray(rows, dtype=np.float64).reshape(-1, 3)


def read_qualisys(path, marker="Osa L"):
    """Qualisys .tsv: 11 header lines, a tab separated column-name line, then frames.  x = -X/1000, y = Z/1000 of the
    marker [m]   (reference qualysis.py:79-108).  Returns [N,3] float64 time, x, y."""
    with open(path) as f:
        lines = f.read().splitlines()
    names = lines[11].rstrip("\t").split("\t")
    it, ix, iz = names.index("Time"), names.index(f"{marker} X"), names.index(f"{marker} Z")
    out = []
    for line in lines[12:]:
        if not line.strip():
            continue
        p = line.split("\t")
        out.append((float(p[it]), -float(p[ix]) / 1000.0, float(p[iz]) / 1000.0))
    return np.asarray(out, dtype=np.float64).reshape(-1, 3)


def window_means(table, windows, device=0):
    """pandas rolling(window, min_periods=1).mean() (window > 0) / expanding().mean() (0) / copy (< 0) per column,
    computed on the GPU."""
    table = np.ascontiguousarray(table, dtype=np.float64)
    T, nc = table.shape
    w = np.ascontiguousarray(windows, dtype=np.int32)
    if w.shape != (nc,):
        raise ValueError("one window per column expected")
    out = np.empty_like(table)
    _lib.check(_lib.lib().vbt_window_means(table.ctypes.data, T, nc, w.ctypes.data, out.ctypes.data, device))
    return out


def metric_trajectory(rows, ref, plate_diameter=0.45, source="kinovea", device=0):
    """rows [T,5] = time, x, y, norm_plate_height, norm_plate_width of ONE track (sorted by time) -> [T,3]
    time, x, y in metres, aligned to `ref` ([N,3] time, x, y).
    kinovea: expanding mean of the plate size, rolling(5) of x and y   (reference kinovea.py:99-116)
    qualisys: rolling(30) of the plate size, raw x and y               (reference qualysis.py:113-131)"""
    rows = np.asarray(rows, dtype=np.float64)
    if source == "kinovea":
        sm = window_means(rows, [-1, 5, 5, 0, 0], device)
    elif source == "qualisys":
        sm = window_means(rows, [-1, -1, -1, 30, 30], device)
    else:
        raise ValueError("source must be 'kinovea' or 'qualisys'")
    x = sm[:, 1] * plate_diameter / sm[:, 4]
    y = -sm[:, 2] * plate_diameter / sm[:, 3]             # image y grows downwards
    y = y + (_mean(ref[:, 2]) - _mean(y))
    x = x + (_mean(ref[:, 1]) - _mean(x))
    return np.stack([rows[:, 0], x, y], axis=1)


def _mean(v):
    # pandas Series.mean(): bottleneck-free nanmean = pairwise np.sum / count
    return np.sum(v) / v.size


def _interp_linear(t, v, ts):
    """scipy.interpolate.interp1d(kind='linear') arithmetic: slope * (t_new - t_lo) + v_lo on the bracketing pair."""
    order = np.argsort(t, kind="mergesort")
    t, v = t[order], v[order]
    if ts.size and (ts.min() < t[0] or ts.max() > t[-1]):
        raise ValueError("A value in x_new is outside the interpolation range.")
    hi = np.clip(np.searchsorted(t, ts), 1, len(t) - 1)
    lo = hi - 1
    slope = (v[hi] - v[lo]) / (t[hi] - t[lo])
    return slope * (ts - t[lo]) + v[lo]


def _pearson(a, b):
    """scipy.stats.pearsonr statistic (normalise the centred vectors, then dot)."""
    am = a - a.mean()
    bm = b - b.mean()
    na, nb = np.linalg.norm(am), np.linalg.norm(bm)
    return float(max(min(np.dot(am / na, bm / nb), 1.0), -1.0))


def compare(ref, traj, rate=30):
    """-> dict(mse_x, mse_y, r_x, r_y, n) on a `rate` Hz grid over the common time span
    (reference kinovea.py:151-172, qualysis.py:166-187; the grid has int(t_max * 30) points)."""
    t_max = min(ref[:, 0].max(), traj[:, 0].max())
    t_min = max(ref[:, 0].min(), traj[:, 0].min())
    ts = np.linspace(t_min, t_max, int(t_max * rate))
    xr, xm = _interp_linear(ref[:, 0], ref[:, 1], ts), _interp_linear(traj[:, 0], traj[:, 1], ts)
    yr, ym = _interp_linear(ref[:, 0], ref[:, 2], ts), _interp_linear(traj[:, 0], traj[:, 2], ts)
    return {"mse_x": float(np.mean((xr - xm) ** 2)), "mse_y": float(np.mean((yr - ym) ** 2)),
            "r_x": _pearson(xr, xm), "r_y": _pearson(yr, ym), "n": int(ts.size)}


def validate_pair(ref, rows, plate_diameter=0.45, source="kinovea", device=0):
    if len(rows) < 2 or not math.isfinite(float(np.sum(rows))):
        raise ValueError("tracked rows are empty or not finite")
    return compare(ref, metric_trajectory(rows, ref, plate_diameter, source, device))
