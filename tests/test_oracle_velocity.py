"""Pins oracle/velocity.py against the reference's own outputs (tests/golden, made by
tools/make_golden.py from reference dfs_ocsort/ + the imported reference VelocityTracker)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import velocity as ov

COLS = ["time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width"]


@pytest.fixture(scope="module")
def golden():
    main = np.load(os.path.join(GOLDEN, "dfs_ocsort_main.npz"))
    pre = np.load(os.path.join(GOLDEN, "pre_ocsort.npz"))
    with open(os.path.join(GOLDEN, "phases_ocsort.json")) as f:
        phases = json.load(f)
    return main, pre, phases


def test_running_average_known_answers():
    with open(os.path.join(GOLDEN, "running_average.json")) as f:
        seqs = json.load(f)
    for s in seqs:
        ra = ov.RunningAverage(s["window"])
        got = [ra.update(float.fromhex(v)) for v in s["in"]]
        assert got == [float.fromhex(v) for v in s["out"]]
    ra = ov.RunningAverage(3)                      # SURVEY.md section 4.5 known answer
    assert [ra.update(float(v)) for v in range(1, 7)] == [1.0, 1.5, 2.0, 3.0, 4.0, 5.0]


@pytest.mark.parametrize("clip", ["001", "005", "009", "030"])
def test_pandas_window_means_bit_exact(golden, clip):
    main, pre, _ = golden
    cols = [main[f"c{clip}_{c}"].tolist() for c in COLS]
    got = ov.preprocess(*cols)
    for c, g in zip(COLS, got):
        want = pre[f"c{clip}_{c}"]
        assert np.array_equal(np.asarray(g), want), c


def test_phases_equal_reference_on_all_34_clips(golden):
    main, _, phases = golden
    clips = sorted(k for k in phases if len(k) == 3)
    assert len(clips) == 34
    exact_counts = 0
    for clip in clips:
        cols = [main[f"c{clip}_{c}"].tolist() for c in COLS]
        got = ov.analyze_track(*cols, plate_diameter=0.45)
        want = phases[clip]["phases"]
        assert len(got) == len(want), clip
        for p, wrow in zip(got, want):
            w = [float.fromhex(v) for v in wrow[:5]]
            assert [p.time_start, p.time_end, p.y_start, p.y_end, p.rom] == w, clip
            assert p.type == wrow[5]
        exact_counts += sum(p.type == ov.CONCENTRIC for p in got) == phases[clip]["reps_in_name"]
    assert exact_counts == 32                     # 009 and 030 miss one rep


def test_clip001_acv_table(golden):
    """SURVEY.md section 4.3 / BASELINE.md: per-rep ACV of 001_squat_6reps, id 1."""
    main, _, _ = golden
    cols = [main[f"c001_{c}"].tolist() for c in COLS]
    ph = [p for p in ov.analyze_track(*cols) if p.type == ov.CONCENTRIC]
    acv = [p.rom / p.duration for p in ph]
    assert np.allclose(acv, [0.437798, 0.481084, 0.455869, 0.445342, 0.391556, 0.400236], atol=5e-7)
    assert np.allclose([p.rom for p in ph], [0.649400, 0.665499, 0.661010, 0.675435, 0.685224, 0.647049], atol=5e-7)


def test_edge_cases():
    vt = ov.VelocityTracker(0.45)
    vt.end_processing()
    assert vt.phases == []
    vt.process_measurements(0.0, 0.5, 0.5, 0, 0, 0.1, 0.1)     # single sample
    vt.end_processing()
    assert vt.phases == []
    assert ov.rolling_mean([], 5) == [] and ov.expanding_mean([]) == []
    # constant signal: pandas returns the repeated value itself (num_consecutive_same_value rule)
    assert ov.rolling_mean([0.1] * 9, 5) == [0.1] * 9
