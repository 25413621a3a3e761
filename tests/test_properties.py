"""Property tests (hypothesis) for the oracle building blocks: SURVEY.md section 4.5."""
import numpy as np
import pandas as pd
from hypothesis import given, settings, strategies as st

from oracle import ocsort_np as oc
from oracle import velocity as ov

coord = st.floats(min_value=-0.5, max_value=1.5, allow_nan=False, allow_infinity=False, width=32)
size = st.floats(min_value=0.015625, max_value=0.75, allow_nan=False, allow_infinity=False, width=32)


@st.composite
def boxes(draw, n_min=1, n_max=6):
    n = draw(st.integers(n_min, n_max))
    out = []
    for _ in range(n):
        x, y, w, h = draw(coord), draw(coord), draw(size), draw(size)
        out.append([x, y, x + w, y + h])
    return np.asarray(out, np.float64)


@settings(max_examples=150, deadline=None)
@given(boxes(), boxes())
def test_iou_diou_symmetry_and_bounds(a, b):
    i1, i2 = oc.iou_batch(a, b), oc.iou_batch(b, a)
    assert np.array_equal(i1, i2.T)                       # bitwise symmetric (max/min and a+b-c are commutative)
    assert np.all(i1 >= 0) and np.all(i1 <= 1 + 1e-12)
    d1, d2 = oc.diou_batch(a, b), oc.diou_batch(b, a)
    assert np.allclose(d1, d2.T, atol=1e-15) and np.all(d1 >= 0) and np.all(d1 <= 1 + 1e-12)
    self_iou = np.diag(oc.iou_batch(a, a))
    assert np.allclose(self_iou, 1.0)
    assert np.all(d1 <= (i1 + 1) / 2 + 1e-15)             # the centre-distance penalty only lowers the score


@settings(max_examples=100, deadline=None)
@given(st.integers(1, 6), st.integers(1, 6), st.integers(0, 2 ** 31 - 1))
def test_linear_assignment_optimal_and_injective(n, m, seed):
    rng = np.random.default_rng(seed)
    c = np.round(rng.uniform(-1, 0, (n, m)), 2) * (rng.random((n, m)) < 0.6)      # ties on purpose
    got = oc.linear_assignment(c)
    assert len(got) == min(n, m) and len(set(got[:, 0])) == len(got) and len(set(got[:, 1])) == len(got)
    import itertools
    if n <= m:
        best = min(sum(c[i, p[i]] for i in range(n)) for p in itertools.permutations(range(m), n))
    else:
        best = min(sum(c[p[j], j] for j in range(m)) for p in itertools.permutations(range(n), m))
    assert abs(sum(c[i, j] for i, j in got) - best) < 1e-12


@settings(max_examples=60, deadline=None)
@given(st.lists(st.floats(min_value=-1.0, max_value=1.0, allow_nan=False, width=64), min_size=1, max_size=60), st.integers(1, 8))
def test_window_means_equal_pandas(values, window):
    s = pd.Series(values, dtype=np.float64)
    assert np.array_equal(np.asarray(ov.rolling_mean(values, window)), s.rolling(window=window, center=False, min_periods=1).mean().to_numpy())
    assert np.array_equal(np.asarray(ov.expanding_mean(values)), s.expanding(min_periods=1).mean().to_numpy())


@settings(max_examples=40, deadline=None)
@given(st.integers(0, 2 ** 31 - 1))
def test_tracker_ids_monotone_and_rows_consistent(seed):
    rng = np.random.default_rng(seed)
    trk = oc.OCSort(max_age=5, asso_func="diou", iou_threshold=0.1)
    seen = 0
    for f in range(25):
        n = int(rng.integers(1, 5))
        c = rng.uniform(0.2, 0.8, (n, 2))
        s = rng.uniform(0.05, 0.2, (n, 2))
        d = np.concatenate([c - s / 2, c + s / 2, rng.uniform(0.5, 1.0, (n, 1)), np.zeros((n, 1))], axis=1)
        out = trk.update(d, [])
        ids = [t.id for t in trk.trackers]
        assert ids == sorted(ids) and len(set(ids)) == len(ids)          # list order = creation order
        assert len(out) <= n and all(1 <= r[4] <= trk._count for r in out)
        assert list(out[:, 4]) == sorted(out[:, 4], reverse=True)         # rows come out in reverse tracker order
        seen = max(seen, trk._count)
    assert seen >= 1


def test_velocity_tracker_is_invariant_to_dx_and_first_dy_only():
    """Reference quirk (VelocityTracker.py:101-102): incoming dx is ignored, dy only matters on the first sample."""
    rng = np.random.default_rng(0)
    t = np.arange(400) / 30.0
    y = 0.5 + 0.2 * np.sin(t * 2.0) + rng.normal(0, 1e-4, len(t))
    rows = [(t[i], 0.5, y[i], 0.0, 0.0, 0.16, 0.28) for i in range(len(t))]
    a, b = ov.VelocityTracker(0.45), ov.VelocityTracker(0.45)
    for i, r in enumerate(rows):
        a.process_measurements(*r)
        b.process_measurements(r[0], r[1], r[2], 123.0, (r[4] if i == 0 else -9.0), r[5], r[6])
    a.end_processing(); b.end_processing()
    assert [p.as_row() for p in a.phases] == [p.as_row() for p in b.phases] and len(a.phases) >= 4
