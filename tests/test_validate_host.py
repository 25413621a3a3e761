"""Host half of the trajectory validation study (N4; reference kinovea.py:73-172, qualysis.py:79-187): export parsers
and the resample / MSE / Pearson arithmetic against numbers produced by the pandas / scipy / sklearn calls the
reference makes (tools/make_golden_validation.py).  The smoothing half runs on the GPU: tests/test_gpu_validate.py."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from vbt_amd import validate as V


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "validation.npz"))


def test_kinovea_parser(gold):
    got = V.read_kinovea(os.path.join(GOLDEN, "kinovea_sample.txt"))
    assert got.shape == gold["k029_ref"].shape and np.array_equal(got, gold["k029_ref"])


def test_qualisys_parser(gold):
    got = V.read_qualisys(os.path.join(GOLDEN, "qualisys_sample.tsv"))
    assert got.shape == (400, 3) and np.array_equal(got, gold["q_squat1_ref"][:400])


def test_compare_matches_scipy_sklearn(gold):
    keys = sorted(k[:-6] for k in gold.files if k.endswith("_stats"))
    assert len(keys) == 37                                   # 32 kinovea clips + 5 qualisys clips
    for k in keys:
        ref, xy, exp = gold[k + "_ref"], gold[k + "_xy"], gold[k + "_stats"]
        t = gold[k + "_rows"][:, 0] if k.startswith("q_") else None
        if t is None:
            main = np.load(os.path.join(GOLDEN, "dfs_ocsort_main.npz"))
            t = main[f"c{k[1:]}_time"]
        r = V.compare(ref, np.column_stack([t, xy]))
        np.testing.assert_allclose([r["mse_x"], r["mse_y"], r["r_x"], r["r_y"]], exp[:4], rtol=1e-12, atol=0)


def test_interp_refuses_extrapolation():
    t = np.array([0.0, 1.0, 2.0])
    with pytest.raises(ValueError):
        V._interp_linear(t, t, np.array([2.5]))
