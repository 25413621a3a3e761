"""GPU half of the validation study (N4): `vbt_window_means` bit-exact against pandas, and the whole
kinovea.py / qualysis.py comparison against the numbers the reference's library calls produce."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLDEN, "validation.npz"))


def test_window_means_equal_pandas(gold):
    import pandas as pd
    from vbt_amd.validate import window_means
    rows = gold["q_squat3_rows"]
    got = window_means(rows, [-1, 5, 0, 30, 30])
    assert np.array_equal(got[:, 0], rows[:, 0])
    assert np.array_equal(got[:, 3:], gold["q_squat3_hw30"])
    assert np.array_equal(got[:, 1], pd.Series(rows[:, 1]).rolling(window=5, center=False, min_periods=1).mean().to_numpy())
    assert np.array_equal(got[:, 2], pd.Series(rows[:, 2]).expanding(min_periods=1).mean().to_numpy())
    rng = np.random.Generator(np.random.PCG64(5))
    t = rng.normal(0.0, 1.0, (777, 6)) * np.array([1e-3, 1.0, 1e3, 1.0, 1.0, 1.0])
    t[100:140, 3] = 0.25                                   # constant run (pandas' `same value` shortcut)
    t[:, 4] = np.abs(t[:, 4])                              # never negative (pandas clamps a negative mean to 0)
    w = [1, 2, 7, 30, 0, 1000]
    got = window_means(t, w)
    for c, k in enumerate(w):
        s = pd.Series(t[:, c])
        exp = (s.expanding(min_periods=1) if k == 0 else s.rolling(window=k, center=False, min_periods=1)).mean().to_numpy()
        assert np.array_equal(got[:, c], exp), (c, k)
    assert window_means(np.empty((0, 3)), [1, 2, 3]).shape == (0, 3)


def test_validation_tables(gold):
    """All 32 Kinovea clips (tracked rows: dfs_ocsort) and the 5 Qualisys clips (qualysis_dfs)."""
    from vbt_amd import validate as V
    main = np.load(os.path.join(GOLDEN, "dfs_ocsort_main.npz"))
    keys = sorted(k[:-6] for k in gold.files if k.endswith("_stats"))
    tot = np.zeros(2)
    for k in keys:
        if k.startswith("q_"):
            rows, src = gold[k + "_rows"], "qualisys"
        else:
            c = "c" + k[1:]
            rows = np.stack([main[f"{c}_{n}"] for n in ("time", "x", "y", "norm_plate_height", "norm_plate_width")], axis=1)
            src = "kinovea"
        traj = V.metric_trajectory(rows, gold[k + "_ref"], 0.45, src)
        np.testing.assert_allclose(traj[:, 1:], gold[k + "_xy"], rtol=1e-13, atol=1e-15)
        r = V.compare(gold[k + "_ref"], traj)
        np.testing.assert_allclose([r["mse_x"], r["mse_y"], r["r_x"], r["r_y"]], gold[k + "_stats"][:4], rtol=1e-10)
        if src == "kinovea":
            tot += [r["mse_x"], r["mse_y"]]
    assert r["r_y"] > 0.98 and tot[1] < 0.05


def test_validate_command(gold, tmp_path):
    import pandas as pd
    import shutil
    from click.testing import CliRunner
    from vbt_amd.cli import main
    rows = gold["q_squat1_rows"]
    df = pd.DataFrame({"id": np.full(len(rows), 23), "time": rows[:, 0], "x": rows[:, 1], "y": rows[:, 2], "dx": 0.0, "dy": 0.0,
                       "norm_plate_height": rows[:, 3], "norm_plate_width": rows[:, 4]})
    dfs = tmp_path / "dfs"
    dfs.mkdir()
    df.to_pickle(str(dfs / "squat1_mobile_side_6reps_id23_efficientdet_lite0_whole.pkl.gz"))
    ex = tmp_path / "exports"
    ex.mkdir()
    shutil.copy(os.path.join(GOLDEN, "qualisys_sample.tsv"), str(ex / "squat1.tsv"))
    (ex / "other.tsv").write_text(open(os.path.join(GOLDEN, "qualisys_sample.tsv")).read())
    res = CliRunner().invoke(main, ["validate", "--qualysis_dir", str(ex), "--df_dir", str(dfs)])
    assert res.exit_code == 0, res.output
    assert "No matching df file found for" in res.output and "squat1_mobile_side_6reps: MSEx" in res.output and "Total MSEx = " in res.output
    res = CliRunner().invoke(main, ["validate", "--df_dir", str(dfs)])
    assert res.exit_code != 0
