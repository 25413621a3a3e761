"""CLI mirror of reference track.py main() / plot.py main() (N3): export naming + DataFrame contract, rep table."""
import os

import numpy as np
import pytest
from click.testing import CliRunner

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def test_analyze_command_reproduces_reference_acv_table(tmp_path):
    import pandas as pd
    from vbt_amd.cli import main
    full = np.load(os.path.join(GOLDEN, "dfs_ocsort_full.npz"))
    cols = ["id", "time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width"]
    df = pd.DataFrame({c: full[f"c001_{c}"] for c in cols}, index=full["c001_index"])
    path = tmp_path / "001_squat_6reps_id1_efficientdet_lite0_whole.pkl.gz"
    df.to_pickle(str(path))
    res = CliRunner().invoke(main, ["analyze", str(path)])
    assert res.exit_code == 0, res.output
    assert "12 phases, 6 concentric reps" in res.output
    for acv in ("0.437798", "0.481084", "0.455869", "0.445342", "0.391556", "0.400236"):
        assert f"ACV {acv}" in res.output
    res = CliRunner().invoke(main, ["analyze", str(tmp_path / "badname.pkl.gz")])
    assert res.exit_code != 0                                                                  # FileNotFoundError like the reference


def test_track_command_exports_reference_style_dataframe(tmp_path, model_path):
    import pandas as pd
    from vbt_amd import synth
    from vbt_amd.cli import main
    frames = synth.clip_frames(12, 0, 12, size=416)            # source resolution != network resolution
    src = tmp_path / "demo_clip.npy"
    np.save(str(src), frames)
    out = tmp_path / "dfs"
    res = CliRunner().invoke(main, ["track", str(src), "--model", model_path, "--df_dir", str(out), "--fps", "60", "--detection_treshold", "0.3"])
    assert res.exit_code == 0, res.output
    files = os.listdir(out)
    assert len(files) == 1 and files[0].startswith("demo_clip_id") and files[0].endswith("_efficientdet_lite0_synth.pkl.gz")
    df = pd.read_pickle(os.path.join(out, files[0]))
    assert list(df.columns) == ["id", "time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width"]
    assert df["id"].dtype == np.int64 and df["time"].dtype == np.float64
    assert df.equals(df.sort_values(by=["id", "time"]))


@pytest.mark.parametrize("stride", [2, 3, 16])
def test_track_command_frame_stride_equals_track_function(tmp_path, model_path, stride):
    """`--frame_stride` (reference track.py:166 `frame_count % 16`) through the CLI's Pipeline path must give the rows of
    the reference-shaped `track(frames, it, frame_stride=...)`: skipped frames advance time, not the pipeline's ring slot
    (an even stride with depth 2 used to land every processed frame in the same slot)."""
    import pandas as pd
    from vbt_amd import synth
    from vbt_amd.cli import main
    from vbt_amd.interpreter import Interpreter
    from vbt_amd.track import COLUMNS, track
    T = 5 * stride + 3 if stride < 16 else 84
    frames = synth.clip_frames(5, 0, T)
    src = tmp_path / "stride_clip.npy"
    np.save(str(src), frames)
    out = tmp_path / "dfs"
    res = CliRunner().invoke(main, ["track", str(src), "--model", model_path, "--df_dir", str(out), "--fps", "30", "--detection_treshold", "0.3",
                                    "--frame_stride", str(stride)])
    assert res.exit_code == 0, res.output
    it = Interpreter(model_path=model_path)
    want = track(frames, it, detection_treshold=0.3, fps=30.0, frame_stride=stride)
    assert len(want["id"]) > 0
    files = os.listdir(out)
    assert len(files) == 1
    df = pd.read_pickle(os.path.join(out, files[0]))
    wdf = pd.DataFrame.from_dict(want).sort_values(by=["id", "time"])
    assert len(df) == len(wdf)
    for c in COLUMNS:
        assert np.array_equal(df[c].to_numpy(), wdf[c].to_numpy()), c
