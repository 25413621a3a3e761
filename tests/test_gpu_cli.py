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


@pytest.mark.parametrize("hw,stride", [((320, 320), 1), ((700, 96), 3)])
def test_c_host_tracks_a_clip_like_the_python_wrapper(tmp_path, model_path, hw, stride):
    """examples/track_clip.c (plain C over include/vbt_hip.h: pinned frames from vbt_host_alloc -> vbt_pipeline_create -> vbt_track_clip)
    prints the same DataFrame rows, to the last bit, as vbt_amd.track.track_frames on the same clip - at the network resolution and at
    a source resolution with a frame stride (row-pair upload + on-device resize, reference odt.py:10-19, track.py:166)."""
    import subprocess
    from test_abi_and_host import _build_c_host
    from vbt_amd import synth
    from vbt_amd.track import COLUMNS, track_frames
    T = 40
    H, W = hw
    if hw == (320, 320):
        frames = synth.clip_frames(21, 0, T)
    else:
        rng = np.random.default_rng(3)
        frames = np.repeat(np.repeat(rng.integers(0, 256, (T, H // 4, W // 4, 3), dtype=np.uint8), 4, 1), 4, 2)
    raw = tmp_path / "clip.raw"
    np.ascontiguousarray(frames).tofile(raw)
    exe = _build_c_host(tmp_path)
    p = subprocess.run([exe, model_path, str(raw), str(T), str(H), str(W), "30", str(stride)], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = p.stdout.strip().split("\n")
    assert lines[0] == ",".join(COLUMNS)
    got = [ln.split(",") for ln in lines[1:]]
    want = track_frames(frames, model_path, fps=30.0, frame_stride=stride, time_batch=64)
    assert len(got) == len(want["id"])
    for i, row in enumerate(got):
        assert int(row[0]) == want["id"][i]
        for j, k in enumerate(COLUMNS[1:]):
            assert float(row[1 + j]) == want[k][i], (i, k)
    if hw == (320, 320):
        assert len(got) > 5
