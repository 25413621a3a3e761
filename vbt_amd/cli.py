"""Command line mirror of the reference's two entry points, minus GUI/plotting:

  python -m vbt_amd.cli track SRC... [--model M] [--detection_treshold 0.5] [--df_dir DIR] [--fps 30] [--frame_stride 1]
      reference track.py:65-126.  SRC = .npy stack of RGB uint8 frames [T,H,W,3] (cv2 / video decode is not a
      dependency here); any source resolution (resized on the GPU like odt.py:10-19).  Writes
      {video}_id{N}_{model}.pkl.gz with the reference's columns, sort order and retained row labels.
  python -m vbt_amd.cli analyze DF.pkl.gz... [--plate_diameter 0.45]
      reference plot.py:50-70,73-95,163-173 without the figure: parses {video}_id{N}_{model}.pkl.gz, applies the
      rolling(5)/expanding preprocessing and the VelocityTracker on the GPU, prints ROM and ACV per concentric rep.
  python -m vbt_amd.cli validate [--kinovea_dir D | --qualysis_dir D] [--df_dir dfs] [--plate_diameter 0.45]
      reference kinovea.py:29-38,57-172,203-215 / qualysis.py:29-38,57-187 without the figures: per export with a
      matching {video}_id{N}_{model}.pkl.gz, MSE and Pearson r of x(t) and y(t) in metres, then the totals line.
Option names (including the reference's `treshold` / `qualysis` spellings) and defaults follow the reference.
"""
import os
import re

import click
import numpy as np

FILENAME_RE = re.compile(r"(\S*)_id(\d+)_(\S*)\.pkl\.gz")           # reference plot.py:19-25
DEFAULT_MODEL = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "models", "efficientdet_lite0_synth.vbtm")


@click.group()
def main():
    pass


@main.command()
@click.argument("src", type=str, nargs=-1)
@click.option("--model", default=DEFAULT_MODEL, show_default=True, type=str, help="VBTM model container.")
@click.option("--detection_treshold", default=0.5, show_default=True, type=float, help="Object detection threshold.")
@click.option("--df_dir", default=None, show_default=True, help="Directory for exporting the dataframes.")
@click.option("--fps", default=30.0, show_default=True, type=float, help="Frame rate of the source (cap.get(CAP_PROP_FPS) in the reference).")
@click.option("--frame_stride", default=1, show_default=True, type=int, help="16 reproduces `frame_count %% 16` of reference track.py:166.")
@click.option("--time_batch", default=64, show_default=True, type=int, help="Consecutive frames of the clip per detector batch (1 = one frame per step).")
def track(src, model, detection_treshold, df_dir, fps, frame_stride, time_batch):
    from .track import export_dataframe, track_frames
    for s in src:
        if not os.path.isfile(s):
            raise FileNotFoundError(s)                                   # reference track.py:89-90
        frames = np.load(s, mmap_mode="r")
        if frames.ndim != 4 or frames.shape[3] != 3 or frames.dtype != np.uint8:
            raise click.ClickException(f"{s}: expected uint8 [T,H,W,3], got {frames.dtype} {frames.shape}")
        data = track_frames(frames, model, fps=fps, detection_treshold=detection_treshold, frame_stride=frame_stride, time_batch=time_batch)
        if not data["id"]:
            click.echo(f"{s}: no tracked rows")
            continue
        df, best, path = export_dataframe(data, s, model, df_dir=df_dir, write=df_dir is not None)
        click.echo(f"{s}: {len(df)} rows, {df['id'].nunique()} ids, export id {best}" + (f" -> {path}" if df_dir is not None else ""))


@main.command()
@click.argument("src", type=str, nargs=-1)
@click.option("--plate_diameter", default=0.45, show_default=True, type=float, help="Diameter of the weight plate used in meters.")
def analyze(src, plate_diameter):
    import pandas as pd
    from .velocity import Phase, analyze_rows
    cols = ["time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width"]
    for s in src:
        if not os.path.isfile(s):
            raise FileNotFoundError(s)                                   # reference plot.py:67-68
        m = FILENAME_RE.match(os.path.basename(s))
        if not m:
            click.echo(f"Couldn't create a plot for file '{s}'.")        # reference plot.py:81-85
            continue
        video, tid, model = m.groups()
        df = pd.read_pickle(s)
        df = df.query(f"id == {tid}").drop(columns=["id"])
        phases = analyze_rows(np.stack([df[c].to_numpy(np.float64) for c in cols], axis=1), plate_diameter, preprocess=True)
        reps = [p for p in phases if p.type == Phase.CONCENTRIC]
        click.echo(f"{video} (id {tid}, {model}): {len(phases)} phases, {len(reps)} concentric reps")
        for i, p in enumerate(reps, 1):
            click.echo(f"  rep {i}: t {p.time_start:.4f}-{p.time_end:.4f} s  ROM {p.rom:.6f} m  ACV {p.rom / p.duration:.6f} m/s")


@main.command()
@click.option("--kinovea_dir", default=None, help="Directory containing the kinovea exports (*.txt).")
@click.option("--qualysis_dir", default=None, help="Directory containing the qualysis exports (*.tsv).")
@click.option("--df_dir", default="dfs", show_default=True, help="Directory containing the dfs.")
@click.option("--plate_diameter", default=0.45, show_default=True, type=float, help="Diameter of the weight plate used in meters.")
def validate(kinovea_dir, qualysis_dir, df_dir, plate_diameter):
    import glob
    import pandas as pd
    from . import validate as V
    if (kinovea_dir is None) == (qualysis_dir is None):
        raise click.UsageError("give exactly one of --kinovea_dir / --qualysis_dir")
    source = "kinovea" if kinovea_dir is not None else "qualisys"
    exports = sorted(glob.glob(os.path.join(kinovea_dir, "*.txt") if source == "kinovea" else os.path.join(qualysis_dir, "*.tsv")))
    df_files = sorted(glob.glob(os.path.join(df_dir, "*.pkl.gz")))
    rows_out = []
    for ex in exports:
        stem = os.path.basename(ex).split(".")[0]
        match = next((x for x in df_files if os.path.basename(x).startswith(stem)), None)
        if match is None:
            click.echo(f"No matching df file found for: {ex}")                    # reference kinovea.py:62-64
            continue
        m = FILENAME_RE.match(os.path.basename(match))
        if not m:
            continue
        video, tid, _ = m.groups()
        df = pd.read_pickle(match).query(f"id == {tid}").sort_values(by="time")
        rows = np.stack([df[c].to_numpy(np.float64) for c in ("time", "x", "y", "norm_plate_height", "norm_plate_width")], axis=1)
        ref = V.read_kinovea(ex) if source == "kinovea" else V.read_qualisys(ex)
        r = V.validate_pair(ref, rows, plate_diameter, source)
        rows_out.append((video, r))
        click.echo(f"{video}: MSEx {r['mse_x']:.4f}  MSEy {r['mse_y']:.4f}  r_x {r['r_x']:.4f}  r_y {r['r_y']:.4f}")
    click.echo(f"Total MSEx = {sum(r['mse_x'] for _, r in rows_out)}, MSEy = {sum(r['mse_y'] for _, r in rows_out)}")


if __name__ == "__main__":
    main()
