#!/usr/bin/env python3
"""Generate tests/golden/* from the reference tree (run in the build container only; the
reference never travels to the GPU box).

Inputs read:  /root/reference/dfs_ocsort/*.pkl.gz, /root/reference/dfs/001_*.pkl.gz (committed pipeline outputs
of reference track.py:103-126) and the importable reference modules VelocityTracker.py /
RunningAverage.py / Phase.py (numpy-only).  Outputs are DATA ONLY (inputs and expected outputs):
  dfs_ocsort_main.npz   per clip, rows of the id named in the file name: time,x,y,dx,dy,h,w + index
  dfs_ocsort_full.npz   clips 001/002/005/008/030 with every id (tracker replay tests)
  pre_ocsort.npz        plot.py:87-95 preprocessing (pandas rolling(5)/expanding means) of 4 clips
  phases_ocsort.json    phases the reference VelocityTracker yields per clip (floats as hex)
  running_average.json  reference RunningAverage input/output sequences
"""
import glob
import json
import os
import re
import sys

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, REF)
from VelocityTracker import VelocityTracker  # noqa: E402  (reference module, imported only here)
from RunningAverage import RunningAverage    # noqa: E402

FN = re.compile(r"(\S*)_id(\d+)_(\S*)\.pkl\.gz")
COLS = ["time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width"]


def preprocess(df, tid):
    """reference plot.py:87-95"""
    df = df.query(f"id == {tid}").drop(columns=["id"])
    for col in ["x", "y", "dx", "dy"]:
        df[col] = df[col].rolling(window=5, center=False, min_periods=1).mean()
    for col in ["norm_plate_height", "norm_plate_width"]:
        df[col] = df[col].expanding(min_periods=1).mean()
    return df


def analyze(df, plate_diameter=0.45):
    """reference plot.py:33-47"""
    vt = VelocityTracker(plate_diameter)
    for _, (time, x, y, dx, dy, h, w) in df.iterrows():
        vt.process_measurements(time, x, y, dx, dy, h, w)
    vt.end_processing()
    return vt.phases


def main():
    os.makedirs(OUT, exist_ok=True)
    main_npz, full_npz, pre_npz, phases = {}, {}, {}, {}
    for f in sorted(glob.glob(os.path.join(REF, "dfs_ocsort", "*.pkl.gz"))):
        video, tid, model = FN.match(os.path.basename(f)).groups()
        tid = int(tid)
        clip = video[:3]
        reps = int(re.search(r"_(\d+)reps", video).group(1))
        df = pd.read_pickle(f)
        sel = df[df["id"] == tid]
        for c in COLS:
            main_npz[f"c{clip}_{c}"] = sel[c].to_numpy(np.float64)
        main_npz[f"c{clip}_index"] = sel.index.to_numpy(np.int64)
        if clip in ("001", "002", "005", "008", "030"):
            for c in ["id"] + COLS:
                full_npz[f"c{clip}_{c}"] = df[c].to_numpy()
            full_npz[f"c{clip}_index"] = df.index.to_numpy(np.int64)
        pre = preprocess(df, tid)
        if clip in ("001", "005", "009", "030"):
            for c in COLS:
                pre_npz[f"c{clip}_{c}"] = pre[c].to_numpy(np.float64)
        ph = analyze(pre)
        t = np.sort(df["time"].unique())
        phases[clip] = {
            "video": video, "id": tid, "model": model, "reps_in_name": reps, "rows": int(len(df)),
            "fps": float(round(1.0 / np.median(np.diff(t)), 3)),
            "phases": [[float(p.time_start).hex(), float(p.time_end).hex(), float(p.y_start).hex(),
                        float(p.y_end).hex(), float(p.rom).hex(), int(p.type)] for p in ph],
        }
        print(clip, video, "rows", len(df), "phases", len(ph), "concentric", sum(p.type == 0 for p in ph), "/", reps)
    np.savez_compressed(os.path.join(OUT, "dfs_ocsort_main.npz"), **main_npz)
    np.savez_compressed(os.path.join(OUT, "dfs_ocsort_full.npz"), **full_npz)
    np.savez_compressed(os.path.join(OUT, "pre_ocsort.npz"), **pre_npz)
    # SORT-variant output of clip 001 (reference dfs/, SURVEY.md section 4.3 right-hand column)
    f = glob.glob(os.path.join(REF, "dfs", "001_*.pkl.gz"))[0]
    video, tid, model = FN.match(os.path.basename(f)).groups()
    df = pd.read_pickle(f)
    ph = analyze(preprocess(df, int(tid)))
    phases["001_sort"] = {"video": video, "id": int(tid), "model": model, "rows": int(len(df)),
                          "acv": [float(p.rom / p.duration) for p in ph if p.type == 0]}
    with open(os.path.join(OUT, "phases_ocsort.json"), "w") as fo:
        json.dump(phases, fo, indent=0)
    # RunningAverage known answers
    rng = np.random.Generator(np.random.PCG64(7))
    seqs = []
    for window, n in ((3, 6), (30, 100), (5, 17), (1, 4)):
        vals = [float(i + 1) for i in range(n)] if window == 3 else [float(v) for v in rng.uniform(0.1, 0.4, n)]
        ra = RunningAverage(window_size=window)
        seqs.append({"window": window, "in": [v.hex() for v in vals], "out": [float(ra.update(v)).hex() for v in vals]})
    with open(os.path.join(OUT, "running_average.json"), "w") as fo:
        json.dump(seqs, fo)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
