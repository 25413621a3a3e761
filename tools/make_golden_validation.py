#!/usr/bin/env python3
"""Golden vectors for the trajectory validation study: run in the build container only.

The reference's kinovea.py / qualysis.py cannot be imported here (seaborn is absent), so the expected numbers are
produced by evaluating, on the reference's own data files, the very library calls those scripts make:
  pandas.read_csv(...)                          kinovea.py:73-84, qualysis.py:79-97
  rolling / expanding means                     kinovea.py:99-105, qualysis.py:113-117
  scipy.interpolate.interp1d(kind='linear')     kinovea.py:155-162, qualysis.py:170-177
  scipy.stats.pearsonr, sklearn mean_squared_error   kinovea.py:164-172, qualysis.py:179-187
Inputs read: /root/reference/{kinovea_exports,qualysis_exports,qualysis_dfs,dfs_ocsort}.  Outputs are DATA ONLY:
  tests/golden/validation.npz    per video: reference trajectory (t, x, y in metres), tracked rows of the export id
                                 (qualisys clips only; the kinovea clips reuse dfs_ocsort_main.npz), the 30-sample
                                 rolling means of the plate size, and the expected MSE / Pearson r
  tests/golden/kinovea_sample.txt, qualisys_sample.tsv   one raw export of each kind (parser tests)
"""
import glob
import os
import re
import shutil

import numpy as np
import pandas as pd
from scipy.interpolate import interp1d
from scipy.stats import pearsonr
from sklearn.metrics import mean_squared_error

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
FN = re.compile(r"(\S*)_id(\d+)_(\S*)\.pkl\.gz")


def stats(ref_df, df):
    t_max = min(ref_df["time"].max(), df["time"].max())
    t_min = max(ref_df["time"].min(), df["time"].min())
    ts = np.linspace(t_min, t_max, int(t_max * 30))
    xr = interp1d(ref_df["time"], ref_df["x"], kind="linear")(ts)
    xm = interp1d(df["time"], df["x"], kind="linear")(ts)
    yr = interp1d(ref_df["time"], ref_df["y"], kind="linear")(ts)
    ym = interp1d(df["time"], df["y"], kind="linear")(ts)
    rx, ry = pearsonr(xr, xm), pearsonr(yr, ym)
    return np.array([mean_squared_error(xr, xm), mean_squared_error(yr, ym), rx.statistic, ry.statistic, rx.pvalue, ry.pvalue])


def main():
    out = {}
    plate = 0.45
    # ---- kinovea.py against dfs_ocsort (the tracker this build reproduces)
    df_files = glob.glob(os.path.join(REF, "dfs_ocsort", "*.pkl.gz"))
    for kf in sorted(glob.glob(os.path.join(REF, "kinovea_exports", "*.txt"))):
        stem = os.path.basename(kf).split(".")[0]
        mf = next((x for x in df_files if os.path.basename(x).startswith(stem)), None)
        if mf is None:
            continue
        video, tid, model = FN.match(os.path.basename(mf)).groups()
        k = pd.read_csv(kf, comment="#", header=None, names=["time", "x", "y"], delimiter=" ", dtype={"time": float},
                        converters={"x": lambda x: float(x.replace(",", ".")), "y": lambda x: float(x.replace(",", "."))},
                        index_col=False)
        k["x"] = k["x"] / 100.0
        k["y"] = k["y"] / 100.0
        m = pd.read_pickle(mf).drop(columns=["dx", "dy"])
        m = m.query(f"id == {tid}").drop(columns=["id"]).sort_values(by="time")
        for col in ["norm_plate_height", "norm_plate_width"]:
            m[col] = m[col].expanding(min_periods=1).mean()
        for col in ["x", "y"]:
            m[col] = m[col].rolling(window=5, center=False, min_periods=1).mean()
        m["x"] = m["x"] * plate / m["norm_plate_width"]
        m["y"] = -m["y"] * plate / m["norm_plate_height"]
        m["y"] += k["y"].mean() - m["y"].mean()
        m["x"] += k["x"].mean() - m["x"].mean()
        key = "k" + video[:3]
        out[key + "_ref"] = np.stack([k["time"].to_numpy(), k["x"].to_numpy(), k["y"].to_numpy()], axis=1)
        out[key + "_xy"] = np.stack([m["x"].to_numpy(), m["y"].to_numpy()], axis=1)          # aligned trajectory
        out[key + "_stats"] = stats(k, m)
        print(key, video, len(k), len(m), out[key + "_stats"][:4])
    # ---- qualysis.py against qualysis_dfs
    df_files = glob.glob(os.path.join(REF, "qualysis_dfs", "*.pkl.gz"))
    for qf in sorted(glob.glob(os.path.join(REF, "qualysis_exports", "*.tsv"))):
        stem = os.path.basename(qf).split(".")[0]
        mf = next((x for x in df_files if os.path.basename(x).startswith(stem)), None)
        if mf is None:
            continue
        video, tid, model = FN.match(os.path.basename(mf)).groups()
        q = pd.read_csv(qf, delimiter="\t", skiprows=11, usecols=["Time", "Osa L X", "Osa L Z"], index_col=False)
        q = q.rename(columns={"Time": "time", "Osa L X": "x", "Osa L Z": "y"})
        q["x"] = -q["x"] / 1000.0
        q["y"] = q["y"] / 1000.0
        raw = pd.read_pickle(mf)
        m = raw.drop(columns=["dx", "dy"]).query(f"id == {tid}").drop(columns=["id"])
        key = "q_" + stem
        out[key + "_rows"] = np.stack([m[c].to_numpy(np.float64) for c in ("time", "x", "y", "norm_plate_height", "norm_plate_width")], axis=1)
        out[key + "_id"] = np.int64(tid)
        m["norm_plate_width"] = m["norm_plate_width"].rolling(window=30, center=False, min_periods=1).mean()
        m["norm_plate_height"] = m["norm_plate_height"].rolling(window=30, center=False, min_periods=1).mean()
        out[key + "_hw30"] = np.stack([m["norm_plate_height"].to_numpy(), m["norm_plate_width"].to_numpy()], axis=1)
        m["x"] = m["x"] * plate / m["norm_plate_width"]
        m["y"] = -m["y"] * plate / m["norm_plate_height"]
        m["y"] += q["y"].mean() - m["y"].mean()
        m["x"] += q["x"].mean() - m["x"].mean()
        out[key + "_ref"] = np.stack([q["time"].to_numpy(), q["x"].to_numpy(), q["y"].to_numpy()], axis=1)
        out[key + "_xy"] = np.stack([m["x"].to_numpy(), m["y"].to_numpy()], axis=1)
        out[key + "_stats"] = stats(q, m)
        print(key, video, len(q), len(m), out[key + "_stats"][:4])
    np.savez_compressed(os.path.join(OUT, "validation.npz"), **out)
    shutil.copyfile(os.path.join(REF, "kinovea_exports", "029_dl_4reps.txt"), os.path.join(OUT, "kinovea_sample.txt"))
    with open(os.path.join(REF, "qualysis_exports", "squat1.tsv")) as f, open(os.path.join(OUT, "qualisys_sample.tsv"), "w") as g:
        g.writelines(f.readlines()[:11 + 1 + 400])          # header block + column names + the first 400 frames
    os.chmod(os.path.join(OUT, "kinovea_sample.txt"), 0o644)
    for fn in ("validation.npz", "kinovea_sample.txt", "qualisys_sample.tsv"):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == "__main__":
    main()
