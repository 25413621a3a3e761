#!/usr/bin/env python3
"""tests/golden/dfs_ocsort_all.npz: EVERY row (all ids) of all 34 reference DataFrames (reference dfs_ocsort/*.pkl.gz, the
committed outputs of reference track.py:103-126), so that the tracker's id numbering, emission order (the retained
DataFrame index), gap structure and Kalman velocities are pinned on the whole corpus and not only on five clips.
Run in the build container only (the reference never travels to the GPU box).  DATA ONLY: columns of the DataFrames."""
import glob
import os
import re

import numpy as np
import pandas as pd

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
FN = re.compile(r"(\S*)_id(\d+)_(\S*)\.pkl\.gz")
COLS = ["time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width"]


def main():
    out = {}
    total = 0
    for f in sorted(glob.glob(os.path.join(REF, "dfs_ocsort", "*.pkl.gz"))):
        video, tid, model = FN.match(os.path.basename(f)).groups()
        clip = video[:3]
        df = pd.read_pickle(f)
        out[f"c{clip}_id"] = df["id"].to_numpy(np.int16)
        out[f"c{clip}_index"] = df.index.to_numpy(np.int32)
        for c in COLS:
            out[f"c{clip}_{c}"] = df[c].to_numpy(np.float64)
        out[f"c{clip}_export_id"] = np.int16(tid)
        total += len(df)
    np.savez_compressed(os.path.join(OUT, "dfs_ocsort_all.npz"), **out)
    print("clips", len(out) // (len(COLS) + 3), "rows", total, "bytes", os.path.getsize(os.path.join(OUT, "dfs_ocsort_all.npz")))


if __name__ == "__main__":
    main()
