#!/usr/bin/env python3
"""Where does the host-fed pipeline lose time against the HBM-resident one?  Runs K steps with the
frames (a) resident, (b) in pinned host memory through Pipeline.step, and prints enqueue time vs total per step.
Under `rocprofv3 --memory-copy-trace --kernel-trace` the copy records show the engine and duration of every H2D.
usage: python tools/h2d_probe.py [--steps 300] [--mode resident|host|both]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
import torch  # noqa: E402
import bench  # noqa: E402
from vbt_amd.track import Pipeline  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--mode", default="both")
    ap.add_argument("--track", type=int, default=1)
    a = ap.parse_args()
    n, K, U = 64, a.steps, 16
    fr_np = bench.make_frames(list(range(n)), 0, U)
    dev_fr = torch.from_numpy(fr_np).cuda()
    host_fr = torch.from_numpy(fr_np).pin_memory()
    pipe = Pipeline(bench.MODEL, n, max_frames=K + 64, fps=60.0, rows_per_frame=8)
    st = torch.cuda.current_stream().cuda_stream
    rows_host = torch.empty(n * pipe.tracker.rows_cap * 64, dtype=torch.uint8).pin_memory()
    out = {}
    for mode in (("resident", "host") if a.mode == "both" else (a.mode,)):
        src = dev_fr if mode == "resident" else host_fr
        for rep in range(2):
            pipe.reset()
            for i in range(2 * U):
                pipe.step(src[i % U], st, track=False)
            pipe.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                pipe.step(src[i % U], st, track=bool(a.track))
            t1 = time.perf_counter()
            tc = tr = t1
            if a.track:
                pipe.close(cap=32)
                tc = time.perf_counter()
                pipe.rows_all(out=rows_host)
                tr = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        out[mode] = {"ms_per_step": (t2 - t0) / K * 1e3, "enqueue_ms_per_step": (t1 - t0) / K * 1e3, "frames_per_s": n * K / (t2 - t0),
                     "close_ms": (tc - t1) * 1e3, "rows_all_ms": (tr - tc) * 1e3}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
