#!/usr/bin/env python3
"""64 clips x F consecutive frames per detector batch (B = 64 F): what time-batching buys on top of the clip batch."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "gpurun_out", "plans", "plan_lite0"))
import torch  # noqa: E402
import bench  # noqa: E402
from vbt_amd.track import Pipeline  # noqa: E402

n = 64
T = 960
for F in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,3,4").split(",")]:
    U = 12 * F
    fr = torch.from_numpy(bench.make_frames(list(range(n)), 0, U)).cuda()          # [U][n]
    frc = fr.transpose(0, 1).contiguous()                                          # [n][U]
    pipe = Pipeline(bench.MODEL, n * F, max_frames=T + 8, fps=60.0, tracker_clips=n, rows_per_frame=8)
    for rep in range(2):
        pipe.reset()
        pipe.frame_count = 0
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(0, T, F):
            s = t % U
            pipe.step_seq(frc[:, s:s + F].contiguous() if F > 1 else frc[:, s:s + 1])
        pipe.close(cap=64)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(json.dumps({"F": F, "batch": n * F, "frames_per_s": n * T / dt, "ms_per_64_frames": dt / T * 1e3}), flush=True)
    del pipe
