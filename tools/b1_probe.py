#!/usr/bin/env python3
"""Batch-1 latency mode (BASELINE config 2 as written): where do 180 us per frame go?  Host enqueue time vs wall time per
step at several pipeline depths, with and without the tracker."""
import json
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
import torch  # noqa: E402
import bench  # noqa: E402
from vbt_amd.track import Pipeline  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
T, U = 1500, 64
fr = torch.from_numpy(bench.make_frames(list(range(nb)), 0, U)).cuda()
fb = fr[0].numel()
st = torch.cuda.current_stream().cuda_stream
for depth in [int(x) for x in os.environ.get("DEPTHS", "1,2,3,4").split(",")]:
    pipe = Pipeline(bench.MODEL, nb, max_frames=T + 8, fps=60.0, depth=depth)
    for track in (False, True):
        for rep in range(2):
            pipe.reset()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for t in range(T):
                pipe.step(fr.data_ptr() + (t % U) * fb, st, track=track)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        print(json.dumps({"batch": nb, "depth": depth, "track": track, "enqueue_us_per_step": (t1 - t0) / T * 1e6, "us_per_step": (t2 - t0) / T * 1e6,
                          "frames_per_s": nb * T / (t2 - t0)}), flush=True)
    del pipe
