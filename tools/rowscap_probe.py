"""Does the capacity of the tracker's row log change the step time?  Resident frames, K = 200, several max_frames."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
import numpy as np, torch
import bench
from vbt_amd.track import Pipeline
n, U = 64, 16
frames = torch.from_numpy(bench.make_frames(list(range(n)), 0, U)).cuda()
fb = frames[0].numel()
stream = torch.cuda.current_stream().cuda_stream
for max_frames, rpf in ((64, 8), (1013, 8), (1013, 1), (4096, 8), (64, 8)):
    pipe = Pipeline(bench.MODEL, n, max_frames=max_frames, fps=60.0, detection_treshold=0.5, device=0, rows_per_frame=rpf)
    for track in (True, False):
        res = []
        for rep in range(3):
            torch.cuda.synchronize(); pipe.reset()
            K = 40
            for i in range(5):
                pipe.step(frames.data_ptr() + (i % U) * fb, stream, track=track)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                pipe.step(frames.data_ptr() + (i % U) * fb, stream, track=track)
            pipe._drain()
            torch.cuda.synchronize()
            res.append((time.perf_counter() - t0) / K * 1e3)
        print(f"max_frames {max_frames} rows_per_frame {rpf} rows_cap {pipe.tracker.rows_cap} track {track}: " + " ".join(f"{r:.4f}" for r in res) + " ms/step", flush=True)
    del pipe
