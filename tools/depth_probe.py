"""Steady-state ms/step of one configuration per process: python tools/depth_probe.py <GPU_MAX_HW_QUEUES> <depth> <detect|own|inline>
(the queue-count / pipeline-depth measurements)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
q, depth, mode = sys.argv[1], int(sys.argv[2]), sys.argv[3]
os.environ["GPU_MAX_HW_QUEUES"] = q
os.environ["VBT_TRACKER_STREAM"] = mode if mode in ("own", "inline") else "own"
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
import numpy as np, torch
import bench
from vbt_amd.track import Pipeline
from vbt_amd.container import Container
n = 64
size = int(Container(bench.MODEL).header["image_size"])
U = 16
frames = torch.from_numpy(bench.make_frames(list(range(n)), 0, U, size)).cuda()
fbytes = frames[0].numel()
stream = torch.cuda.current_stream().cuda_stream
pipe = Pipeline(bench.MODEL, n, max_frames=1500, fps=60.0, detection_treshold=0.5, device=0, rows_per_frame=8, depth=depth)
res = []
for rep in range(int(os.environ.get('PROBE_REPS', '3'))):
    pipe.reset()
    for i in range(20):
        pipe.step(frames.data_ptr() + (i % U) * fbytes, stream, track=mode != "detect")
    torch.cuda.synchronize()
    K = int(os.environ.get('PROBE_K', '300'))
    t0 = time.perf_counter()
    for i in range(K):
        pipe.step(frames.data_ptr() + (i % U) * fbytes, stream, track=mode != "detect")
    pipe._drain()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res.append(1e3 * dt / K)
print(f"queues {q} depth {depth} {mode}: " + " ".join(f"{r:.4f}" for r in res) + f" ms/step  best {n/min(res)*1e3:.0f} fps", flush=True)
