"""Scratch: detector-only throughput against pipeline depth (is the depth-4 cliff the fifth active stream?)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", sys.argv[1] if len(sys.argv) > 1 else "8")
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
for depth in (3, 4, 5):
    pipe = Pipeline(bench.MODEL, n, max_frames=700, fps=60.0, detection_treshold=0.5, device=0, rows_per_frame=8, depth=depth)
    for track in (False, True, "own", "inline"):
        if isinstance(track, str):
            pipe._trk_inline = track == "inline"
        pipe.reset()
        for i in range(20):
            pipe.step(frames.data_ptr() + (i % U) * fbytes, stream, track=bool(track))
        torch.cuda.synchronize()
        K = 300
        t0 = time.perf_counter()
        for i in range(K):
            pipe.step(frames.data_ptr() + (i % U) * fbytes, stream, track=bool(track))
        pipe._drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"queues {os.environ['GPU_MAX_HW_QUEUES']} depth {depth} track {track}: {1e3*dt/K:.4f} ms/step  {n*K/dt:.0f} fps", flush=True)
    del pipe
