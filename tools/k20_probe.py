"""Where the fixed cost of a 20-step timed region goes: enqueue / drain / close per repetition, K = 20, 40, 80."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "plan_lite0"))
import numpy as np, torch
import bench
from vbt_amd.track import Pipeline
from vbt_amd.container import Container
n = 64
size = int(Container(bench.MODEL).header["image_size"])
U = 25
frames_np = bench.make_frames(list(range(n)), 0, U, size)
frames = torch.from_numpy(frames_np).cuda()
fbytes = frames[0].numel()
stream = torch.cuda.current_stream().cuda_stream
for depth in (3, 2, 4):
    os.environ["VBT_PIPELINE_DEPTH"] = str(depth)
    pipe = Pipeline(bench.MODEL, n, max_frames=400, fps=60.0, detection_treshold=0.5, device=0, rows_per_frame=8)
    for rep in range(4):
        for K, W in ((20, 5), (40, 5), (80, 5), (20, 50)):
            pipe.reset()
            for i in range(W):
                pipe.step(frames.data_ptr() + (i % U) * fbytes, stream)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(K):
                pipe.step(frames.data_ptr() + ((W + i) % U) * fbytes, stream)
            t1 = time.perf_counter()
            pipe._drain()
            t2 = time.perf_counter()
            pipe.close(cap=32)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            print(f"depth {depth} rep {rep} K {K} W {W}: total {1e3*(t3-t0):.3f} ms = {1e3*(t3-t0)/K:.4f}/step  enq {1e3*(t1-t0):.2f} drain {1e3*(t2-t1):.2f} close {1e3*(t3-t2):.3f}", flush=True)
    del pipe
