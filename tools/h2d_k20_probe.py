"""Where the host-fed K = 20 run (SURVEY 8d's metric at the driver's --steps 20) spends its time: the pass of bench.extra_measurements
repeated, with the enqueue / close / rows split, for several sizes of the tracker's row log and of the pinned row buffer."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
import numpy as np, torch
import bench
from vbt_amd.track import Pipeline
n, W, PH = 64, 5, 32
Uh = 16
frames_np = bench.make_frames(list(range(n)), 0, Uh)
stream = torch.cuda.current_stream().cuda_stream
variants = sys.argv[1:] or ["big,pinned", "big,pageable", "small,pinned", "small,dummy64", "big,pinned_first"]
for v in variants:
    cap, mode = v.split(",")
    max_frames = 1013 if cap == "big" else 64
    pre = torch.empty(64 * 8179 * 64, dtype=torch.uint8).pin_memory() if mode == "pinned_first" else None
    host = torch.from_numpy(frames_np).pin_memory()
    pipe = Pipeline(bench.MODEL, n, max_frames=max_frames, fps=60.0, detection_treshold=0.5, device=0, rows_per_frame=8)
    rows_host = None
    if mode == "pinned":
        rows_host = torch.empty(n * pipe.tracker.rows_cap * 64, dtype=torch.uint8).pin_memory()
    elif mode == "pinned_first":
        rows_host = pre
    dummy = torch.empty(64 << 20, dtype=torch.uint8).pin_memory() if mode == "dummy64" else None
    for i in range(2 * Uh):
        pipe.step(host[i % Uh], stream, track=False)
    for rep in range(4):
        K = 20 if rep < 3 else 200
        torch.cuda.synchronize(); pipe.reset()
        for i in range(W):
            pipe.step(host[i % Uh], stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            pipe.step(host[(W + i) % Uh], stream)
        t1 = time.perf_counter()
        pipe.close(cap=PH)
        t2 = time.perf_counter()
        counts, rows = pipe.rows_all(out=rows_host)
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f"{v:18s} rows_cap {pipe.tracker.rows_cap} K {K}: total {1e3*(t3-t0):.3f} ms = {n*K/(t3-t0):.0f} fps  enqueue {1e3*(t1-t0):.2f} close {1e3*(t2-t1):.3f} rows {1e3*(t3-t2):.3f}", flush=True)
    del pipe, host, rows_host, dummy, pre
    import gc; gc.collect()
