"""Timeline of the host-fed K = 20 run: for every timed step, when the host entered / left Pipeline.step (and how long it sat in the
staging-buffer gate), and when the GPU finished that step's forward (event times against the run's start)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
import numpy as np, torch
import bench
from vbt_amd.track import Pipeline
n, W, PH, K, Uh = 64, 5, 32, 20, 16
frames_np = bench.make_frames(list(range(n)), 0, Uh)
stream = torch.cuda.current_stream().cuda_stream
pipe = Pipeline(bench.MODEL, n, max_frames=1013, fps=60.0, detection_treshold=0.5, device=0, rows_per_frame=8)
rows_host = torch.empty(n * pipe.tracker.rows_cap * 64, dtype=torch.uint8).pin_memory()
host = torch.from_numpy(frames_np).pin_memory()
gate_t = [0.0]
orig_gate = pipe._host_copy_gate
def gate(j):
    t = time.perf_counter(); orig_gate(j); gate_t[0] += time.perf_counter() - t
pipe._host_copy_gate = gate
for i in range(2 * Uh):
    pipe.step(host[i % Uh], stream, track=False)
for rep in range(3):
    torch.cuda.synchronize(); pipe.reset()
    for i in range(W):
        pipe.step(host[i % Uh], stream)
    torch.cuda.synchronize()
    start = torch.cuda.Event(enable_timing=True); start.record()
    evs, host_t = [], []
    t0 = time.perf_counter()
    for i in range(K):
        gate_t[0] = 0.0
        a = time.perf_counter()
        pipe.step(host[(W + i) % Uh], stream)
        b = time.perf_counter()
        o = (pipe._step_idx - 1) % pipe._ring
        e = torch.cuda.Event(enable_timing=True); e.record(pipe._det_streams[o % pipe.depth]); evs.append(e)
        host_t.append((1e3 * (a - t0), 1e3 * (b - t0), 1e3 * gate_t[0]))
    t1 = time.perf_counter()
    pipe.close(cap=PH)
    t2 = time.perf_counter()
    counts, rows = pipe.rows_all(out=rows_host)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"rep {rep}: total {1e3*(t3-t0):.2f} ms = {n*K/(t3-t0):.0f} fps; enqueue {1e3*(t1-t0):.2f} close {1e3*(t2-t1):.2f} rows {1e3*(t3-t2):.2f}")
    if rep == 2:
        for i in range(K):
            print(f"  step {i:2d}: host in {host_t[i][0]:6.2f} out {host_t[i][1]:6.2f} (gate {host_t[i][2]:5.2f})  gpu done {start.elapsed_time(evs[i]):6.2f} ms")
