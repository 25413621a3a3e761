"""Scratch: which part of the tracker step costs the throughput at depth 4?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
import numpy as np, torch
import bench
from vbt_amd import _lib
from vbt_amd.track import Pipeline
from vbt_amd.container import Container
n = 64
size = int(Container(bench.MODEL).header["image_size"])
U = 16
frames = torch.from_numpy(bench.make_frames(list(range(n)), 0, U, size)).cuda()
fbytes = frames[0].numel()
stream = torch.cuda.current_stream().cuda_stream

def make_enq(pipe, mode):
    def enq(k):
        T = pipe._det_streams[k]
        if mode in ("wait+kernel", "wait-only") and pipe._last_trk_ev is not None:
            T.wait_event(pipe._last_trk_ev)
        b, s, c, cnt = pipe._bufs[k]
        if mode in ("wait+kernel", "kernel-only"):
            _lib.check(_lib.lib().vbt_tracker_update_from_detections(pipe.tracker.handle, b.data_ptr(), s.data_ptr(), cnt.data_ptr(),
                                                                     pipe._times[k].ctypes.data, pipe.thr, T.cuda_stream))
        if mode != "nothing":
            ev = torch.cuda.Event()
            ev.record(T)
            pipe._ev_trk[k] = ev
            pipe._last_trk_ev = ev
    return enq

for depth in (4, 3):
    pipe = Pipeline(bench.MODEL, n, max_frames=3000, fps=60.0, detection_treshold=0.5, device=0, rows_per_frame=8, depth=depth)
    pipe._trk_inline = True
    for mode in ("nothing", "event-only", "wait-only", "kernel-only", "wait+kernel"):
        pipe.reset()
        pipe._enqueue_tracker = make_enq(pipe, mode)
        for i in range(20):
            pipe.step(frames.data_ptr() + (i % U) * fbytes, stream)
        torch.cuda.synchronize()
        K = 300
        t0 = time.perf_counter()
        for i in range(K):
            pipe.step(frames.data_ptr() + (i % U) * fbytes, stream)
        pipe._drain()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"depth {depth} {mode}: {1e3*dt/K:.4f} ms/step  {n*K/dt:.0f} fps", flush=True)
    del pipe
