"""Developer probe: EfficientDet-Lite2 448x448, 64 clips per step (BASELINE config 4) at several pipeline depths / plans.
usage: python tools/lite2_probe.py <depth> [plan prefix]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
depth = int(sys.argv[1]) if len(sys.argv) > 1 else 3
os.environ["VBT_PLAN_FILE"] = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "profiles", "plan_lite2")
import numpy as np, torch
import bench
from vbt_amd.track import Pipeline
nb, T, U = 64, 120, 4
fr = torch.from_numpy(bench.make_frames(list(range(nb)), 0, U, 448)).cuda()
fb = fr[0].numel()
stream = torch.cuda.current_stream().cuda_stream
pipe = Pipeline(bench.MODEL_LITE2, nb, max_frames=T, fps=30.0, depth=depth)
res = []
for rep in range(3):
    torch.cuda.synchronize(); pipe.reset()
    t0 = time.perf_counter()
    for t in range(T):
        pipe.step(fr.data_ptr() + (t % U) * fb, stream)
    pipe.close(cap=64); pipe.rows_all()
    torch.cuda.synchronize()
    res.append(nb * T / (time.perf_counter() - t0))
print(f"lite2 depth {depth} plan {os.path.basename(os.environ['VBT_PLAN_FILE'])} launches {pipe.interpreter.num_launches()}: " + " ".join(f"{r:.0f}" for r in res) + " frames/s", flush=True)
