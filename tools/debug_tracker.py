import sys, os, pickle
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
from test_gpu_tracker import _random_scene, _pack
from vbt_amd.ocsort import MultiClipTracker
asso = sys.argv[1] if len(sys.argv) > 1 else "diou"
scenes = [_random_scene(s) for s in range(24)]
dets, counts, times = _pack([s[0] for s in scenes], [s[1] for s in scenes])
mc = MultiClipTracker(len(scenes), 4096, max_age=30, asso_func=asso, iou_threshold=0.1)
mc.update_frames(dets, counts, times)
out = [mc.rows(ci) for ci in range(len(scenes))]
pickle.dump(out, open(f"gpurun_out/tracker_rows_{asso}.pkl", "wb"))
print("ok")
