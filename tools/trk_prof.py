#!/usr/bin/env python3
"""Phase profile of the on-device OC-SORT step on the one-clip time-batched workload.  Needs a library built with
VBT_EXTRA_CXXFLAGS=-DVBT_TRK_PROF (python -m vbt_amd.build --force)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
import torch  # noqa: E402
from vbt_amd import _lib, synth  # noqa: E402
from vbt_amd.track import Pipeline  # noqa: E402

L = _lib.lib()
T, F, U = 2048, 64, 256
bg = synth.background(0)
base = np.stack([synth.render(bg, t) for t in range(U)])
fr = torch.from_numpy(np.concatenate([base, base[:F]])).cuda()
pipe = Pipeline(os.path.join(ROOT, "models", "efficientdet_lite0_synth.vbtm"), F, max_frames=T, fps=60.0, tracker_clips=1)
out = (ctypes.c_ulonglong * 16)()
L.vbt_tracker_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
for rep in range(2):
    pipe.reset()
    L.vbt_tracker_prof_read(out, 1)
    for f0 in range(0, T, F):
        pipe.step_runs(fr[f0 % U:f0 % U + F], [(0, 0, F, f0 + 1)])
    pipe.close(cap=512)
    torch.cuda.synchronize()
L.vbt_tracker_prof_read(out, 0)
v = list(out)
steps = max(v[8], 1)
names = ["predict", "cost matrix", "assignment", "unmatched lists (after the matched updates)", "second association: rest", "update(None) + births", "emission + deletion",
         None, None, None, None, "matched Kalman updates", "second association: cost entries", "second association: assignment", "second association: recovered updates"]
print(f"steps {steps}  mean detections {v[9] / steps:.2f}  mean live trackers {v[10] / steps:.2f}   (s_memtime counts shader cycles: 2.4 GHz when the kernel runs alone)")
tot = sum(v[i] for i, nme in enumerate(names) if nme)
for i, nme in enumerate(names):
    if not nme:
        continue
    print(f"  {nme:52s} {v[i] / steps:8.0f} cycles = {v[i] / steps / 2400:6.2f} us  {100 * v[i] / tot:5.1f} %")
print(f"  total                        {tot / steps:8.0f} cycles = {tot / steps / 2400:6.2f} us per stepped frame")
