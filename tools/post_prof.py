#!/usr/bin/env python3
"""Phase profile of the decode + NMS kernel (TFLite_Detection_PostProcess) at a small batch.  Needs a library built with
VBT_EXTRA_CXXFLAGS=-DVBT_POST_PROF (python -m vbt_amd.build --force)."""
import ctypes
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
import torch  # noqa: E402
import bench  # noqa: E402
from vbt_amd import _lib  # noqa: E402
from vbt_amd.interpreter import Interpreter  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = _lib.lib()
L.vbt_post_prof_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
it = Interpreter(bench.MODEL, max_batch=B)
fr = torch.from_numpy(bench.make_frames(list(range(B)), 0, 8)).cuda()      # [8, B, ...]
out = (ctypes.c_ulonglong * 16)()
for rep in range(2):
    L.vbt_post_prof_read(out, 1)
    for t in range(64):
        it.detect(fr[t % 8].cpu().numpy())
    torch.cuda.synchronize()
L.vbt_post_prof_read(out, 0)
v = list(out)
calls = max(v[10], 1)
names = {0: "class bytes -> LDS, tables", 1: "pass 1: histogram", 2: "bin range selection", 11: "class-byte range of the ranks", 12: "pass 2: count + scan",
         3: "pass 2: keys", 4: "sort", 13: "decode", 5: "greedy suppression", 6: "output tail"}
tot = sum(v[i] for i in names)
print(f"calls {calls}  rounds per call {v[8] / calls:.2f}  candidates per round {v[9] / max(v[8], 1):.1f}   ")
for i, n in names.items():      # s_memtime counts shader cycles (2.4 GHz when the kernel runs alone)
    print(f"  {n:32s} {v[i] / calls:9.0f} cycles = {v[i] / calls / 2400:6.2f} us  {100 * v[i] / max(tot, 1):5.1f} %")
print(f"  total {tot / calls:9.0f} cycles = {tot / calls / 2400:6.2f} us per call")
