#!/usr/bin/env python3
"""Developer probe: what a small-batch forward is made of.
  run mode   (under `rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 tools/b1_trace.py run B`):
             N forwards of batch B on one stream (hipGraph replay), nothing else on the GPU.
  parse mode (`python tools/b1_trace.py parse DIR [n_launches]`): per launch position of the forward, the median kernel
             duration and the median gap to the previous kernel's end; totals."""
import collections
import csv
import glob
import os
import statistics
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)


def run(B, n=60):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
    os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
    import torch
    import bench
    from vbt_amd import _lib
    from vbt_amd.interpreter import Interpreter
    it = Interpreter(bench.MODEL, max_batch=B)
    fr = torch.from_numpy(bench.make_frames(list(range(B)), 0, 1)).cuda()
    dev = torch.device("cuda:0")
    b = torch.empty((B, 25, 4), dtype=torch.float32, device=dev)
    s = torch.empty((B, 25), dtype=torch.float32, device=dev)
    c = torch.empty((B, 25), dtype=torch.float32, device=dev)
    k = torch.empty((B,), dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(n):
        _lib.check(_lib.lib().vbt_detect_async(it.handle, fr.data_ptr(), B, st, b.data_ptr(), s.data_ptr(), c.data_ptr(), k.data_ptr()))
    torch.cuda.synchronize()
    print("launches per forward:", _lib.lib().vbt_model_num_launches(it.handle))


def parse(d, first="stem_block_kernel"):
    f = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)[0]
    tr = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(tr) if first in r["Kernel_Name"]]
    L = idx[-1] - idx[-2]
    dur, gap, names = collections.defaultdict(list), collections.defaultdict(list), {}
    for a in idx[len(idx) // 2:-1]:
        for j in range(L):
            r = tr[a + j]
            names[j] = r["Kernel_Name"].split("(")[0][:60]
            dur[j].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
            gap[j].append(int(r["Start_Timestamp"]) - int(tr[a + j - 1]["End_Timestamp"]))
    td = tg = 0
    for j in range(L):
        dj, gj = statistics.median(dur[j]), statistics.median(gap[j])
        td += dj
        tg += gj
        print(f"{j:3d} {names[j]:60s} dur {dj / 1e3:7.2f} us  gap {gj / 1e3:6.2f} us")
    print(f"launches {L}: kernel time {td / 1e3:.1f} us + gaps {tg / 1e3:.1f} us = {(td + tg) / 1e3:.1f} us per forward")


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    else:
        parse(sys.argv[2])
