#!/usr/bin/env python3
"""BASELINE config 5: the reference's 34-clip corpus as a synthetic stand-in with the REAL per-clip frame counts and frame
rates (derived from tests/golden: last time stamp x fps of every dfs_ocsort clip), clip-sharded over the ranks by
longest-processing-time packing (vbt_amd/shard.py) and run as one ragged batch per rank (Pipeline.step(active=...)).
Developer tool, not the contract bench.   usage: python tools/corpus_bench.py   (or under torch.distributed.run)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from vbt_amd import shard, synth  # noqa: E402
from vbt_amd.track import Pipeline  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
MODEL = os.path.join(ROOT, "models", "efficientdet_lite0_synth.vbtm")
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "gpurun_out", "plan_corpus"))


def corpus():
    meta = json.load(open(os.path.join(GOLD, "phases_ocsort.json")))
    main = np.load(os.path.join(GOLD, "dfs_ocsort_main.npz"))
    out = {}
    for k, v in meta.items():
        if k == "001_sort":
            continue
        out[k] = (int(round(float(main[f"c{k}_time"].max()) * v["fps"])), float(v["fps"]))
    return out


def main():
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("VBT_BENCH_SAME_DEVICE") != "1" else 0)
    clips = corpus()
    mine = shard.shard_clips({k: v[0] for k, v in clips.items()}, world)[rank]
    n = len(mine)
    lengths = np.array([clips[k][0] for k in mine])
    fps = np.array([clips[k][1] for k in mine])
    T, U = int(lengths.max()), 8
    frames = torch.from_numpy(np.stack([np.stack([synth.render(synth.background(int(k), 320), 7 * u) for k in mine]) for u in range(U)])).cuda()
    pipe = Pipeline(MODEL, n, max_frames=T, fps=fps, detection_treshold=0.5)
    st = torch.cuda.current_stream().cuda_stream
    fb = frames[0].numel()
    for t in range(6):                         # warm-up on a throw-away pipeline state is not possible: use masked steps
        pipe.step(frames.data_ptr() + (t % U) * fb, st, active=np.zeros(n, bool))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(T):
        pipe.step(frames.data_ptr() + (t % U) * fb, st, active=t < lengths)
    pipe.finish(st)
    best, rows, nph, ovf, ph = pipe.tracker.summary(cap=512)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    total = int(lengths.sum())
    print(json.dumps({"rank": rank, "world": world, "clips": n, "frames": total, "longest_clip": T, "seconds": round(dt, 3),
                      "frames_per_s": round(total / dt), "batch_slots_per_s": round(n * T / dt), "rows": int(rows.sum()),
                      "overflow": int((ovf != 0).sum())}))


if __name__ == "__main__":
    main()
