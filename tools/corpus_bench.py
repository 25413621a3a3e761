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
    U = 8
    slots = int(os.environ.get("VBT_CORPUS_SLOTS", "0")) or n         # 0 / unset: one slot per clip (ragged batch)
    frames = torch.from_numpy(np.stack([np.stack([synth.render(synth.background(int(k), 320), 7 * u) for u in range(U)]) for k in mine])).cuda()  # [clip][U]
    st = torch.cuda.current_stream().cuda_stream
    if slots >= n:
        T = int(lengths.max())
        pipe = Pipeline(MODEL, n, max_frames=T, fps=fps, detection_treshold=0.5)
        fr = frames.transpose(0, 1).contiguous()                       # [U][clip]
        fb = fr[0].numel()
        for t in range(6):
            pipe.step(fr.data_ptr() + (t % U) * fb, st, active=np.zeros(n, bool))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(T):
            pipe.step(fr.data_ptr() + (t % U) * fb, st, active=t < lengths)
    else:
        cmap, fidx = shard.slot_schedule(lengths, slots)
        T = len(cmap)
        pipe = Pipeline(MODEL, slots, max_frames=int(lengths.max()), fps=fps, detection_treshold=0.5, tracker_clips=n)
        cm_dev = torch.from_numpy(np.maximum(cmap, 0).astype(np.int64)).cuda()
        idle = torch.zeros((slots, 320, 320, 3), dtype=torch.uint8, device="cuda")
        for t in range(6):
            pipe.step(idle, st, clip_map=np.full(slots, -1), frame_idx=np.zeros(slots))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(T):
            buf = torch.index_select(frames[:, t % U], 0, cm_dev[t])      # this step's frame of the clip sitting in each slot
            pipe.step(buf, st, clip_map=cmap[t], frame_idx=fidx[t])          # (a tensor: Pipeline ties its lifetime to the slot's stream)
    pipe.finish(st)
    best, rows, nph, ovf, ph = pipe.tracker.summary(cap=512)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    total = int(lengths.sum())
    print(json.dumps({"rank": rank, "world": world, "clips": n, "frames": total, "steps": T, "slots": slots, "seconds": round(dt, 3),
                      "frames_per_s": round(total / dt), "slot_steps_per_s": round(slots * T / dt), "rows": int(rows.sum()),
                      "overflow": int((ovf != 0).sum())}))


if __name__ == "__main__":
    main()
