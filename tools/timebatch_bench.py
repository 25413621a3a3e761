#!/usr/bin/env python3
"""Time-batched path (Pipeline.step_runs): one clip of T frames with F consecutive frames per detector batch (BASELINE
config 2's clip at batch-64 throughput) and the 34-clip corpus on one GPU through shard.run_schedule (config 5's per-GPU
shape).  Developer tool; bench.py prints the same figures as extra keys.
usage: python tools/timebatch_bench.py [--clip-frames 4096] [--F 64] [--host] [--no-corpus]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
import torch  # noqa: E402
from vbt_amd import shard, synth  # noqa: E402
from vbt_amd.track import Pipeline  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
MODEL = os.path.join(ROOT, "models", "efficientdet_lite0_synth.vbtm")


def corpus():
    meta = json.load(open(os.path.join(GOLD, "phases_ocsort.json")))
    main = np.load(os.path.join(GOLD, "dfs_ocsort_main.npz"))
    return {k: (int(round(float(main[f"c{k}_time"].max()) * v["fps"])), float(v["fps"])) for k, v in meta.items() if k != "001_sort"}


def one_clip(T, F, host, U=256):
    bg = synth.background(0)
    base = np.stack([synth.render(bg, t) for t in range(U)])
    frames = torch.from_numpy(np.concatenate([base, base[:F]]))          # cycle of U frames; any run of F frames is contiguous
    frames = frames.pin_memory() if host else frames.cuda()
    pipe = Pipeline(MODEL, F, max_frames=T, fps=60.0, tracker_clips=1)
    for rep in range(2):
        pipe.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f0 in range(0, T, F):
            nf = min(F, T - f0)
            s = f0 % U
            pipe.step_runs(frames[s:s + nf], [(0, 0, nf, f0 + 1)])
        best, rows, nph, ovf, ph = pipe.close(cap=512)
        cnt, rr = pipe.rows_all()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    return {"frames": T, "F": F, "host_fed": bool(host), "seconds": round(dt, 4), "frames_per_s": round(T / dt), "rows": int(rows.sum()), "phases": int(nph.sum())}


def whole_corpus(slots, host, U=8):
    clips = corpus()
    keys = sorted(clips)
    lengths = np.array([clips[k][0] for k in keys])
    fps = np.array([clips[k][1] for k in keys])
    base = np.stack([np.stack([synth.render(synth.background(int(k[:3]), 320), 11 * u) for u in range(U)]) for k in keys])
    frames = torch.from_numpy(np.concatenate([base, base], axis=1))      # [clip][2U]
    frames = frames.pin_memory() if host else frames.cuda()
    steps = shard.run_schedule(lengths, slots)
    pipe = Pipeline(MODEL, slots, max_frames=int(lengths.max()), fps=fps, tracker_clips=len(keys))
    for rep in range(2):
        pipe.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for step in steps:
            pipe.step_runs([frames[c, (f0 - 1) % U:(f0 - 1) % U + nf] for c, _, nf, f0 in step], step)
        best, rows, nph, ovf, ph = pipe.close(cap=512)
        cnt, rr = pipe.rows_all()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    total = int(lengths.sum())
    return {"clips": len(keys), "frames": total, "steps": len(steps), "slots": slots, "host_fed": bool(host), "seconds": round(dt, 4),
            "frames_per_s": round(total / dt), "rows": int(rows.sum()), "longest_run": max(nf for s in steps for _, _, nf, _ in s)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clip-frames", type=int, default=4096)
    ap.add_argument("--F", type=int, default=64)
    ap.add_argument("--host", action="store_true")
    ap.add_argument("--no-corpus", action="store_true")
    a = ap.parse_args()
    print(json.dumps({"one_clip": one_clip(a.clip_frames, a.F, a.host)}), flush=True)
    if not a.no_corpus:
        print(json.dumps({"corpus_1gpu": whole_corpus(a.F, a.host)}), flush=True)


if __name__ == "__main__":
    main()
