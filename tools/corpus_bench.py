#!/usr/bin/env python3
"""BASELINE config 5: the reference's 34-clip corpus as a synthetic stand-in with the REAL per-clip frame counts and frame
rates (tests/golden/corpus_meta.json, from dfs_ocsort), clip-sharded over the ranks by longest-processing-time packing
(vbt_amd/shard.shard_clips) and run TIME-BATCHED on every rank: each step deals the rank's 64 detector slots to its 4-5
clips in proportion to the frames they have left (shard.run_schedule), the OC-SORT steps of a clip's run are walked in frame
order inside one launch.  One all-gather of fixed-size per-clip result records at the end (RCCL on GPUs, gloo in rehearsal).
Developer tool, not the contract bench (bench.py prints the one-GPU figure as configs.corpus_1gpu).
  python tools/corpus_bench.py                                  one rank, the whole corpus
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/corpus_bench.py
  rehearsal on one GPU: VBT_BENCH_SAME_DEVICE=1 VBT_BENCH_BACKEND=gloo"""
import datetime
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

MODEL = os.path.join(ROOT, "models", "efficientdet_lite0_synth.vbtm")
META = os.path.join(ROOT, "tests", "golden", "corpus_meta.json")
SLOTS, U, PH = 64, 8, 32


def main():
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = 0 if os.environ.get("VBT_BENCH_SAME_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("VBT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device(f"cuda:{local}")
    dist = None
    if world > 1:
        import torch.distributed as dist
        tmo = datetime.timedelta(seconds=int(os.environ.get("VBT_BENCH_TIMEOUT_S", "180")))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=tmo)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=tmo)
    cdev = dev if backend == "nccl" else torch.device("cpu")
    clips = {k: (int(v[0]), float(v[1])) for k, v in json.load(open(META)).items()}
    shards = shard.shard_clips({k: v[0] for k, v in clips.items()}, world)
    mine = shards[rank]
    n = len(mine)
    lengths = np.array([clips[k][0] for k in mine])
    fps = np.array([clips[k][1] for k in mine])
    base = np.stack([np.stack([synth.render(synth.background(int(k), 320), 11 * u) for u in range(U)]) for k in mine])
    frames = torch.from_numpy(np.concatenate([base] * (2 + SLOTS // U), axis=1)).to(dev)      # [clip][cycle]: any run of <= 64 frames is contiguous
    steps = shard.run_schedule(lengths, SLOTS)
    pipe = Pipeline(MODEL, SLOTS, max_frames=int(lengths.max()), fps=fps, detection_treshold=0.5, tracker_clips=n, device=local)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for rep in range(2):                                             # the first pass warms plans, pinned buffers and clocks
        pipe.reset()
        fence()
        t0 = time.perf_counter()
        for step in steps:
            pipe.step_runs([frames[c, (f0 - 1) % U:(f0 - 1) % U + nf] for c, _, nf, f0 in step], step)
        best, rows, nph, ovf, ph = pipe.close(cap=512)                # (the synthetic detector's noisy tracks give up to ~200 short phases)
        ph = ph[:, :PH]                                               # the fixed-size record keeps the first PH
        rec = np.zeros((n, 4 + PH * 6), np.float64)
        rec[:, 0] = [int(k) for k in mine]
        rec[:, 1], rec[:, 2], rec[:, 3] = best, rows, nph
        rec[:, 4:] = ph.reshape(n, -1)
        allrec = shard.gather_records(torch.from_numpy(rec).to(cdev), dist, pad_to=max(len(sh) for sh in shards)) if dist is not None else rec
        fence()
        dt = time.perf_counter() - t0
    per_rank, rccl_ranks = None, None
    if dist is not None:
        tall = torch.zeros(world, dtype=torch.float64, device=cdev)
        tall[rank] = dt
        dist.all_reduce(tall)
        per_rank = [float(t) for t in tall.tolist()]
        dt = max(per_rank)
        ones = torch.ones(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
    if rank == 0:
        total = sum(v[0] for v in clips.values())
        print(json.dumps({"config": "34-clip corpus, clip-sharded, time-batched", "n_gpus": world, "rccl_ranks": rccl_ranks, "clips": len(allrec),
                          "frames": total, "seconds": round(dt, 4), "frames_per_s": round(total / dt), "per_rank_seconds": per_rank,
                          "rank0": {"clips": n, "frames": int(lengths.sum()), "steps": len(steps), "longest_run": max(nf for s in steps for _, _, nf, _ in s),
                                    "rows": int(rows.sum()), "overflow": int((ovf != 0).sum())}}))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
