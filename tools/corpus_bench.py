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

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
os.environ.setdefault("VBT_PLAN_FILE", os.path.join(ROOT, "profiles", "plan_lite0"))
import torch  # noqa: E402
import bench  # noqa: E402


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
    res = bench.corpus_sharded(torch, dist, dev, cdev, rank, world, local)      # the body lives in bench.py (configs.corpus_sharded at N > 1)
    if rank == 0:
        print(json.dumps(dict(res, config="34-clip corpus, clip-sharded, time-batched")))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
