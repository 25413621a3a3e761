"""Multi-GPU partitioning of the hot path: clips are independent (detector
has no state; Kalman + phase state are per clip), so ranks own disjoint clips and never exchange
data on the path.  The only exchange is one fixed-size all-gather of the per-clip result records at
the end (RCCL over xGMI on the GPUs, gloo in the CPU tests): latency-bound, a few KB per rank."""
import numpy as np


def shard_clips(work, world):
    """Longest-processing-time bin packing of {clip: frames} over `world` ranks -> list of clip lists."""
    order = sorted(work, key=lambda c: (-work[c], c))
    loads = [0] * world
    shards = [[] for _ in range(world)]
    for c in order:
        r = min(range(world), key=lambda i: (loads[i], i))
        shards[r].append(c)
        loads[r] += work[c]
    return shards


def gather_records(rec, dist, pad_to):
    """rec: tensor [n_i, k] of this rank's records (n_i <= pad_to).  Returns the valid rows of all ranks
    (numpy) on every rank via ONE all_gather of equal-size blocks (row 0 of each block carries n_i)."""
    import torch
    world = dist.get_world_size()
    k = rec.shape[1] if rec.ndim == 2 and rec.shape[0] else 1
    kk = torch.tensor([k], dtype=torch.int64, device=rec.device)
    dist.all_reduce(kk, op=dist.ReduceOp.MAX)
    k = int(kk.item())
    block = torch.zeros((pad_to + 1, k), dtype=torch.float64, device=rec.device)
    n = rec.shape[0]
    block[0, 0] = n
    if n:
        block[1:1 + n] = rec.to(torch.float64)
    out = [torch.empty_like(block) for _ in range(world)]
    dist.all_gather(out, block)
    rows = [o[1:1 + int(o[0, 0].item())].cpu().numpy() for o in out]
    return np.concatenate(rows, axis=0) if rows else np.zeros((0, k))
