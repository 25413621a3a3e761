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


def slot_schedule(lengths, n_slots):
    """Clips of `lengths[c]` frames run through `n_slots` detector slots: LPT assignment of clips to slot queues, a slot
    starts its next clip the step after the previous one ended.  Returns (clip_map [T, n_slots] int32 with -1 for idle,
    frame_idx [T, n_slots] int32, 1-based), T = makespan in steps."""
    queues = shard_clips({c: int(n) for c, n in enumerate(lengths)}, n_slots)
    T = max(sum(lengths[c] for c in q) for q in queues) if queues else 0
    cmap = np.full((T, n_slots), -1, np.int32)
    fidx = np.zeros((T, n_slots), np.int32)
    for s, q in enumerate(queues):
        t = 0
        for c in q:
            n = int(lengths[c])
            cmap[t:t + n, s] = c
            fidx[t:t + n, s] = np.arange(1, n + 1)
            t += n
    return cmap, fidx


def run_schedule(lengths, n_slots, max_run=None):
    """Time-batched schedule of clips of `lengths[c]` frames through a detector batch of `n_slots` frames per step: every
    step hands each unfinished clip a RUN of consecutive frames, the slots being dealt in proportion to the frames the clips
    have left (largest remainders first, ties to the lower clip), so that every slot of every step but the last carries a
    frame and all clips end together - the sequential OC-SORT walk of a step is then as short as it can be
    (a clip's run in a step is at most ceil(n_slots * frames it has left / frames left in all) + 1: its quota rounded
    down plus one spare slot).  max_run (>= 1) caps the frames of one clip per step.
    Returns a list of steps; a step is a list of (clip, slot0, n_frames, frame0) with frame0 1-based."""
    if int(n_slots) < 1:
        raise ValueError("run_schedule: n_slots must be at least 1")
    if max_run is not None and int(max_run) < 1:
        raise ValueError("run_schedule: max_run must be at least 1 (a clip that may take no frame never ends)")
    left = np.asarray(lengths, np.int64).copy()
    if (left < 0).any():
        raise ValueError("run_schedule: negative clip length")
    done = np.zeros_like(left)
    steps = []
    while left.sum() > 0:
        total = int(left.sum())
        if total <= n_slots and (max_run is None or left.max() <= max_run):
            give = left.copy()
        else:
            quota = left * (n_slots / total)
            give = np.minimum(np.floor(quota).astype(np.int64), left)
            if max_run is not None:
                give = np.minimum(give, max_run)
            spare = n_slots - int(give.sum())
            cap = left if max_run is None else np.minimum(left, max_run)
            order = sorted(range(len(left)), key=lambda c: (-(quota[c] - np.floor(quota[c])), c))
            while spare > 0:
                moved = False
                for c in order:
                    if spare > 0 and give[c] < cap[c]:
                        give[c] += 1
                        spare -= 1
                        moved = True
                if not moved:
                    break
        step, slot = [], 0
        for c in range(len(left)):
            if give[c] > 0:
                step.append((c, slot, int(give[c]), int(done[c]) + 1))
                slot += int(give[c])
        done += give
        left -= give
        steps.append(step)
    return steps


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


# ---- frame-major mode: ONE long clip ----------------------------------------------
# Detection has no state, so rank r detects the contiguous frame chunk r; tracking is sequential per clip, so
# a single owner consumes all detections.  Exchange = one all-gather of fixed-size per-frame records
# [25 x (4 box + 1 score) f32 + count] = 504 B/frame (latency-bound: 10 k frames = 5 MB in total).
RECORD_FLOATS = 25 * 5 + 1


def frame_chunks(n_frames, world):
    """Contiguous, balanced [start, stop) per rank."""
    base, rem = divmod(n_frames, world)
    out, s = [], 0
    for r in range(world):
        e = s + base + (1 if r < rem else 0)
        out.append((s, e))
        s = e
    return out


def pack_detection_records(boxes, scores, counts):
    """boxes [F,25,4], scores [F,25], counts [F] -> float32 records [F,126]."""
    F = len(counts)
    rec = np.zeros((F, RECORD_FLOATS), np.float32)
    rec[:, :100] = np.asarray(boxes, np.float32).reshape(F, 100)
    rec[:, 100:125] = np.asarray(scores, np.float32)
    rec[:, 125] = np.asarray(counts, np.float32)
    return rec


def unpack_detection_records(rec):
    rec = np.asarray(rec, np.float32)
    return rec[:, :100].reshape(-1, 25, 4), rec[:, 100:125], rec[:, 125].astype(np.int32)


def gather_detection_records(rec, n_frames, dist, device=None):
    """rec: this rank's records for its chunk of frame_chunks(n_frames, world).  Returns all n_frames records in
    frame order on every rank via ONE all_gather_into_tensor of equal (padded) blocks."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    chunks = frame_chunks(n_frames, world)
    per = max(e - s for s, e in chunks)
    block = torch.zeros((per, RECORD_FLOATS), dtype=torch.float32, device=device)
    n = chunks[rank][1] - chunks[rank][0]
    assert rec.shape == (n, RECORD_FLOATS)
    if n:
        block[:n] = torch.as_tensor(rec, dtype=torch.float32, device=device)
    out = torch.empty((world * per, RECORD_FLOATS), dtype=torch.float32, device=device)
    dist.all_gather_into_tensor(out, block)
    out = out.cpu().numpy().reshape(world, per, RECORD_FLOATS)
    return np.concatenate([out[r, :chunks[r][1] - chunks[r][0]] for r in range(world)], axis=0)


def records_to_tracker_inputs(rec, fps, threshold=0.5):
    """Gathered records -> (dets [F,1,25,6] float64, counts [F,1], times [F,1]) for MultiClipTracker.update_frames:
    threshold of reference odt.py:70-75 and reorder of odt.py:102-118, time = frame_count / fps (track.py:169)."""
    boxes, scores, counts = unpack_detection_records(rec)
    F = len(counts)
    dets = np.zeros((F, 1, 25, 6), np.float64)
    cnt = np.zeros((F, 1), np.int32)
    for f in range(F):
        k = 0
        for i in range(counts[f]):
            if scores[f, i] >= threshold:
                dets[f, 0, k] = (boxes[f, i, 1], boxes[f, i, 0], boxes[f, i, 3], boxes[f, i, 2], scores[f, i], 0.0)
                k += 1
        cnt[f, 0] = k
    times = ((np.arange(F, dtype=np.float64) + 1.0) / fps).reshape(F, 1)
    return dets, cnt, times
