"""BASELINE config 5: the reference's 34-clip corpus (reference track.py:85-126 loops over it; dfs_ocsort/ holds its
outputs) sharded over 8 ranks.  The clips here are synthetic stand-ins with the REAL per-clip frame counts and frame rates
(tests/golden, from dfs_ocsort); the 8 ranks' shards run one after the other on the one GPU of the test box."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

COLS = ("time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width")


def corpus():
    meta = json.load(open(os.path.join(GOLDEN, "phases_ocsort.json")))
    main = np.load(os.path.join(GOLDEN, "dfs_ocsort_main.npz"))
    return {k: (int(round(float(main[f"c{k}_time"].max()) * v["fps"])), float(v["fps"])) for k, v in meta.items() if k != "001_sort"}


def test_corpus_shape_and_sharding():
    from vbt_amd import shard
    clips = corpus()
    lens = {k: v[0] for k, v in clips.items()}
    assert len(clips) == 34 and min(lens.values()) >= 699 and max(lens.values()) >= 3200 and sum(lens.values()) > 53931
    assert sorted(v[1] for v in clips.values()).count(60.0) == 3
    shards = shard.shard_clips(lens, 8)
    assert sorted(c for s in shards for c in s) == sorted(clips) and all(4 <= len(s) <= 5 for s in shards)
    loads = [sum(lens[c] for c in s) for s in shards]
    assert max(loads) / (sum(loads) / 8) < 1.08                      # LPT packing: the slowest rank is within 8 % of the mean


@pytest.mark.gpu
def test_full_length_corpus_rank_by_rank(model_path):
    """Every rank's shard at FULL clip lengths (699...3243 frames) as a ragged batch, then again through fewer detector
    slots than clips (slot_schedule): no track / row / phase overflow, every clip stepped exactly its own number of
    frames, the slot-queued run reproduces the ragged run, and the 8 result blocks cover the corpus exactly once."""
    import torch
    from vbt_amd import shard, synth
    from vbt_amd.track import Pipeline
    clips = corpus()
    shards = shard.shard_clips({k: v[0] for k, v in clips.items()}, 8)
    st = torch.cuda.current_stream().cuda_stream
    U = 4
    records = {}
    for rank, mine in enumerate(shards):
        n = len(mine)
        lengths = np.array([clips[k][0] for k in mine])
        fps = np.array([clips[k][1] for k in mine])
        frames = torch.from_numpy(np.stack([np.stack([synth.render(synth.background(int(k[:3]), 320), 9 * u) for k in mine]) for u in range(U)])).cuda()   # [U][clip]
        T = int(lengths.max())
        pipe = Pipeline(model_path, n, max_frames=T, fps=fps, detection_treshold=0.5)
        for t in range(T):
            pipe.step(frames[t % U], st, active=t < lengths)
        best, rows_n, nph, ovf, ph = pipe.close(cap=64)
        assert np.all(ovf == 0) and np.all(rows_n <= pipe.tracker.rows_cap) and np.all(nph <= 64)
        for c in range(n):
            stt = pipe.tracker.status(c)
            assert stt["overflow"] == 0 and stt["rows_overflow"] == 0 and stt["trackers"] <= 64
            assert stt["frame_count"] <= lengths[c]                  # tracker steps = frames with a detection (track.py:180-181)
        counts, rows = pipe.rows_all()
        assert np.array_equal(counts, rows_n)
        for c in range(n):                                           # no row beyond the clip's last frame; times on the clip's own fps grid
            tt = rows[c, :counts[c]]["time"]
            if counts[c]:
                assert tt.max() <= lengths[c] / fps[c] + 1e-12 and np.allclose(np.round(tt * fps[c]), tt * fps[c], atol=1e-6)
        if rank in (0, 5):                                           # the same shard through 2 detector slots (LPT clip queues per slot)
            cmap, fidx = shard.slot_schedule(lengths, 2)
            assert cmap.shape[0] >= int(np.ceil(lengths.sum() / 2)) and all(((cmap == c).sum() == lengths[c]) for c in range(n))
            p2 = Pipeline(model_path, 2, max_frames=T, fps=fps, detection_treshold=0.5, tracker_clips=n)
            for t in range(cmap.shape[0]):
                cm = cmap[t]
                sel = torch.stack([frames[(int(fidx[t, s]) - 1) % U, max(int(cm[s]), 0)] for s in range(2)])
                p2.step(sel, st, clip_map=cm, frame_idx=fidx[t])
            b2, r2, n2, o2, ph2 = p2.close(cap=64)
            assert np.array_equal(b2, best) and np.array_equal(r2, rows_n) and np.array_equal(n2, nph) and np.array_equal(ph2, ph)
            c2, rows2 = p2.rows_all()
            assert all(np.array_equal(rows2[c, :c2[c]], rows[c, :counts[c]]) for c in range(n))
        for c, k in enumerate(mine):
            records[k] = (rank, int(best[c]), int(rows_n[c]), int(nph[c]))
        del pipe
    assert sorted(records) == sorted(clips) and sum(r[2] for r in records.values()) > 34 * 100


@pytest.mark.gpu
def test_truncated_corpus_equals_per_clip_oracle(oracle_lib, model_path):
    """Ragged shards with the corpus' length RATIOS (lengths / 150 -> 4...21 frames): every clip's rows equal the oracle
    chain (oracle detector -> oracle OC-SORT) run on that clip alone, like the reference's per-clip loop."""
    import torch
    from oracle import ocsort_np
    from vbt_amd import shard, synth
    from vbt_amd.track import Pipeline
    clips = corpus()
    shards = shard.shard_clips({k: v[0] for k, v in clips.items()}, 8)
    st = torch.cuda.current_stream().cuda_stream
    total_rows = 0
    for rank in (0, 3, 7):
        mine = shards[rank]
        n = len(mine)
        lengths = np.array([max(clips[k][0] // 150, 3) for k in mine])
        fps = np.array([clips[k][1] for k in mine])
        T = int(lengths.max())
        frames = np.stack([np.stack([synth.render(synth.background(int(k[:3]) + 40, 320), 5 * t + int(k[:3])) for k in mine]) for t in range(T)])   # [T][clip]
        pipe = Pipeline(model_path, n, max_frames=T, fps=fps, detection_treshold=0.4, rows_per_frame=25)
        fd = torch.from_numpy(frames).cuda()
        for t in range(T):
            pipe.step(fd[t], st, active=t < lengths)
        pipe.close()
        counts, rows = pipe.rows_all()
        for c in range(n):
            L = int(lengths[c])
            ob, os_, oc, on = oracle_lib.run_batch(model_path, np.ascontiguousarray(frames[:L, c]), threads=4)
            dets = [np.asarray([[ob[t, i, 1], ob[t, i, 0], ob[t, i, 3], ob[t, i, 2], os_[t, i], 0.0] for i in range(on[t]) if os_[t, i] >= np.float32(0.4)],
                               np.float64).reshape(-1, 6) for t in range(L)]
            want = ocsort_np.track_boxes(dets, [(t + 1) / fps[c] for t in range(L)])
            got = rows[c, :counts[c]]
            assert got["id"].tolist() == want["id"], (rank, c)
            for k in COLS:
                assert np.array_equal(got[k], np.asarray(want[k])), (rank, c, k)
            total_rows += counts[c]
    assert total_rows > 50
