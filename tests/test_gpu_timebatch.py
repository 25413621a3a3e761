"""Time-batched device path (VERDICT r02 row e'): the reference's unit of work is ONE video (track.py:85-126, loop
track.py:159-247), so the detector batch must be fillable with consecutive frames of one (or a few) clips and the tracker
has to walk them in order on the device.  Everything here is compared with the per-frame pipeline (one frame of a clip per
step, the form every other parity test pins against the oracle) and, on a prefix, with the oracle chain itself."""
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


def test_run_schedule_properties():
    from vbt_amd import shard
    rng = np.random.default_rng(3)
    for lengths, slots in ([[4096], 64], [[10, 3, 7], 4], [[1] * 70, 64], [rng.integers(1, 400, 34).tolist(), 64], [[5, 5], 64],
                           [[v[0] for v in corpus().values()], 64]):
        steps = shard.run_schedule(lengths, slots)
        nxt = [1] * len(lengths)
        for i, step in enumerate(steps):
            used = 0
            clips = set()
            for clip, slot0, nf, frame0 in step:
                assert slot0 == used and nf >= 1 and frame0 == nxt[clip] and clip not in clips   # dense batch, frames in order, a clip once per step
                clips.add(clip)
                used += nf
                nxt[clip] += nf
            assert used == slots or (i == len(steps) - 1 and used <= slots)   # every step but the last is full
        assert [n - 1 for n in nxt] == list(lengths)
        assert len(steps) == -(-sum(lengths) // slots)
        longest = max(nf for step in steps for _, _, nf, _ in step)
        assert longest <= -(-slots * max(lengths) // sum(lengths)) + 1        # the sequential tracker walk of a step stays short
    capped = shard.run_schedule([100, 2], 64, max_run=8)
    assert max(nf for st in capped for _, _, nf, _ in st) == 8 and sum(nf for st in capped for c, _, nf, _ in st if c == 0) == 100
    for bad in (dict(n_slots=0), dict(n_slots=4, max_run=0), dict(n_slots=4, max_run=-3)):   # used to loop for ever
        with pytest.raises(ValueError):
            shard.run_schedule([10, 3], **bad)
    assert shard.run_schedule([], 8) == [] and shard.run_schedule([0, 0], 8) == []


def _rows_equal(a, b):
    return a["id"] == b["id"] and all(np.array_equal(np.asarray(a[k]), np.asarray(b[k])) for k in COLS)


@pytest.mark.gpu
def test_abi_rejects_bad_runs(model_path):
    import ctypes
    import torch
    from vbt_amd import _lib
    from vbt_amd.ocsort import MultiClipTracker
    L = _lib.lib()
    trk = MultiClipTracker(2, 64, max_age=30, asso_func="diou", iou_threshold=0.1)
    b = torch.zeros((8, 25, 4), dtype=torch.float32, device="cuda")
    s = torch.zeros((8, 25), dtype=torch.float32, device="cuda")
    c = torch.zeros((8,), dtype=torch.int32, device="cuda")

    def call(runs):
        ra = (_lib.Run * len(runs))(*[_lib.Run(*r) for r in runs])
        return L.vbt_tracker_update_from_detections_seq(trk.handle, b.data_ptr(), s.data_ptr(), c.data_ptr(), 8, ra, len(runs), 0.5, None)

    assert call([(0, 0, 1, 8, 1, 1, 30.0)]) == 0
    assert call([(0, 0, 1, 9, 1, 1, 30.0)]) == -1          # runs past the batch
    assert call([(0, 0, 1, 4, 1, 1, 30.0), (0, 4, 1, 4, 5, 1, 30.0)]) == -1   # one clip twice in a call
    assert call([(2, 0, 1, 4, 1, 1, 30.0)]) == -1          # clip outside the tracker
    assert call([(0, 0, 1, 4, 0, 1, 30.0)]) == -1          # frame numbers are 1-based
    assert call([(0, 0, 1, 4, 1, 1, 0.0)]) == -1           # fps
    assert call([(-1, 0, 1, 4, 1, 1, 30.0), (1, 4, 1, 4, 1, 1, 30.0)]) == 0   # an empty descriptor is skipped
    torch.cuda.synchronize()
    assert trk.status(0)["rows"] == 0 and trk.status(0)["frame_count"] == 0   # empty frames never step the tracker (track.py:180-181)


@pytest.mark.gpu
def test_one_clip_time_batched_equals_per_frame_and_oracle(oracle_lib, model_path):
    """One clip, 200 frames, 64 consecutive frames per detector batch (the last batch is partial) == the per-frame pipeline ==
    the oracle chain on the first 96 frames; host-fed and per-run sources give the same rows."""
    import torch
    from oracle import ocsort_np
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    T, F = 200, 64
    frames = synth.clip_frames(77, 0, T)
    fd = torch.from_numpy(frames).cuda()
    ref = Pipeline(model_path, 1, max_frames=T, fps=60.0, rows_per_frame=25)
    for t in range(T):
        ref.step(fd[t:t + 1])
    ref.finish()
    want = ref.rows(0)
    want_ph = ref.phases(0)
    assert len(want["id"]) > 30
    host = torch.from_numpy(frames).pin_memory()
    for mode in ("device", "host", "device_sources", "host_sources"):
        pipe = Pipeline(model_path, F, max_frames=T, fps=60.0, rows_per_frame=25, tracker_clips=1)
        for t0 in range(0, T, F):
            nf = min(F, T - t0)
            src = (fd if mode.startswith("device") else host)[t0:t0 + nf]
            pipe.step_runs([src] if mode.endswith("sources") else src, [(0, 0, nf, t0 + 1)])
        pipe.finish()
        got = pipe.rows(0)
        assert _rows_equal(got, want), mode
        gp = pipe.phases(0)
        assert gp[0] == want_ph[0] and np.array_equal(gp[1], want_ph[1]), mode
    P = 96
    ob, os_, oc, on = oracle_lib.run_batch(model_path, frames[:P], threads=16)
    dets = [np.asarray([[ob[t, i, 1], ob[t, i, 0], ob[t, i, 3], ob[t, i, 2], os_[t, i], 0.0] for i in range(on[t]) if os_[t, i] >= 0.5],
                       np.float64).reshape(-1, 6) for t in range(P)]
    orc = ocsort_np.track_boxes(dets, [(t + 1) / 60.0 for t in range(P)])
    npre = len(orc["id"])
    assert got["id"][:npre] == orc["id"] and all(np.array_equal(np.asarray(got[k])[:npre], np.asarray(orc[k])) for k in COLS)


@pytest.mark.gpu
def test_frame_step_and_empty_frames(model_path):
    """frame_step = the reference's `frame_count % 16` stride (track.py:161-169): skipped frames advance time only; frames
    without a detection inside a run leave the tracker untouched (track.py:180-181)."""
    import torch
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    T, stride = 96, 3
    frames = synth.clip_frames(5, 0, T)
    blank = np.zeros_like(frames[0])
    frames[30:36] = blank                                      # six frames the detector finds nothing in
    fd = torch.from_numpy(frames).cuda()
    ref = Pipeline(model_path, 1, max_frames=T, fps=30.0, rows_per_frame=25)
    for t in range(T):
        if (t + 1) % stride:
            ref.skip_frames(1)
            continue
        ref.step(fd[t:t + 1])
    ref.finish()
    want = ref.rows(0)
    picked = fd[stride - 1::stride].contiguous()              # frames 3, 6, 9, ...
    pipe = Pipeline(model_path, 16, max_frames=T, fps=30.0, rows_per_frame=25, tracker_clips=1)
    for i0 in range(0, picked.shape[0], 16):
        nf = min(16, picked.shape[0] - i0)
        pipe.step_runs(picked[i0:i0 + nf], [(0, 0, nf, (i0 + 1) * stride, stride)])
    pipe.finish()
    assert _rows_equal(pipe.rows(0), want) and len(want["id"]) > 10
    assert pipe.tracker.status(0)["frame_count"] == ref.tracker.status(0)["frame_count"] <= T // stride


@pytest.mark.gpu
def test_step_runs_validates_per_run_sources(model_path):
    """Per-run source lists: raw pointers base + f * frame_bytes go to the gather kernel, so a short, strided,
    wrongly shaped or wrongly typed source must be refused on the host; a source resolution whose frame size is not a multiple
    of 16 bytes (7 x 5 x 3 = 105) takes the copy path and gives the rows of the assembled-batch form."""
    import torch
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    bg = synth.background(4)
    fd = torch.from_numpy(np.stack([synth.render(bg, t) for t in range(12)])).cuda()
    pipe = Pipeline(model_path, 8, max_frames=32, fps=30.0, rows_per_frame=25, tracker_clips=2)
    runs = [(0, 0, 4, 1), (1, 4, 4, 1)]
    for bad in ([fd[:3], fd[4:8]],                                     # too few frames for the run
                [fd[:8:2], fd[4:8]],                                   # strided
                [fd[:4, :, :160], fd[4:8]],                            # another shape (and not contiguous)
                [fd[:4].to(torch.int8), fd[4:8]],                      # another dtype
                [fd[:4], fd[4:8].cpu()],                               # mixed devices
                [fd[:4]]):                                             # one source per run
        with pytest.raises(ValueError):
            pipe.step_runs(bad, runs)
    with pytest.raises(ValueError):
        pipe.step_runs([fd[:4], fd[4:7]], [(0, 0, 4, 1), (1, 5, 3, 1)])   # hole at slot 4
    pipe.step_runs([fd[:4], fd[4:8]], runs)                             # the valid call still runs after the refusals
    pipe.finish()
    assert len(pipe.rows(0)["id"]) > 0
    small = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (8, 7, 5, 3), dtype=np.uint8)).cuda()
    a = Pipeline(model_path, 8, max_frames=32, fps=30.0, rows_per_frame=25, tracker_clips=2)
    a.step_runs([small[:4], small[4:]], runs, src_hw=(7, 5))
    a.finish()
    b = Pipeline(model_path, 8, max_frames=32, fps=30.0, rows_per_frame=25, tracker_clips=2)
    b.step_runs(small, runs, src_hw=(7, 5))
    b.finish()
    assert all(_rows_equal(a.rows(c), b.rows(c)) for c in range(2))
    da, db = a.detections(), b.detections()
    assert all(np.array_equal(x, y) for x, y in zip(da, db))


@pytest.mark.gpu
def test_config2_full_length_clip(model_path):
    """BASELINE config 2 at its full T = 4096: one clip, 64 consecutive frames per step == one frame per step, row for row
    and phase for phase; step_seq (F frames of every clip) on two clips gives the same rows as well."""
    import torch
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    T, F = 4096, 64
    bg = synth.background(0)
    frames = np.stack([synth.render(bg, t) for t in range(T)])
    fd = torch.from_numpy(frames).cuda()
    ref = Pipeline(model_path, 1, max_frames=T, fps=60.0)
    for t in range(T):
        ref.step(fd[t:t + 1])
    rb, rr, rn, ro, rph = ref.close(cap=512)
    rc, rrows = ref.rows_all()
    assert rr[0] > 1000 and rn[0] >= 2 and ro[0] == 0
    pipe = Pipeline(model_path, F, max_frames=T, fps=60.0, tracker_clips=1)
    for t0 in range(0, T, F):
        pipe.step_runs(fd[t0:t0 + F], [(0, 0, F, t0 + 1)])
    b, r, n, o, ph = pipe.close(cap=512)
    c, rows = pipe.rows_all()
    assert np.array_equal(b, rb) and np.array_equal(r, rr) and np.array_equal(n, rn) and np.array_equal(ph, rph)
    assert np.array_equal(rows[0, :c[0]], rrows[0, :rc[0]])
    two = Pipeline(model_path, 2 * 32, max_frames=T // 2, fps=60.0, tracker_clips=2)
    halves = fd.reshape(2, T // 2, 320, 320, 3)               # clip 1 = the second half of the sequence, as its own clip
    for t0 in range(0, T // 2, 32):
        two.step_seq(halves[:, t0:t0 + 32].contiguous())
    two.close(cap=512)
    c2, rows2 = two.rows_all()
    assert np.array_equal(rows2[0, :c2[0]], rrows[0, :c2[0]]) and c2[1] > 500


@pytest.mark.gpu
def test_corpus_time_batched_equals_ragged_per_frame(model_path):
    """The 34-clip corpus (real frame counts / frame rates) on ONE GPU: run_schedule's time-batched steps (64 slots dealt in
    proportion to the frames left) reproduce the per-frame ragged batch exactly - export ids, row counts, phases, rows."""
    import torch
    from vbt_amd import shard, synth
    from vbt_amd.track import Pipeline
    clips = corpus()
    keys = sorted(clips)
    lengths = np.array([clips[k][0] for k in keys])
    fps = np.array([clips[k][1] for k in keys])
    n, U = len(keys), 8
    frames = torch.from_numpy(np.stack([np.stack([synth.render(synth.background(int(k[:3]), 320), 11 * u) for u in range(U)]) for k in keys])).cuda()   # [clip][U]
    T = int(lengths.max())
    ref = Pipeline(model_path, n, max_frames=T, fps=fps, detection_treshold=0.5)
    fr = frames.transpose(0, 1).contiguous()
    for t in range(T):
        ref.step(fr[t % U], active=t < lengths)
    rb, rr, rn, ro, rph = ref.close(cap=512)
    rc, rrows = ref.rows_all()
    del ref
    pipe = Pipeline(model_path, 64, max_frames=T, fps=fps, detection_treshold=0.5, tracker_clips=n)
    steps = shard.run_schedule(lengths, 64)
    assert len(steps) == -(-int(lengths.sum()) // 64)
    cyc = torch.cat([frames, frames], dim=1)                  # [clip][2U]: a run of <= U frames starting anywhere in the cycle is contiguous
    for step in steps:
        assert all(nf <= U for _, _, nf, _ in step)
        pipe.step_runs([cyc[c, (f0 - 1) % U:(f0 - 1) % U + nf] for c, _, nf, f0 in step], step)
    b, r, nph, o, ph = pipe.close(cap=512)
    c, rows = pipe.rows_all()
    assert np.all(o == 0) and np.array_equal(b, rb) and np.array_equal(r, rr) and np.array_equal(nph, rn) and np.array_equal(ph, rph)
    for i in range(n):
        assert np.array_equal(rows[i, :c[i]], rrows[i, :rc[i]]), keys[i]
    assert int(r.sum()) > 34 * 300
