"""GPU end-to-end: frames -> detect -> NMS -> OC-SORT -> export id -> rep analysis through the fused
Pipeline, against the oracle chain on the same seeded frames; plus the odt/track call shapes."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
COLS = ("time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width")


def _oracle_chain(oracle_lib, model_path, frames, fps=60.0, thr=0.5):
    from oracle import ocsort_np
    T, n = frames.shape[:2]
    ob, os_, oc, on = oracle_lib.run_batch(model_path, frames.reshape(-1, *frames.shape[2:]), threads=8)
    ob, os_, on = ob.reshape(T, n, 25, 4), os_.reshape(T, n, 25), on.reshape(T, n)
    rows = []
    for c in range(n):
        dets = [np.asarray([[ob[t, c, i, 1], ob[t, c, i, 0], ob[t, c, i, 3], ob[t, c, i, 2], os_[t, c, i], 0.0]
                            for i in range(on[t, c]) if os_[t, c, i] >= thr], np.float64).reshape(-1, 6) for t in range(T)]
        rows.append(ocsort_np.track_boxes(dets, [(t + 1) / fps for t in range(T)]))
    return (ob, os_, on), rows


def test_pipeline_equals_oracle_chain(oracle_lib, model_path):
    import torch
    from oracle import velocity as ov
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    n, T = 6, 24
    frames = np.stack([np.stack([synth.render(synth.background(40 + c), 9 * c + t) for c in range(n)]) for t in range(T)])
    fd = torch.from_numpy(frames).to("cuda:0")
    pipe = Pipeline(model_path, n, max_frames=T, fps=60.0)
    st = torch.cuda.current_stream().cuda_stream
    dets = []
    for t in range(T):
        pipe.step(fd[t].data_ptr(), st)
        dets.append(pipe.detections())
    pipe.finish(st)
    (ob, os_, on), rows = _oracle_chain(oracle_lib, model_path, frames)
    for t in range(T):
        b, s, c, k = dets[t]
        assert np.array_equal(k, on[t]) and np.array_equal(s, os_[t]) and np.array_equal(b, ob[t])
    total = 0
    for c in range(n):
        got, want = pipe.rows(c), rows[c]
        assert got["id"] == want["id"]
        for k in COLS:
            assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), (c, k)
        total += len(got["id"])
        # export id + phases
        ids = np.asarray(want["id"])
        cum = {}
        for tid in np.unique(ids):
            m = ids == tid
            d = np.sqrt(np.diff(np.asarray(want["x"])[m]) ** 2 + np.diff(np.asarray(want["y"])[m]) ** 2)
            if len(d):
                cum[int(tid)] = d.sum()
        best, ph = pipe.phases(c)
        if cum:
            want_id = max(cum, key=cum.get)
            assert best == want_id
            m = ids == want_id
            wp = ov.analyze_track(*[np.asarray(want[k])[m].tolist() for k in COLS])
            assert [list(r) for r in ph] == [p.as_row() for p in wp]
        else:
            assert best == -1 and len(ph) == 0
    assert total > 20


def test_track_function_and_run_odt_call_shapes(oracle_lib, model_path):
    """track(src, interpreter, ...) with Interpreter / run_odt / OCSort drop-ins (reference track.py:129-260)."""
    from vbt_amd import synth
    from vbt_amd.interpreter import Interpreter
    from vbt_amd.odt import results_to_sorttracker_inputs, run_odt
    from vbt_amd.track import track
    frames = synth.clip_frames(3, 0, 10)
    it = Interpreter(model_path=model_path, num_threads=4)
    it.allocate_tensors()
    assert tuple(it.get_input_details()[0]["shape"]) == (1, 320, 320, 3)
    res = run_odt(frames[0], it, threshold=0.0)
    det = oracle_lib.OracleDetector(model_path)
    ob, os_, oc, on = det.run(frames[0])
    assert len(res) == on == 25
    assert all(np.array_equal(r["bounding_box"], ob[i]) and r["score"] == os_[i] for i, r in enumerate(res))
    assert results_to_sorttracker_inputs(res).shape == (25, 6) and results_to_sorttracker_inputs([]).shape == (0, 6)
    data = track(frames, it, detection_treshold=0.5, fps=60.0)
    _, rows = _oracle_chain(oracle_lib, model_path, frames[:, None])
    assert data["id"] == rows[0]["id"]
    for k in COLS:
        assert np.array_equal(np.asarray(data[k]), np.asarray(rows[0][k])), k
    data16 = track(frames, it, frame_stride=16)      # the snapshot's `frame_count % 16` (track.py:166)
    assert data16["id"] == []


def test_preprocess_resize_matches_oracle():
    """reference odt.py:10-19 at real source sizes (portrait 1080x1920 like the reference clips), up and down."""
    from oracle.preprocess import preprocess_image as ref
    from vbt_amd.odt import preprocess_image
    rng = np.random.default_rng(5)
    for (H, W), (h, w) in (((1920, 1080), (320, 320)), ((416, 416), (320, 320)), ((200, 333), (448, 448)), ((320, 320), (320, 320)), ((7, 5), (320, 320))):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        got, orig = preprocess_image(img, (h, w))
        assert got.shape == (1, h, w, 3) and got.dtype == np.uint8 and orig is not None
        assert np.array_equal(got, ref(img, (h, w))), ((H, W), (h, w))


def test_pipeline_source_resolution_frames(oracle_lib, model_path):
    """N2: 416x416 BGR source frames (the size of the reference's data/test images) resized + channel-swapped on the
    device inside the pipeline == oracle preprocess (reference odt.py:10-19, track.py:171) + oracle detector."""
    import torch
    from oracle.preprocess import preprocess_image as ref_pre
    from vbt_amd.track import Pipeline
    rng = np.random.default_rng(11)
    n, T = 3, 4
    src = rng.integers(0, 256, (T, n, 416, 416, 3), dtype=np.uint8)
    src = (src // 32 * 32 + np.repeat(np.repeat(rng.integers(0, 32, (T, n, 52, 52, 3), dtype=np.uint8), 8, 2), 8, 3)).astype(np.uint8)
    fd = torch.from_numpy(src).to("cuda:0")
    pipe = Pipeline(model_path, n, max_frames=T, fps=30.0)
    st = torch.cuda.current_stream().cuda_stream
    got = []
    for t in range(T):
        pipe.step(fd[t].data_ptr(), st, src_hw=(416, 416), swap_rb=True)
        got.append(pipe.detections())
    pipe.finish(st)
    want_frames = np.stack([np.stack([ref_pre(src[t, c], (320, 320), swap_rb=True)[0] for c in range(n)]) for t in range(T)])
    ob, os_, oc, on = oracle_lib.run_batch(model_path, want_frames.reshape(-1, 320, 320, 3), threads=8)
    ob, os_, on = ob.reshape(T, n, 25, 4), os_.reshape(T, n, 25), on.reshape(T, n)
    for t in range(T):
        b, s, c, k = got[t]
        assert np.array_equal(k, on[t]) and np.array_equal(s, os_[t]) and np.array_equal(b, ob[t]), t


def test_frame_major_mode_equals_clip_mode(model_path):
    """SURVEY 8e second mode on one long clip: detect contiguous frame chunks independently (as two ranks would),
    exchange the 504-byte records, track on the owner == the fused per-frame pipeline on the same clip."""
    import torch
    from vbt_amd import shard, synth
    from vbt_amd.interpreter import Interpreter
    from vbt_amd.ocsort import MultiClipTracker
    from vbt_amd.track import Pipeline
    T = 21
    frames = synth.clip_frames(33, 0, T)
    recs = []
    for s, e in shard.frame_chunks(T, 2):                               # each "rank" batches its whole chunk
        it = Interpreter(model_path, max_batch=e - s)
        b, sc, c, k = it.detect(frames[s:e])
        recs.append(shard.pack_detection_records(b, sc, k))
    dets, cnt, times = shard.records_to_tracker_inputs(np.concatenate(recs), fps=60.0)
    mc = MultiClipTracker(1, 25 * T, max_age=30, asso_func="diou", iou_threshold=0.1)
    mc.update_frames(dets, cnt, times)
    pipe = Pipeline(model_path, 1, max_frames=T, fps=60.0, rows_per_frame=25)
    fd = torch.from_numpy(frames).to("cuda:0")
    for t in range(T):
        pipe.step(fd[t:t + 1].data_ptr())
    pipe.finish()
    a, b2 = mc.rows(0), pipe.rows(0)
    assert a["id"] == b2["id"] and len(a["id"]) > 0
    for k in COLS:
        assert np.array_equal(np.asarray(a[k]), np.asarray(b2[k])), k


def test_errors_are_loud(model_path):
    from vbt_amd import _lib
    from vbt_amd.interpreter import Interpreter
    with pytest.raises(_lib.VbtError):
        Interpreter("/nonexistent/model.vbtm")
    it = Interpreter(model_path, max_batch=2)
    with pytest.raises(_lib.VbtError):
        it.detect(np.zeros((3, 320, 320, 3), np.uint8))          # batch > max_batch
    with pytest.raises(ValueError):
        it.get_signature_runner()(images=np.zeros((1, 100, 100, 3), np.uint8))


def test_summary_equals_per_clip_queries(model_path):
    """vbt_tracker_summary (all clips, four copies) against vbt_tracker_phases / vbt_tracker_status clip by clip."""
    import torch
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    n, T = 5, 40
    pipe = Pipeline(model_path, n, max_frames=T, fps=60.0, detection_treshold=0.3, rows_per_frame=25)
    st = torch.cuda.current_stream().cuda_stream
    for t in range(T):
        fd = torch.from_numpy(synth.batch_frames(range(20, 20 + n), 2 * t)).cuda()
        pipe.step(fd, st)
        torch.cuda.current_stream().synchronize()
    pipe.finish(st)
    best, rows, nph, ovf, ph = pipe.tracker.summary(cap=48)
    assert rows.sum() > 0
    for c in range(n):
        b, p = pipe.phases(c)
        s = pipe.tracker.status(c)
        assert (b, s["rows"], len(p)) == (best[c], rows[c], nph[c]) and ovf[c] == 0
        assert np.array_equal(ph[c, :nph[c]], p)


def test_ragged_clips(model_path, oracle_lib):
    """Clips of different lengths in one batch (BASELINE config 5 has 34 clips of 699-3243 frames): a clip that has
    ended is masked out of the tracker step; every clip's rows equal the oracle run on that clip alone."""
    import torch
    from oracle import ocsort_np
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    lengths = [9, 5, 7]
    n, T = len(lengths), max(lengths)
    pipe = Pipeline(model_path, n, max_frames=T, fps=30.0, detection_treshold=0.3, rows_per_frame=25)
    st = torch.cuda.current_stream().cuda_stream
    clips = [synth.clip_frames(40 + c, 3 * c, lengths[c]) for c in range(n)]
    for t in range(T):
        batch = np.stack([clips[c][t] if t < lengths[c] else np.zeros_like(clips[c][0]) for c in range(n)])
        fd = torch.from_numpy(batch).cuda()
        pipe.step(fd, st, active=[t < lengths[c] for c in range(n)])
        torch.cuda.current_stream().synchronize()
    pipe.finish(st)
    for c in range(n):
        ob, os_, oc, on = oracle_lib.run_batch(model_path, clips[c], threads=4)
        dets = [np.asarray([[ob[t, i, 1], ob[t, i, 0], ob[t, i, 3], ob[t, i, 2], os_[t, i], 0.0]
                            for i in range(on[t]) if os_[t, i] >= 0.3], np.float64).reshape(-1, 6) for t in range(lengths[c])]
        want = ocsort_np.track_boxes(dets, [(t + 1) / 30.0 for t in range(lengths[c])])
        got = pipe.rows(c)
        assert got["id"] == want["id"], c
        for k in ("time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width"):
            assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), (c, k)


def test_slot_reuse_equals_per_clip_runs(model_path, oracle_lib):
    """Five clips of different lengths through TWO detector slots (a slot takes the next clip of its queue when one ends,
    vbt_tracker_update_from_slots): every clip's rows equal the oracle run on that clip alone."""
    import torch
    from oracle import ocsort_np
    from vbt_amd import shard, synth
    from vbt_amd.track import Pipeline
    lengths = [7, 4, 6, 3, 5]
    fps = [30.0, 60.0, 30.0, 30.0, 60.0]
    clips = [synth.clip_frames(60 + c, 2 * c, lengths[c]) for c in range(len(lengths))]
    cmap, fidx = shard.slot_schedule(lengths, 2)
    assert (cmap >= 0).sum() == sum(lengths) and len(cmap) == 14          # LPT queues {7, 4, 3} and {6, 5}: makespan 14
    pipe = Pipeline(model_path, 2, max_frames=max(lengths), fps=fps, detection_treshold=0.3, rows_per_frame=25, tracker_clips=len(lengths))
    st = torch.cuda.current_stream().cuda_stream
    for t in range(len(cmap)):
        batch = np.stack([clips[cmap[t, s]][fidx[t, s] - 1] if cmap[t, s] >= 0 else np.zeros((320, 320, 3), np.uint8) for s in range(2)])
        fd = torch.from_numpy(batch).cuda()
        pipe.step(fd, st, clip_map=cmap[t], frame_idx=fidx[t])
        torch.cuda.current_stream().synchronize()
    pipe.finish(st)
    for c in range(len(lengths)):
        ob, os_, oc, on = oracle_lib.run_batch(model_path, clips[c], threads=4)
        dets = [np.asarray([[ob[t, i, 1], ob[t, i, 0], ob[t, i, 3], ob[t, i, 2], os_[t, i], 0.0]
                            for i in range(on[t]) if os_[t, i] >= 0.3], np.float64).reshape(-1, 6) for t in range(lengths[c])]
        want = ocsort_np.track_boxes(dets, [(t + 1) / fps[c] for t in range(lengths[c])])
        got = pipe.rows(c)
        assert got["id"] == want["id"], c
        for k in ("time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width"):
            assert np.array_equal(np.asarray(got[k]), np.asarray(want[k])), (c, k)


def test_host_frames_close_and_rows_all_equal_the_per_clip_accessors(model_path):
    """The SURVEY 8d path (frames in pinned host memory -> rows on the host): step() on a pinned host tensor, close() (one
    packed D2H) and rows_all() (one strided D2H) give what device-resident frames + finish() + rows(clip)/phases(clip) give;
    reset() starts the same clips over."""
    import torch
    from vbt_amd import synth
    from vbt_amd.ocsort import ROW_DTYPE
    from vbt_amd.track import Pipeline
    n, T = 3, 9
    frames = np.stack([np.stack([synth.render(synth.background(300 + c), 4 * c + t) for c in range(n)]) for t in range(T)])
    st = torch.cuda.current_stream().cuda_stream
    pipe = Pipeline(model_path, n, max_frames=T, fps=60.0, detection_treshold=0.3, rows_per_frame=25)
    fd = torch.from_numpy(frames).to("cuda:0")
    for t in range(T):
        pipe.step(fd[t], st)
    pipe.finish(st)
    want_rows = [pipe.rows(c) for c in range(n)]
    want_ph = [pipe.phases(c) for c in range(n)]
    assert sum(len(r["id"]) for r in want_rows) > 10
    for rep in range(2):                                                  # second round: after reset()
        pipe.reset()
        host = torch.from_numpy(frames).pin_memory()
        for t in range(T):
            pipe.step(host[t], st)
        best, rows_n, nph, ovf, ph = pipe.close(cap=8)
        counts, rows = pipe.rows_all()
        assert rows.dtype == ROW_DTYPE and rows.shape[0] == n
        for c in range(n):
            assert counts[c] == rows_n[c] == len(want_rows[c]["id"]) and ovf[c] == 0
            for k in ROW_DTYPE.names:
                assert np.array_equal(rows[c, :counts[c]][k], np.asarray(want_rows[c][k])), (rep, c, k)
            assert best[c] == want_ph[c][0] and nph[c] == len(want_ph[c][1])
            assert np.array_equal(ph[c, :nph[c]], want_ph[c][1])
    pinned = torch.empty(n * pipe.tracker.rows_cap * 64, dtype=torch.uint8).pin_memory()
    counts2, rows2 = pipe.rows_all(out=pinned)
    assert np.array_equal(counts2, counts) and all(np.array_equal(rows2[c, :counts[c]], rows[c, :counts[c]]) for c in range(n))


def test_more_than_64_clips_step_in_chunks_of_kernel_argument_metadata(model_path):
    """The tracker step carries frame times / clip ids in its kernel arguments, 64 slots per launch: 130 clips take three
    launches per step and must behave like 130 independent single-clip pipelines (checked on a sample of clips)."""
    import torch
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    n, T = 130, 5
    bg = [synth.background(500 + (c % 7)) for c in range(n)]
    frames = np.stack([np.stack([synth.render(bg[c], 3 * (c % 11) + t) for c in range(n)]) for t in range(T)])
    st = torch.cuda.current_stream().cuda_stream
    pipe = Pipeline(model_path, n, max_frames=T, fps=np.where(np.arange(n) % 2 == 0, 30.0, 60.0), detection_treshold=0.3, rows_per_frame=25)
    fd = torch.from_numpy(frames).to("cuda:0")
    for t in range(T):
        pipe.step(fd[t], st)
    best, rows_n, nph, ovf, ph = pipe.close(cap=8)
    counts, rows = pipe.rows_all()
    assert np.all(ovf == 0) and int(counts.sum()) > 100
    for c in (0, 63, 64, 65, 127, 128, 129):
        one = Pipeline(model_path, 1, max_frames=T, fps=30.0 if c % 2 == 0 else 60.0, detection_treshold=0.3, rows_per_frame=25)
        for t in range(T):
            one.step(fd[t, c:c + 1].contiguous(), st)
        one.close()
        c1, r1 = one.rows_all()
        assert c1[0] == counts[c] and np.array_equal(r1[0, :c1[0]], rows[c, :counts[c]]), c


@pytest.mark.parametrize("depth", [1, 3, 4])
def test_tracker_on_the_detector_streams_gives_the_same_rows(model_path, monkeypatch, depth):
    """VBT_TRACKER_STREAM=inline (the OC-SORT step at the end of the slot's own stream, ordered by events) against the
    tracker stream of the default configuration: identical DataFrame rows and phases at every depth."""
    import torch
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    n, T = 5, 20
    frames = np.stack([np.stack([synth.render(synth.background(70 + c), 7 * c + t) for c in range(n)]) for t in range(T)])
    fd = torch.from_numpy(frames).to("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for mode in ("own", "inline"):
        monkeypatch.setenv("VBT_TRACKER_STREAM", mode)
        pipe = Pipeline(model_path, n, max_frames=T, fps=60.0, depth=depth)
        assert pipe._trk_inline == (mode == "inline")
        for t in range(T):
            pipe.step(fd[t], st)
        best, n_rows, nph, ovf, ph = pipe.close(cap=16)
        counts, rows = pipe.rows_all()
        out[mode] = (best.copy(), n_rows.copy(), nph.copy(), ph.copy(), counts.copy(), [rows[c][:counts[c]].tobytes() for c in range(n)])
    a, b = out["own"], out["inline"]
    for i in range(5):
        assert np.array_equal(a[i], b[i])
    assert int(a[1].sum()) > 10
    assert a[5] == b[5]


@pytest.mark.parametrize("n,depth", [(1, 4), (3, 2), (8, 3)])
def test_deferred_tracker_steps_give_the_same_rows(model_path, monkeypatch, n, depth):
    """Small batches hand the detections of `depth` consecutive steps to ONE launch of the time-batched walk (VBT_TRACKER_DEFER,
    default on for <= 8 clips) instead of one tracker launch per step: identical rows, export ids and phases - with a group cut
    short by skip_frames() (another frame stride), by a step with a clip map, by a read of the rows in the middle, and with
    a partial last group."""
    import torch
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    T = 27
    frames = np.stack([np.stack([synth.render(synth.background(90 + c), 5 * c + t) for c in range(n)]) for t in range(T)])
    fd = torch.from_numpy(frames).to("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    out = {}
    for defer in ("0", "1"):
        monkeypatch.setenv("VBT_TRACKER_DEFER", defer)
        monkeypatch.setenv("VBT_TRACKER_STREAM", "inline")
        pipe = Pipeline(model_path, n, max_frames=4 * T, fps=60.0, depth=depth)
        assert pipe._defer == (depth if defer == "1" else 0) and pipe._ring == (2 * depth if defer == "1" else depth)
        mid = None
        for t in range(T):
            if t in (5, 6, 13):
                pipe.skip_frames(2)                                  # the frame stride changes inside / between groups
            if t == 17:
                # not a plain step (clip map + explicit frame numbers): flushes the group, runs per step
                pipe.step(fd[t], st, clip_map=np.arange(n, dtype=np.int32), frame_idx=np.full(n, pipe.frame_count + 1))
            else:
                pipe.step(fd[t], st)
            if t == 10:
                mid = pipe.rows(0)                                   # a read in the middle drains the deferred steps first
        best, n_rows, nph, ovf, ph = pipe.close(cap=16)
        counts, rows = pipe.rows_all()
        out[defer] = (best.copy(), n_rows.copy(), nph.copy(), ph.copy(), counts.copy(), [rows[c][:counts[c]].tobytes() for c in range(n)],
                      {k: list(v) for k, v in mid.items()})
    a, b = out["0"], out["1"]
    for i in range(5):
        assert np.array_equal(a[i], b[i])
    assert int(a[1].sum()) > 10 * n
    assert a[5] == b[5] and a[6] == b[6]


def test_busy_streams_sit_on_distinct_hardware_queues(model_path):
    """Pipeline._place_streams: after creation every pair of the streams that carry kernels side by side (detector slots +
    the copy stream) runs two spinning waves concurrently; a stream against itself reads as shared (the probe's control)."""
    import ctypes
    from vbt_amd import _lib
    from vbt_amd.track import Pipeline
    pipe = Pipeline(model_path, 2, max_frames=4, fps=60.0, depth=3)   # three forwards + the copy stream = the four hardware queues
    L = _lib.lib()

    def shared(a, b):
        sh = ctypes.c_int()
        _lib.check(L.vbt_streams_share_queue(a.cuda_stream, b.cuda_stream, 150, ctypes.byref(sh)))
        return bool(sh.value)

    busy = list(pipe._det_streams) + [pipe._copy_stream]
    assert shared(busy[0], busy[0])
    for i in range(len(busy)):
        for j in range(i + 1, len(busy)):
            assert not shared(busy[i], busy[j]), (i, j)


def test_sixty_four_clips_without_torch(model_path, tmp_path):
    """The pipeline behind the C ABI needs no framework: a child process that never imports torch drives 64 clips through
    vbt_pipeline_create / _step / _close / _rows_all - frames resident in device memory from vbt_device_alloc, and once more from
    pinned host memory (vbt_host_alloc) - and its rows, export ids and phases equal those of the same clips driven from this
    (torch-using) process."""
    import pickle
    import subprocess
    import sys
    import torch
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    n, T = 64, 6
    frames = np.stack([np.stack([synth.render(synth.background(300 + c), 3 * c + t) for c in range(n)]) for t in range(T)])
    src = tmp_path / "frames.npy"
    np.save(src, frames)
    code = (
        "import sys, pickle, numpy as np\n"
        "from vbt_amd.track import Pipeline\n"
        "from vbt_amd import mem\n"
        "frames = np.load(sys.argv[1])\n"
        "T, n = frames.shape[:2]\n"
        "out = {}\n"
        "pipe = Pipeline(sys.argv[2], n, max_frames=T, fps=60.0, rows_per_frame=25)\n"
        "dev = [mem.DeviceBuffer.from_host(frames[t]) for t in range(T)]\n"
        "pinned = mem.pinned_empty(frames.shape)\n"
        "pinned[...] = frames\n"
        "for name, srcs in (('device', [d.ptr for d in dev]), ('pinned', [pinned[t] for t in range(T)])):\n"
        "    pipe.reset()\n"
        "    for t in range(T):\n"
        "        pipe.step(srcs[t])\n"
        "    best, rows_n, nph, ovf, ph = pipe.close(cap=16)\n"
        "    counts, rows = pipe.rows_all()\n"
        "    out[name] = (best, rows_n, nph, ovf, ph, counts, [rows[c, :counts[c]].tobytes() for c in range(n)])\n"
        "assert 'torch' not in sys.modules, 'torch was imported'\n"
        "pickle.dump(out, open(sys.argv[3], 'wb'))\n")
    dst = tmp_path / "out.pkl"
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    subprocess.run([sys.executable, "-c", code, str(src), model_path, str(dst)], check=True, cwd=root, timeout=300)
    got = pickle.load(open(dst, "rb"))
    pipe = Pipeline(model_path, n, max_frames=T, fps=60.0, rows_per_frame=25)
    fd = torch.from_numpy(frames).to("cuda:0")
    for t in range(T):
        pipe.step(fd[t])
    best, rows_n, nph, ovf, ph = pipe.close(cap=16)
    counts, rows = pipe.rows_all()
    assert int(counts.sum()) > 100 and np.all(ovf == 0)
    for name in ("device", "pinned"):
        g = got[name]
        for a, b in zip(g[:6], (best, rows_n, nph, ovf, ph, counts)):
            assert np.array_equal(a, b), name
        assert g[6] == [rows[c, :counts[c]].tobytes() for c in range(n)], name


@pytest.mark.parametrize("H,W", [(1280, 96), (960, 333), (700, 64), (1000, 41), (641, 80)])
def test_host_frames_at_source_resolution_upload_only_the_rows_the_resize_reads(oracle_lib, model_path, H, W):
    """N2 from host memory (the reference's situation: cap.read() hands over 1080 x 1920 frames, track.py:160): when the source holds more
    than twice the network's rows, only the row pairs the bilinear resize of odt.py:15-16 reads are uploaded (vbt_amd/csrc/pipeline.hip:
    one strided copy for integer scales - 1280 = 4 x 320, 960 = 3 x 320 where the two rows of a pair coincide -, a row table otherwise) and the
    resize kernel indexes the compact buffer.  Detections == oracle preprocess + oracle detector, through step() and through step_runs() with
    one host source per run; the uploaded byte count is the compact one."""
    from oracle.preprocess import preprocess_image as ref_pre
    from vbt_amd import mem
    from vbt_amd.track import Pipeline
    rng = np.random.default_rng(H + W)
    n, T = 2, 2
    src = rng.integers(0, 256, (T, n, H, W, 3), dtype=np.uint8)
    src[..., : W // 2, :] = (src[..., : W // 2, :] // 64 * 64)             # flat regions next to noise: interpolated values on and off integers
    pinned = mem.pinned_empty(src.shape)
    pinned[...] = src
    want_frames = np.stack([np.stack([ref_pre(src[t, c], (320, 320), swap_rb=True)[0] for c in range(n)]) for t in range(T)])
    ob, os_, oc, on = oracle_lib.run_batch(model_path, want_frames.reshape(-1, 320, 320, 3), threads=8)
    ob, os_, on = ob.reshape(T, n, 25, 4), os_.reshape(T, n, 25), on.reshape(T, n)
    pipe = Pipeline(model_path, n, max_frames=T, fps=30.0, tracker_clips=n)
    up0 = pipe.info().h2d_bytes
    for t in range(T):
        pipe.step(pinned[t], src_hw=(H, W), swap_rb=True)
        b, s, c, k = pipe.detections()
        assert np.array_equal(k, on[t]) and np.array_equal(s, os_[t]) and np.array_equal(b, ob[t]), ("step", t)
    assert pipe.info().h2d_bytes - up0 == T * n * 2 * 320 * W * 3          # row pairs, not T * n * H * W * 3
    pipe.reset()
    for t in range(T):                                                     # pageable numpy sources, one per run
        pipe.step_runs([src[t, 0:1], src[t, 1:2]], [(0, 0, 1, t + 1), (1, 1, 1, t + 1)], src_hw=(H, W), swap_rb=True)
        b, s, c, k = pipe.detections()
        assert np.array_equal(k, on[t]) and np.array_equal(s, os_[t]) and np.array_equal(b, ob[t]), ("step_runs", t)


def test_pipeline_abi_refuses_bad_calls_and_stays_usable(model_path):
    """Error behaviour of the vbt_pipeline_* entry points through ctypes: bad arguments come back as VBT_ERR_ARG (-1) with a message, a
    too-small phase buffer as VBT_ERR_CAPACITY (-4), and the pipeline keeps working after every refusal."""
    import ctypes
    from vbt_amd import _lib, mem, synth
    from vbt_amd.track import Pipeline
    L = _lib.lib()
    prm = _lib.PipelineParams()
    L.vbt_pipeline_default_params(ctypes.byref(prm))
    h = ctypes.c_void_p()
    fps = np.array([30.0, 30.0])
    prm.n_slots, prm.n_clips, prm.rows_cap = 0, 2, 100
    assert L.vbt_pipeline_create(model_path.encode(), ctypes.byref(prm), fps.ctypes.data, ctypes.byref(h)) == -1 and not h.value
    prm.n_slots = 2
    bad_fps = np.array([30.0, 0.0])
    assert L.vbt_pipeline_create(model_path.encode(), ctypes.byref(prm), bad_fps.ctypes.data, ctypes.byref(h)) == -1
    assert L.vbt_pipeline_create(b"/nonexistent.vbtm", ctypes.byref(prm), fps.ctypes.data, ctypes.byref(h)) == -2
    pipe = Pipeline(model_path, 2, max_frames=8, fps=30.0, rows_per_frame=25)
    frames = np.stack([synth.render(synth.background(40 + c), 3 * c) for c in range(2)])
    dev = mem.DeviceBuffer.from_host(frames)
    p = pipe._h
    assert L.vbt_pipeline_step(p, None, 1, 0, 0, 0, None, None, None, 1, None) == -1
    assert L.vbt_pipeline_step(p, dev.ptr, 1, 320, 0, 0, None, None, None, 1, None) == -1                  # src_h without src_w
    cm = np.array([0, 5], np.int32)
    fi = np.array([1, 1], np.int32)
    assert L.vbt_pipeline_step(p, dev.ptr, 1, 0, 0, 0, None, cm.ctypes.data, fi.ctypes.data, 1, None) == -1   # clip 5 of 2
    assert L.vbt_pipeline_step(p, dev.ptr, 1, 0, 0, 0, None, cm.ctypes.data, None, 1, None) == -1             # clip_map without frame_idx
    runs = (_lib.Run * 2)(_lib.Run(0, 0, 1, 2, 1, 1, 30.0), _lib.Run(1, 1, 1, 2, 1, 1, 30.0))               # second run leaves the batch
    assert L.vbt_pipeline_step_runs(p, dev.ptr, None, 1, runs, 2, 0, 0, 0, 1, None, None, None, None, None) == -1
    assert b"outside" in L.vbt_last_error()
    assert L.vbt_pipeline_step_runs(p, None, None, 1, runs, 1, 0, 0, 0, 1, None, None, None, None, None) == -1  # neither frames nor sources
    assert int(pipe.info().steps_enqueued) == 0                          # nothing was enqueued by the refused calls
    for t in range(3):
        pipe.step(dev.ptr)
    best, rows, nph, ovf = (np.zeros(2, np.int32) for _ in range(4))
    ph = np.zeros((2, 1, 6))
    rc = L.vbt_pipeline_close(p, best.ctypes.data, rows.ctypes.data, nph.ctypes.data, ovf.ctypes.data, ph.ctypes.data, 0)
    assert rc == -1                                                      # cap 0 is not a buffer
    best2, rows2, nph2, ovf2, ph2 = pipe.close(cap=16)                   # ... and the pipeline is still usable
    assert int(rows2.sum()) >= 0 and np.all(ovf2 == 0)
    with pytest.raises(_lib.VbtError):
        pipe.tracker_only_steps(1, slot=99)


def test_device_frames_are_read_after_the_callers_stream_has_produced_them(model_path, oracle_lib):
    """vbt_pipeline_step with device frames: the forward waits for the point of `caller_stream` at which the call is made.  The frame
    buffer is filled on a side stream BEHIND a long spin, then handed over with that stream: the detections are those of the frames
    that arrive late, not of the zeros the buffer held when step() was called."""
    import torch
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    n = 2
    frames = np.stack([synth.render(synth.background(60 + c), 5 * c) for c in range(n)])
    want = oracle_lib.run_batch(model_path, frames, threads=2)
    src = torch.from_numpy(frames).cuda()
    buf = torch.zeros_like(src)
    pipe = Pipeline(model_path, n, max_frames=4, fps=30.0)
    side = torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        torch.cuda._sleep(200_000_000)                                   # ~0.1 s of spinning in front of the copy
        buf.copy_(src, non_blocking=True)
    pipe.step(buf, stream=side.cuda_stream)
    b, s, c, k = pipe.detections()
    assert np.array_equal(k, want[3]) and np.array_equal(s, want[1]) and np.array_equal(b, want[0])
    assert int(want[3].sum()) > 0


def test_many_pipelines_in_one_process_share_the_classified_streams(model_path):
    """The library's stream pool is per process and per device: pipelines created one after the other take their busy streams from the
    hardware-queue groups classified once (vbt_amd/csrc/pipeline.hip), so the tenth pipeline is placed as well as the first and no more
    than GPU_MAX_HW_QUEUES groups are ever seen."""
    from vbt_amd.track import Pipeline
    seen = []
    for i in range(10):
        pipe = Pipeline(model_path, 2, max_frames=4, fps=30.0, depth=3)
        inf = pipe.info()
        assert inf.placement_ok == 1, i
        seen.append(int(inf.queue_groups_seen))
        busy = {pipe._det_streams[k].cuda_stream for k in range(3)} | {pipe._copy_stream.cuda_stream}
        assert len(busy) == 4
        del pipe
    assert max(seen) <= 4 and seen[-1] == seen[2]
