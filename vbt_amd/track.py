"""`track()` + export: the reference's per-clip hot loop (reference track.py:129-260, 103-126),
and `Pipeline`, its batched MI355X form (n clips frame-wise through detect -> NMS -> tracker on
one stream, rep analysis at the end).

No GUI: drawing / imshow / VideoWriter (reference track.py:28-62,201-207,237-247) are out of
scope.  Frame sources are arrays / iterables of RGB uint8 frames instead of
cv2.VideoCapture (cv2 is not a dependency here).
"""
import collections
import ctypes
import os
import sys

import numpy as np

from . import _lib
from .interpreter import Interpreter
from .ocsort import ROW_DTYPE, MultiClipTracker, OCSort
from .odt import (calc_bounding_box_center, calc_plate_height, calc_plate_width, results_to_sorttracker_inputs,
                  run_odt)

MAX_AGE = 30                                   # reference track.py:22
COLUMNS = ("id", "time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width")


def track(src, interpreter, detection_treshold=0.5, display_image_height=720, video_path=None, fps=30.0, frame_stride=1):
    """Per-frame loop of reference track.py:159-234 with the reference's object call shapes.
    src: iterable of RGB uint8 frames.  frame_stride=16 reproduces `frame_count % 16` of the
    snapshot (track.py:166); the committed DataFrames were made with stride 1."""
    data = {k: [] for k in COLUMNS}
    tracker = OCSort(max_age=MAX_AGE, asso_func="diou", iou_threshold=0.1)
    frame_count = 0
    for frame in src:
        frame_count += 1
        if frame_stride > 1 and frame_count % frame_stride:
            continue
        time = frame_count / fps
        results = run_odt(frame=frame, interpreter=interpreter, threshold=detection_treshold)
        if results == []:
            continue
        tracker_out = tracker.update(results_to_sorttracker_inputs(results), [], time=time)
        trackers = tracker.trackers
        for res in tracker_out:
            xmin, ymin, xmax, ymax, tracking_id, _, score = res
            bounding_box = [ymin, xmin, ymax, xmax]
            tracking_id = int(tracking_id)
            kf = None
            for trk in trackers:
                if trk.id == tracking_id - 1:
                    kf = trk.kf
                    break
            dx, dy = kf.x.flatten()[4:6]
            x_center, y_center = calc_bounding_box_center(bounding_box)
            data["id"].append(tracking_id)
            data["time"].append(time)
            data["x"].append(x_center)
            data["y"].append(y_center)
            data["dx"].append(dx)
            data["dy"].append(dy)
            data["norm_plate_height"].append(calc_plate_height(bounding_box))
            data["norm_plate_width"].append(calc_plate_width(bounding_box))
    return data


def track_frames(frames, model_path, fps=30.0, detection_treshold=0.5, frame_stride=1, time_batch=64, device=0):
    """The whole clip loop of reference track.py:129-260 on the time-batched device path: `time_batch` consecutive (kept)
    frames of the clip per detector batch, OC-SORT walking each batch in frame order on the device, nothing but the finished
    rows coming back.  frames: uint8 [T,H,W,3] RGB (numpy array or memmap; any resolution - resized on the GPU like
    odt.py:10-19).  frame_stride = the `frame_count % 16` of track.py:166: frames whose 1-based number is not a multiple are
    read and dropped, they only advance the clip time.  Returns the reference's dict of lists (track.py:144-145)."""
    T, H, W = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    stride = max(int(frame_stride), 1)
    kept = T // stride                                               # frames that are processed
    F = max(1, min(int(time_batch), max(kept, 1)))
    pipe = Pipeline(model_path, F, max_frames=max(kept, 1), fps=fps, detection_treshold=detection_treshold, device=device,
                    rows_per_frame=25, tracker_clips=1)
    size = pipe._size
    src_hw = None if (H, W) == (size, size) else (H, W)
    if isinstance(frames, np.ndarray) and frames.dtype == np.uint8 and frames.flags.c_contiguous:
        return pipe.track_clip(frames, frame_stride=stride, src_hw=src_hw)      # vbt_track_clip: the whole loop inside the library
    # any other sequence: chunk by chunk through contiguous host copies
    idx_all = np.arange(stride - 1, T, stride)
    for i0 in range(0, len(idx_all), F):
        idx = idx_all[i0:i0 + F]
        chunk = np.ascontiguousarray(frames[idx[0]:idx[-1] + 1:stride] if stride > 1 else frames[idx[0]:idx[-1] + 1], dtype=np.uint8)
        pipe.step_runs(chunk, [(0, 0, len(idx), int(idx[0]) + 1, stride)], src_hw=src_hw)
    pipe.finish()
    return pipe.rows(0)


def export_dataframe(data, src_name, model_path, df_dir=None, write=True):
    """reference track.py:103-126: sort by (id,time) keeping the original row labels, pick the id with
    the largest cumulative path length, name the file f'{video}_id{id}_{model}.pkl.gz'."""
    import pandas as pd
    df = pd.DataFrame.from_dict(data)
    df = df.sort_values(by=["id", "time"])
    df2 = df.copy()
    df2["distance"] = np.where(df2["id"] == df2["id"].shift(),
                               ((df2["x"] - df2["x"].shift()) ** 2 + (df2["y"] - df2["y"].shift()) ** 2) ** 0.5, np.nan)
    df2["cumulative_distance"] = df2.groupby("id")["distance"].cumsum()
    max_distance_id = df2.loc[df2["cumulative_distance"].idxmax(), "id"]
    model_name = os.path.basename(model_path).split(".")[0]
    df_filename = f'{os.path.basename(src_name).split(".")[0]}_id{max_distance_id}_{model_name}.pkl.gz'
    df_path = df_filename if df_dir is None else os.path.join(df_dir, df_filename)
    if write:
        if df_dir is not None:
            os.makedirs(df_dir, exist_ok=True)
        df.to_pickle(df_path)
    return df, int(max_distance_id), df_path


class StreamPlacementError(RuntimeError):
    pass


def _torch():
    """torch, if the caller's process already uses it (never imported from here: the pipeline itself needs no framework)."""
    return sys.modules.get("torch")


def _host_or_device_ptr(x):
    """(pointer, on_device, keepalive) of a frame source: a torch tensor (device or host / pinned), a numpy array, an object with
    __cuda_array_interface__, or a raw DEVICE pointer (int) the caller keeps alive."""
    if isinstance(x, int):
        return x, True, None
    if hasattr(x, "data_ptr"):                                      # torch tensor
        if x.dtype != _torch().uint8 or not x.is_contiguous():
            raise ValueError("frames must be a contiguous uint8 tensor")
        return x.data_ptr(), x.device.type != "cpu", x
    if hasattr(x, "__cuda_array_interface__"):
        return int(x.__cuda_array_interface__["data"][0]), True, x
    a = np.asarray(x)
    if a.dtype != np.uint8 or not a.flags.c_contiguous:
        raise ValueError("frames must be a C-contiguous uint8 array")
    return a.ctypes.data, False, a


class Pipeline:
    """n clips processed frame-wise - the batched form of the clip loop of reference track.py:129-260 - as a thin ctypes wrapper of
    `vbt_pipeline` (include/vbt_hip.h): streams and their placement on hardware queues, the ring of detector outputs, the staging
    ring of the host-fed mode, the deferred tracker groups and the clip close all live in libvbt_hip.so (vbt_amd/csrc/pipeline.hip).
    step(frames) enqueues detect + NMS + one OC-SORT step per clip; close() drains, selects each clip's export id and runs the rep
    analysis on the device.  Nothing leaves the GPU until rows() / phases() / close() are read.

    Frame sources: numpy arrays (host memory; `vbt_amd.mem.pinned_empty` gives DMA-able ones), raw device pointers, and - a
    convenience for callers who hold them - torch tensors (device or pinned host).  torch is never imported here."""

    def __init__(self, model_path, n_clips, max_frames, fps=60.0, detection_treshold=0.5, device=0, rows_per_frame=4,
                 plate_diameter=0.45, depth=None, tracker_clips=None):
        L = _lib.lib()
        self.n = int(n_clips)                       # slots of the detector batch
        # tracker_clips > n_clips: more clips than batch slots; step(clip_map=...) says which clip sits in which slot
        self.n_trk = int(tracker_clips) if tracker_clips is not None else self.n
        self.fps = np.broadcast_to(np.asarray(fps, np.float64), (self.n_trk,)).copy()
        self.thr = float(detection_treshold)
        self.plate_diameter = plate_diameter
        self._dev = int(device)
        prm = _lib.PipelineParams()
        L.vbt_pipeline_default_params(ctypes.byref(prm))
        prm.n_slots, prm.n_clips, prm.device = self.n, self.n_trk, self._dev
        prm.rows_cap = int(max_frames) * rows_per_frame + 3 * 25          # frames 1-3 may emit 25 rows each
        prm.depth = int(depth) if depth is not None else 0                 # 0: VBT_PIPELINE_DEPTH, else 4 for <= 8 slots, else 3
        prm.detection_threshold = self.thr
        prm.plate_diameter = float(plate_diameter)
        prm.model_flags = int(os.environ.get("VBT_FUSION_FLAGS", "0"))
        # --model may name a TFLite flatbuffer like the reference's (track.py:67): converted to the container format on the fly
        from .tflite_import import as_container_path
        path, temporary = as_container_path(str(model_path))
        h = ctypes.c_void_p()
        try:
            rc = L.vbt_pipeline_create(path.encode(), ctypes.byref(prm), self.fps.ctypes.data, ctypes.byref(h))
        finally:
            if temporary:
                os.unlink(path)
        if rc == -5 and "hardware queue" in L.vbt_last_error().decode():
            raise StreamPlacementError(L.vbt_last_error().decode())
        _lib.check(rc)
        self._owner = _PipelineHandle(h)            # shared with the borrowed Interpreter / tracker views: the library object lives as long as any of them
        self._h = h
        info = self.info()
        self.depth, self._ring, self._defer, self._trk_inline = info.depth, info.ring, info.defer, bool(info.tracker_inline)
        self._size = info.image_size
        self._det_streams = [_StreamHandle(info.det_streams[k]) for k in range(self.depth)]
        self._copy_stream = _StreamHandle(info.copy_stream)
        self._trk_stream = _StreamHandle(info.tracker_stream)
        self.interpreters = [Interpreter._borrowed(L.vbt_pipeline_model(self._h, k), model_path, device, self.n, self._owner) for k in range(self.depth)]
        self.interpreter = self.interpreters[0]
        self.tracker = MultiClipTracker._borrowed(L.vbt_pipeline_tracker(self._h), self.n_trk, prm.rows_cap, self._owner)
        self._step_idx = 0                          # steps enqueued so far (mirror of the library's counter: which stream a step runs on)
        self._keep = collections.deque(maxlen=2 * self._ring + 4)   # host sources / foreign device arrays of the steps in flight
        self._ext = {}                              # torch.cuda.ExternalStream views of the detector streams (record_stream)

    def info(self):
        out = _lib.PipelineInfo()
        _lib.check(_lib.lib().vbt_pipeline_get_info(self._h, ctypes.byref(out)))
        return out

    @property
    def frame_count(self):
        """time counter of the clips: time = frame_count / fps (track.py:161,169)"""
        return int(self.info().frame_count)

    @frame_count.setter
    def frame_count(self, v):
        _lib.check(_lib.lib().vbt_pipeline_set_frame_count(self._h, int(v)))

    # ---- frame sources ----
    def _caller_stream(self, stream):
        if stream is not None:
            return int(stream)
        t = _torch()
        if t is not None and t.cuda.is_available() and t.cuda.is_initialized():
            return t.cuda.current_stream().cuda_stream
        return 0

    def _source(self, x, k, n_frames=None, hw=None):
        """pointer / residency of one source; a torch device tensor is told to the caching allocator (its storage is used on the slot's
        stream up to `depth` steps after the call returns), anything else that owns memory is kept referenced while steps are in flight"""
        ptr, on_dev, keep = _host_or_device_ptr(x)
        if keep is not None:
            shp = tuple(keep.shape)
            if len(shp) != 4 or shp[3] != 3 or (hw is not None and shp[1:3] != tuple(hw)) or (n_frames is not None and shp[0] < n_frames):
                raise ValueError(f"frames must be uint8 [{n_frames if n_frames is not None else 'B'}, {hw[0] if hw else 'H'}, {hw[1] if hw else 'W'}, 3], got {shp}")
            if on_dev and hasattr(keep, "record_stream"):
                t = _torch()
                if k not in self._ext:
                    self._ext[k] = t.cuda.ExternalStream(self._det_streams[k].cuda_stream, device=keep.device)
                keep.record_stream(self._ext[k])
            else:
                self._keep.append(keep)
        return ptr, on_dev

    def _enqueued(self, rc):
        """after a step call: the mirror of the library's step counter (it selects the stream a torch tensor is recorded on)"""
        if rc == 0:
            self._step_idx += 1
            return
        self._step_idx = int(self.info().steps_enqueued)
        _lib.check(rc)

    def _hw(self, src_hw):
        return (int(src_hw[0]), int(src_hw[1])) if src_hw is not None else (self._size, self._size)

    def step(self, frames_dev_ptr, stream=None, src_hw=None, swap_rb=False, active=None, clip_map=None, frame_idx=None, track=True):
        """frames: uint8 [n,H,W,3], frame `frame_count+1` of every clip - a device tensor / raw device pointer valid on the caller's
        stream (`stream`, default: torch's current stream if torch is in use, else the null stream) and left unmodified until the step
        has run (up to `depth` steps later), or host memory (numpy array, pinned tensor), which must be final when the call is made
        and stay untouched until the step has run (vbt_pipeline_step, include/vbt_hip.h).  src_hw=(H, W) of source-resolution frames:
        the bilinear resize + truncating cast of reference odt.py:10-19 (and, with swap_rb, the BGR->RGB of track.py:171) then run on
        the device ahead of the detector.
        active: optional bool [n] - clips that still have a frame in this step (clips of different lengths batched together; the
        reference processes them one after the other, track.py:85-126).
        clip_map / frame_idx: int [n] - slot i carries frame number frame_idx[i] (1-based) of tracker clip clip_map[i] (-1: empty)."""
        k = (self._step_idx % self._ring) % self.depth
        H, W = self._hw(src_hw)
        ptr, on_dev = self._source(frames_dev_ptr, k, self.n, (H, W))
        act = cm = fi = None
        if active is not None:
            act = np.ascontiguousarray(np.asarray(active, bool).astype(np.uint8))
            if act.shape != (self.n,):
                raise ValueError("active must have one entry per slot")
        if clip_map is not None:
            cm = np.ascontiguousarray(clip_map, dtype=np.int32)
            fi = np.ascontiguousarray(frame_idx, dtype=np.int32)
            if cm.shape != (self.n,) or fi.shape != (self.n,):
                raise ValueError("clip_map / frame_idx must have one entry per slot")
        self._enqueued(_lib.lib().vbt_pipeline_step(self._h, ptr, int(on_dev), H if src_hw is not None else 0, W if src_hw is not None else 0, int(bool(swap_rb)),
                                                    act.ctypes.data if act is not None else None, cm.ctypes.data if cm is not None else None,
                                                    fi.ctypes.data if fi is not None else None, int(bool(track)), self._caller_stream(stream) if on_dev else None))

    def step_runs(self, frames, runs, stream=None, src_hw=None, swap_rb=False, track=True, outputs=None):
        """Time-batched step (the reference's unit of work is ONE video, track.py:85-126,159-247): the detector batch holds RUNS of
        consecutive frames of a clip; the OC-SORT steps of a run are walked in frame order by one wavefront inside ONE tracker launch.
        runs: sequence of (clip, slot0, n_frames, frame0[, frame_step]) - frame f of the run sits in batch slot slot0 + f and is frame
        number frame0 + f * frame_step (1-based) of tracker clip `clip`; its time stamp is frame number / fps[clip].
        frames: the assembled batch [B,H,W,3] (device or host), or a list with one source [n_frames,H,W,3] per run (all device or all
        host, contiguous): the batch is then assembled in the library (one gather launch / one H2D copy per run on the copy stream).
        outputs (with track=False): (boxes [B,25,4] f32, scores [B,25] f32, classes [B,25] f32, counts [B] i32) device arrays that
        receive this step's detections instead of the pipeline's ring - the frame-major multi-GPU mode."""
        L = _lib.lib()
        if outputs is not None and track:
            raise ValueError("step_runs: outputs= is for detector-only steps (track=False)")
        k = (self._step_idx % self._ring) % self.depth
        H, W = self._hw(src_hw)
        ra = (_lib.Run * len(runs))()
        B = 0
        for i, r in enumerate(runs):
            clip, slot0, nf, frame0 = (int(v) for v in r[:4])
            fstep = int(r[4]) if len(r) > 4 else 1
            if not (0 <= clip < self.n_trk) or nf < 1 or slot0 < 0 or slot0 + nf > self.n:
                raise ValueError(f"run {i}: clip {clip}, slots {slot0}..{slot0 + nf - 1} outside {self.n_trk} clips / {self.n} slots")
            ra[i] = _lib.Run(clip, slot0, 1, nf, frame0, fstep, float(self.fps[clip]))
            B = max(B, slot0 + nf)
        fptr = srcs = None
        if isinstance(frames, (list, tuple)):
            if len(frames) != len(runs):
                raise ValueError("one source tensor per run")
            used = np.zeros(B, bool)
            for r in ra:
                used[r.slot0:r.slot0 + r.n_frames] = True
            if not used.all():
                raise ValueError("step_runs: the runs leave a hole in the detector batch")
            srcs = (ctypes.c_void_p * len(runs))()
            on = set()
            for i, (src, r) in enumerate(zip(frames, ra)):
                # raw pointers base + f * frame_bytes go to the gather kernel / the copies: a short, strided or differently shaped
                # source would make them read past its allocation
                try:
                    srcs[i], d = self._source(src, k, r.n_frames, (H, W))
                except ValueError as e:
                    raise ValueError(f"step_runs: source {i}: {e}") from None
                on.add((d, str(getattr(src, "device", "host"))))
            if len(on) != 1:
                raise ValueError(f"step_runs: the sources live in different memories: {sorted(on)}")
            on_dev = on.pop()[0]
        else:
            fptr, on_dev = self._source(frames, k, B, (H, W))
        outs = [None] * 4
        if outputs is not None:
            for j, (t_, shp_, sz) in enumerate(zip(outputs, ((B, 25, 4), (B, 25), (B, 25), (B,)), (4, 4, 4, 4))):
                if tuple(t_.shape) != shp_ or t_.element_size() != sz or not t_.is_contiguous() or t_.device.type != "cuda":
                    raise ValueError(f"step_runs: outputs must be contiguous 4-byte device tensors {shp_}")
                if k not in self._ext:
                    self._ext[k] = _torch().cuda.ExternalStream(self._det_streams[k].cuda_stream, device=t_.device)
                t_.record_stream(self._ext[k])
                outs[j] = t_.data_ptr()
        self._enqueued(L.vbt_pipeline_step_runs(self._h, fptr, srcs, int(on_dev), ra, len(runs), H if src_hw is not None else 0, W if src_hw is not None else 0,
                                                int(bool(swap_rb)), int(bool(track)), outs[0], outs[1], outs[2], outs[3], self._caller_stream(stream) if on_dev else None))

    def join_detectors(self, stream=None):
        """The caller's stream waits for every forward enqueued so far (after detector-only steps their outputs are then safe to read on it)."""
        _lib.check(_lib.lib().vbt_pipeline_join_detectors(self._h, self._caller_stream(stream)))

    def step_seq(self, frames, frame0=None, stream=None, **kw):
        """F consecutive frames of EVERY clip in one step: frames [n_clips, F, H, W, 3] (clip-major, device or host); the clips' frame
        counters advance by F."""
        ncl, F = int(frames.shape[0]), int(frames.shape[1])
        if ncl != self.n_trk or ncl * F > self.n:
            raise ValueError(f"step_seq: {ncl} clips x {F} frames do not fit {self.n_trk} clips / {self.n} slots")
        f0 = self.frame_count + 1 if frame0 is None else int(frame0)
        self.step_runs(frames.reshape((ncl * F,) + tuple(frames.shape[2:])), [(c, c * F, F, f0) for c in range(ncl)], stream, **kw)
        self.frame_count = f0 + F - 1

    def track_clip(self, frames, frame_stride=1, src_hw=None, swap_rb=False):
        """vbt_track_clip: the whole loop of reference track.py:129-260 for ONE clip held in memory (frames uint8 [T,H,W,3], host or
        device), `n` consecutive kept frames per detector batch.  Returns the reference's dict of lists (track.py:144-145)."""
        ptr, on_dev, keep = _host_or_device_ptr(frames)
        T = int(frames.shape[0]) if keep is not None else None
        if T is None:
            raise ValueError("track_clip needs an array (its length is the clip's frame count)")
        H, W = self._hw(src_hw)
        if tuple(frames.shape[1:]) != (H, W, 3):
            raise ValueError(f"track_clip: frames must be [T, {H}, {W}, 3], got {tuple(frames.shape)}")
        cap = self.tracker.rows_cap
        ids = np.empty(cap, np.int64)
        cols = np.empty((cap, 7), np.float64)
        n = ctypes.c_int()
        _lib.check(_lib.lib().vbt_track_clip(self._h, ptr, int(on_dev), T, H if src_hw is not None else 0, W if src_hw is not None else 0, int(bool(swap_rb)),
                                             int(frame_stride), ids.ctypes.data, cols.ctypes.data, cap, ctypes.byref(n)))
        self._step_idx = int(self.info().steps_enqueued)
        d = {"id": ids[:n.value].tolist()}
        for j, nm in enumerate(COLUMNS[1:]):
            d[nm] = cols[:n.value, j].tolist()
        return d

    def reset(self):
        """Back to frame 0 of fresh clips (tracker state cleared); models, streams and buffers are kept."""
        _lib.check(_lib.lib().vbt_pipeline_reset(self._h))
        self._step_idx = 0
        self._keep.clear()

    def tracker_only_steps(self, count, slot=0):
        """Measurement split: `count` tracker steps of all clips on the detections sitting in ring slot `slot`."""
        _lib.check(_lib.lib().vbt_pipeline_tracker_only_steps(self._h, int(count), int(slot)))

    def skip_frames(self, n=1):
        """Frames read from the source but not processed (`frame_count % 16` of reference track.py:161-167): they advance the clip
        time and nothing else - no ring slot is used."""
        _lib.check(_lib.lib().vbt_pipeline_skip_frames(self._h, int(n)))

    def _drain(self):
        _lib.check(_lib.lib().vbt_pipeline_drain(self._h))

    def finish(self, stream=None):
        _lib.check(_lib.lib().vbt_pipeline_finish(self._h))

    def close(self, cap=32):
        """Clip close in one go: drain the pipeline, export-id selection + rep analysis on the device, then ONE packed device-to-host
        copy and ONE stream synchronisation.  Returns (best_ids[n], n_rows[n], n_phases[n], overflow[n], phases[n, cap, 6]) - per clip
        the id of reference track.py:107-115 and the Phase list of plot.py:33-47."""
        n = self.n_trk
        best, rows, nph, ovf = (np.zeros(n, np.int32) for _ in range(4))
        ph = np.zeros((n, cap, 6), np.float64)
        _lib.check(_lib.lib().vbt_pipeline_close(self._h, best.ctypes.data, rows.ctypes.data, nph.ctypes.data, ovf.ctypes.data, ph.ctypes.data, int(cap)))
        return best, rows, nph, ovf, ph

    def rows_all(self, cap=None, out=None):
        """DataFrame rows of every clip, one strided copy (after close() / finish()): (counts[n], rows[n, cap] of ROW_DTYPE).  `out`:
        optional preallocated (pinned) buffer of at least n * cap * 64 bytes (torch tensor or numpy array)."""
        n = self.n_trk
        cap = int(cap or self.tracker.rows_cap)
        counts = np.zeros(n, np.int32)
        if out is None:
            rows = np.zeros((n, cap), ROW_DTYPE)
            ptr = rows.ctypes.data
        elif hasattr(out, "data_ptr"):
            rows = out.numpy().view(np.uint8).reshape(-1)[:n * cap * 64].view(ROW_DTYPE).reshape(n, cap)
            ptr = out.data_ptr()
        else:
            rows = np.frombuffer(out, np.uint8)[:n * cap * 64].view(ROW_DTYPE).reshape(n, cap)
            ptr = rows.ctypes.data
        _lib.check(_lib.lib().vbt_pipeline_rows_all(self._h, counts.ctypes.data, ptr, cap))
        return counts, rows

    def rows(self, clip):
        cap = self.tracker.rows_cap
        ids = np.empty(cap, np.int64)
        cols = np.empty((cap, 7), np.float64)
        n = ctypes.c_int()
        _lib.check(_lib.lib().vbt_pipeline_rows(self._h, int(clip), ids.ctypes.data, cols.ctypes.data, cap, ctypes.byref(n)))
        d = {"id": ids[:n.value].tolist()}
        for j, nm in enumerate(COLUMNS[1:]):
            d[nm] = cols[:n.value, j].tolist()
        return d

    def phases(self, clip):
        return self.tracker.phases(clip)

    def detections(self):
        """Most recent step's detector outputs (host copies) - for tests."""
        b = np.empty((self.n, 25, 4), np.float32)
        s = np.empty((self.n, 25), np.float32)
        c = np.empty((self.n, 25), np.float32)
        cnt = np.empty((self.n,), np.int32)
        B = ctypes.c_int()
        _lib.check(_lib.lib().vbt_pipeline_detections(self._h, b.ctypes.data, s.ctypes.data, c.ctypes.data, cnt.ctypes.data, self.n, ctypes.byref(B)))
        return b[:B.value], s[:B.value], c[:B.value], cnt[:B.value]


class _PipelineHandle:
    """Owner of a vbt_pipeline handle (destroyed with the last reference: the Pipeline and every view borrowed from it)."""

    def __init__(self, h):
        self.h = h

    def __del__(self):
        h, self.h = self.h, None
        if h and _lib is not None and getattr(_lib, "_lib", None) is not None:
            _lib._lib.vbt_pipeline_destroy(h)


class _StreamHandle:
    """A HIP stream owned by the library, as the raw handle (`.cuda_stream`, the attribute name torch uses)."""

    def __init__(self, handle):
        self.cuda_stream = int(handle or 0)

    def synchronize(self):
        _lib.check(_lib.lib().vbt_stream_synchronize(self.cuda_stream))
