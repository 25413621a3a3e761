"""`track()` + export: the reference's per-clip hot loop (reference track.py:129-260, 103-126),
and `Pipeline`, its batched MI355X form (n clips frame-wise through detect -> NMS -> tracker on
one stream, rep analysis at the end).

No GUI: drawing / imshow / VideoWriter (reference track.py:28-62,201-207,237-247) are out of
scope.  Frame sources are arrays / iterables of RGB uint8 frames instead of
cv2.VideoCapture (cv2 is not a dependency here).
"""
import ctypes
import os

import numpy as np

from . import _lib
from .interpreter import Interpreter
from .ocsort import MultiClipTracker, OCSort
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
    import torch
    T, H, W = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
    stride = max(int(frame_stride), 1)
    kept = np.arange(stride - 1, T, stride)                          # 0-based indices of the frames that are processed
    F = max(1, min(int(time_batch), max(len(kept), 1)))
    pipe = Pipeline(model_path, F, max_frames=max(len(kept), 1), fps=fps, detection_treshold=detection_treshold, device=device,
                    rows_per_frame=25, tracker_clips=1)
    size = int(pipe.interpreter.get_input_details()[0]["shape"][1])
    src_hw = None if (H, W) == (size, size) else (H, W)
    for i0 in range(0, len(kept), F):
        idx = kept[i0:i0 + F]
        chunk = frames[idx[0]:idx[-1] + 1:stride] if stride > 1 else frames[idx[0]:idx[-1] + 1]
        fd = torch.from_numpy(np.ascontiguousarray(chunk)).to(f"cuda:{device}", non_blocking=False)
        pipe.step_runs(fd, [(0, 0, len(idx), int(idx[0]) + 1, stride)], src_hw=src_hw)
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


# device -> {"streams": [torch.cuda.ExternalStream], "free": [indices], "group": {index: hardware-queue group}, "reps": [one
# stream index per known group]}: the process-wide stream pool of every Pipeline (see Pipeline._new_stream / _place_streams)
_STREAMS = {}


class StreamPlacementError(RuntimeError):
    pass


class Pipeline:
    """n clips processed frame-wise: step(frames[n,H,W,3] on the device) enqueues detect+NMS for this frame
    set and the tracker step of the PREVIOUS one; finish() drains, selects each clip's export id and runs the
    rep analysis on the device.  Nothing leaves the GPU until rows()/phases() are read.

    Software pipeline over HIP streams: the detector is stateless, so `depth` consecutive steps are in
    flight at once, each on its own stream with its own model instance (activation arena) and output
    buffers; the OC-SORT steps stay strictly ordered on a dedicated tracker stream and trail the detector,
    chained by events (detector(t) -> tracker(t) -> slot reuse).  At batch 64 a single forward leaves the
    GPU latency-bound (measured: throughput = B / (0.65 ms + 16.8 us * B)); two or more forwards in flight
    recover most of that without changing the per-step batch."""

    def __init__(self, model_path, n_clips, max_frames, fps=60.0, detection_treshold=0.5, device=0, rows_per_frame=4,
                 plate_diameter=0.45, depth=None, tracker_clips=None):
        import torch
        self.n = int(n_clips)                       # slots of the detector batch
        # tracker_clips > n_clips: more clips than batch slots; step(clip_map=...) says which clip sits in which slot
        self.n_trk = int(tracker_clips) if tracker_clips is not None else self.n
        self.fps = np.broadcast_to(np.asarray(fps, np.float64), (self.n_trk,)).copy()
        self.thr = float(detection_treshold)
        self.plate_diameter = plate_diameter
        # forwards in flight: 3 at batch 64 (four hardware queues: three forwards + the copy stream); a batch of one
        # to eight frames is launch latency (56 launches of ~6.5 us), where a fourth forward still pays (batch 8: 40.1 k -> 46.7 k frames/s)
        self.depth = int(depth if depth is not None else os.environ.get("VBT_PIPELINE_DEPTH", "4" if self.n <= 8 else "3"))
        self.depth = max(1, min(self.depth, 8))
        self._dev = device
        self._torch = torch
        tdev = torch.device(f"cuda:{device}")
        self.interpreters = [Interpreter(model_path, device=device, max_batch=self.n) for _ in range(self.depth)]
        self.interpreter = self.interpreters[0]
        self.tracker = MultiClipTracker(self.n_trk, int(max_frames) * rows_per_frame + 3 * 25, max_age=MAX_AGE,   # frames 1-3 may emit 25 rows each
                                        asso_func="diou", iou_threshold=0.1, device=device)
        self.frame_count = 0                        # time counter of the clips: time = frame_count / fps (track.py:161,169)
        self._step_idx = 0                          # steps enqueued so far: selects the ring slot, independent of time
        n = self.n
        # Deferred tracker steps (small batches): a forward of <= 8 frames is a chain of launch-latency-sized kernels, and one
        # single-wave tracker launch plus its cross-stream event at the end of EVERY forward costs a fifth of the step (batch 1,
        # depth 4: 147 vs 121 us per step).  With deferral the detections of `depth` consecutive steps stay in a ring of output
        # slots and ONE launch of the time-batched walk (vbt_tracker_update_from_detections_seq: one wavefront per clip steps
        # through the group in frame order) follows the group's last forward.  Rows, ids and phases are those of the per-step
        # form (same kernel code per frame); only plain steps (no clip_map / active) are deferred.
        dflt = "1" if (self.n <= 8 and self.n_trk == self.n and self.depth >= 2) else "0"
        inline = os.environ.get("VBT_TRACKER_STREAM", "inline" if self.depth >= 3 else "own") == "inline"
        self._defer = self.depth if os.environ.get("VBT_TRACKER_DEFER", dflt) == "1" and self.n_trk == self.n and self.depth >= 2 and inline else 0
        self._ring = R = 2 * self.depth if self._defer else self.depth     # output slots: a group may still be read while the next one fills
        # one block per output tensor, [ring slot][clip]...: the walk addresses frame f of clip c as slot (o0 + f) * n + c
        self._out = (torch.empty((R, n, 25, 4), dtype=torch.float32, device=tdev), torch.empty((R, n, 25), dtype=torch.float32, device=tdev),
                     torch.empty((R, n, 25), dtype=torch.float32, device=tdev), torch.empty((R, n), dtype=torch.int32, device=tdev))
        self._bufs = [tuple(t[o] for t in self._out) for o in range(R)]
        self._times = [np.zeros(n, np.float64) for _ in range(R)]
        self._maps = [None] * R                      # per-slot clip maps of the steps in flight
        self._fc = [0] * R                           # per-slot frame number of a plain step
        self._group = []                             # deferred plain steps (output slots, ascending), not yet handed to the tracker
        # The pipeline's own HIP streams, created back to back (detector slots, tracker, copy): each is bound to its
        # hardware queue at creation (vbt_stream_create), so they sit on distinct queues.  Streams from torch's pool may have
        # been used before and then share a queue with a neighbour - measured 89 k -> 58 k frames/s.
        self._own_streams = []
        self._det_streams = [self._new_stream(tdev) for _ in range(self.depth)]
        self._copy_stream = self._new_stream(tdev)
        self._trk_stream = self._new_stream(tdev)
        # Where the OC-SORT step of a frame runs.  "own": on the tracker stream (it waits for the slot's detections).  "inline":
        # at the end of the slot's own stream, after an event wait on the previous frame's tracker step.  The GPU runs four
        # hardware queues side by side; a fifth active one costs a quarter of the throughput.  Inline keeps the pipeline on
        # `depth` queues: at depth 3 that leaves the fourth to the copy stream of the host-fed mode (MI355X, 64 clips: 92.9 k
        # vs 92.4 k frames/s device-resident, 77.8 k vs 71.7 k with frames from pinned host memory).  Depth 4 is slower either
        # way (81-83 k).  Default: inline from depth 3, own stream below (there the tracker overlaps the next forward).
        mode = os.environ.get("VBT_TRACKER_STREAM", "inline" if self.depth >= 3 else "own")
        if mode not in ("own", "inline"):
            raise ValueError("VBT_TRACKER_STREAM must be 'own' or 'inline'")
        self._trk_inline = mode == "inline"
        self._place_streams(tdev)
        self._last_trk_ev = None                    # the most recent tracker step (inline mode orders the steps through it)
        self._ev_in = [torch.cuda.Event() for _ in range(self.depth)]
        self._ev_det = [torch.cuda.Event() for _ in range(self._ring)]
        self._ev_trk = [None] * self._ring          # tracker finished reading output slot o
        self._pending = []                          # slots whose tracker step has not been enqueued yet
        self._resized = [None] * self.depth         # per-slot network-resolution frames (source-resolution input)
        # step() on pinned host memory: H2D copies run on their own stream into a ring of depth + 2 staging buffers, i.e. up
        # to two steps ahead of the forwards, so that a slot's forward never waits for its own copy
        self._stage = [None] * (self.depth + 2)
        self._stage_free = [None] * (self.depth + 2)
        self._stage_idx = 0
        # Self-check: every slot runs its whole plan on a blank batch before the first real frame, so a plan the kernels
        # reject (LDS budget, tile shape) fails here and not in the middle of a clip; the first real step then also finds
        # code objects, arenas and GPU clocks warm.  Detector only: no tracker state is touched.
        n_check = int(os.environ.get("VBT_PIPELINE_SELFCHECK", "1"))
        if n_check > 0:
            size = int(self.interpreter.get_input_details()[0]["shape"][1])
            blank = torch.zeros((self.n, size, size, 3), dtype=torch.uint8, device=tdev)
            torch.cuda.current_stream().synchronize()
            for _ in range(n_check):
                for k in range(self.depth):
                    b, s_, c, cnt = self._bufs[k]
                    _lib.check(_lib.lib().vbt_detect_async(self.interpreters[k].handle, blank.data_ptr(), self.n,
                                                           self._det_streams[k].cuda_stream, b.data_ptr(), s_.data_ptr(), c.data_ptr(), cnt.data_ptr()))
            for S in self._det_streams:
                S.synchronize()

    def _pool(self):
        return _STREAMS.setdefault(self._dev, {"streams": [], "free": [], "group": {}, "reps": []})

    def _take(self, tdev, i=None, create=False):
        """Stream i of the pool (None: the lowest free index, or - none free / create - a new stream), now owned by this pipeline."""
        pool = self._pool()
        if i is None:
            if pool["free"] and not create:
                i = min(pool["free"])
            else:
                h = ctypes.c_void_p()
                _lib.check(_lib.lib().vbt_stream_create(self._dev, ctypes.byref(h)))
                pool["streams"].append(self._torch.cuda.ExternalStream(h.value, device=tdev))
                i = len(pool["streams"]) - 1
                pool["free"].append(i)
        pool["free"].remove(i)
        self._own_streams.append(i)
        return i

    def _new_stream(self, tdev):
        torch = self._torch
        if os.environ.get("VBT_TORCH_POOL_STREAMS") == "1":
            return torch.cuda.Stream(device=tdev)
        # Streams are kept for the life of the process and handed out again when a pipeline goes away.  They are never
        # destroyed: torch's caching allocator may still hold record_stream() references to them.
        return self._pool()["streams"][self._take(tdev)]

    def _place_streams(self, tdev):
        """The streams that carry kernels side by side (detector slots, the copy stream, the tracker stream unless its step
        runs inline) must sit on distinct hardware queues.  HIP binds a stream to one of GPU_MAX_HW_QUEUES hardware queues when
        it is created (a zig-zag that also counts streams created by others) and the queue cannot be queried, so a pool stream
        is CLASSIFIED once per process: timed with a spinning wave against one representative of every queue group known so far
        (vbt_streams_share_queue, ~0.3 ms per probe).  A pipeline then takes its busy streams from distinct groups - a stream that
        once collided is simply left for another role - and only creates streams while some group is still unseen, so that
        any number of pipelines created one after the other in a process end up on the same few streams (round 3: the twelve-
        stream budget of the old swap loop ran out after a few pipelines and the pipeline silently shared queues: -35 %).
        VBT_STRICT_PLACEMENT=1 turns a failed placement into StreamPlacementError (bench.py sets it)."""
        if os.environ.get("VBT_TORCH_POOL_STREAMS") == "1" or os.environ.get("VBT_PLACE_STREAMS", "1") == "0":
            return
        L = _lib.lib()
        pool = self._pool()
        self._torch.cuda.synchronize()

        def shared(i, j):
            sh = ctypes.c_int()
            a, b = pool["streams"][i], pool["streams"][j]
            for _ in range(2):      # host-timed: a descheduled host thread can make one probe read "shared"; two in a row cannot
                _lib.check(L.vbt_streams_share_queue(a.cuda_stream, b.cuda_stream, 150, ctypes.byref(sh)))
                if not sh.value:
                    return False
            return True

        def group_of(i):
            if i not in pool["group"]:
                g = next((g for g, rep in enumerate(pool["reps"]) if shared(i, rep)), None)
                if g is None:
                    g = len(pool["reps"])
                    pool["reps"].append(i)
                pool["group"][i] = g
            return pool["group"][i]

        # (four hardware queues: with four forwards in flight the copy stream has to share one, which costs a small batch nothing)
        busy = [("det", k) for k in range(self.depth)] + ([("copy", 0)] if self.depth < 4 else []) + ([] if self._trk_inline else [("trk", 0)])
        index_of = {id(st): i for i, st in enumerate(pool["streams"])}
        roles = {("det", k): index_of[id(self._det_streams[k])] for k in range(self.depth)}
        roles[("copy", 0)] = index_of[id(self._copy_stream)]
        roles[("trk", 0)] = index_of[id(self._trk_stream)]
        nq = max(1, int(os.environ.get("GPU_MAX_HW_QUEUES", "4")))
        used, failed = set(), False
        for role in busy:
            cur = roles[role]
            if group_of(cur) in used:
                # another stream of a group this pipeline does not use yet: one it already holds for an idle role, a free pool
                # stream, or - while fewer groups than hardware queues are known, and at most 3 nq times - a new one
                spare = [i for r_, i in roles.items() if r_ not in busy and group_of(i) not in used]
                cand = spare or [i for i in sorted(pool["free"]) if group_of(i) not in used]
                created = 0
                while not cand and len(pool["reps"]) < nq and created < 3 * nq:
                    i = self._take(tdev, create=True)
                    created += 1
                    if group_of(i) not in used:
                        cand = [i]
                    else:
                        pool["free"].append(i)          # stays in the pool for a later pipeline / another role
                        self._own_streams.remove(i)
                if not cand:
                    failed = True
                    continue
                new = cand[0]
                if new in pool["free"]:
                    self._take(tdev, new)
                if spare:                                # swap the two roles' streams
                    other = next(r_ for r_, i in roles.items() if i == new)
                    roles[other] = cur
                roles[role] = new
                cur = new
            used.add(group_of(cur))
        for k in range(self.depth):
            self._det_streams[k] = pool["streams"][roles[("det", k)]]
        self._copy_stream = pool["streams"][roles[("copy", 0)]]
        self._trk_stream = pool["streams"][roles[("trk", 0)]]
        # streams taken but left without a role go back to the pool
        held = set(roles.values())
        for i in [i for i in self._own_streams if i not in held]:
            self._own_streams.remove(i)
            pool["free"].append(i)
        if failed:
            msg = (f"vbt_amd: could not give every pipeline stream its own hardware queue: {len(busy)} busy streams (depth {self.depth}"
                   f"{'' if self._trk_inline else ' + tracker stream'}{' + copy stream' if self.depth < 4 else ''}), {len(pool['reps'])} distinct "
                   f"queues seen, GPU_MAX_HW_QUEUES={nq} (too few queues for this configuration, or kernels are being serialised by a "
                   "profiler); throughput will be lower")
            if os.environ.get("VBT_STRICT_PLACEMENT") == "1":
                raise StreamPlacementError(msg)
            import warnings
            warnings.warn(msg)

    def __del__(self):
        try:
            idx, self._own_streams = getattr(self, "_own_streams", []), []
            if idx and _STREAMS is not None:
                _STREAMS[self._dev]["free"].extend(idx)
        except Exception:
            pass

    def _enqueue_tracker(self, o):
        if self._trk_inline:
            T = self._det_streams[o % self.depth]                    # stream order gives "after this slot's detections"
            if self._last_trk_ev is not None:
                T.wait_event(self._last_trk_ev)                      # tracker steps run in frame order
        else:
            T = self._trk_stream
            T.wait_event(self._ev_det[o])
        k = o
        b, s, c, cnt = self._bufs[k]
        # frame times / clip map of the step travel in the kernel arguments (read during the call, no copy in flight)
        tm = self._times[k]
        mp = self._maps[k]
        if isinstance(mp, tuple):                                    # time-batched step: runs of consecutive frames
            _, ra, B = mp
            _lib.check(_lib.lib().vbt_tracker_update_from_detections_seq(self.tracker.handle, b.data_ptr(), s.data_ptr(), cnt.data_ptr(), B,
                                                                         ra, len(ra), self.thr, T.cuda_stream))
        elif mp is not None:
            _lib.check(_lib.lib().vbt_tracker_update_from_slots(self.tracker.handle, b.data_ptr(), s.data_ptr(), cnt.data_ptr(),
                                                                mp.ctypes.data, tm.ctypes.data, self.n, self.thr, T.cuda_stream))
        else:
            _lib.check(_lib.lib().vbt_tracker_update_from_detections(self.tracker.handle, b.data_ptr(), s.data_ptr(), cnt.data_ptr(),
                                                                     tm.ctypes.data, self.thr, T.cuda_stream))
        ev = self._torch.cuda.Event()
        ev.record(T)
        self._ev_trk[k] = ev
        self._last_trk_ev = ev

    def _flush_group(self):
        """Hand the deferred plain steps to the tracker: ONE launch of the time-batched walk on the stream of the group's last
        forward, after the other members' forwards (events) and the previous tracker launch."""
        g, self._group = self._group, []
        if not g:
            return
        for o in g:
            self._pending.remove(o)
        if len(g) == 1:
            self._enqueue_tracker(g[0])
            return
        n, last = self.n, g[-1]
        T = self._det_streams[last % self.depth]
        for o in g[:-1]:
            T.wait_event(self._ev_det[o])
        if self._last_trk_ev is not None:
            T.wait_event(self._last_trk_ev)
        fstep = self._fc[g[1]] - self._fc[g[0]]
        ra = (_lib.Run * n)()
        for c in range(n):
            ra[c] = _lib.Run(c, c, n, len(g), self._fc[g[0]], fstep, float(self.fps[c]))
        b, s, _, cnt = self._bufs[g[0]]                              # slot (o - g[0]) * n + c of the block that starts here
        _lib.check(_lib.lib().vbt_tracker_update_from_detections_seq(self.tracker.handle, b.data_ptr(), s.data_ptr(), cnt.data_ptr(), len(g) * n,
                                                                     ra, n, self.thr, T.cuda_stream))
        ev = self._torch.cuda.Event()
        ev.record(T)
        for o in g:
            self._ev_trk[o] = ev
        self._last_trk_ev = ev

    def step(self, frames_dev_ptr, stream=None, src_hw=None, swap_rb=False, active=None, clip_map=None, frame_idx=None, track=True):
        """frames_dev_ptr: uint8 [n,H,W,3] on the device (frame `frame_count+1` of every clip), valid on the caller's current
        torch stream: either a torch tensor (preferred: its lifetime is then handled here) or a raw device pointer, which
        the caller must keep alive and unmodified until the step has run (up to `depth` steps later).  src_hw=(H, W) of the source frames when they are not at the network
        resolution: the bilinear resize + truncating cast of reference odt.py:10-19 (and, with swap_rb, the
        BGR->RGB of track.py:171) then run on the device, on the slot's stream, ahead of the detector.
        active: optional bool [n] - clips that still have a frame in this step (clips of different lengths batched together;
        the reference processes them one after the other, track.py:85-126).  Inactive clips keep their tracker state and
        their frame counter; whatever sits in their slot of the frame batch is detected on but ignored.
        clip_map / frame_idx: int [n] - slot i carries frame number frame_idx[i] (1-based) of tracker clip clip_map[i] (-1:
        empty slot).  With tracker_clips > n_clips a slot moves on to the next clip of its queue when one ends, so a corpus
        of ragged clips keeps the whole detector batch busy."""
        torch = self._torch
        o = self._step_idx % self._ring                              # output slot; k = forward slot (model instance, stream)
        k = o % self.depth
        plain = clip_map is None and active is None and track
        if self._group and (not plain or o <= self._group[-1] or
                            (len(self._group) >= 2 and self.frame_count + 1 - self._fc[self._group[-1]] != self._fc[self._group[1]] - self._fc[self._group[0]])):
            self._flush_group()                                      # (ring wrap, another kind of step, or skip_frames() changed the frame stride)
        if o in self._pending:
            raise RuntimeError(f"Pipeline: ring slot {o} still holds a step whose tracker update has not been enqueued")
        self._step_idx += 1
        self.frame_count += 1
        S = self._det_streams[k]
        host_frames = None
        if hasattr(frames_dev_ptr, "data_ptr"):
            if frames_dev_ptr.device.type == "cpu":
                # frames in (pinned) host memory, the reference's situation (track.py:160 hands every frame over from the
                # host): the H2D copy is enqueued on the slot's stream, so it overlaps the forwards of the other slots.  The
                # caller must leave the host buffer untouched until the step has run (up to `depth` steps later).
                host_frames = frames_dev_ptr
            else:
                # a device tensor: its storage is used on the slot's stream, up to `depth` steps after this call returns; tell
                # the caching allocator, so that dropping the tensor does not hand the memory to a later batch too early
                frames_dev_ptr.record_stream(S)
                frames_dev_ptr = frames_dev_ptr.data_ptr()
        self._ev_in[k].record(torch.cuda.current_stream())           # frames are ready once the caller's stream gets here
        S.wait_event(self._ev_in[k])
        if self._ev_trk[o] is not None:
            S.wait_event(self._ev_trk[o])                            # the tracker is done with this slot's previous outputs
        stage_j = None
        if host_frames is not None:
            stage_j = j = self._stage_idx % len(self._stage)
            self._stage_idx += 1
            self._staging(j, host_frames.shape)
            C = self._copy_stream
            self._host_copy_gate(j)
            with torch.cuda.stream(C):
                self._stage[j].copy_(host_frames, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(C)
            S.wait_event(ev)
            frames_dev_ptr = self._stage[j].data_ptr()
        size = int(self.interpreter.get_input_details()[0]["shape"][1])
        if src_hw is not None and (tuple(src_hw) != (size, size) or swap_rb):
            if self._resized[k] is None:
                self._resized[k] = torch.empty((self.n, size, size, 3), dtype=torch.uint8, device=torch.device(f"cuda:{self._dev}"))
            _lib.check(_lib.lib().vbt_resize_frames(frames_dev_ptr, self.n, int(src_hw[0]), int(src_hw[1]), 1, self._resized[k].data_ptr(),
                                                    size, size, 1, int(bool(swap_rb)), self._dev, S.cuda_stream))
            frames_dev_ptr = self._resized[k].data_ptr()
        self._maps[o] = None
        self._fc[o] = self.frame_count
        if clip_map is not None:
            cm = np.ascontiguousarray(clip_map, dtype=np.int32)
            fi = np.asarray(frame_idx, np.float64)
            self._maps[o] = cm
            self._times[o][:] = np.where(cm >= 0, fi / self.fps[np.maximum(cm, 0)], -1.0)
        elif active is None:
            np.divide(float(self.frame_count), self.fps, out=self._times[o])  # time = frame_count / fps (track.py:169)
        else:
            act = np.asarray(active, bool)
            self._clip_frames = getattr(self, "_clip_frames", np.zeros(self.n, np.int64))
            self._clip_frames[act] += 1
            np.divide(self._clip_frames.astype(np.float64), self.fps, out=self._times[o])
            self._times[o][~act] = -1.0
        b, s, c, cnt = self._bufs[o]
        _lib.check(_lib.lib().vbt_detect_async(self.interpreters[k].handle, frames_dev_ptr, self.n, S.cuda_stream, b.data_ptr(),
                                               s.data_ptr(), c.data_ptr(), cnt.data_ptr()))
        self._ev_det[o].record(S)
        if stage_j is not None:
            ev = torch.cuda.Event()
            ev.record(S)
            self._stage_free[stage_j] = ev
        if not track:                                                # detector-only step (measurement splits)
            return
        self._pending.append(o)
        if self._defer and plain:
            self._group.append(o)
            if len(self._group) >= self._defer or o % self._defer == self._defer - 1:    # groups are aligned: their slots never wrap
                self._flush_group()
            return
        # own stream: keep depth-1 detector steps ahead of the tracker; inline: the step follows its forward directly
        while len(self._pending) >= (1 if self._trk_inline else self.depth):
            self._enqueue_tracker(self._pending.pop(0))

    def _host_copy_gate(self, j):
        """Before an H2D copy into staging buffer j is enqueued: the forward that last read the buffer must be done.  The wait is on
        the HOST (the event is `depth + 2` steps old: it has completed unless the caller is that many steps ahead of the GPU, and then
        blocking the caller is the back-pressure wanted), NOT a stream wait on the copy stream: a cross-stream event wait in front of
        a DMA copy makes hipMemcpyAsync itself block the calling thread on this stack - 0.6-0.9 ms per step instead of 0.2 - and costs
        the host-fed pipeline 6 % (97.5 k -> 103.9 k frames/s without it, profiles/r04_h2d_pinned_order.md).  Frames in host memory are
        ready when the call is made, so the copy stream does not wait for the caller's stream either."""
        ev = self._stage_free[j]
        if ev is not None:
            ev.synchronize()

    def _staging(self, j, shape):
        """Staging buffer j of the host-fed / gathered input ring with (at least) the given shape.  A buffer that has to be
        replaced may still be read by a forward or written by a copy in flight (up to depth + 2 steps): it is handed back
        to the caching allocator only after every stream of the pipeline has been told about it."""
        torch = self._torch
        cur = self._stage[j]
        if cur is None or tuple(cur.shape) != tuple(shape):
            if cur is not None:
                for S in self._det_streams + [self._copy_stream]:
                    cur.record_stream(S)
            self._stage[j] = torch.empty(tuple(shape), dtype=torch.uint8, device=torch.device(f"cuda:{self._dev}"))
        return self._stage[j]

    def step_runs(self, frames, runs, stream=None, src_hw=None, swap_rb=False, track=True, outputs=None):
        """Time-batched step (the reference's unit of work is ONE video, track.py:85-126,159-247): the detector batch holds
        RUNS of consecutive frames of a clip instead of one frame of each clip; the OC-SORT steps of a run are walked in
        frame order by one wavefront inside ONE tracker launch (vbt_tracker_update_from_detections_seq).
        runs: sequence of (clip, slot0, n_frames, frame0[, frame_step]) - frame f of the run sits in batch slot slot0 + f and
        is frame number frame0 + f * frame_step (1-based) of tracker clip `clip`; its time stamp is frame number / fps[clip].
        frames: the assembled batch - a device tensor / raw device pointer or a (pinned) host tensor [B,H,W,3], B = slots used
        - or a list with one tensor [n_frames,H,W,3] per run (device or host, contiguous): the batch is then assembled here,
        by one gather launch (device sources) or one H2D copy per run on the copy stream (host sources).
        outputs (with track=False): (boxes [B,25,4] f32, scores [B,25] f32, classes [B,25] f32, counts [B] i32) device tensors that
        receive this step's detections instead of the pipeline's ring buffers - the frame-major multi-GPU mode
        collects a whole frame chunk's detections for the gather; the caller keeps them alive until the stream has run."""
        torch = self._torch
        L = _lib.lib()
        if outputs is not None and track:
            raise ValueError("step_runs: outputs= is for detector-only steps (track=False)")
        self._flush_group()
        o = self._step_idx % self._ring
        k = o % self.depth
        if o in self._pending:
            raise RuntimeError(f"Pipeline: ring slot {o} still holds a step whose tracker update has not been enqueued")
        ra = (_lib.Run * len(runs))()
        B = 0
        for i, r in enumerate(runs):
            clip, slot0, nf, frame0 = (int(v) for v in r[:4])
            fstep = int(r[4]) if len(r) > 4 else 1
            if not (0 <= clip < self.n_trk) or nf < 1 or slot0 < 0 or slot0 + nf > self.n:
                raise ValueError(f"run {i}: clip {clip}, slots {slot0}..{slot0 + nf - 1} outside {self.n_trk} clips / {self.n} slots")
            ra[i] = _lib.Run(clip, slot0, 1, nf, frame0, fstep, float(self.fps[clip]))
            B = max(B, slot0 + nf)
        self._step_idx += 1
        S = self._det_streams[k]
        self._ev_in[k].record(torch.cuda.current_stream())
        S.wait_event(self._ev_in[k])
        if self._ev_trk[o] is not None:
            S.wait_event(self._ev_trk[o])
        size = int(self.interpreter.get_input_details()[0]["shape"][1])
        stage_j = None
        if isinstance(frames, (list, tuple)):
            if len(frames) != len(runs):
                raise ValueError("one source tensor per run")
            shp = tuple(frames[0].shape[1:])
            dev0 = frames[0].device
            for i, (src, r) in enumerate(zip(frames, ra)):
                # raw pointers base + f * frame_bytes go to the gather kernel / the copies: a short, strided or differently
                # shaped source would make them read past its allocation
                if src.dtype != torch.uint8 or src.dim() != 4 or tuple(src.shape[1:]) != shp or not src.is_contiguous():
                    raise ValueError(f"step_runs: source {i} must be a contiguous uint8 tensor [n_frames, {shp[0]}, {shp[1]}, {shp[2]}]")
                if int(src.shape[0]) < r.n_frames:
                    raise ValueError(f"step_runs: source {i} holds {int(src.shape[0])} frames, its run needs {r.n_frames}")
                if src.device != dev0:
                    raise ValueError(f"step_runs: source {i} is on {src.device}, source 0 on {dev0}")
            used = np.zeros(B, bool)
            for r in ra:
                used[r.slot0:r.slot0 + r.n_frames] = True
            if not used.all():
                raise ValueError("step_runs: the runs leave a hole in the detector batch")
            stage_j = j = self._stage_idx % len(self._stage)
            self._stage_idx += 1
            st = self._staging(j, (self.n,) + shp)
            fb = int(np.prod(shp))
            on_host = dev0.type == "cpu"
            if on_host:
                C = self._copy_stream
                self._host_copy_gate(j)
                with torch.cuda.stream(C):
                    for src, r in zip(frames, ra):
                        st[r.slot0:r.slot0 + r.n_frames].copy_(src[:r.n_frames], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(C)
                S.wait_event(ev)
            else:
                if self._stage_free[j] is not None:
                    S.wait_event(self._stage_free[j])
                if fb % 16 == 0 and all(src.data_ptr() % 16 == 0 for src in frames):
                    ptrs = (ctypes.c_void_p * B)()
                    for src, r in zip(frames, ra):
                        src.record_stream(S)
                        base = src.data_ptr()
                        for f in range(r.n_frames):
                            ptrs[r.slot0 + f] = base + f * fb
                    _lib.check(L.vbt_gather_frames(st.data_ptr(), ptrs, B, fb, S.cuda_stream))
                else:
                    # a frame size that is not a multiple of 16 bytes (any source resolution is allowed): the gather kernel
                    # moves 16-byte pieces, so the batch is assembled by one device copy per run instead
                    with torch.cuda.stream(S):
                        for src, r in zip(frames, ra):
                            src.record_stream(S)
                            st[r.slot0:r.slot0 + r.n_frames].copy_(src[:r.n_frames], non_blocking=True)
            frames_ptr = st.data_ptr()
        elif hasattr(frames, "data_ptr"):
            if frames.device.type == "cpu":
                stage_j = j = self._stage_idx % len(self._stage)
                self._stage_idx += 1
                st = self._staging(j, (self.n,) + tuple(frames.shape[1:]))
                C = self._copy_stream
                self._host_copy_gate(j)
                with torch.cuda.stream(C):
                    st[:frames.shape[0]].copy_(frames, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(C)
                S.wait_event(ev)
                frames_ptr = st.data_ptr()
            else:
                frames.record_stream(S)
                frames_ptr = frames.data_ptr()
        else:
            frames_ptr = frames
        if src_hw is not None and (tuple(src_hw) != (size, size) or swap_rb):
            if self._resized[k] is None:
                self._resized[k] = torch.empty((self.n, size, size, 3), dtype=torch.uint8, device=torch.device(f"cuda:{self._dev}"))
            _lib.check(L.vbt_resize_frames(frames_ptr, B, int(src_hw[0]), int(src_hw[1]), 1, self._resized[k].data_ptr(), size, size, 1,
                                           int(bool(swap_rb)), self._dev, S.cuda_stream))
            frames_ptr = self._resized[k].data_ptr()
        b, s_, c, cnt = self._bufs[o] if outputs is None else outputs
        if outputs is not None:
            for t_, shp_, dt_ in zip(outputs, ((B, 25, 4), (B, 25), (B, 25), (B,)), (torch.float32, torch.float32, torch.float32, torch.int32)):
                if tuple(t_.shape) != shp_ or t_.dtype != dt_ or not t_.is_contiguous() or t_.device.type != "cuda":
                    raise ValueError(f"step_runs: outputs must be contiguous device tensors {shp_} {dt_}")
                t_.record_stream(S)
        _lib.check(L.vbt_detect_async(self.interpreters[k].handle, frames_ptr, B, S.cuda_stream, b.data_ptr(), s_.data_ptr(), c.data_ptr(),
                                      cnt.data_ptr()))
        self._ev_det[o].record(S)
        if stage_j is not None:
            ev = torch.cuda.Event()
            ev.record(S)
            self._stage_free[stage_j] = ev
        self._maps[o] = ("runs", ra, B)
        self._last_B = B
        if not track:
            return
        self._pending.append(o)
        while len(self._pending) >= (1 if self._trk_inline else self.depth):
            self._enqueue_tracker(self._pending.pop(0))

    def join_detectors(self):
        """The caller's current torch stream waits for every forward enqueued so far (after detector-only steps their outputs are
        then safe to read on it)."""
        cur = self._torch.cuda.current_stream()
        for ev in self._ev_det:
            cur.wait_event(ev)

    def step_seq(self, frames, frame0=None, stream=None, **kw):
        """F consecutive frames of EVERY clip in one step: frames [n_clips, F, H, W, 3] (clip-major, device or pinned host
        tensor); the clips' frame counters advance by F."""
        ncl, F = int(frames.shape[0]), int(frames.shape[1])
        if ncl != self.n_trk or ncl * F > self.n:
            raise ValueError(f"step_seq: {ncl} clips x {F} frames do not fit {self.n_trk} clips / {self.n} slots")
        f0 = self.frame_count + 1 if frame0 is None else int(frame0)
        self.step_runs(frames.reshape((ncl * F,) + tuple(frames.shape[2:])), [(c, c * F, F, f0) for c in range(ncl)], stream, **kw)
        self.frame_count = f0 + F - 1

    def reset(self):
        """Back to frame 0 of fresh clips (tracker state cleared); models, streams and buffers are kept."""
        self._drain()
        self._torch.cuda.synchronize()
        self.tracker.reset()
        self.frame_count = 0
        self._step_idx = 0
        self._ev_trk = [None] * self._ring
        self._last_trk_ev = None
        if hasattr(self, "_clip_frames"):
            self._clip_frames[:] = 0

    def tracker_only_steps(self, count, slot=0):
        """Measurement split: `count` tracker steps of all clips on the detections sitting in ring slot `slot`."""
        if self.n_trk != self.n:
            raise RuntimeError("tracker_only_steps needs one tracker clip per detector slot")
        self._flush_group()                                          # (deferred steps first: tracker launches stay in frame order)
        b, s, c, cnt = self._bufs[slot]
        T = self._trk_stream
        T.wait_event(self._ev_det[slot])
        if self._last_trk_ev is not None:
            T.wait_event(self._last_trk_ev)
        tm = np.empty(self.n, np.float64)
        for i in range(count):
            self.frame_count += 1
            np.divide(float(self.frame_count), self.fps[:self.n], out=tm)
            _lib.check(_lib.lib().vbt_tracker_update_from_detections(self.tracker.handle, b.data_ptr(), s.data_ptr(), cnt.data_ptr(),
                                                                     tm.ctypes.data, self.thr, T.cuda_stream))
        ev = self._torch.cuda.Event()
        ev.record(T)
        self._ev_trk[slot] = ev
        self._last_trk_ev = ev

    def skip_frames(self, n=1):
        """Frames read from the source but not processed (`frame_count % 16` of reference track.py:161-167): they advance
        the clip time and nothing else - no ring slot is used."""
        self.frame_count += int(n)

    def _drain(self):
        self._flush_group()
        while self._pending:
            self._enqueue_tracker(self._pending.pop(0))
        if self._trk_inline and self._last_trk_ev is not None:       # clip close / row reads run on the tracker stream
            self._trk_stream.wait_event(self._last_trk_ev)

    def finish(self, stream=None):
        self._drain()
        self.tracker.finish(self.plate_diameter, stream=self._trk_stream.cuda_stream)
        self._trk_stream.synchronize()

    def close(self, cap=32):
        """Clip close in one go: drain the pipeline, export-id selection + rep analysis on the device, then ONE packed
        device-to-host copy and ONE stream synchronisation.  Returns (best_ids[n], n_rows[n], n_phases[n], overflow[n],
        phases[n, cap, 6]) - per clip the id of reference track.py:107-115 and the Phase list of plot.py:33-47."""
        self._drain()
        self.tracker.finish(self.plate_diameter, stream=self._trk_stream.cuda_stream)
        return self.tracker.summary(cap=cap)

    def rows_all(self, cap=None, out=None):
        """DataFrame rows of every clip, one strided copy (after close() / finish())."""
        self._drain()
        return self.tracker.rows_all(cap=cap, out=out, stream=self._trk_stream.cuda_stream)

    def rows(self, clip):
        self._drain()
        self._trk_stream.synchronize()
        return self.tracker.rows(clip)

    def phases(self, clip):
        return self.tracker.phases(clip)

    def detections(self):
        """Most recent step's detector outputs (host copies) - for tests."""
        o = (self._step_idx - 1) % self._ring
        self._det_streams[o % self.depth].synchronize()
        b, s, c, cnt = self._bufs[o]
        return b.cpu().numpy(), s.cpu().numpy(), c.cpu().numpy(), cnt.cpu().numpy()
