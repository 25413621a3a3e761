"""ORACLE - TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline).

numpy restatement of the tracker half of the reference hot path:
  OCSort(max_age=30, asso_func="diou", iou_threshold=0.1)      reference track.py:157
  tracker.update(dets[N,6], []) -> rows [x1,y1,x2,y2,id,cls,score]   reference track.py:186-190
  tracker.trackers[i].id / .kf.x (7x1)                          reference track.py:194-199
The implementation lives in the un-vendored, UNPINNED `ocsort` package (imported at reference
track.py:17, absent from requirements.txt).  This file restates the published OC-SORT algorithm
[EXTERNAL: Cao et al., "Observation-Centric SORT", public reference implementation] with the
call shape the reference uses: SORT 7-state constant-velocity Kalman filter
(x = [cx, cy, s, r, vcx, vcy, vs]), observation-centric re-update (ORU) after a lost period,
velocity-direction consistency cost (OCM, inertia 0.2, delta_t 3), observation-centric recovery
(OCR) with the configured asso_func, min_hits = 3 emission rule, deletion after max_age misses.

Pinned by the reference's committed outputs (tests/test_oracle_ocsort.py, fixtures from reference
dfs_ocsort/): Kalman dx,dy bit-for-bit on contiguous-from-birth segments, first-frame rows emitted
from the filter state (the 1e-6 in convert_bbox_to_z is visible in the data), ids, emission rule.
The association COST (IoU + OCM term; DIoU in OCR) is only weakly exercised by those clips
(<= 3 well separated plates): "parity unpinned" for the exact cost formula.
"""
from __future__ import annotations

import numpy as np


def linear_assignment(cost):
    """Optimal assignment of a rectangular cost matrix -> array of (row, col)."""
    from scipy.optimize import linear_sum_assignment
    r, c = linear_sum_assignment(cost)
    return np.array(list(zip(r, c)), dtype=int).reshape(-1, 2)


def iou_batch(b1, b2):
    b2 = np.expand_dims(b2, 0)
    b1 = np.expand_dims(b1, 1)
    xx1 = np.maximum(b1[..., 0], b2[..., 0])
    yy1 = np.maximum(b1[..., 1], b2[..., 1])
    xx2 = np.minimum(b1[..., 2], b2[..., 2])
    yy2 = np.minimum(b1[..., 3], b2[..., 3])
    w = np.maximum(0.0, xx2 - xx1)
    h = np.maximum(0.0, yy2 - yy1)
    wh = w * h
    return wh / ((b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1]) + (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1]) - wh)


def diou_batch(b1, b2):
    b2 = np.expand_dims(b2, 0)
    b1 = np.expand_dims(b1, 1)
    xx1 = np.maximum(b1[..., 0], b2[..., 0])
    yy1 = np.maximum(b1[..., 1], b2[..., 1])
    xx2 = np.minimum(b1[..., 2], b2[..., 2])
    yy2 = np.minimum(b1[..., 3], b2[..., 3])
    w = np.maximum(0.0, xx2 - xx1)
    h = np.maximum(0.0, yy2 - yy1)
    wh = w * h
    iou = wh / ((b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1]) + (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1]) - wh)
    cx1 = (b1[..., 0] + b1[..., 2]) / 2.0
    cy1 = (b1[..., 1] + b1[..., 3]) / 2.0
    cx2 = (b2[..., 0] + b2[..., 2]) / 2.0
    cy2 = (b2[..., 1] + b2[..., 3]) / 2.0
    inner = (cx1 - cx2) ** 2 + (cy1 - cy2) ** 2
    xc1 = np.minimum(b1[..., 0], b2[..., 0])
    yc1 = np.minimum(b1[..., 1], b2[..., 1])
    xc2 = np.maximum(b1[..., 2], b2[..., 2])
    yc2 = np.maximum(b1[..., 3], b2[..., 3])
    outer = (xc2 - xc1) ** 2 + (yc2 - yc1) ** 2
    return (iou - inner / outer + 1) / 2.0


ASSO = {"iou": iou_batch, "diou": diou_batch}


def convert_bbox_to_z(bbox):
    w = bbox[2] - bbox[0]
    h = bbox[3] - bbox[1]
    x = bbox[0] + w / 2.0
    y = bbox[1] + h / 2.0
    s = w * h
    r = w / float(h + 1e-6)
    return np.array([x, y, s, r]).reshape((4, 1))


def convert_x_to_bbox(x):
    w = np.sqrt(x[2] * x[3])
    h = x[2] / w
    return np.array([x[0] - w / 2.0, x[1] - h / 2.0, x[0] + w / 2.0, x[1] + h / 2.0]).reshape((1, 4))


def speed_direction(b1, b2):
    cx1, cy1 = (b1[0] + b1[2]) / 2.0, (b1[1] + b1[3]) / 2.0
    cx2, cy2 = (b2[0] + b2[2]) / 2.0, (b2[1] + b2[3]) / 2.0
    speed = np.array([cy2 - cy1, cx2 - cx1])
    norm = np.sqrt((cy2 - cy1) ** 2 + (cx2 - cx1) ** 2) + 1e-6
    return speed / norm


def k_previous_obs(observations, cur_age, k):
    if len(observations) == 0:
        return [-1, -1, -1, -1, -1]
    for i in range(k):
        dt = k - i
        if cur_age - dt in observations:
            return observations[cur_age - dt]
    return observations[max(observations.keys())]


class KalmanFilter7:
    """filterpy-style linear KF with OC-SORT's freeze/unfreeze (observation-centric re-update)."""

    def __init__(self):
        self.x = np.zeros((7, 1))
        self.P = np.eye(7)
        self.Q = np.eye(7)
        self.F = np.array([[1, 0, 0, 0, 1, 0, 0], [0, 1, 0, 0, 0, 1, 0], [0, 0, 1, 0, 0, 0, 1], [0, 0, 0, 1, 0, 0, 0],
                           [0, 0, 0, 0, 1, 0, 0], [0, 0, 0, 0, 0, 1, 0], [0, 0, 0, 0, 0, 0, 1]], dtype=float)
        self.H = np.array([[1, 0, 0, 0, 0, 0, 0], [0, 1, 0, 0, 0, 0, 0], [0, 0, 1, 0, 0, 0, 0], [0, 0, 0, 1, 0, 0, 0]], dtype=float)
        self.R = np.eye(4)
        self._I = np.eye(7)
        self.R[2:, 2:] *= 10.0
        self.P[4:, 4:] *= 1000.0
        self.P *= 10.0
        self.Q[-1, -1] *= 0.01
        self.Q[4:, 4:] *= 0.01
        self.history_obs = []
        self.observed = False
        self.saved = None

    def predict(self):
        self.x = np.dot(self.F, self.x)
        self.P = 1.0 * np.dot(np.dot(self.F, self.P), self.F.T) + self.Q

    def _freeze(self):
        self.saved = (self.x.copy(), self.P.copy(), list(self.history_obs), self.observed)

    def _unfreeze(self):
        if self.saved is None:
            return
        new_history = list(self.history_obs)
        self.x, self.P, hist, self.observed = self.saved[0].copy(), self.saved[1].copy(), list(self.saved[2]), self.saved[3]
        self.history_obs = hist[:-1]
        idx = [i for i, d in enumerate(new_history) if d is not None]
        i1, i2 = idx[-2], idx[-1]
        x1, y1, s1, r1 = new_history[i1].reshape(-1)
        w1, h1 = np.sqrt(s1 * r1), np.sqrt(s1 / r1)
        x2, y2, s2, r2 = new_history[i2].reshape(-1)
        w2, h2 = np.sqrt(s2 * r2), np.sqrt(s2 / r2)
        gap = i2 - i1
        dx, dy, dw, dh = (x2 - x1) / gap, (y2 - y1) / gap, (w2 - w1) / gap, (h2 - h1) / gap
        for i in range(gap):
            x, y = x1 + (i + 1) * dx, y1 + (i + 1) * dy
            w, h = w1 + (i + 1) * dw, h1 + (i + 1) * dh
            self.update(np.array([x, y, w * h, w / float(h)]).reshape((4, 1)))
            if i != gap - 1:
                self.predict()

    def update(self, z):
        self.history_obs.append(z)
        if z is None:
            if self.observed:
                self._freeze()
            self.observed = False
            return
        if not self.observed:
            self._unfreeze()
        self.observed = True
        y = z - np.dot(self.H, self.x)
        PHT = np.dot(self.P, self.H.T)
        S = np.dot(self.H, PHT) + self.R
        SI = np.linalg.inv(S)
        K = np.dot(PHT, SI)
        self.x = self.x + np.dot(K, y)
        I_KH = self._I - np.dot(K, self.H)
        self.P = np.dot(np.dot(I_KH, self.P), I_KH.T) + np.dot(np.dot(K, self.R), K.T)


class KalmanBoxTracker:
    def __init__(self, bbox, cls, tid, delta_t=3):
        self.kf = KalmanFilter7()
        self.kf.x[:4] = convert_bbox_to_z(bbox)
        self.time_since_update = 0
        self.id = tid
        self.hits = 0
        self.hit_streak = 0
        self.age = 0
        self.conf = bbox[-1]
        self.cls = cls
        self.last_observation = np.array([-1, -1, -1, -1, -1])
        self.observations = {}
        self.velocity = None
        self.delta_t = delta_t

    def update(self, bbox, cls):
        if bbox is not None:
            self.conf = bbox[-1]
            self.cls = cls
            if self.last_observation.sum() >= 0:
                prev = None
                for i in range(self.delta_t):
                    dt = self.delta_t - i
                    if self.age - dt in self.observations:
                        prev = self.observations[self.age - dt]
                        break
                if prev is None:
                    prev = self.last_observation
                self.velocity = speed_direction(prev, bbox)
            self.last_observation = bbox
            self.observations[self.age] = bbox
            self.time_since_update = 0
            self.hits += 1
            self.hit_streak += 1
            self.kf.update(convert_bbox_to_z(bbox))
        else:
            self.kf.update(None)

    def predict(self):
        if (self.kf.x[6] + self.kf.x[2]) <= 0:
            self.kf.x[6] *= 0.0
        self.kf.predict()
        self.age += 1
        if self.time_since_update > 0:
            self.hit_streak = 0
        self.time_since_update += 1
        return convert_x_to_bbox(self.kf.x)

    def get_state(self):
        return convert_x_to_bbox(self.kf.x)


def associate(dets, trks, iou_threshold, velocities, previous_obs, vdc_weight):
    if len(trks) == 0:
        return np.empty((0, 2), dtype=int), np.arange(len(dets)), np.empty((0,), dtype=int)
    # velocity-direction consistency (OCM)
    cx1, cy1 = (dets[:, 0] + dets[:, 2]) / 2.0, (dets[:, 1] + dets[:, 3]) / 2.0
    cx2, cy2 = (previous_obs[:, 0] + previous_obs[:, 2]) / 2.0, (previous_obs[:, 1] + previous_obs[:, 3]) / 2.0
    dx = cx1[None, :] - cx2[:, None]
    dy = cy1[None, :] - cy2[:, None]
    norm = np.sqrt(dx ** 2 + dy ** 2) + 1e-6
    X, Y = dx / norm, dy / norm                       # [trk, det]
    inertia_Y, inertia_X = velocities[:, 0][:, None], velocities[:, 1][:, None]
    diff_angle_cos = np.clip(inertia_X * X + inertia_Y * Y, -1, 1)
    diff_angle = (np.pi / 2.0 - np.abs(np.arccos(diff_angle_cos))) / np.pi
    valid = np.ones(previous_obs.shape[0])
    valid[np.where(previous_obs[:, 4] < 0)] = 0
    iou_matrix = iou_batch(dets, trks)               # [det, trk]
    # OC-SORT weights the direction term by the detection SCORE (column 4 of [x1,y1,x2,y2,score,cls]; the
    # original takes `detections[:, -1]` of 5-column detections)
    scores = np.repeat(dets[:, 4][:, None], trks.shape[0], axis=1)
    angle_cost = ((valid[:, None] * diff_angle) * vdc_weight).T * scores
    if min(iou_matrix.shape) > 0:
        a = (iou_matrix > iou_threshold).astype(np.int32)
        if a.sum(1).max() == 1 and a.sum(0).max() == 1:
            matched = np.stack(np.where(a), axis=1)
        else:
            matched = linear_assignment(-(iou_matrix + angle_cost))
    else:
        matched = np.empty((0, 2), dtype=int)
    um_d = [d for d in range(len(dets)) if d not in matched[:, 0]]
    um_t = [t for t in range(len(trks)) if t not in matched[:, 1]]
    matches = []
    for m in matched:
        if iou_matrix[m[0], m[1]] < iou_threshold:
            um_d.append(m[0])
            um_t.append(m[1])
        else:
            matches.append(m.reshape(1, 2))
    matches = np.concatenate(matches, axis=0) if matches else np.empty((0, 2), dtype=int)
    return matches, np.array(um_d, dtype=int), np.array(um_t, dtype=int)


class OCSort:
    def __init__(self, det_thresh=0.2, max_age=30, min_hits=3, iou_threshold=0.3, delta_t=3, asso_func="iou", inertia=0.2):
        self.max_age, self.min_hits, self.iou_threshold = max_age, min_hits, iou_threshold
        self.trackers = []
        self.frame_count = 0
        self.det_thresh, self.delta_t, self.inertia = det_thresh, delta_t, inertia
        self.asso_func = ASSO[asso_func]
        self._count = 0                                  # KalmanBoxTracker.count, reset per tracker object

    def update(self, dets, _=None):
        self.frame_count += 1
        dets = np.asarray(dets, dtype=np.float64).reshape(-1, 6)
        dets = dets[dets[:, 4] > self.det_thresh]
        trks = np.zeros((len(self.trackers), 5))
        to_del, ret = [], []
        for t in range(len(trks)):
            pos = self.trackers[t].predict()[0]
            trks[t, :] = [pos[0], pos[1], pos[2], pos[3], 0]
            if np.any(np.isnan(pos)):
                to_del.append(t)
        trks = trks[~np.isnan(trks).any(axis=1)]
        for t in reversed(to_del):
            self.trackers.pop(t)
        velocities = np.array([trk.velocity if trk.velocity is not None else np.array((0, 0)) for trk in self.trackers]).reshape(-1, 2)
        last_boxes = np.array([trk.last_observation for trk in self.trackers]).reshape(-1, 5)
        k_obs = np.array([k_previous_obs(trk.observations, trk.age, self.delta_t) for trk in self.trackers]).reshape(-1, 5)
        matched, um_d, um_t = associate(dets, trks, self.iou_threshold, velocities, k_obs, self.inertia)
        for m in matched:
            self.trackers[m[1]].update(dets[m[0], :5], dets[m[0], 5])
        # observation-centric recovery on the last observations
        if um_d.shape[0] > 0 and um_t.shape[0] > 0:
            left_dets, left_trks = dets[um_d], last_boxes[um_t]
            iou_left = np.array(self.asso_func(left_dets, left_trks))
            if iou_left.max() > self.iou_threshold:
                rem = linear_assignment(-iou_left)
                rd, rt = [], []
                for m in rem:
                    di, ti = um_d[m[0]], um_t[m[1]]
                    if iou_left[m[0], m[1]] < self.iou_threshold:
                        continue
                    self.trackers[ti].update(dets[di, :5], dets[di, 5])
                    rd.append(di)
                    rt.append(ti)
                um_d = np.setdiff1d(um_d, np.array(rd, dtype=int))
                um_t = np.setdiff1d(um_t, np.array(rt, dtype=int))
        for t in um_t:
            self.trackers[t].update(None, None)
        for i in um_d:
            self.trackers.append(KalmanBoxTracker(dets[i, :5], dets[i, 5], self._count, delta_t=self.delta_t))
            self._count += 1
        i = len(self.trackers)
        for trk in reversed(self.trackers):
            d = trk.get_state()[0] if trk.last_observation.sum() < 0 else trk.last_observation[:4]
            if trk.time_since_update < 1 and (trk.hit_streak >= self.min_hits or self.frame_count <= self.min_hits):
                ret.append(np.concatenate((d, [trk.id + 1], [trk.cls], [trk.conf])).reshape(1, -1))
            i -= 1
            if trk.time_since_update > self.max_age:
                self.trackers.pop(i)
        return np.concatenate(ret) if ret else np.empty((0, 7))


def track_boxes(frames_dets, times, **kw):
    """Replay of reference track.py:159-234 on precomputed detections.
    frames_dets: list of [N_i,6] arrays [x1,y1,x2,y2,score,cls]; frames with N_i == 0 are skipped
    entirely (reference track.py:180-181).  Returns the 8-column dict of reference track.py:144-145."""
    kw.setdefault("max_age", 30)
    kw.setdefault("asso_func", "diou")
    kw.setdefault("iou_threshold", 0.1)
    trk = OCSort(**kw)
    data = {k: [] for k in ("id", "time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width")}
    for dets, t in zip(frames_dets, times):
        if len(dets) == 0:
            continue
        out = trk.update(np.asarray(dets, dtype=np.float64), [])
        for res in out:
            xmin, ymin, xmax, ymax, tid, _, score = res
            tid = int(tid)
            kf = None
            for tk in trk.trackers:
                if tk.id == tid - 1:
                    kf = tk.kf
                    break
            dx, dy = kf.x.flatten()[4:6]
            data["id"].append(tid)
            data["time"].append(t)
            data["x"].append((xmin + xmax) / 2)
            data["y"].append((ymin + ymax) / 2)
            data["dx"].append(dx)
            data["dy"].append(dy)
            data["norm_plate_height"].append(abs(ymax - ymin))
            data["norm_plate_width"].append(abs(xmax - xmin))
    return data
