"""The tracker oracle against EVERY reference DataFrame (reference dfs_ocsort/*.pkl.gz, all 34 clips, all ids ->
tests/golden/dfs_ocsort_all.npz, written by tools/make_golden_corpus.py).  `ocsort` itself is an unpinned third-party
package (reference track.py:17,157), so these committed outputs of reference track.py:103-126 are the only evidence the
reference holds about its id numbering, emission order, min_hits rule and Kalman recipe.

What a replay can and cannot reproduce.  The DataFrame stores a track's observation only on frames where the track was
EMITTED (reference track.py:189-234), and OC-SORT hides a track until its third consecutive hit (min_hits = 3) unless
frame_count <= 3.  Feeding the stored boxes back therefore lacks (a) the two hidden observations after every miss and
(b) the two hidden observations of every track born after frame 3 (and whole tracks that never reached three hits, which
still consumed an id: clip 012 has ids 1-4, 6-11, 13).  So:
  * 22 clips hold ONE track that was never lost: the replay must reproduce the DataFrame completely - row count, ids,
    emission order (= retained index), time, x, y, dx, dy bit for bit; h, w bit for bit except the birth row, which the
    reference emits from the filter state (sqrt(s*r), not the float32 detection);
  * 12 clips hold several tracks / losses: every track visible from frame <= 3 must be bit-exact up to its first loss and
    stay within 1.2e-4 afterwards (two ids named below need more: their hidden observations moved the velocity estimate
    further); every visible loss of d stepped frames in the reference is a loss of d + 2 in the replay, so the replay
    has about 2 rows less per loss - checked as a bound.
"""
import json
import os
from collections import Counter

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import ocsort_np as oc
from test_oracle_ocsort import frames_from_rows

KEYS = ("id", "time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width", "index")
EXACT = ("time", "x", "y", "dx", "dy")
# (clip, id) -> bound on |dx|,|dy| differences after the track's first loss (default 1.2e-4)
LOOSE_AFTER_GAP = {("029", 3): 2.5e-3, ("032", 2): 2.0e-4}


def load_all():
    a = np.load(os.path.join(GOLDEN, "dfs_ocsort_all.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "corpus_meta.json")))
    return {clip: ({k: a[f"c{clip}_{k}"] for k in KEYS}, int(a[f"c{clip}_export_id"]), meta[clip][1]) for clip in sorted(meta)}


@pytest.fixture(scope="module")
def corpus():
    return load_all()


@pytest.fixture(scope="module")
def replays(corpus):
    out = {}
    for clip, (g, _, _) in corpus.items():
        frames, times = frames_from_rows(g)
        out[clip] = {k: np.asarray(v) for k, v in oc.track_boxes(frames, times).items()}
    return out


def stepped_gaps(rows):
    """Histogram of a track's losses, in frames on which the tracker was stepped (= distinct time stamps of the clip)."""
    ut = np.unique(rows["time"])
    pos = {t: i for i, t in enumerate(ut)}
    h = Counter()
    for tid in np.unique(rows["id"]):
        d = np.diff([pos[t] for t in rows["time"][rows["id"] == tid]])
        h.update(int(v) for v in d[d > 1])
    return h


def emission_order_ok(ids, times):
    """Rows in emission order: time never decreases; rows of one frame come in DESCENDING id order (OC-SORT walks its
    tracker list backwards when it collects the output, reference track.py:189-190 then appends in that order)."""
    if np.any(np.diff(times) < 0):
        return False
    same = np.diff(times) == 0
    return bool(np.all(np.diff(ids)[same] < 0))


def test_reference_dataframes_structure(corpus):
    """Properties of the fixture itself (they define what the tracker must do): the retained index is the emission order;
    no track ever reappears after exactly 2 stepped frames (min_hits = 3 makes the shortest visible loss 1 miss + 2 hidden
    hits = a step of 4); a step of 2 happens only inside the first 3 frames, where everything is emitted."""
    hist = Counter()
    assert len(corpus) == 34 and sum(len(g["id"]) for g, _, _ in corpus.values()) == 61461
    for clip, (g, export_id, fps) in corpus.items():
        n = len(g["id"])
        assert np.array_equal(np.sort(g["index"]), np.arange(n)), clip               # labels 0..n-1, kept through sort_values
        order = np.argsort(g["index"])
        assert emission_order_ok(g["id"][order].astype(int), g["time"][order]), clip
        assert np.all(np.diff(g["id"].astype(int)) >= 0), clip                         # stored sorted by (id, time): track.py:105
        assert all(np.all(np.diff(g["time"][g["id"] == t]) > 0) for t in np.unique(g["id"])), clip
        assert export_id in g["id"]
        hist.update(stepped_gaps(g))
        for tid in np.unique(g["id"]):                                              # KAT-2: a track's first row has zero velocity
            j = np.flatnonzero(g["id"] == tid)[0]
            if round(g["time"][j] * fps) <= 3:
                assert g["dx"][j] == 0.0 and g["dy"][j] == 0.0, (clip, tid)
    assert hist[3] == 0 and hist[2] == 2 and hist[4] == 45 and hist[5] == 17 and max(hist, key=hist.get) == 4, sorted(hist.items())[:6]


def single_track_clips(corpus):
    return [c for c, (g, _, _) in corpus.items() if len(np.unique(g["id"])) == 1 and not stepped_gaps(g)]


def test_single_track_clips_reproduce_the_dataframe_completely(corpus, replays):
    clips = single_track_clips(corpus)
    assert len(clips) == 22
    rows = 0
    for clip in clips:
        g, export_id, _ = corpus[clip]
        o = replays[clip]
        order = np.argsort(g["index"])
        assert len(o["id"]) == len(g["id"]) and np.array_equal(o["id"], g["id"][order]) and set(o["id"]) == {export_id}, clip
        for col in EXACT:
            assert np.array_equal(o[col], g[col][order]), (clip, col)
        for col in ("norm_plate_height", "norm_plate_width"):                        # birth row comes from the filter state
            assert np.array_equal(o[col][1:], g[col][order][1:]) and abs(o[col][0] - g[col][order][0]) < 3e-6, (clip, col)   # the 1e-6 of convert_bbox_to_z, applied twice
        rows += len(o["id"])
    assert rows == 34143


def test_multi_track_clips_pinned_where_the_fixture_allows(corpus, replays):
    clips = [c for c in corpus if c not in single_track_clips(corpus)]
    assert len(clips) == 12
    checked_rows = 0
    for clip in clips:
        g, _, fps = corpus[clip]
        o = replays[clip]
        assert emission_order_ok(o["id"].astype(int), o["time"]), clip
        ut = np.unique(g["time"])
        pos = {t: i for i, t in enumerate(ut)}
        losses = late = 0
        for tid in np.unique(g["id"]):
            m = g["id"] == tid
            t = g["time"][m]
            p = np.array([pos[v] for v in t])
            d = np.diff(p)
            losses += int((d > 1).sum())
            if round(t[0] * fps) > 3:
                late += 1                                                            # born hidden: the replay sees it two hits later, if at all
                continue
            twins = [i for i in np.unique(o["id"]) if o["time"][o["id"] == i][0] == t[0] and o["x"][o["id"] == i][0] == g["x"][m][0]]
            assert twins == [tid], (clip, tid, twins)                                # same id numbering for the tracks born on frames 1-3
            om = o["id"] == tid
            pre = len(t) if not np.any(d > 1) else int(np.flatnonzero(d > 1)[0]) + 1
            assert om.sum() >= pre
            for col in EXACT:                                                        # bit-exact from birth to the first loss
                assert np.array_equal(g[col][m][:pre], o[col][om][:pre]), (clip, tid, col)
            checked_rows += pre
            ref = dict(zip(t, zip(g["dx"][m], g["dy"][m])))
            got = dict(zip(o["time"][om], zip(o["dx"][om], o["dy"][om])))
            common = sorted(set(ref) & set(got))
            assert len(common) >= 0.65 * len(t), (clip, tid)
            err = max(max(abs(ref[k][0] - got[k][0]), abs(ref[k][1] - got[k][1])) for k in common)
            assert err < LOOSE_AFTER_GAP.get((clip, tid), 1.2e-4), (clip, tid, err)
        deficit = len(g["id"]) - len(o["id"])
        assert 0 <= deficit <= 2 * losses + 4 * late + 2, (clip, deficit, losses, late)
        assert deficit >= 2 * losses - 6 or late, (clip, deficit, losses)            # every loss really costs the replay ~2 rows
        h = stepped_gaps(o)
        assert h[3] == 0 and h[5] == 0, (clip, sorted(h.items())[:6])                # d -> d + 2: the reference's 4 / 5 / ... arrive as 6 / 7 / ...
    assert checked_rows > 12000
