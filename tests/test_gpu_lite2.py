"""BASELINE config 4: EfficientDet-Lite2 448x448 + OC-SORT association, 1x MI355X.
The Lite2 container is generated on the fly (tools/make_model.py, seeded) into a temp dir and read by both
the oracle and the HIP library, so nothing large is committed; the kernels and the planner are generic in
(image size, widths, depths, BiFPN width/cells)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lite2_model(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("models") / "efficientdet_lite2_synth.vbtm")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_model.py"), "--arch", "2", "--out", out, "--calib", "4"])
    return out


def test_lite2_detector_and_tracker_parity(oracle_lib, lite2_model):
    import torch
    from oracle import ocsort_np
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    n, T, S = 2, 5, 448
    frames = np.stack([np.stack([synth.render(synth.background(70 + c, S), 6 * c + t) for c in range(n)]) for t in range(T)])
    det = oracle_lib.OracleDetector(lite2_model)
    assert det.size == 448
    pipe = Pipeline(lite2_model, n, max_frames=T, fps=30.0)
    assert tuple(pipe.interpreter.get_input_details()[0]["shape"]) == (1, 448, 448, 3)
    fd = torch.from_numpy(frames).to("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    got = []
    for t in range(T):
        pipe.step(fd[t].data_ptr(), st)
        got.append(pipe.detections())
    ob, os_, oc, on = oracle_lib.run_batch(lite2_model, frames.reshape(-1, S, S, 3), threads=8)
    ob, os_, on = ob.reshape(T, n, 25, 4), os_.reshape(T, n, 25), on.reshape(T, n)
    for t in range(T):
        b, s, c, k = got[t]
        assert np.array_equal(k, on[t]) and np.array_equal(s, os_[t]) and np.array_equal(b, ob[t]), t
    for c in range(n):
        dets = [np.asarray([[ob[t, c, i, 1], ob[t, c, i, 0], ob[t, c, i, 3], ob[t, c, i, 2], os_[t, c, i], 0.0]
                            for i in range(on[t, c]) if os_[t, c, i] >= 0.5], np.float64).reshape(-1, 6) for t in range(T)]
        want = ocsort_np.track_boxes(dets, [(t + 1) / 30.0 for t in range(T)])
        g = pipe.rows(c)
        assert g["id"] == want["id"]
        for k in ("time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width"):
            assert np.array_equal(np.asarray(g[k]), np.asarray(want[k])), (c, k)


@pytest.mark.parametrize("flags", [8, 8 | 2048, 0])
def test_lite1_detector_parity(oracle_lib, tmp_path_factory, flags):
    """EfficientDet-Lite1 (384x384, BiFPN width 88 -> two 64-channel output blocks, 4 cells): every tensor that reaches
    HBM and the detections, bit-exact against the oracle (most-fused, 48-channel chunks, autotuned)."""
    from vbt_amd import synth
    from vbt_amd.interpreter import Interpreter
    global _LITE1
    try:
        _LITE1
    except NameError:
        _LITE1 = str(tmp_path_factory.mktemp("models") / "efficientdet_lite1_synth.vbtm")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "make_model.py"), "--arch", "1", "--out", _LITE1, "--calib", "2"])
    frames = np.stack([synth.render(synth.background(90 + c, 384), 4 * c) for c in range(2)])
    det = oracle_lib.OracleDetector(_LITE1)
    it = Interpreter(_LITE1, max_batch=2, flags=flags)
    boxes, scores, classes, counts = it.detect(frames)
    for b in range(2):
        ob, os_, oc, on = det.run(frames[b])
        assert counts[b] == on and np.array_equal(scores[b], os_) and np.array_equal(boxes[b], ob)
        for tid in range(1, it.num_tensors() - 1):
            if it.materialized(tid):
                assert np.array_equal(it.read_tensor(tid, 2)[b], det.tensor(tid)), (flags, tid)


def test_lite2_batch64_equals_oracle_with_the_pinned_plan(oracle_lib):
    """BASELINE config 4 at its bench shape: 64 frames of 448x448 in one forward of the committed Lite2 container under the
    plan bench.py pins (profiles/plan_lite2.b64.f0: row-band expand + depthwise on b11-b19, row-band BiFPN nodes / head layers
    of width 112) - boxes, scores and counts of every frame equal the oracle's, and a second pass through the depth-3
    pipeline with OC-SORT gives the oracle chain's rows for four of the clips.  16 steps of 64 frames (1024 Lite2 frames on the oracle: ~20 s on 16 cores)."""
    import torch
    from oracle import ocsort_np
    from vbt_amd import synth
    from vbt_amd.track import Pipeline
    model = os.path.join(ROOT, "models", "efficientdet_lite2_synth.vbtm")
    old = os.environ.get("VBT_PLAN_FILE")
    os.environ["VBT_PLAN_FILE"] = os.path.join(ROOT, "profiles", "plan_lite2")
    try:
        n, T, S = 64, 16, 448                      # 16 steps: the tracker is past min_hits on every clip, the ring of the pipeline wraps five times
        frames = np.stack([np.stack([synth.render(synth.background(300 + c, S), 5 * c + 2 * t) for c in range(n)]) for t in range(T)])
        pipe = Pipeline(model, n, max_frames=T, fps=30.0, rows_per_frame=25)
    finally:
        if old is None:
            os.environ.pop("VBT_PLAN_FILE", None)
        else:
            os.environ["VBT_PLAN_FILE"] = old
    fd = torch.from_numpy(frames).cuda()
    got = []
    for t in range(T):
        pipe.step(fd[t])
        got.append(pipe.detections())
    pipe.finish()
    ob, os_, oc, on = oracle_lib.run_batch(model, frames.reshape(-1, S, S, 3), threads=16)
    ob, os_, on = ob.reshape(T, n, 25, 4), os_.reshape(T, n, 25), on.reshape(T, n)
    for t in range(T):
        b, s, c, k = got[t]
        assert np.array_equal(k, on[t]) and np.array_equal(s, os_[t]) and np.array_equal(b, ob[t]), t
    assert int(on.sum()) > 64
    for c in (0, 17, 40, 63):
        dets = [np.asarray([[ob[t, c, i, 1], ob[t, c, i, 0], ob[t, c, i, 3], ob[t, c, i, 2], os_[t, c, i], 0.0]
                            for i in range(on[t, c]) if os_[t, c, i] >= 0.5], np.float64).reshape(-1, 6) for t in range(T)]
        want = ocsort_np.track_boxes(dets, [(t + 1) / 30.0 for t in range(T)])
        g = pipe.rows(c)
        assert g["id"] == want["id"]
        for k in ("time", "x", "y", "dx", "dy", "norm_plate_height", "norm_plate_width"):
            assert np.array_equal(np.asarray(g[k]), np.asarray(want[k])), (c, k)
