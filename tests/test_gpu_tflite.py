"""GPU parity for a model that arrives as a `.tflite` flatbuffer (reference track.py:67,93): the HIP path on the imported
graph (binary ADD chains, importer-derived multipliers and LUTs) against the oracle on the same imported container."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


@pytest.fixture(scope="module")
def imported(tmp_path_factory, model_path):
    import export_tflite
    from vbt_amd.tflite_import import convert
    d = tmp_path_factory.mktemp("tfl")
    tfl, vbtm = str(d / "lite0.tflite"), str(d / "lite0.vbtm")
    export_tflite.export(model_path, tfl)
    convert(tfl, vbtm)
    return tfl, vbtm


@pytest.mark.parametrize("flags", [1, 8, 0])
def test_tflite_model_bit_exact(imported, oracle_lib, flags):
    from vbt_amd import synth
    from vbt_amd.interpreter import Interpreter
    tfl, vbtm = imported
    frames = np.concatenate([synth.clip_frames(s, 5 * s, 2) for s in range(2)])
    det = oracle_lib.OracleDetector(vbtm)
    it = Interpreter(model_path=tfl, max_batch=len(frames), flags=flags)       # the .tflite itself
    assert it.num_tensors() == det.num_tensors
    boxes, scores, classes, counts = it.detect(frames)
    checked = 0
    for b, f in enumerate(frames):
        ob, os_, oc, on = det.run(f)
        assert counts[b] == on and np.array_equal(scores[b], os_) and np.array_equal(boxes[b], ob)
        for tid in range(1, it.num_tensors() - 1):
            if it.materialized(tid):
                checked += 1
                assert np.array_equal(it.read_tensor(tid, len(frames))[b], det.tensor(tid)), (tid, b)
    if flags == 1:
        assert checked == len(frames) * (it.num_tensors() - 2)
    else:
        assert it.num_launches() < 120


def test_cli_accepts_tflite(imported, tmp_path):
    """`track --model x.tflite` end to end (reference track.py:67)."""
    from click.testing import CliRunner
    from vbt_amd import synth
    from vbt_amd.cli import main
    tfl, _ = imported
    clip = tmp_path / "clip.npy"
    np.save(str(clip), synth.clip_frames(3, 0, 12))
    out = tmp_path / "dfs"
    res = CliRunner().invoke(main, ["track", str(clip), "--model", tfl, "--df_dir", str(out), "--fps", "60", "--detection_treshold", "0.3"])
    assert res.exit_code == 0, res.output
    assert "rows" in res.output
    files = os.listdir(out)
    assert len(files) <= 1 and all(f.endswith("_lite0.pkl.gz") for f in files)


def test_foreign_encoded_tiny_model_hip_equals_oracle(tmp_path, oracle_lib):
    """A .tflite written by the independent encoder of tests/tflite_minienc.py (32x32 image, 8-channel graph, shared head
    weights, non-unit box scales, score plateaus in the LOGISTIC table): imported, then HIP == oracle for every tensor and
    every detection in three plan modes."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tflite_minienc import TinyModel
    from vbt_amd.interpreter import Interpreter
    m = TinyModel(S=32, seed=5).build()
    path = str(tmp_path / "tiny.tflite")
    open(path, "wb").write(m.serialize())
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 256, (5, 32, 32, 3), dtype=np.uint8)
    frames[1] = 0
    frames[2] = 255
    for flags in (1, 8, 0):
        it = Interpreter(model_path=path, max_batch=5, flags=flags)
        assert tuple(it.get_input_details()[0]["shape"]) == (1, 32, 32, 3)
        boxes, scores, classes, counts = it.detect(frames)
        from vbt_amd.tflite_import import convert
        vb = str(tmp_path / f"tiny{flags}.vbtm")
        convert(path, vb)
        det = oracle_lib.OracleDetector(vb)
        for b in range(5):
            ob, os_, oc, on = det.run(frames[b])
            assert counts[b] == on and np.array_equal(scores[b], os_) and np.array_equal(boxes[b], ob) and np.array_equal(classes[b], oc), (flags, b)
            if b == 0:
                assert 0 < oc[:on].sum() < on      # two class columns per anchor (vbt_amd/spec.py): both win somewhere on this frame
            for tid in range(1, it.num_tensors() - 1):
                if it.materialized(tid):
                    assert np.array_equal(it.read_tensor(tid, 5)[b], det.tensor(tid)), (flags, tid, b)


def test_converter_style_full_size_file_at_batch_64(tmp_path, model_path, oracle_lib):
    """a FULL-SIZE Lite0 file in the conventions of a converter-written one (tests/tflite_fullenc.py: independent
    encoder, constants ahead of reversed activations, operators in a random topological order, per-channel quantized_dimension, fused
    activations, FlexBuffers post-process options) goes through `Interpreter(model_path=x.tflite)` at the bench's batch of 64: every
    detection equals the oracle's on the imported container, every materialised tensor of two frames too, and the plan is as fused as
    the native container's (the importer untangles the operator order)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from tflite_fullenc import ConverterStyleModel
    from vbt_amd import synth
    from vbt_amd.interpreter import Interpreter
    from vbt_amd.tflite_import import convert
    m = ConverterStyleModel(model_path, order=7).build()
    tfl, vb = str(tmp_path / "conv_style.tflite"), str(tmp_path / "conv_style.vbtm")
    open(tfl, "wb").write(m.serialize())
    convert(tfl, vb)
    B = 64
    frames = np.concatenate([synth.clip_frames(s, 2 * s, 4) for s in range(B // 4)])
    it = Interpreter(model_path=tfl, max_batch=B, flags=8)
    native = Interpreter(model_path=model_path, max_batch=B, flags=8)
    assert it.num_launches() <= native.num_launches() + 2
    boxes, scores, classes, counts = it.detect(frames)
    ob, os_, oc, on = oracle_lib.run_batch(vb, frames, threads=8)
    assert np.array_equal(counts, on) and np.array_equal(scores, os_) and np.array_equal(boxes, ob)
    nb, ns, nc, nn = native.detect(frames)                  # same weights, same quantisation: the native container detects the same
    assert np.array_equal(counts, nn) and np.array_equal(scores, ns) and np.array_equal(boxes, nb)
    det = oracle_lib.OracleDetector(vb)
    for b in (0, B - 1):
        det.run(frames[b])
        for tid in range(1, it.num_tensors() - 1):
            if it.materialized(tid):
                assert np.array_equal(it.read_tensor(tid, B)[b], det.tensor(tid)), (tid, b)
