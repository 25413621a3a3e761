"""`.tflite` importer (reference track.py:67,93 loads models/*.tflite, all absent from the
tree).  Files come from tools/export_tflite.py, which writes a container in the converter's layout; the importer and
the exporter only share the flatbuffer codec, so a file that round-trips exercises the schema field numbers twice
(once writing, once reading) plus the graph pattern matching."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from vbt_amd import spec  # noqa: E402
from vbt_amd.container import Container  # noqa: E402
from vbt_amd.flatbuf import flex_map, flex_root  # noqa: E402
from vbt_amd.tflite_import import (BO_NAMES, TfModel, UnsupportedModel, convert, import_tflite, is_tflite)  # noqa: E402
import export_tflite  # noqa: E402

MODEL = os.path.join(ROOT, "models", "efficientdet_lite0_synth.vbtm")


@pytest.fixture(scope="module")
def paths(tmp_path_factory):
    d = tmp_path_factory.mktemp("tfl")
    p = {k: str(d / k) for k in ("a.tflite", "a.vbtm", "b.tflite", "b.vbtm")}
    export_tflite.export(MODEL, p["a.tflite"])
    convert(p["a.tflite"], p["a.vbtm"])
    export_tflite.export(p["a.vbtm"], p["b.tflite"])
    convert(p["b.tflite"], p["b.vbtm"])
    return p


def test_flexbuffer_map_roundtrip():
    d = {"max_detections": 25, "nms_iou_threshold": 0.5, "use_regular_nms": False, "x_scale": 10.0, "a": -7}
    back = flex_root(flex_map(d))
    assert set(back) == set(d)
    for k, v in d.items():
        assert type(back[k]) is type(v) and back[k] == pytest.approx(v)


def test_file_structure(paths):
    assert is_tflite(paths["a.tflite"]) and not is_tflite(MODEL)
    m = TfModel(paths["a.tflite"])
    names = [o.name for o in m.ops]
    assert names[0] == "QUANTIZE" and names[1] == "CONV_2D" and names[-1] == "TFLite_Detection_PostProcess"
    count = {n: names.count(n) for n in set(names)}
    g = spec.build_graph(0)
    three = sum(1 for o in g.ops if o.type == spec.OP_ADD and len(o.inputs) == 3)
    assert count["DEPTHWISE_CONV_2D"] == sum(o.type == spec.OP_DW for o in g.ops)
    assert count["CONV_2D"] == sum(o.type in (spec.OP_PW, spec.OP_STEM) for o in g.ops)
    assert count["ADD"] == sum(o.type == spec.OP_ADD for o in g.ops) + three      # chained binary adds
    assert count["RESHAPE"] == 10 and count["CONCATENATION"] == 2 and count["LOGISTIC"] == 1 and count["DEQUANTIZE"] == 2
    t = m.tensors[m.inputs[0]]
    assert t.shape == (1, 320, 320, 3) and t.q() == (np.float32(1 / 128), 127)
    w = m.tensors[m.ops[1].inputs[1]]
    assert w.shape == (32, 3, 3, 3) and w.scale.size == 32 and w.data.dtype == np.int8


def test_import_matches_source_container(paths):
    """Same ops (3-input sums become two binary ADDs), identical int8 weights / int32 biases / zero points / clamps;
    float32 multipliers within two ulps (s_w is not stored in the container, so the exporter re-derives it)."""
    A, B = Container(MODEL), Container(paths["a.vbtm"])
    assert int(B.header["image_size"]) == 320 and int(B.header["num_anchors"]) == 19206
    assert float(B.header["nms_score_threshold"]) == float(A.header["nms_score_threshold"])
    ib = 0
    for ra in A.ops:
        ta = int(ra["type"])
        if ta == spec.OP_ADD and int(ra["n_inputs"]) == 3:
            assert int(B.ops[ib]["type"]) == spec.OP_ADD and int(B.ops[ib]["n_inputs"]) == 2
            ib += 1
        rb = B.ops[ib]
        ib += 1
        assert int(rb["type"]) == ta
        for f in ("k", "stride", "pad_t", "pad_l", "act_min", "act_max", "level"):
            assert int(ra[f]) == int(rb[f]), (f, ib)
        oa, ob = A.tensors[int(ra["output"])], B.tensors[int(rb["output"])]
        assert (oa["h"], oa["w"], oa["c"], oa["zero_point"], oa["scale"]) == (ob["h"], ob["w"], ob["c"], ob["zero_point"], ob["scale"])
        if ta in (spec.OP_STEM, spec.OP_PW, spec.OP_DW):
            cout = int(oa["c"])
            cin = int(A.tensors[int(ra["inputs"][0])]["c"])
            nw = {spec.OP_STEM: cout * 27, spec.OP_PW: cout * cin, spec.OP_DW: cout * int(ra["k"]) ** 2}[ta]
            assert np.array_equal(A.i8(int(ra["w_off"]), nw), B.i8(int(rb["w_off"]), nw))
            assert np.array_equal(A.i32(int(ra["b_off"]), cout), B.i32(int(rb["b_off"]), cout))
            ma, mb = A.f32(int(ra["m_off"]), cout), B.f32(int(rb["m_off"]), cout)
            assert np.abs(ma.view(np.int32).astype(np.int64) - mb.view(np.int32)).max() <= 2
        if ta == spec.OP_POSTPROCESS:
            n = int(A.header["num_anchors"])
            assert np.array_equal(A.f32(int(ra["aux_off"]), 4 * n), B.f32(int(rb["aux_off"]), 4 * n))
            assert np.array_equal(A.f32(int(ra["aux2_off"]), 768), B.f32(int(rb["aux2_off"]), 768))
    assert ib == len(B.ops)


def test_import_is_a_fixed_point(paths):
    """export -> import of an imported container reproduces it byte for byte."""
    assert open(paths["a.vbtm"], "rb").read() == open(paths["b.vbtm"], "rb").read()


def test_oracle_runs_imported_model(paths):
    from oracle.detector_ref import OracleDetector
    from vbt_amd import synth
    a, b = OracleDetector(MODEL), OracleDetector(paths["a.vbtm"])
    f = synth.clip_frames(0, 0, 1)[0]
    a.run(f)
    boxes, scores, classes, n = b.run(f)
    assert 0 < n <= 25 and np.all(np.diff(scores[:n]) <= 0)
    # the stem has exactly recoverable multipliers for most channels: tensors agree except where a multiplier moved an ulp
    ta, tb = a.tensor(1), b.tensor(1)
    assert np.abs(ta.astype(int) - tb.astype(int)).max() <= 1


@pytest.mark.parametrize("code", [19, 25])   # RELU, SOFTMAX
def test_unsupported_operator_is_named(tmp_path, code):
    p = str(tmp_path / "x.tflite")
    export_tflite.export(MODEL, p, extra_op=(3, code))
    with pytest.raises(UnsupportedModel, match=BO_NAMES[code]):
        import_tflite(p)


def test_half_pixel_resize_is_refused_only_when_it_matters(tmp_path):
    p = str(tmp_path / "x.tflite")
    export_tflite.export(MODEL, p, half_pixel=True)
    with pytest.raises(UnsupportedModel, match="half_pixel"):       # the 3 -> 5 up-sampling is not an integral factor
        import_tflite(p)


def test_not_a_flatbuffer(tmp_path):
    p = tmp_path / "x.tflite"
    p.write_bytes(b"\0" * 64)
    with pytest.raises(UnsupportedModel, match="TFL3"):
        TfModel(str(p))
    from vbt_amd.tflite_import import as_container_path
    assert as_container_path(MODEL) == (MODEL, False)
