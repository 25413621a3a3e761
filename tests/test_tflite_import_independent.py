"""N1 hardening: the importer on bytes that tools/export_tflite.py did NOT write.  tests/tflite_minienc.py is a second,
independently written encoder (back-to-front FlatBuffers builder, FlexBuffers map writer, schema.fbs field ids spelled out).
Covered: per-channel `quantized_dimension` (0 for CONV_2D, 3 for DEPTHWISE_CONV_2D), UINT8 input + QUANTIZE, buffers shared
by several tensors, vtable de-duplication, the FlexBuffers custom-options map of TFLite_Detection_PostProcess with non-unit
box scales - and the imported graph evaluated by the CPU oracle against a from-scratch numpy evaluation of the same model."""
import math

import numpy as np
import pytest

from tflite_minienc import TinyModel, flexbuffer_map
from test_quant_kat import ref_add, ref_add_params

from vbt_amd import spec
from vbt_amd.container import Container
from vbt_amd.flatbuf import flex_root
from vbt_amd.tflite_import import TfModel, convert, is_tflite

from conftest import MODEL_LITE0 as MODEL


@pytest.fixture(scope="module")
def tiny(tmp_path_factory):
    d = tmp_path_factory.mktemp("tiny")
    m = TinyModel(S=32, seed=5).build()
    path = str(d / "tiny.tflite")
    open(path, "wb").write(m.serialize())
    out = str(d / "tiny.vbtm")
    convert(path, out)
    return m, path, out


def test_flexbuffer_options_decoded_from_foreign_bytes():
    opts = {"max_detections": 25, "nms_iou_threshold": 0.45, "use_regular_nms": False, "y_scale": 10.0, "h_scale": 5.0, "num_classes": 1,
            "a_rather_long_key_name_to_push_the_root_offset_past_one_byte_" * 4: 3}
    back = flex_root(flexbuffer_map(opts))
    assert set(back) == set(opts)
    for k, v in opts.items():
        assert type(back[k]) is type(v)
        assert back[k] == (np.float32(v) if isinstance(v, float) else v)


def test_file_is_parsed_by_schema_field_ids(tiny):
    m, path, _ = tiny
    assert is_tflite(path)
    t = TfModel(path)
    assert t.version == 3 and [o.name for o in t.ops][:5] == ["QUANTIZE", "CONV_2D", "DEPTHWISE_CONV_2D", "CONV_2D", "ADD"]
    assert t.ops[-1].name == "TFLite_Detection_PostProcess" and len(t.tensors) == len(m.tensors)
    for a, b in zip(t.tensors, m.tensors):
        assert a.name == b["name"] and list(a.shape) == b["shape"] and a.type == b["type"] and a.qdim == b["qdim"]
        if b["scale"] is not None:
            assert np.array_equal(a.scale, np.asarray(b["scale"], np.float32)) and np.array_equal(a.zero_point, np.asarray(b["zp"], np.int64))
    dw = next(x for x in t.tensors if x.name == "dw/w")
    assert dw.qdim == 3 and dw.scale.size == 8 and dw.shape == (1, 3, 3, 8)
    cls_w = [x for x in t.tensors if x.name.startswith("class") and x.name.endswith("/w")]
    assert len(cls_w) == 5 and len({x.buffer for x in cls_w}) == 1                      # one buffer shared by five tensors


def test_container_fields_against_independent_arithmetic(tiny):
    m, _, out = tiny
    c = Container(out)
    h = c.header
    assert int(h["image_size"]) == 32 and int(h["num_anchors"]) == m.n_anchor == 3069 and int(h["max_detections"]) == 25
    assert np.float32(h["nms_iou_threshold"]) == np.float32(0.45) and np.float32(h["nms_score_threshold"]) == np.float32(0.0625)
    types = [int(r["type"]) for r in c.ops]
    assert types == [spec.OP_STEM, spec.OP_DW, spec.OP_PW, spec.OP_ADD] + [spec.OP_MAXPOOL] * 4 + [spec.OP_PW] * 10 + [spec.OP_POSTPROCESS]
    t0 = c.tensors[0]
    assert (int(t0["h"]), int(t0["c"]), int(t0["zero_point"])) == (32, 3, -1) and np.float32(t0["scale"]) == np.float32(1 / 128)   # QUANTIZE folded
    for r, name in zip(c.ops[:3], ("stem", "dw", "pw")):
        me = m.meta[name]
        cout = me["sw"].size
        want_m = ((me["xs"] * me["sw"]).astype(np.float32) / me["so"]).astype(np.float32)                 # XNNPACK: (s_x * s_w) / s_y in float32
        assert np.array_equal(c.f32(int(r["m_off"]), cout), want_m), name
        assert np.array_equal(c.i32(int(r["b_off"]), cout), me["b"]), name
        w = me["w"]
        want_w = w[0].reshape(-1) if name == "dw" else w.reshape(-1)                                      # [ky][kx][C] / [Cout][ky][kx][Cin]
        assert np.array_equal(c.i8(int(r["w_off"]), want_w.size), want_w), name
    stem, dw = c.ops[0], c.ops[1]
    assert (int(stem["k"]), int(stem["stride"]), int(stem["pad_t"]), int(stem["pad_l"])) == (3, 2, 0, 0)   # SAME on 32 with 3x3/2: pad after only
    assert (int(dw["k"]), int(dw["stride"]), int(dw["pad_t"])) == (3, 1, 1)
    assert (int(stem["act_min"]), int(stem["act_max"])) == (-128, min(127, -128 + round(6 / float(np.float32(0.0235)))))
    add = c.ops[3]
    prm = ref_add_params(float(np.float32(0.031)), float(np.float32(0.0235)), float(np.float32(0.0235)), 4, -128)
    assert tuple(int(v) for v in add["add_q"]) == prm and int(add["n_inputs"]) == 2
    heads = c.ops[8:18]
    assert len({int(r["w_off"]) for r in heads[:5]}) == 1 and len({int(r["w_off"]) for r in heads[5:]}) == 1      # shared weights de-duplicated
    assert [int(r["level"]) for r in heads] == [3, 4, 5, 6, 7, 3, 4, 5, 6, 7]
    post = c.ops[-1]
    assert np.array_equal(c.f32(int(post["aux_off"]), m.n_anchor * 4).reshape(-1, 4), m.anchors)
    tab = c.blob[int(post["aux2_off"]):int(post["aux2_off"]) + 6160]
    box = tab[1024:2048].view("<f4")
    q = np.arange(-128, 128)
    assert np.array_equal(box, (np.float32(0.021) * (q + 7).astype(np.float32)).astype(np.float32))
    assert tab[2048:4096].view("<f8").tolist() == [float(v) / 10.0 for v in box]                           # y_scale = x_scale = 10
    assert tab[4096:6144].view("<f8").tolist() == [math.exp(float(v) / 5.0) for v in box]                  # h_scale = w_scale = 5
    assert tab[6144:6160].view("<f4").tolist() == [10.0, 10.0, 5.0, 5.0]


def _conv(x, w, b, zx, stride, pad_t, pad_l, oh):
    """int conv NHWC single image, w [Cout][k][k][Cin], zero-point padding == zero contribution"""
    k = w.shape[1]
    H = x.shape[0]
    xp = np.zeros((H + k, H + k, x.shape[2]), np.int64)
    xp[pad_t:pad_t + H, pad_l:pad_l + H] = x.astype(np.int64) - zx
    acc = np.zeros((oh, oh, w.shape[0]), np.int64)
    for ky in range(k):
        for kx in range(k):
            acc += xp[ky:ky + stride * oh:stride, kx:kx + stride * oh:stride] @ w[:, ky, kx].astype(np.int64).T
    return acc + b.astype(np.int64)


def _requant(acc, mult, zo, lo, hi):
    t = (acc.astype(np.float32) * mult.astype(np.float32)).astype(np.float32)
    return np.clip(np.rint(t).astype(np.int64) + zo, lo, hi).astype(np.int8)


def test_oracle_on_imported_graph_equals_numpy_evaluation_of_the_source_model(tiny, oracle_lib):
    m, _, out = tiny
    det = oracle_lib.OracleDetector(out)
    assert det.size == 32
    rng = np.random.default_rng(9)
    frame = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    boxes, scores, classes, count = det.run(frame)
    f32 = np.float32
    x = frame.astype(np.int64) - 128                                                     # QUANTIZE uint8 -> int8 (zero point 127 -> -1)
    me = m.meta["stem"]
    relu6 = lambda zo, so: (max(-128, zo), min(127, zo + int(np.rint(6.0 / float(so)))))
    mul = lambda me: ((me["xs"] * me["sw"]).astype(f32) / me["so"]).astype(f32)
    stem = _requant(_conv(x, me["w"], me["b"], -1, 2, 0, 0, 16), mul(me), -128, *relu6(-128, me["so"]))
    me = m.meta["dw"]
    wd = me["w"][0]                                                                      # [k][k][C]
    xp = np.zeros((18, 18, 8), np.int64)
    xp[1:17, 1:17] = stem.astype(np.int64) + 128
    acc = sum(xp[ky:ky + 16, kx:kx + 16] * wd[ky, kx].astype(np.int64) for ky in range(3) for kx in range(3)) + me["b"]
    dw = _requant(acc, mul(me), -3, -128, 127)
    me = m.meta["pw"]
    pw = _requant(_conv(dw, me["w"], me["b"], -3, 1, 0, 0, 16), mul(me), 4, -128, 127)
    prm = ref_add_params(float(f32(0.031)), float(f32(0.0235)), float(f32(0.0235)), 4, -128)
    lo, hi = relu6(-128, f32(0.0235))
    p3 = np.vectorize(lambda a, b: ref_add(int(a), int(b), prm, -128, lo, hi))(pw, stem).astype(np.int8)
    assert np.array_equal(det.tensor(1), stem) and np.array_equal(det.tensor(2), dw) and np.array_equal(det.tensor(3), pw)
    assert np.array_equal(det.tensor(4), p3)
    levels = [p3]
    for i in range(4):
        a = levels[-1]
        H = a.shape[0]
        oh = (H + 1) // 2
        ap = np.full((2 * oh + 2, 2 * oh + 2, 8), -128, np.int64)
        ap[:H, :H] = a                                                                   # SAME padding on even sizes: after only
        levels.append(np.max([ap[ky:ky + 2 * oh:2, kx:kx + 2 * oh:2] for ky in range(3) for kx in range(3)], axis=0).astype(np.int8))
        assert np.array_equal(det.tensor(5 + i), levels[-1])
    cls, box = [], []
    for li, t in enumerate(levels):
        for name, store, zo in (("class", cls, 12), ("box", box, -7)):
            me = m.meta[f"{name}{li}"]
            store.append(_requant(_conv(t, me["w"], me["b"], -128, 1, 0, 0, t.shape[0]), mul(me), zo, -128, 127).reshape(-1))
    cls, box = np.concatenate(cls).reshape(-1, m.num_classes), np.concatenate(box).reshape(-1, 4)
    assert m.num_classes == 2
    # LOGISTIC (float32 table) -> DEQUANTIZE -> decode in double -> fast NMS, all from the source model's numbers
    sig = np.asarray([min(max(f32(256.0) / (f32(1.0) + f32(math.exp(-float(f32(0.09) * f32(int(q) - 12))))), f32(0)), f32(255)) for q in range(-128, 128)], f32)
    score = (np.rint(sig).astype(np.int64)).astype(f32) * f32(1 / 256)
    # two class columns per anchor: the anchor scores with its best column, its class is the first column holding that score
    col_sc = score[cls.astype(int) + 128]                                                 # [A, 2]
    sc = col_sc.max(axis=1)
    best_col = np.argmax(col_sc, axis=1)
    order = sorted([i for i in range(len(cls)) if sc[i] >= f32(0.0625)], key=lambda i: (-sc[i], i))
    bq = (f32(0.021) * (box.astype(np.int64) + 7).astype(f32)).astype(f32)
    sel, sel_cls = [], []
    for i in order:
        an = m.anchors[i].astype(np.float64)
        yc, xc = f32(float(bq[i, 0]) / 10.0 * an[2] + an[0]), f32(float(bq[i, 1]) / 10.0 * an[3] + an[1])
        hh, hw = f32(0.5 * math.exp(float(bq[i, 2]) / 5.0) * an[2]), f32(0.5 * math.exp(float(bq[i, 3]) / 5.0) * an[3])
        b = np.asarray([yc - hh, xc - hw, yc + hh, xc + hw], f32)
        ok = True
        for s, _ in sel:
            ih = max(min(s[2], b[2]) - max(s[0], b[0]), f32(0)); iw = max(min(s[3], b[3]) - max(s[1], b[1]), f32(0))
            inter = f32(ih * iw)
            ua = f32((s[2] - s[0]) * (s[3] - s[1])); ub = f32((b[2] - b[0]) * (b[3] - b[1]))
            if ua > 0 and ub > 0 and f32(inter / f32(f32(ua + ub) - inter)) > f32(0.45):
                ok = False
                break
        if ok:
            sel.append((b, sc[i]))
            sel_cls.append(float(best_col[i]))
        if len(sel) == 25:
            break
    assert count == len(sel) and count > 3
    assert classes[:count].tolist() == sel_cls and 0 < sum(sel_cls) < count               # both columns win somewhere
    assert np.array_equal(scores[:count], np.asarray([s for _, s in sel], f32))
    assert np.array_equal(boxes[:count], np.stack([b for b, _ in sel]))


# ---- a FULL-SIZE file in the conventions of a converter-written one (tests/tflite_fullenc.py) ----------------------------------
@pytest.fixture(scope="module")
def converter_style(tmp_path_factory):
    from tflite_fullenc import ConverterStyleModel
    from vbt_amd.tflite_import import convert
    d = tmp_path_factory.mktemp("conv_style")
    out = {}
    for tag, order in (("max", "max"), ("rnd", 7)):
        m = ConverterStyleModel(MODEL, order=order).build()
        tfl, vb = str(d / f"{tag}.tflite"), str(d / f"{tag}.vbtm")
        open(tfl, "wb").write(m.serialize())
        convert(tfl, vb)
        out[tag] = (m, tfl, vb)
    return out


def _op_signatures(c):
    return [(int(r["type"]), int(r["k"]), int(c.tensors[int(r["output"])]["h"]), int(c.tensors[int(r["output"])]["c"])) for r in c.ops]


def test_converter_style_file_keeps_every_fusable_chain_contiguous(converter_style):
    """Operators shuffled (highest native index first / a random topological order), constants ahead of reversed activations in the tensor
    table: the importer's chain-first ordering must hand the planner the same contiguous expand -> depthwise -> project [-> add] and
    add [-> add] -> depthwise -> project runs as the native container (the planner fuses only adjacent operators)."""
    from vbt_amd import spec
    from vbt_amd.container import Container
    native = Container(MODEL)

    def chains(c):
        """(depthwise convs whose producer sits directly before them and whose one consumer, a conv, directly after;
            of those, the ones fed by an ADD = BiFPN nodes; ADDs directly behind the ADD that produced one of their inputs = partial sums)"""
        prod = {int(r["output"]): i for i, r in enumerate(c.ops)}
        cons = {}
        for i, r in enumerate(c.ops):
            for t in r["inputs"][:int(r["n_inputs"])]:
                cons.setdefault(int(t), []).append(i)
        n_dw = n_node = n_pair = 0
        for i, r in enumerate(c.ops):
            typ = int(r["type"])
            ins = [int(v) for v in r["inputs"][:int(r["n_inputs"])]]
            if typ == spec.OP_DW and prod.get(ins[0]) == i - 1 and cons.get(int(r["output"])) == [i + 1] and int(c.ops[i + 1]["type"]) == spec.OP_PW:
                n_dw += 1
                n_node += int(c.ops[i - 1]["type"]) == spec.OP_ADD
            if typ == spec.OP_ADD and i >= 1 and int(c.ops[i - 1]["type"]) == spec.OP_ADD and int(c.ops[i - 1]["output"]) in ins:
                n_pair += 1
        return n_dw, n_node, n_pair

    want = chains(native)
    assert want[0] >= 15 + 24 + 30 and want[1] == 24 and want[2] == 9       # MBConv blocks + BiFPN nodes + inner head layers; 3 cells x 8 nodes; 3 three-input sums per cell
    for tag in ("max", "rnd"):
        m, tfl, vb = converter_style[tag]
        c = Container(vb)
        assert len(c.ops) == len(native.ops) and m.order != sorted(m.order)     # same operators, and the file really is in another order
        got = chains(c)
        assert got[0] >= want[0] and got[1:] == want[1:], (tag, got, want)     # (another order may make MORE depthwise convs contiguous, never fewer)


def test_converter_style_file_detects_like_the_native_container(converter_style, oracle_lib):
    """Same weights, same quantisation, another file layout: the oracle on the imported container returns the native container's
    detections, bit for bit."""
    from vbt_amd import synth
    frames = np.concatenate([synth.clip_frames(s, 3 * s, 1) for s in range(2)])
    want = [oracle_lib.OracleDetector(MODEL).run(f) for f in frames]
    for tag in ("max", "rnd"):
        det = oracle_lib.OracleDetector(converter_style[tag][2])
        for f, (wb, ws, wc, wn) in zip(frames, want):
            b, s_, c, n = det.run(f)
            assert n == wn and np.array_equal(s_, ws) and np.array_equal(b, wb), tag
