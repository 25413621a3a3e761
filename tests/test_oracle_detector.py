"""oracle/detector.c sanity (parity unpinned: no reference model, frame or tensor exists - SURVEY.md 8c).
Structural pins only: 25 detections, k/256 score lattice, NMS invariants; plus an independent numpy
re-computation of single ops from the container to guard the C restatement itself."""
import numpy as np
import pytest

from vbt_amd import spec, synth
from vbt_amd.container import Container


@pytest.fixture(scope="module")
def run(oracle_lib, model_path):
    det = oracle_lib.OracleDetector(model_path)
    frame = synth.clip_frames(0, 0, 1)[0]
    out = det.run(frame)
    return det, frame, out


def test_output_signature(run):
    det, frame, (boxes, scores, classes, count) = run
    assert boxes.shape == (25, 4) and scores.shape == (25,) and 0 <= count <= 25
    s = scores[:count]
    assert np.all(np.diff(s) <= 0)                                   # sorted by score
    assert np.all((s * 256) % 1 == 0) and np.all(s >= 1 / 256)        # reference scores are multiples of 1/256
    assert np.all(np.isin(classes[:count], (0.0, 1.0))) and np.all(classes[count:] == 0) and np.all(boxes[count:] == 0)   # two class columns (vbt_amd/spec.py)
    for i in range(count):                                            # greedy NMS invariant: IoU <= 0.5 between survivors
        for j in range(i):
            a, b = boxes[i], boxes[j]
            ih = max(min(a[2], b[2]) - max(a[0], b[0]), 0)
            iw = max(min(a[3], b[3]) - max(a[1], b[1]), 0)
            inter = ih * iw
            union = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter
            assert inter / union <= 0.5 + 1e-6


def _requant(acc, mult, zp, lo, hi):
    t = (acc.astype(np.float32) * mult.astype(np.float32)).astype(np.float32)
    return np.clip(np.rint(t).astype(np.int64) + zp, lo, hi).astype(np.int8)


def test_single_ops_against_numpy(run, model_path):
    det, frame, _ = run
    c = Container(model_path)
    g = spec.build_graph(0)
    checked = set()
    for oi, op in enumerate(g.ops):
        rec = c.ops[oi]
        to = c.tensors[op.output]
        if op.type == spec.OP_PW and op.name in ("b1.expand", "c0.n3.pw", "cls.l5.out.pw"):
            ti = c.tensors[op.inputs[0]]
            x = det.tensor(op.inputs[0]).astype(np.int64) - int(ti["zero_point"])
            w = c.i8(int(rec["w_off"]), int(to["c"]) * int(ti["c"])).reshape(int(to["c"]), int(ti["c"])).astype(np.int64)
            acc = x.reshape(-1, int(ti["c"])) @ w.T + c.i32(int(rec["b_off"]), int(to["c"])).astype(np.int64)
            want = _requant(acc, c.f32(int(rec["m_off"]), int(to["c"]))[None, :], int(to["zero_point"]), int(rec["act_min"]), int(rec["act_max"]))
            assert np.array_equal(det.tensor(op.output).reshape(-1, int(to["c"])), want), op.name
            checked.add("pw")
        if op.type == spec.OP_DW and op.name == "b3.dw":            # 5x5 stride 2, asymmetric SAME padding
            ti = c.tensors[op.inputs[0]]
            C, k, s = int(ti["c"]), op.k, op.stride
            x = det.tensor(op.inputs[0]).astype(np.int64) - int(ti["zero_point"])
            w = c.i8(int(rec["w_off"]), k * k * C).reshape(k, k, C).astype(np.int64)
            oh, ow = int(to["h"]), int(to["w"])
            pb = max((oh - 1) * s + k - int(ti["h"]) - op.pad_t, 0)
            xp = np.pad(x, ((op.pad_t, pb), (op.pad_l, pb), (0, 0)))
            acc = np.zeros((oh, ow, C), np.int64)
            for ky in range(k):
                for kx in range(k):
                    acc += xp[ky:ky + s * oh:s, kx:kx + s * ow:s] * w[ky, kx]
            acc += c.i32(int(rec["b_off"]), C).astype(np.int64)
            want = _requant(acc, c.f32(int(rec["m_off"]), C)[None, None, :], int(to["zero_point"]), int(rec["act_min"]), int(rec["act_max"]))
            assert np.array_equal(det.tensor(op.output), want)
            checked.add("dw")
        if op.type == spec.OP_ADD and op.name in ("c0.n4.sum.partial1", "c0.n4.sum", "b2.skip"):
            # binary integer ADD (XNNPACK qs8-vadd): parameters re-derived from the tensor scales by the exact-rational
            # statement of tests/test_quant_kat.py, every element by integer arithmetic
            from test_quant_kat import ref_add_params
            ta, tb = c.tensors[op.inputs[0]], c.tensors[op.inputs[1]]
            prm = ref_add_params(float(ta["scale"]), float(tb["scale"]), float(to["scale"]), int(ta["zero_point"]), int(tb["zero_point"]))
            assert tuple(int(v) for v in rec["add_q"]) == prm, op.name
            acc = prm[0] + det.tensor(op.inputs[0]).astype(np.int64) * prm[1] + det.tensor(op.inputs[1]).astype(np.int64) * prm[2]
            want = np.clip((acc >> prm[3]) + int(to["zero_point"]), int(rec["act_min"]), int(rec["act_max"])).astype(np.int8)
            assert np.array_equal(det.tensor(op.output), want), op.name
            checked.add("add:" + op.name.split(".")[-1])
        if op.type == spec.OP_RESIZE_NN and "up" in op.name and "resize" not in checked:
            x = det.tensor(op.inputs[0])
            oh = int(to["h"])
            idx = (np.arange(oh) * x.shape[0]) // oh
            assert np.array_equal(det.tensor(op.output), x[idx][:, idx])
            checked.add("resize")
    assert checked == {"pw", "dw", "add:partial1", "add:sum", "add:skip", "resize"}


def test_batch_helper_equals_single(oracle_lib, model_path):
    frames = synth.clip_frames(5, 3, 3)
    b, s, c, n = oracle_lib.run_batch(model_path, frames, threads=3)
    det = oracle_lib.OracleDetector(model_path)
    for i in range(3):
        ob, os_, oc, on = det.run(frames[i])
        assert n[i] == on and np.array_equal(b[i], ob) and np.array_equal(s[i], os_)
