"""Graph spec vs the only structural pins the reference holds for the detector."""
import numpy as np
import pytest

from vbt_amd import spec


LOG_OPS = {0: ("efficientdet_lite0_whole.log", "1.752", "0.876"), 1: ("efficientdet_lite1_whole.log", "3.547", "1.773"), 2: ("efficientdet_lite2.log", "6.066", "3.033")}


@pytest.mark.parametrize("arch,macs_m,anchors", [(0, 865.4, 19206), (1, 1752.8, 27621), (2, 3000.4, 37629)])
def test_graph_reproduces_the_converters_op_counts(arch, macs_m, anchors):
    """The one reference-held number that constrains the detector graph (A4-A6): the TFLite converter's "Estimated count of arithmetic
    ops: 1.752 G ops, equivalently 0.876 G MACs" (reference models/efficientdet_lite0_whole.log:110; Lite1 3.547 / 1.773, Lite2
    6.066 / 3.033).  Under the converter's counting rule (Graph.tflite_arithmetic_ops) the restated graph gives all six printed figures
    to the last digit - with the class net the model maker builds for `label_map={1: "barbell"}` (train.py:30-46): two class columns per
    anchor.  A one-column class net (rounds 1-4) misses every one of them by the same 0.2 %: that was the unresolved residual of
    SURVEY.md 8a (its -1.3 % compared MACs with the converter's ops / 2, which also count bias adds, ADDs, pools and LOGISTIC)."""
    g = spec.build_graph(arch)
    assert g.num_classes == 2
    ops = g.tflite_arithmetic_ops()
    name, want_ops, want_macs = LOG_OPS[arch]
    assert f"{ops / 1e9:.3f}" == want_ops and f"{ops / 2e9:.3f}" == want_macs
    one = spec.build_graph(arch, num_classes=1).tflite_arithmetic_ops()
    assert f"{one / 1e9:.3f}" != want_ops and 0.997 < one / ops < 0.999
    assert round(g.total_macs() / 1e6, 1) == macs_m
    assert g.num_anchors() == anchors == len(spec.make_anchors(arch))
    import os
    log = os.path.join("/root/reference/models", name)           # (the reference tree is not on the GPU box; this is a CPU-suite check)
    if os.path.exists(log):
        line = [ln for ln in open(log, errors="replace").read().replace("\r", "\n").split("\n") if "Estimated count of arithmetic ops" in ln][0]
        assert f"{want_ops} G  ops" in line and f"{want_macs} G  MACs" in line


def test_lite0_op_inventory():
    g = spec.build_graph(0)
    from collections import Counter
    c = Counter(spec.OP_NAMES[o.type] for o in g.ops)
    assert (c["dw"], c["pw"], c["stem"]) == (80, 101, 1)
    # 24 BiFPN sums (9 of them 3-input = two chained binary ADDs, like the converter's graph) + 9 MBConv residual adds
    assert c["add"] == 24 + 9 + 9 and c["maxpool"] + c["resize"] == 24 + 2
    stages = {st: sum(o.macs(g.tensors) for o in g.ops if o.stage == st) / 1e6 for st in ("backbone", "fpn", "head")}
    assert [round(stages[k], 1) for k in ("backbone", "fpn", "head")] == [742.5, 53.2, 69.7]     # (SURVEY 8a's 68.4 is the one-class head)
    p = {g.tensors[o.output].name: g.tensors[o.output] for o in g.ops}
    assert (p["b4.skip"].h, p["b4.skip"].c) == (40, 40) and (p["b10.skip"].h, p["b10.skip"].c) == (20, 112) and (p["b15.project"].h, p["b15.project"].c) == (10, 320)


def test_anchor_geometry():
    a = spec.make_anchors(0)
    assert a.dtype == np.float32
    # first location of level 3: stride 8, centre 4 px, base size 3*8 = 24 px, aspects 1, 2, 1/2
    assert np.allclose(a[0], [4 / 320, 4 / 320, 24 / 320, 24 / 320])
    assert np.allclose(a[1], [4 / 320, 4 / 320, 24 / np.sqrt(2) / 320, 24 * np.sqrt(2) / 320])
    assert np.allclose(a[3, 2:], np.array([24, 24]) * 2 ** (1 / 3) / 320)
    assert np.allclose(a[-1][:2], [320 / 320, 320 / 320])                  # level 7: 3x3 grid, last centre at 64+2*128


def test_container_roundtrip(model_path):
    from vbt_amd.container import Container
    c = Container(model_path)
    g = spec.build_graph(0)
    assert int(c.header["num_ops"]) == len(g.ops) and int(c.header["num_tensors"]) == len(g.tensors)
    assert int(c.header["num_anchors"]) == 19206 and int(c.header["max_detections"]) == 25
    for rec, op in zip(c.ops, g.ops):
        assert rec["type"] == op.type and rec["output"] == op.output and list(rec["inputs"][:rec["n_inputs"]]) == op.inputs
    post = c.ops[-1]
    lut = c.f32(int(post["aux2_off"]), 768)
    assert np.all(np.diff(lut[:256]) >= 0) and set(np.unique(lut[:256] * 256) % 1) == {0.0}     # k/256 score lattice
