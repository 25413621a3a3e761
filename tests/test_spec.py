"""Graph spec vs the only structural pins the reference holds for the detector."""
import numpy as np
import pytest

from vbt_amd import spec


@pytest.mark.parametrize("arch,macs_m,log_macs_m,anchors", [(0, 864.2, 876.0, 19206), (1, 1750.3, 1773.0, 27621), (2, 2996.2, 3033.0, 37629)])
def test_mac_and_anchor_counts(arch, macs_m, log_macs_m, anchors):
    g = spec.build_graph(arch)
    assert round(g.total_macs() / 1e6, 1) == macs_m
    # reference models/efficientdet_lite*_whole.log:110 prints 2*MACs as "ops"; our restatement sits 1.2-1.4% below it
    assert 0.98 < g.total_macs() / 1e6 / log_macs_m < 1.0
    assert g.num_anchors() == anchors == len(spec.make_anchors(arch))


def test_lite0_op_inventory():
    g = spec.build_graph(0)
    from collections import Counter
    c = Counter(spec.OP_NAMES[o.type] for o in g.ops)
    assert (c["dw"], c["pw"], c["stem"]) == (80, 101, 1)
    # 24 BiFPN sums (9 of them 3-input = two chained binary ADDs, like the converter's graph) + 9 MBConv residual adds
    assert c["add"] == 24 + 9 + 9 and c["maxpool"] + c["resize"] == 24 + 2
    stages = {st: sum(o.macs(g.tensors) for o in g.ops if o.stage == st) / 1e6 for st in ("backbone", "fpn", "head")}
    assert [round(stages[k], 1) for k in ("backbone", "fpn", "head")] == [742.5, 53.2, 68.4]
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
