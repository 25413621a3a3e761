"""EfficientDet-Lite0/1/2 graph specification (shapes only, no arithmetic).

The reference never states the network: it loads ``models/efficientdet_lite*.tflite``
(reference track.py:67,93-94) which are absent from the tree (.MISSING_LARGE_BLOBS).
The layer table is therefore restated from the public EfficientDet-Lite spec
(SURVEY.md section 8a rows A4-A6 [EXTERNAL]) and cross-checked against the static op
counts printed by the TFLite converter in reference models/efficientdet_lite0_whole.log:110
(see tests/test_spec.py).

This module is pure host logic.  It is used by
  * tools/make_model.py  - to generate the synthetic quantised model container,
  * tests               - MAC / op / anchor counts,
  * bench.py/DESIGN.md  - algorithmic bytes per op (roofline accounting).
The product (csrc/) and the oracle (oracle/) both execute the *container* and never
import this file at run time.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

# op type ids (shared with the container format, see vbt_amd/container.py and csrc/vbt_model.h)
OP_STEM = 1       # 3x3 stride-2 conv on the uint8 frame (Cin = 3)
OP_PW = 2         # 1x1 conv
OP_DW = 3         # depthwise kxk conv
OP_ADD = 4        # binary elementwise int8 ADD (TFLite ADD / XNNPACK qs8-vadd); 3-input BiFPN sums are two chained ADDs
OP_MAXPOOL = 5    # 3x3 stride-2 SAME max pool
OP_RESIZE_NN = 6  # nearest-neighbour resize (legacy TF semantics: src = floor(dst*in/out))
OP_POSTPROCESS = 7

OP_NAMES = {OP_STEM: "stem", OP_PW: "pw", OP_DW: "dw", OP_ADD: "add", OP_MAXPOOL: "maxpool",
            OP_RESIZE_NN: "resize", OP_POSTPROCESS: "postprocess"}

ACT_NONE = 0
ACT_RELU6 = 1

MAX_DETECTIONS = 25          # pinned by reference dfs/eval_detections.pkl.gz (61 x 25 rows / model)
NUM_ANCHORS_PER_LOC = 9      # 3 octave scales x 3 aspect ratios
# Class columns per anchor in the head's class tensors.  reference train.py:30-46 trains ONE label, `label_map={1: "barbell"}`, with
# tflite-model-maker, whose EfficientDet head has max(label id) + 1 columns (label ids are 1-based, column 0 is never a target): TWO.
# Pinned by the converter's op counts in reference models/*.log:110 (1.752 / 3.547 / 6.066 G): only the two-column class net reproduces
# all three to the printed digits under the converter's counting rule (tflite_arithmetic_ops below, tests/test_spec.py).
NUM_CLASSES = 2


@dataclass
class Tensor:
    id: int
    name: str
    h: int
    w: int
    c: int

    @property
    def elems(self) -> int:
        return self.h * self.w * self.c


@dataclass
class Op:
    type: int
    name: str
    inputs: List[int]
    output: int
    k: int = 1
    stride: int = 1
    pad_t: int = 0
    pad_l: int = 0
    act: int = ACT_NONE
    stage: str = ""           # "backbone" | "fpn" | "head"
    level: int = -1           # feature level for head ops
    share: str = ""           # weight-sharing key (head convs share across levels)

    def macs(self, tensors: List[Tensor]) -> int:
        o = tensors[self.output]
        if self.type == OP_STEM:
            return o.elems * self.k * self.k * tensors[self.inputs[0]].c
        if self.type == OP_PW:
            return o.elems * tensors[self.inputs[0]].c
        if self.type == OP_DW:
            return o.elems * self.k * self.k
        return 0

    def weight_elems(self, tensors: List[Tensor]) -> int:
        o = tensors[self.output]
        if self.type == OP_STEM:
            return o.c * self.k * self.k * tensors[self.inputs[0]].c
        if self.type == OP_PW:
            return o.c * tensors[self.inputs[0]].c
        if self.type == OP_DW:
            return o.c * self.k * self.k
        return 0


@dataclass
class ArchConfig:
    name: str
    image_size: int
    width: float
    depth: float
    fpn_ch: int
    fpn_cells: int
    head_repeats: int = 3
    anchor_scale: float = 3.0


ARCHS = {
    0: ArchConfig("efficientdet_lite0", 320, 1.0, 1.0, 64, 3),
    1: ArchConfig("efficientdet_lite1", 384, 1.0, 1.1, 88, 4),
    2: ArchConfig("efficientdet_lite2", 448, 1.1, 1.2, 112, 5),
}

# (repeats, kernel, stride, expand, cout) - EfficientNet-Lite stage table
_STAGES = [(1, 3, 1, 1, 16), (2, 3, 2, 6, 24), (2, 5, 2, 6, 40), (3, 3, 2, 6, 80),
           (3, 5, 1, 6, 112), (4, 5, 2, 6, 192), (1, 3, 1, 6, 320)]


def _round_filters(f: int, width: float) -> int:
    if width == 1.0:
        return f
    f2 = f * width
    new = max(8, int(f2 + 4) // 8 * 8)
    if new < 0.9 * f2:
        new += 8
    return int(new)


def _same_pad(in_size: int, k: int, s: int) -> Tuple[int, int]:
    """TF 'SAME' padding: returns (out_size, pad_before)."""
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2


class Graph:
    def __init__(self, cfg: ArchConfig):
        self.cfg = cfg
        self.tensors: List[Tensor] = []
        self.ops: List[Op] = []

    def tensor(self, name, h, w, c) -> int:
        t = Tensor(len(self.tensors), name, h, w, c)
        self.tensors.append(t)
        return t.id

    def add(self, op: Op) -> int:
        self.ops.append(op)
        return op.output

    # ---- layer helpers -------------------------------------------------
    def conv_pw(self, name, x, cout, act, stage, level=-1, share=""):
        t = self.tensors[x]
        o = self.tensor(name, t.h, t.w, cout)
        return self.add(Op(OP_PW, name, [x], o, act=act, stage=stage, level=level, share=share))

    def conv_dw(self, name, x, k, s, act, stage, level=-1, share=""):
        t = self.tensors[x]
        oh, pt = _same_pad(t.h, k, s)
        ow, pl = _same_pad(t.w, k, s)
        o = self.tensor(name, oh, ow, t.c)
        return self.add(Op(OP_DW, name, [x], o, k=k, stride=s, pad_t=pt, pad_l=pl, act=act,
                           stage=stage, level=level, share=share))

    def addn(self, name, xs, act, stage):
        """Sum of 2 or 3 tensors as the Keras EfficientDet writes it (`sum(nodes)`: ((n0 + n1) + n2)) and the TFLite
        converter keeps it: binary ADD ops, the partial sum with a quantisation of its own and no activation, the
        activation fused into the last ADD."""
        t = self.tensors[xs[0]]
        for x in xs[1:]:
            u = self.tensors[x]
            assert (u.h, u.w, u.c) == (t.h, t.w, t.c), (name, t, u)
        assert 2 <= len(xs) <= 3
        acc = xs[0]
        for j, x in enumerate(xs[1:], 1):
            last = j == len(xs) - 1
            o = self.tensor(name if last else f"{name}.partial{j}", t.h, t.w, t.c)
            acc = self.add(Op(OP_ADD, name if last else f"{name}.partial{j}", [acc, x], o, act=act if last else ACT_NONE, stage=stage))
        return acc

    def maxpool(self, name, x, stage):
        t = self.tensors[x]
        oh, pt = _same_pad(t.h, 3, 2)
        ow, pl = _same_pad(t.w, 3, 2)
        o = self.tensor(name, oh, ow, t.c)
        return self.add(Op(OP_MAXPOOL, name, [x], o, k=3, stride=2, pad_t=pt, pad_l=pl, stage=stage))

    def resize(self, name, x, oh, ow, stage):
        t = self.tensors[x]
        o = self.tensor(name, oh, ow, t.c)
        return self.add(Op(OP_RESIZE_NN, name, [x], o, stage=stage))

    # ---- accounting ----------------------------------------------------
    def total_macs(self) -> int:
        return sum(op.macs(self.tensors) for op in self.ops)

    def activation_elems(self) -> int:
        """Compulsory activation traffic: every op reads its inputs once and writes its
        output once."""
        return sum(self.alg_elems(op) for op in self.ops)

    def alg_elems(self, op) -> int:
        """Activation elements an op moves in the survey's accounting.  A 3-input BiFPN sum is ONE add there (three
        reads, one write): the partial sum between its two binary ADDs is not counted on either side."""
        if op.type == OP_POSTPROCESS:
            return 0
        partial = lambda t: self.tensors[t].name.rsplit(".", 1)[-1].startswith("partial")
        n = sum(self.tensors[i].elems for i in op.inputs if not partial(i))
        return n + (0 if partial(op.output) else self.tensors[op.output].elems)

    def weight_elems(self) -> int:
        seen = set()
        n = 0
        for op in self.ops:
            key = op.share or op.name
            if key in seen:
                continue
            seen.add(key)
            n += op.weight_elems(self.tensors)
        return n

    def tflite_arithmetic_ops(self) -> int:
        """"Estimated count of arithmetic ops" as the TFLite converter prints it when it exports a model (reference
        models/efficientdet_lite0_whole.log:110 and the two sibling logs) [EXTERNAL: TFLite MLIR converter, the per-op
        GetArithmeticCount rules]: CONV_2D / DEPTHWISE_CONV_2D 2 x MACs + one bias add per output element; ADD one per output
        element; MAX_POOL_2D filter_h x filter_w per output element; LOGISTIC 64 per element (on the concatenated class scores);
        QUANTIZE / DEQUANTIZE / RESIZE_NEAREST_NEIGHBOR / RESHAPE / CONCATENATION / the post-process custom op count nothing."""
        T = self.tensors
        n = 0
        for op in self.ops:
            o = T[op.output]
            if op.type in (OP_STEM, OP_PW, OP_DW):
                n += 2 * op.macs(T) + o.elems
            elif op.type == OP_ADD:
                n += o.elems
            elif op.type == OP_MAXPOOL:
                n += o.elems * op.k * op.k
            elif op.type == OP_POSTPROCESS:
                n += 64 * sum(T[t].elems for t in op.inputs[:len(op.inputs) // 2])
        return n

    def num_anchors(self) -> int:
        return sum(s * s for s in self.level_sizes()) * NUM_ANCHORS_PER_LOC

    def level_sizes(self) -> List[int]:
        return [math.ceil(self.cfg.image_size / 2 ** l) for l in range(3, 8)]


def build_graph(arch: int, num_classes: int = NUM_CLASSES) -> Graph:
    cfg = ARCHS[arch]
    g = Graph(cfg)
    g.num_classes = int(num_classes)
    S = cfg.image_size
    x = g.tensor("image", S, S, 3)

    # ---------------- backbone: EfficientNet-Lite ------------------------
    oh, pt = _same_pad(S, 3, 2)
    stem = g.tensor("stem", oh, oh, 32)          # stem width is not scaled in the Lite variants
    g.add(Op(OP_STEM, "stem", [x], stem, k=3, stride=2, pad_t=pt, pad_l=pt, act=ACT_RELU6, stage="backbone"))
    x = stem
    feats: Dict[int, int] = {}
    nst = len(_STAGES)
    bi = 0
    for si, (r, k, s, e, cout) in enumerate(_STAGES):
        cout = _round_filters(cout, cfg.width)
        if si not in (0, nst - 1):                # first/last stage repeats are not scaled
            r = int(math.ceil(r * cfg.depth))
        for j in range(r):
            stride = s if j == 0 else 1
            cin = g.tensors[x].c
            name = f"b{bi}"
            y = x
            if e != 1:
                y = g.conv_pw(f"{name}.expand", y, cin * e, ACT_RELU6, "backbone")
            y = g.conv_dw(f"{name}.dw", y, k, stride, ACT_RELU6, "backbone")
            y = g.conv_pw(f"{name}.project", y, cout, ACT_NONE, "backbone")
            if stride == 1 and cin == cout:
                y = g.addn(f"{name}.skip", [y, x], ACT_NONE, "backbone")
            x = y
            bi += 1
        red = int(round(math.log2(S / g.tensors[x].h)))
        feats[red] = x                            # last tensor at each reduction level
    p3, p4, p5 = feats[3], feats[4], feats[5]

    # ---------------- P6 / P7 --------------------------------------------
    W = cfg.fpn_ch
    p6c = g.conv_pw("p6.conv", p5, W, ACT_NONE, "fpn")
    p6 = g.maxpool("p6.pool", p6c, "fpn")
    p7 = g.maxpool("p7.pool", p6, "fpn")
    nodes_cfg = [(6, [3, 4]), (5, [2, 5]), (4, [1, 6]), (3, [0, 7]),
                 (4, [1, 7, 8]), (5, [2, 6, 9]), (6, [3, 5, 10]), (7, [4, 11])]
    cur = [p3, p4, p5, p6, p7]
    for cell in range(cfg.fpn_cells):
        fl = list(cur)
        levels = [3, 4, 5, 6, 7]
        for ni, (lvl, offs) in enumerate(nodes_cfg):
            tgt = g.level_sizes()[lvl - 3]
            ins = []
            for o in offs:
                t = fl[o]
                tt = g.tensors[t]
                nm = f"c{cell}.n{ni}.in{o}"
                if tt.c != W:                                  # lateral 1x1 conv (first cell only)
                    t = g.conv_pw(nm + ".lat", t, W, ACT_NONE, "fpn")
                    tt = g.tensors[t]
                if tt.h < tgt:
                    t = g.resize(nm + ".up", t, tgt, tgt, "fpn")
                elif tt.h > tgt:
                    t = g.maxpool(nm + ".down", t, "fpn")
                    assert g.tensors[t].h == tgt
                ins.append(t)
            s = g.addn(f"c{cell}.n{ni}.sum", ins, ACT_RELU6, "fpn")
            d = g.conv_dw(f"c{cell}.n{ni}.dw", s, 3, 1, ACT_NONE, "fpn")
            p = g.conv_pw(f"c{cell}.n{ni}.pw", d, W, ACT_NONE, "fpn")
            fl.append(p)
            levels.append(lvl)
        cur = []
        for lvl in range(3, 8):
            idx = max(i for i, l in enumerate(levels) if l == lvl)
            cur.append(fl[idx])

    # ---------------- heads ----------------------------------------------
    cls_out, box_out = [], []
    for li, f in enumerate(cur):
        lvl = li + 3
        for head, cfin, outs in (("cls", NUM_ANCHORS_PER_LOC * g.num_classes, cls_out),
                                 ("box", NUM_ANCHORS_PER_LOC * 4, box_out)):
            y = f
            for i in range(cfg.head_repeats):
                y = g.conv_dw(f"{head}.l{lvl}.r{i}.dw", y, 3, 1, ACT_NONE, "head", lvl, share=f"{head}.r{i}.dw")
                y = g.conv_pw(f"{head}.l{lvl}.r{i}.pw", y, W, ACT_RELU6, "head", lvl, share=f"{head}.r{i}.pw")
            y = g.conv_dw(f"{head}.l{lvl}.out.dw", y, 3, 1, ACT_NONE, "head", lvl, share=f"{head}.out.dw")
            y = g.conv_pw(f"{head}.l{lvl}.out.pw", y, cfin, ACT_NONE, "head", lvl, share=f"{head}.out.pw")
            outs.append(y)
    det = g.tensor("detections", 1, MAX_DETECTIONS, 6)
    g.add(Op(OP_POSTPROCESS, "postprocess", cls_out + box_out, det, stage="post"))
    return g


def make_anchors(arch: int):
    """Anchor boxes (ycenter, xcenter, h, w) normalised to the image, in head output order:
    level-major, then y, x, then (octave, aspect).  Restates the public EfficientDet anchor
    rule named in SURVEY.md section 8a row A7 [EXTERNAL]: levels 3-7, 3 octave scales
    2^{0,1/3,2/3}, aspects {1, 2, 1/2}, anchor_scale 3.0 (Lite0-2)."""
    import numpy as np
    cfg = ARCHS[arch]
    S = cfg.image_size
    out = []
    for lvl in range(3, 8):
        fs = math.ceil(S / 2 ** lvl)
        stride = S / fs if False else 2 ** lvl
        for y in range(fs):
            for x in range(fs):
                cy = stride / 2.0 + y * stride
                cx = stride / 2.0 + x * stride
                for octave in range(3):
                    for aspect in (1.0, 2.0, 0.5):
                        base = cfg.anchor_scale * stride * 2 ** (octave / 3.0)
                        ax = base * math.sqrt(aspect)
                        ay = base / math.sqrt(aspect)
                        out.append((cy / S, cx / S, ay / S, ax / S))
    return np.asarray(out, dtype=np.float32)


def per_op_bytes(g: Graph, act_bytes: int = 1, w_bytes: int = 1, batch: int = 1):
    """Algorithmic HBM bytes per op for a batch: inputs read once,
    output written once, weights read once per launch."""
    rows = []
    for op in g.ops:
        if op.type == OP_POSTPROCESS:
            continue
        a = g.alg_elems(op)
        w = op.weight_elems(g.tensors)
        rows.append((op.name, OP_NAMES[op.type], a * act_bytes * batch + w * w_bytes, op.macs(g.tensors) * batch))
    return rows
