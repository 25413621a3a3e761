"""`.tflite` importer: a full-integer EfficientDet-Lite flatbuffer -> the VBTM container the library executes.

The reference loads `models/efficientdet_lite{0,1,2}[_whole].tflite` through tflite_runtime
(reference track.py:67,88-94; eval.py:160-168).  Those files are absent from the tree (.MISSING_LARGE_BLOBS), so
this importer is written against the public TFLite schema (schema.fbs field numbers are listed next to each accessor)
and is exercised by files produced with tools/export_tflite.py; SURVEY.md section 8f row N1.

What is mapped (everything else raises `UnsupportedModel` naming the operator):
  QUANTIZE uint8->int8 on the input (same scale, zero point shifted by 128)  -> folded into the stem's input tensor
  CONV_2D 3x3/2 on the 3-channel image                                       -> OP_STEM
  CONV_2D 1x1/1                                                              -> OP_PW
  DEPTHWISE_CONV_2D (depth multiplier 1, SAME)                               -> OP_DW
  ADD (binary, fused NONE/RELU6)                                             -> OP_ADD
  MAX_POOL_2D 3x3/2 SAME                                                     -> OP_MAXPOOL
  RESIZE_NEAREST_NEIGHBOR (legacy rounding)                                  -> OP_RESIZE_NN
  per-level head outputs -> RESHAPE -> CONCATENATION -> [LOGISTIC] -> DEQUANTIZE -> TFLite_Detection_PostProcess
                                                                             -> OP_POSTPROCESS (+ anchors and LUTs)
Requantisation multipliers are float32 `s_x * s_w[c] / s_y` (the XNNPACK "fp32" scheme the oracle restates,
DESIGN.md section 2).
"""
from __future__ import annotations

import os
import tempfile

import numpy as np

from . import quant, spec
from .container import OP_DTYPE, TENSOR_DTYPE, BlobWriter, write_container
from .flatbuf import Table, flex_root

# BuiltinOperator values (schema.fbs `enum BuiltinOperator`)
BO_ADD, BO_CONCATENATION, BO_CONV_2D, BO_DEPTHWISE_CONV_2D, BO_DEQUANTIZE = 0, 2, 3, 4, 6
BO_LOGISTIC, BO_MAX_POOL_2D, BO_RESHAPE, BO_CUSTOM, BO_RESIZE_NEAREST_NEIGHBOR, BO_QUANTIZE = 14, 17, 22, 32, 97, 114
BO_NAMES = {0: "ADD", 1: "AVERAGE_POOL_2D", 2: "CONCATENATION", 3: "CONV_2D", 4: "DEPTHWISE_CONV_2D", 6: "DEQUANTIZE",
            9: "FULLY_CONNECTED", 14: "LOGISTIC", 17: "MAX_POOL_2D", 18: "MUL", 19: "RELU", 21: "RELU6", 22: "RESHAPE",
            23: "RESIZE_BILINEAR", 25: "SOFTMAX", 32: "CUSTOM", 97: "RESIZE_NEAREST_NEIGHBOR", 114: "QUANTIZE"}
# TensorType
TT_FLOAT32, TT_INT32, TT_UINT8, TT_INT64, TT_INT8 = 0, 2, 3, 4, 9
TT_NP = {TT_FLOAT32: np.float32, TT_INT32: np.int32, TT_UINT8: np.uint8, TT_INT64: np.int64, TT_INT8: np.int8}
ACT_NONE, ACT_RELU6 = 0, 3          # ActivationFunctionType
PAD_SAME, PAD_VALID = 0, 1          # Padding

POSTPROCESS_NAME = "TFLite_Detection_PostProcess"


class UnsupportedModel(ValueError):
    pass


def is_tflite(path) -> bool:
    try:
        with open(path, "rb") as f:
            head = f.read(8)
    except OSError:
        return False
    return len(head) == 8 and head[4:8] == b"TFL3"


class TfTensor:
    def __init__(self, t: Table, buffers):
        self.shape = tuple(int(v) for v in t.vector(0, np.int32))            # Tensor.shape:0
        self.type = t.scalar(1, "b", 0)                                       # Tensor.type:1
        self.buffer = t.scalar(2, "I", 0)                                     # Tensor.buffer:2
        self.name = t.string(3)                                               # Tensor.name:3
        q = t.table(4)                                                        # Tensor.quantization:4
        self.scale = np.array(q.vector(2, np.float32)) if q else np.empty(0, np.float32)   # .scale:2
        self.zero_point = np.array(q.vector(3, np.int64)) if q else np.empty(0, np.int64)  # .zero_point:3
        self.qdim = q.scalar(6, "i", 0) if q else 0                                        # .quantized_dimension:6
        raw = buffers[self.buffer] if 0 <= self.buffer < len(buffers) else None
        self.data = None
        if raw is not None and raw.size:
            dt = TT_NP.get(self.type)
            if dt is not None:
                self.data = raw.view(np.dtype(dt).newbyteorder("<")).reshape(self.shape if self.shape else (-1,))

    @property
    def is_const(self):
        return self.data is not None

    def q(self):
        if self.scale.size != 1 or self.zero_point.size != 1:
            raise UnsupportedModel(f"tensor '{self.name}': per-tensor quantisation expected")
        return np.float32(self.scale[0]), int(self.zero_point[0])


class TfOp:
    def __init__(self, o: Table, codes):
        idx = o.scalar(0, "I", 0)                                             # Operator.opcode_index:0
        self.code, self.custom = codes[idx]
        self.inputs = [int(v) for v in o.vector(1, np.int32)]                 # Operator.inputs:1
        self.outputs = [int(v) for v in o.vector(2, np.int32)]                # Operator.outputs:2
        self.options = o.table(4)                                             # Operator.builtin_options:4 (union value)
        self.custom_options = bytes(o.vector(5, np.uint8))                    # Operator.custom_options:5

    @property
    def name(self):
        return self.custom if self.code == BO_CUSTOM else BO_NAMES.get(self.code, f"builtin#{self.code}")

    def opt(self, idx, fmt, default=0):
        return self.options.scalar(idx, fmt, default) if self.options is not None else default


class TfModel:
    def __init__(self, path):
        buf = np.fromfile(path, dtype=np.uint8)
        if buf.size < 8 or bytes(buf[4:8]) != b"TFL3":
            raise UnsupportedModel(f"{path}: not a TFLite flatbuffer (no 'TFL3' identifier)")
        mv = memoryview(buf)
        root = Table.root(mv)
        self.version = root.scalar(0, "I", 0)                                 # Model.version:0
        codes = []
        for c in root.tables(1):                                              # Model.operator_codes:1
            dep = c.scalar(0, "b", 0)                                         # OperatorCode.deprecated_builtin_code:0
            new = c.scalar(3, "i", 0)                                         # OperatorCode.builtin_code:3
            codes.append((max(dep, new), c.string(1)))                        # OperatorCode.custom_code:1
        buffers = []
        for b in root.tables(4):                                              # Model.buffers:4
            data = b.vector(0, np.uint8)                                      # Buffer.data:0
            off, size = b.scalar(1, "Q", 0), b.scalar(2, "Q", 0)              # Buffer.offset:1 / size:2 (>2 GB files)
            if data.size == 0 and off > 1 and size:
                data = buf[off:off + size]
            buffers.append(np.asarray(data))
        subgraphs = root.tables(2)                                            # Model.subgraphs:2
        if len(subgraphs) != 1:
            raise UnsupportedModel(f"expected one subgraph, found {len(subgraphs)}")
        sg = subgraphs[0]
        self.tensors = [TfTensor(t, buffers) for t in sg.tables(0)]           # SubGraph.tensors:0
        self.inputs = [int(v) for v in sg.vector(1, np.int32)]                # SubGraph.inputs:1
        self.outputs = [int(v) for v in sg.vector(2, np.int32)]               # SubGraph.outputs:2
        self.ops = [TfOp(o, codes) for o in sg.tables(3)]                     # SubGraph.operators:3


def _same_pad(in_size, k, s):
    out = -(-in_size // s)
    total = max((out - 1) * s + k - in_size, 0)
    return out, total // 2


def _act_range(act, scale, zp, where):
    if act == ACT_NONE:
        return -128, 127
    if act == ACT_RELU6:
        return max(-128, zp), min(127, zp + int(np.rint(6.0 / float(scale))))
    raise UnsupportedModel(f"{where}: fused activation {act} is not supported (NONE / RELU6 only)")


def import_tflite(path):
    """Returns (header dict, tensors, ops, blob bytes) ready for container.write_container."""
    m = TfModel(path)
    T = m.tensors
    producer = {}
    for oi, op in enumerate(m.ops):
        for t in op.outputs:
            producer[t] = oi
    consumers = {}
    for oi, op in enumerate(m.ops):
        for t in op.inputs:
            consumers.setdefault(t, []).append(oi)

    if len(m.inputs) != 1:
        raise UnsupportedModel("expected exactly one model input")
    tin = T[m.inputs[0]]
    if len(tin.shape) != 4 or tin.shape[0] != 1 or tin.shape[3] != 3 or tin.shape[1] != tin.shape[2]:
        raise UnsupportedModel(f"input shape {tin.shape}: expected [1,S,S,3]")
    S = tin.shape[1]

    post = [oi for oi, op in enumerate(m.ops) if op.code == BO_CUSTOM and op.custom == POSTPROCESS_NAME]
    if len(post) != 1:
        raise UnsupportedModel(f"expected one {POSTPROCESS_NAME} operator, found {len(post)}")
    post_op = m.ops[post[0]]
    if len(post_op.inputs) != 3:
        raise UnsupportedModel(f"{POSTPROCESS_NAME}: expected 3 inputs")

    # ---- walk the post-process inputs back to the per-level head tensors
    tail_ops = {post[0]}

    def trace_back(t):
        """-> (list of per-level activation tensors, concat quant (scale, zp), LOGISTIC output tensor or None)."""
        logistic = None
        while True:
            if t not in producer:
                raise UnsupportedModel(f"{POSTPROCESS_NAME}: input '{T[t].name}' is not computed by the graph")
            oi = producer[t]
            op = m.ops[oi]
            tail_ops.add(oi)
            if op.code == BO_DEQUANTIZE:
                t = op.inputs[0]
            elif op.code == BO_LOGISTIC:
                logistic = op.outputs[0]
                t = op.inputs[0]
            elif op.code == BO_RESHAPE and len(consumers.get(op.inputs[0], [])) == 1 and producer.get(op.inputs[0]) is not None \
                    and m.ops[producer[op.inputs[0]]].code == BO_CONCATENATION:
                t = op.inputs[0]
            elif op.code == BO_CONCATENATION:
                cq = T[op.outputs[0]].q()
                levels = []
                for ci in op.inputs:
                    src = ci
                    if src in producer and m.ops[producer[src]].code == BO_RESHAPE:
                        tail_ops.add(producer[src])
                        src = m.ops[producer[src]].inputs[0]
                    if T[ci].q() != cq or T[src].q() != cq:
                        raise UnsupportedModel("CONCATENATION inputs must share the output's quantisation")
                    levels.append(src)
                return levels, cq, logistic
            else:
                raise UnsupportedModel(f"unexpected {op.name} between the heads and {POSTPROCESS_NAME}")

    box_levels, box_q, box_log = trace_back(post_op.inputs[0])
    cls_levels, cls_q, cls_log = trace_back(post_op.inputs[1])
    if box_log is not None:
        raise UnsupportedModel("LOGISTIC on the box branch")
    if len(box_levels) != 5 or len(cls_levels) != 5:
        raise UnsupportedModel(f"expected 5 pyramid levels, found {len(cls_levels)} class / {len(box_levels)} box")
    anchors_t = T[post_op.inputs[2]]
    if not anchors_t.is_const:
        raise UnsupportedModel("anchors must be a constant tensor")
    anchors = np.asarray(anchors_t.data)
    if anchors_t.type == TT_UINT8 or anchors_t.type == TT_INT8:
        s, z = anchors_t.q()
        anchors = (anchors.astype(np.float32) - np.float32(z)) * s
    anchors = np.ascontiguousarray(anchors.reshape(-1, 4), dtype=np.float32)

    opts = flex_root(post_op.custom_options) if post_op.custom_options else {}
    if not isinstance(opts, dict):
        raise UnsupportedModel(f"{POSTPROCESS_NAME}: custom options are not a map")
    # class predictions [1, A, C]: C = num_classes_with_background of detection_postprocess.cc; the op scores columns label_offset = C -
    # num_classes onwards.  The reference's models (one label with id 1 under tflite-model-maker: two columns, num_classes = 2,
    # vbt_amd/spec.py) have label_offset 0, and so must any model mapped here.
    num_classes = int(opts.get("num_classes", 1))
    cls_shape = list(T[post_op.inputs[1]].shape)
    n_cols = int(cls_shape[2]) if len(cls_shape) == 3 else 1
    if num_classes < 1 or num_classes > 4 or n_cols != num_classes:
        raise UnsupportedModel(f"num_classes = {num_classes} with {n_cols} class columns: 1..4 classes without a background column are mapped")
    if int(opts.get("max_classes_per_detection", 1)) != 1:
        raise UnsupportedModel("max_classes_per_detection != 1")
    if bool(opts.get("use_regular_nms", False)):
        raise UnsupportedModel("use_regular_nms = true is not supported (fast NMS only)")
    max_det = int(opts.get("max_detections", spec.MAX_DETECTIONS))
    y_scale, x_scale = float(opts.get("y_scale", 1.0)), float(opts.get("x_scale", 1.0))
    h_scale, w_scale = float(opts.get("h_scale", 1.0)), float(opts.get("w_scale", 1.0))
    if y_scale != x_scale or h_scale != w_scale:
        raise UnsupportedModel("y_scale != x_scale or h_scale != w_scale")

    # ---- activation tensors and ops, in file order
    tmap = {}                     # tflite tensor index -> container tensor id
    tensors = []                  # (h, w, c, zp, scale)
    ops = []
    blob = BlobWriter()
    wcache = {}

    def new_tensor(ti, qp=None):
        t = T[ti]
        if len(t.shape) != 4 or t.shape[0] != 1:
            raise UnsupportedModel(f"tensor '{t.name}' shape {t.shape}: expected [1,H,W,C]")
        if t.type != TT_INT8 and qp is None:
            raise UnsupportedModel(f"tensor '{t.name}': int8 activations expected")
        s, z = qp if qp is not None else t.q()
        tmap[ti] = len(tensors)
        tensors.append((t.shape[1], t.shape[2], t.shape[3], int(z), np.float32(s)))
        return tmap[ti]

    def src(ti, where):
        if ti not in tmap:
            raise UnsupportedModel(f"{where}: input '{T[ti].name}' is not an activation produced earlier")
        return tmap[ti]

    def new_op(typ, ins, out, **kw):
        r = np.zeros(1, OP_DTYPE)[0]
        r["type"], r["n_inputs"], r["output"] = typ, len(ins), out
        r["inputs"][:len(ins)] = ins
        r["k"], r["stride"], r["level"] = kw.get("k", 1), kw.get("stride", 1), -1
        r["pad_t"], r["pad_l"] = kw.get("pad_t", 0), kw.get("pad_l", 0)
        r["act_min"], r["act_max"] = kw.get("act", (-128, 127))
        ops.append(r)
        return r

    def add_const(arr):
        key = (arr.dtype.str, arr.shape, arr.tobytes())
        if key not in wcache:
            wcache[key] = blob.add(arr)
        return wcache[key]

    def conv_params(op, where, w_t, b_t, cout, sx, so):
        if w_t.type != TT_INT8 or not w_t.is_const:
            raise UnsupportedModel(f"{where}: constant int8 weights expected")
        sw = w_t.scale.astype(np.float32)
        if sw.size == 1:
            sw = np.repeat(sw, cout)
        if sw.size != cout or np.any(w_t.zero_point != 0):
            raise UnsupportedModel(f"{where}: symmetric per-channel weight quantisation expected")
        if b_t is None:
            bias = np.zeros(cout, "<i4")
        else:
            if b_t.type != TT_INT32 or not b_t.is_const or b_t.data.size != cout:
                raise UnsupportedModel(f"{where}: constant int32 bias expected")
            bias = np.asarray(b_t.data, "<i4").reshape(-1)
        mult = quant.conv_requant_scales(sx, sw, so)        # XNNPACK: (s_x * s_w[c]) / s_y in float32
        return bias, mult

    # ---- operator order.  A .tflite lists its operators in SOME topological order of the TF graph; which one depends on the converter.
    # The library's planner fuses chains of consecutive operators (expand -> depthwise -> project -> add, add -> depthwise -> project),
    # so the body is re-ordered chain-first: an operator whose result has exactly ONE consumer, along one of the edges the planner fuses
    # (conv -> depthwise, depthwise -> conv, conv -> add, add -> depthwise, add -> add), is followed by that consumer as soon as it is
    # ready; otherwise the ready operator that comes first in the file runs next.  A file whose chains are contiguous keeps its order
    # (a result with several consumers - a pyramid level feeding two heads and a resample - never pulls anything forward); one that
    # interleaves independent branches (two heads layer by layer, a lateral conv between the halves of a block) is untangled.
    chain_edges = {(BO_CONV_2D, BO_DEPTHWISE_CONV_2D), (BO_DEPTHWISE_CONV_2D, BO_CONV_2D), (BO_CONV_2D, BO_ADD), (BO_ADD, BO_DEPTHWISE_CONV_2D),
                   (BO_ADD, BO_ADD)}
    body = [oi for oi in range(len(m.ops)) if oi not in tail_ops]
    is_act = lambda t: t >= 0 and (t in producer or t == m.inputs[0])      # noqa: E731  (constants carry no ordering)
    waiting = {oi: {t for t in m.ops[oi].inputs if is_act(t)} for oi in body}
    avail = {m.inputs[0]}
    order, left, last = [], set(body), None
    while left:
        ready = [oi for oi in sorted(left) if waiting[oi] <= avail]
        if not ready:
            raise UnsupportedModel("operators are not in a valid order (an input is never produced)")
        def completes(oi):
            # a partial sum (an ADD whose only consumer is another ADD) stays next to the sum it feeds (the planner folds `partial, final`
            # pairs that are adjacent): it runs only once that second ADD's other input is there - it is neither pulled forward along a
            # chain nor taken in file order before (unless nothing else is ready)
            if m.ops[oi].code != BO_ADD:
                return True
            cons = consumers.get(m.ops[oi].outputs[0], [])
            if len(cons) != 1 or m.ops[cons[0]].code != BO_ADD:
                return True
            return all(t in avail or t in m.ops[oi].outputs for t in waiting[cons[0]])
        pick = None
        if last is not None:
            only = consumers.get(m.ops[last].outputs[0], []) if len(m.ops[last].outputs) == 1 else []
            follow = [oi for oi in ready if len(only) == 1 and oi == only[0] and (m.ops[last].code, m.ops[oi].code) in chain_edges and completes(oi)]
            pick = follow[0] if follow else None
        if pick is None:
            pick = next((oi for oi in ready if completes(oi)), ready[0])
        order.append(pick)
        left.remove(pick)
        avail.update(m.ops[pick].outputs)
        last = pick

    in_container = None
    for oi in order:
        op = m.ops[oi]
        where = f"op {oi} ({op.name})"
        if op.code == BO_QUANTIZE:
            ti, to = op.inputs[0], op.outputs[0]
            if ti != m.inputs[0] or T[ti].type != TT_UINT8 or T[to].type != TT_INT8:
                raise UnsupportedModel(f"{where}: only the uint8->int8 input conversion is supported")
            (s0, z0), (s1, z1) = T[ti].q(), T[to].q()
            if s0 != s1 or z0 - 128 != z1:
                raise UnsupportedModel(f"{where}: input re-scaling ({s0},{z0}) -> ({s1},{z1}) is not a pure offset")
            in_container = new_tensor(to)
            continue
        if op.code == BO_CONV_2D:
            x_t, w_t = T[op.inputs[0]], T[op.inputs[1]]
            b_t = T[op.inputs[2]] if len(op.inputs) > 2 and op.inputs[2] >= 0 else None
            pad, sw_, sh_ = op.opt(0, "b"), op.opt(1, "i", 1), op.opt(2, "i", 1)      # Conv2DOptions 0,1,2
            act = op.opt(3, "b")                                                     # fused_activation_function:3
            if op.opt(4, "i", 1) != 1 or op.opt(5, "i", 1) != 1:
                raise UnsupportedModel(f"{where}: dilation")
            cout, kh, kw_, cin = w_t.shape
            if op.inputs[0] == m.inputs[0] and in_container is None:
                # uint8 graph input feeding the stem directly is the pre-TF2 uint8 scheme
                raise UnsupportedModel(f"{where}: uint8 activations (no QUANTIZE on the input)")
            xi = src(op.inputs[0], where)
            so, zo = T[op.outputs[0]].q()
            sx = tensors[xi][4]
            bias, mult = conv_params(op, where, w_t, b_t, cout, sx, so)
            if kh == 1 and kw_ == 1 and sw_ == 1 and sh_ == 1:
                o = new_tensor(op.outputs[0])
                r = new_op(spec.OP_PW, [xi], o, act=_act_range(act, so, zo, where))
                r["w_off"] = add_const(np.ascontiguousarray(w_t.data.reshape(cout, cin)))
            elif kh == 3 and kw_ == 3 and sw_ == 2 and sh_ == 2 and cin == 3 and pad == PAD_SAME and xi == in_container:
                o = new_tensor(op.outputs[0])
                _, pt = _same_pad(x_t.shape[1], 3, 2)
                _, pl = _same_pad(x_t.shape[2], 3, 2)
                r = new_op(spec.OP_STEM, [xi], o, k=3, stride=2, pad_t=pt, pad_l=pl, act=_act_range(act, so, zo, where))
                r["w_off"] = add_const(np.ascontiguousarray(w_t.data))               # [Cout][ky][kx][Cin]
            else:
                raise UnsupportedModel(f"{where}: {kh}x{kw_} stride {sh_} convolution with Cin={cin} "
                                       "(only the 3x3/2 stem on the image and 1x1/1 convolutions are mapped)")
            r["b_off"], r["m_off"] = blob.add(bias), blob.add(mult)
            continue
        if op.code == BO_DEPTHWISE_CONV_2D:
            x_t, w_t = T[op.inputs[0]], T[op.inputs[1]]
            b_t = T[op.inputs[2]] if len(op.inputs) > 2 and op.inputs[2] >= 0 else None
            pad, sw_, sh_, dm = op.opt(0, "b"), op.opt(1, "i", 1), op.opt(2, "i", 1), op.opt(3, "i", 1)   # DepthwiseConv2DOptions 0..3
            act = op.opt(4, "b")
            if op.opt(5, "i", 1) != 1 or op.opt(6, "i", 1) != 1:
                raise UnsupportedModel(f"{where}: dilation")
            _, kh, kw_, c = w_t.shape
            if dm != 1 and c != x_t.shape[3]:
                raise UnsupportedModel(f"{where}: depth multiplier {dm}")
            if kh != kw_ or sw_ != sh_ or pad != PAD_SAME or c != x_t.shape[3]:
                raise UnsupportedModel(f"{where}: square kernel, equal strides and SAME padding expected")
            xi = src(op.inputs[0], where)
            so, zo = T[op.outputs[0]].q()
            bias, mult = conv_params(op, where, w_t, b_t, c, tensors[xi][4], so)
            o = new_tensor(op.outputs[0])
            _, pt = _same_pad(x_t.shape[1], kh, sh_)
            _, pl = _same_pad(x_t.shape[2], kh, sh_)
            r = new_op(spec.OP_DW, [xi], o, k=kh, stride=sh_, pad_t=pt, pad_l=pl, act=_act_range(act, so, zo, where))
            r["w_off"] = add_const(np.ascontiguousarray(w_t.data.reshape(kh, kw_, c)))   # [ky][kx][C]
            r["b_off"], r["m_off"] = blob.add(bias), blob.add(mult)
            continue
        if op.code == BO_ADD:
            if len(op.inputs) != 2:
                raise UnsupportedModel(f"{where}: binary ADD expected")
            ins = [src(t, where) for t in op.inputs]
            if T[op.inputs[0]].shape != T[op.inputs[1]].shape:
                raise UnsupportedModel(f"{where}: broadcasting ADD")
            so, zo = T[op.outputs[0]].q()
            o = new_tensor(op.outputs[0])
            r = new_op(spec.OP_ADD, ins, o, act=_act_range(op.opt(0, "b"), so, zo, where))   # AddOptions.fused_activation_function:0
            (sa, za), (sb, zb) = (tensors[ins[0]][4], tensors[ins[0]][3]), (tensors[ins[1]][4], tensors[ins[1]][3])
            r["in_mult"][0], r["in_mult"][1] = np.float32(sa) / np.float32(so), np.float32(sb) / np.float32(so)
            try:
                r["add_q"][:] = quant.xnn_qs8_add_params(sa, sb, so, za, zb)     # XNNPACK qs8-vadd-minmax, integer
            except ValueError as e:
                raise UnsupportedModel(f"{where}: {e}")
            continue
        if op.code == BO_MAX_POOL_2D:
            pad, sw_, sh_ = op.opt(0, "b"), op.opt(1, "i", 1), op.opt(2, "i", 1)      # Pool2DOptions 0,1,2
            fw, fh, act = op.opt(3, "i", 1), op.opt(4, "i", 1), op.opt(5, "b")       # filter_width:3 filter_height:4 act:5
            if (fw, fh, sw_, sh_) != (3, 3, 2, 2) or pad != PAD_SAME or act != ACT_NONE:
                raise UnsupportedModel(f"{where}: only 3x3 stride-2 SAME max pooling is mapped")
            xi = src(op.inputs[0], where)
            if T[op.inputs[0]].q() != T[op.outputs[0]].q():
                raise UnsupportedModel(f"{where}: input and output quantisation differ")
            o = new_tensor(op.outputs[0])
            x_t = T[op.inputs[0]]
            _, pt = _same_pad(x_t.shape[1], 3, 2)
            _, pl = _same_pad(x_t.shape[2], 3, 2)
            new_op(spec.OP_MAXPOOL, [xi], o, k=3, stride=2, pad_t=pt, pad_l=pl)
            continue
        if op.code == BO_RESIZE_NEAREST_NEIGHBOR:
            xi = src(op.inputs[0], where)
            x_t, o_t = T[op.inputs[0]], T[op.outputs[0]]
            integral = o_t.shape[1] % x_t.shape[1] == 0 and o_t.shape[2] % x_t.shape[2] == 0
            if op.opt(0, "?", False) or (op.opt(1, "?", False) and not integral):    # align_corners:0 half_pixel_centers:1
                raise UnsupportedModel(f"{where}: align_corners / half_pixel_centers rounding is not mapped")
            if x_t.q() != o_t.q():
                raise UnsupportedModel(f"{where}: input and output quantisation differ")
            o = new_tensor(op.outputs[0])
            new_op(spec.OP_RESIZE_NN, [xi], o)
            continue
        raise UnsupportedModel(f"{where}: operator is not part of the EfficientDet-Lite int8 graph this library maps")

    # ---- heads: feature level of every op that belongs to one of the 10 chains (lets the planner batch them)
    out_of = {int(r["output"]): i for i, r in enumerate(ops)}
    ncons = {}
    for r in ops:
        for j in range(int(r["n_inputs"])):
            ncons[int(r["inputs"][j])] = ncons.get(int(r["inputs"][j]), 0) + 1
    heads_in = []
    for li in range(5):
        for lv in (cls_levels[li], box_levels[li]):
            t = src(lv, POSTPROCESS_NAME)
            while t in out_of:
                r = ops[out_of[t]]
                if r["type"] not in (spec.OP_PW, spec.OP_DW) or ncons.get(t, 0) > 1:
                    break
                r["level"] = 3 + li
                t = int(r["inputs"][0])
    for li in range(5):
        c, b = tensors[tmap[cls_levels[li]]], tensors[tmap[box_levels[li]]]
        if (c[0], c[1]) != (b[0], b[1]) or c[2] % num_classes or b[2] != 4 * (c[2] // num_classes):
            raise UnsupportedModel(f"class / box head shapes do not match (A anchors x {num_classes} classes, A x 4)")
    n_anch = sum(tensors[tmap[t]][0] * tensors[tmap[t]][1] * tensors[tmap[t]][2] // num_classes for t in cls_levels)
    if anchors.shape[0] != n_anch:
        raise UnsupportedModel(f"{anchors.shape[0]} anchors for {n_anch} head outputs")
    if max_det != spec.MAX_DETECTIONS:
        raise UnsupportedModel(f"max_detections = {max_det}: the boundary is fixed at {spec.MAX_DETECTIONS} "
                               "(reference dfs/eval_detections.pkl.gz)")

    det = len(tensors)
    tensors.append((1, spec.MAX_DETECTIONS, 6, 0, np.float32(1.0)))
    r = new_op(spec.OP_POSTPROCESS, [tmap[t] for t in cls_levels] + [tmap[t] for t in box_levels], det)
    r["aux_off"] = blob.add(anchors)
    sc, zc = cls_q
    sb, zb = box_q
    if cls_log is None:
        raise UnsupportedModel("class branch without LOGISTIC: raw logits as scores are not mapped")
    so, zo = T[cls_log].q()
    try:
        # LOGISTIC (XNNPACK x8-lut), DEQUANTIZE and the double-precision decode tables of detection_postprocess.cc
        r["aux2_off"] = blob.add(quant.pack_postprocess_tables(sc, zc, sb, zb, y_scale=y_scale, h_scale=h_scale))
        quant.xnn_qs8_sigmoid_lut(sc, zc, so, zo)
    except ValueError as e:
        raise UnsupportedModel(f"{POSTPROCESS_NAME}: {e}")

    tarr = np.zeros(len(tensors), TENSOR_DTYPE)
    for i, (h, w, c, z, s) in enumerate(tensors):
        tarr[i] = (h, w, c, z, s, (0, 0, 0))
    oarr = np.zeros(len(ops), OP_DTYPE)
    for i, r in enumerate(ops):
        oarr[i] = r
    arch = {320: 0, 384: 1, 448: 2}.get(S, -1)
    header = dict(arch=arch, image_size=S, num_anchors=n_anch, max_detections=max_det,
                  nms_iou_threshold=float(opts.get("nms_iou_threshold", 0.5)),
                  nms_score_threshold=float(opts.get("nms_score_threshold", 0.0)), input_tensor=in_container or 0, num_classes=num_classes)
    if in_container != 0:
        raise UnsupportedModel("the quantised image must be the first activation of the graph")
    return header, tarr, oarr, blob.bytes()


def convert(tflite_path, out_path):
    header, tensors, ops, blob = import_tflite(tflite_path)
    write_container(out_path, header, tensors, ops, blob)
    return out_path


def as_container_path(model_path):
    """`model_path` itself when it is a VBTM container; for a .tflite, the path of a converted temporary container
    (the caller removes it after vbt_model_create has read it)."""
    if not is_tflite(model_path):
        return str(model_path), False
    fd, tmp = tempfile.mkstemp(suffix=".vbtm", prefix="vbt_import_")
    os.close(fd)
    try:
        convert(model_path, tmp)
    except Exception:
        os.unlink(tmp)
        raise
    return tmp, True
