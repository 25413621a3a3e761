#!/usr/bin/env python3
"""Write a VBTM container as a TFLite flatbuffer shaped like the converter's full-integer EfficientDet-Lite export.

Test tooling for vbt_amd/tflite_import.py: the reference's own `.tflite` files are absent
(.MISSING_LARGE_BLOBS), so the importer is exercised on files produced here.  The layout follows what the TFLite
converter emits for the reference's `export(... quantization_config=...)` call (reference train.py:58-70):

  uint8 image -> QUANTIZE -> int8 graph of CONV_2D / DEPTHWISE_CONV_2D / ADD / MAX_POOL_2D / RESIZE_NEAREST_NEIGHBOR
  -> per level RESHAPE -> CONCATENATION -> (class branch) LOGISTIC -> DEQUANTIZE -> TFLite_Detection_PostProcess

A 3-input BiFPN sum is written as two chained binary ADDs (TFLite's ADD is binary), the first with its own
quantisation of the partial sum.  Per-channel weight scales are recovered from the container's requantisation
multipliers so that `s_x * s_w[c] / s_y` evaluates back to the stored float32 value.

usage: python tools/export_tflite.py models/efficientdet_lite0_synth.vbtm out.tflite
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from vbt_amd import spec  # noqa: E402
from vbt_amd.container import Container  # noqa: E402
from vbt_amd.flatbuf import Scalar, TableSpec, Vec, Writer, flex_map  # noqa: E402
from vbt_amd.tflite_import import (ACT_NONE, ACT_RELU6, BO_ADD, BO_CONCATENATION, BO_CONV_2D, BO_CUSTOM,  # noqa: E402
                                   BO_DEPTHWISE_CONV_2D, BO_DEQUANTIZE, BO_LOGISTIC, BO_MAX_POOL_2D, BO_QUANTIZE,
                                   BO_RESHAPE, BO_RESIZE_NEAREST_NEIGHBOR, PAD_SAME, POSTPROCESS_NAME, TT_FLOAT32,
                                   TT_INT8, TT_INT32, TT_UINT8)

# BuiltinOptions union tags (schema.fbs `union BuiltinOptions`)
OPT_CONV, OPT_DW, OPT_POOL, OPT_CONCAT, OPT_ADD, OPT_RESHAPE, OPT_RESIZE_NN = 1, 2, 5, 10, 11, 17, 74


def weight_scales(mult, sx, so):
    """float32 s_w[c] with float32((sx*s_w)/so) == mult[c] (searching a few ulps around the quotient)."""
    sx, so = np.float32(sx), np.float32(so)
    base = (mult.astype(np.float32) * so / sx).astype(np.float32)
    cands = [base]
    for sign in (1, -1):
        cand = base
        for _ in range(4):
            cand = np.nextafter(cand, np.float32(np.inf * sign), dtype=np.float32)
            cands.append(cand)
    cands = np.stack(cands)
    err = np.abs(((sx * cands) / so).astype(np.float32).view(np.int32).astype(np.int64) - mult.view(np.int32).astype(np.int64))
    pick = err.argmin(axis=0)
    best = cands[pick, np.arange(mult.size)]
    ok = err.min(axis=0) == 0
    return best, ok


class Exporter:
    def __init__(self, c: Container, chain_adds=True, extra_op=None, half_pixel=False):
        self.c = c
        self.tensors = []        # TableSpec
        self.buffers = [np.empty(0, np.uint8)]
        self.ops = []
        self.codes = []
        self.code_idx = {}
        self.chain_adds = chain_adds
        self.extra_op = extra_op
        self.half_pixel = half_pixel
        self.inexact = 0

    def tensor(self, name, shape, ttype, scale=None, zp=None, data=None, qdim=0):
        buf = 0
        if data is not None:
            buf = len(self.buffers)
            self.buffers.append(np.frombuffer(np.ascontiguousarray(data).tobytes(), np.uint8))
        quant = None
        if scale is not None:
            quant = TableSpec({2: Vec(np.atleast_1d(np.asarray(scale, np.float32))),
                               3: Vec(np.atleast_1d(np.asarray(zp, np.int64)), align=8),
                               6: Scalar("i", qdim) if qdim else None})
        self.tensors.append(TableSpec({0: Vec(np.asarray(shape, np.int32)), 1: Scalar("b", ttype) if ttype else None,
                                       2: Scalar("I", buf) if buf else None, 3: name, 4: quant}))
        return len(self.tensors) - 1

    def code(self, builtin, custom=None):
        key = (builtin, custom)
        if key not in self.code_idx:
            self.code_idx[key] = len(self.codes)
            self.codes.append(TableSpec({0: Scalar("b", min(builtin, 127)), 1: custom, 2: Scalar("i", 1),
                                         3: Scalar("i", builtin)}))
        return self.code_idx[key]

    def op(self, builtin, inputs, outputs, opt_type=0, options=None, custom=None, custom_options=None):
        self.ops.append(TableSpec({
            0: Scalar("I", self.code(builtin, custom)) if self.code(builtin, custom) else None,
            1: Vec(np.asarray(inputs, np.int32)), 2: Vec(np.asarray(outputs, np.int32)),
            3: Scalar("B", opt_type) if opt_type else None, 4: options,
            5: Vec(np.frombuffer(custom_options, np.uint8)) if custom_options is not None else None}))

    def build(self):
        c = self.c
        S = int(c.header["image_size"])
        T = c.tensors
        tid = {}

        def act_of(r, t):
            return ACT_NONE if (int(r["act_min"]), int(r["act_max"])) == (-128, 127) else ACT_RELU6

        img = self.tensor("serving_default_images:0", [1, S, S, 3], TT_UINT8, T[0]["scale"], int(T[0]["zero_point"]) + 128)
        tid[0] = self.tensor("tfl.quantize", [1, S, S, 3], TT_INT8, T[0]["scale"], int(T[0]["zero_point"]))
        self.op(BO_QUANTIZE, [img], [tid[0]])

        def act_tensor(i, name):
            t = T[i]
            tid[i] = self.tensor(name, [1, int(t["h"]), int(t["w"]), int(t["c"])], TT_INT8, t["scale"], int(t["zero_point"]))
            return tid[i]

        post = None
        for oi, r in enumerate(c.ops):
            typ = int(r["type"])
            ins = [int(v) for v in r["inputs"][:int(r["n_inputs"])]]
            out = int(r["output"])
            if typ in (spec.OP_STEM, spec.OP_PW, spec.OP_DW):
                ti, to = T[ins[0]], T[out]
                cin, cout, k = int(ti["c"]), int(to["c"]), int(r["k"])
                mult = np.array(c.f32(int(r["m_off"]), cout))
                sw, ok = weight_scales(mult, ti["scale"], to["scale"])
                self.inexact += int((~ok).sum())
                bias = np.array(c.i32(int(r["b_off"]), cout))
                if typ == spec.OP_DW:
                    w = np.array(c.i8(int(r["w_off"]), k * k * cout)).reshape(1, k, k, cout)
                    wt = self.tensor(f"op{oi}/depthwise_weights", w.shape, TT_INT8, sw, np.zeros(cout, np.int64), w, qdim=3)
                else:
                    w = np.array(c.i8(int(r["w_off"]), cout * k * k * cin)).reshape(cout, k, k, cin)
                    wt = self.tensor(f"op{oi}/weights", w.shape, TT_INT8, sw, np.zeros(cout, np.int64), w)
                bt = self.tensor(f"op{oi}/bias", [cout], TT_INT32, (np.float32(ti["scale"]) * sw).astype(np.float32),
                                 np.zeros(cout, np.int64), bias.astype("<i4"))
                o = act_tensor(out, f"op{oi}/out")
                s = int(r["stride"])
                if typ == spec.OP_DW:
                    opts = TableSpec({0: None, 1: Scalar("i", s), 2: Scalar("i", s), 3: Scalar("i", 1),
                                      4: Scalar("b", act_of(r, to)) if act_of(r, to) else None})
                    self.op(BO_DEPTHWISE_CONV_2D, [tid[ins[0]], wt, bt], [o], OPT_DW, opts)
                else:
                    opts = TableSpec({1: Scalar("i", s), 2: Scalar("i", s),
                                      3: Scalar("b", act_of(r, to)) if act_of(r, to) else None})
                    self.op(BO_CONV_2D, [tid[ins[0]], wt, bt], [o], OPT_CONV, opts)
            elif typ == spec.OP_ADD:
                to = T[out]
                cur = tid[ins[0]]
                lo = float(T[ins[0]]["scale"]) * (-128 - int(T[ins[0]]["zero_point"]))
                hi = float(T[ins[0]]["scale"]) * (127 - int(T[ins[0]]["zero_point"]))
                for j in range(1, len(ins) - 1):
                    tj = T[ins[j]]
                    lo += float(tj["scale"]) * (-128 - int(tj["zero_point"]))
                    hi += float(tj["scale"]) * (127 - int(tj["zero_point"]))
                    sc = np.float32((hi - lo) / 255.0)
                    zp = int(np.clip(np.rint(-128 - lo / float(sc)), -128, 127))
                    part = self.tensor(f"op{oi}/partial{j}", [1, int(to["h"]), int(to["w"]), int(to["c"])], TT_INT8, sc, zp)
                    self.op(BO_ADD, [cur, tid[ins[j]]], [part], OPT_ADD, TableSpec({}))
                    cur = part
                o = act_tensor(out, f"op{oi}/add")
                a = act_of(r, to)
                self.op(BO_ADD, [cur, tid[ins[-1]]], [o], OPT_ADD, TableSpec({0: Scalar("b", a) if a else None}))
            elif typ == spec.OP_MAXPOOL:
                o = act_tensor(out, f"op{oi}/pool")
                self.op(BO_MAX_POOL_2D, [tid[ins[0]]], [o], OPT_POOL,
                        TableSpec({1: Scalar("i", 2), 2: Scalar("i", 2), 3: Scalar("i", 3), 4: Scalar("i", 3)}))
            elif typ == spec.OP_RESIZE_NN:
                o = act_tensor(out, f"op{oi}/resize")
                size = self.tensor(f"op{oi}/size", [2], TT_INT32, data=np.array([T[out]["h"], T[out]["w"]], "<i4"))
                self.op(BO_RESIZE_NEAREST_NEIGHBOR, [tid[ins[0]], size], [o], OPT_RESIZE_NN,
                        TableSpec({1: Scalar("?", True) if self.half_pixel else None}))
            elif typ == spec.OP_POSTPROCESS:
                post = (r, ins)
            else:
                raise ValueError(f"op type {typ}")
            if self.extra_op is not None and oi == self.extra_op[0]:
                t = T[out]
                x = self.tensor(f"op{oi}/extra", [1, int(t["h"]), int(t["w"]), int(t["c"])], TT_INT8, t["scale"], int(t["zero_point"]))
                self.op(self.extra_op[1], [tid[out]], [x])

        r, ins = post
        A = int(c.header["num_anchors"])
        NC = max(int(c.header["num_classes"]), 1)          # class columns per anchor
        branches = []
        for name, levels, width in (("class", ins[:5], NC), ("box", ins[5:], 4)):
            parts = []
            q = T[levels[0]]
            for li, t in enumerate(levels):
                n = int(T[t]["h"]) * int(T[t]["w"]) * int(T[t]["c"]) // width
                shp = self.tensor(f"{name}{li}/shape", [3], TT_INT32, data=np.array([1, n, width], "<i4"))
                rs = self.tensor(f"{name}{li}/reshape", [1, n, width], TT_INT8, T[t]["scale"], int(T[t]["zero_point"]))
                self.op(BO_RESHAPE, [tid[t], shp], [rs], OPT_RESHAPE, TableSpec({}))
                parts.append(rs)
            cat = self.tensor(f"{name}/concat", [1, A, width], TT_INT8, q["scale"], int(q["zero_point"]))
            self.op(BO_CONCATENATION, parts, [cat], OPT_CONCAT, TableSpec({0: Scalar("i", 1)}))
            if name == "class":
                lg = self.tensor("class/logistic", [1, A, NC], TT_INT8, np.float32(1.0 / 256.0), -128)
                self.op(BO_LOGISTIC, [cat], [lg])
                cat = lg
            dq = self.tensor(f"{name}/dequantize", [1, A, width], TT_FLOAT32)
            self.op(BO_DEQUANTIZE, [cat], [dq])
            branches.append(dq)
        anchors = np.array(c.f32(int(r["aux_off"]), A * 4)).reshape(A, 4)
        at = self.tensor("anchors", [A, 4], TT_FLOAT32, data=anchors.astype("<f4"))
        outs = [self.tensor(n, s, TT_FLOAT32) for n, s in (("StatefulPartitionedCall:3", [1, 25, 4]), ("StatefulPartitionedCall:2", [1, 25]),
                                                           ("StatefulPartitionedCall:1", [1, 25]), ("StatefulPartitionedCall:0", [1]))]
        fo = flex_map({"max_detections": int(c.header["max_detections"]), "max_classes_per_detection": 1,
                       "detections_per_class": 100, "use_regular_nms": False,
                       "nms_score_threshold": float(c.header["nms_score_threshold"]),
                       "nms_iou_threshold": float(c.header["nms_iou_threshold"]), "num_classes": NC,
                       "y_scale": 1.0, "x_scale": 1.0, "h_scale": 1.0, "w_scale": 1.0})
        self.op(BO_CUSTOM, [branches[1], branches[0], at], outs, custom=POSTPROCESS_NAME, custom_options=fo)

        sub = TableSpec({0: Vec(self.tensors), 1: Vec(np.asarray([img], np.int32)), 2: Vec(np.asarray(outs, np.int32)),
                         3: Vec(self.ops), 4: "main"})
        bufs = [TableSpec({0: Vec(b, align=16) if b.size else None}) for b in self.buffers]
        root = TableSpec({0: Scalar("I", 3), 1: Vec(self.codes), 2: Vec([sub]), 3: "vbt_amd synthetic export", 4: Vec(bufs)})
        return Writer().finish(root)


def export(container_path, out_path, **kw):
    e = Exporter(Container(container_path), **kw)
    data = e.build()
    with open(out_path, "wb") as f:
        f.write(data)
    return e


if __name__ == "__main__":
    e = export(sys.argv[1], sys.argv[2])
    print(f"wrote {sys.argv[2]}: {os.path.getsize(sys.argv[2]) / 1e6:.2f} MB, {len(e.ops)} operators, {len(e.tensors)} tensors, "
          f"{e.inexact} multipliers not exactly recoverable")
