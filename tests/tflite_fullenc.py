"""Test infrastructure: a FULL-SIZE EfficientDet-Lite `.tflite` written by the independent encoder of tests/tflite_minienc.py in the
conventions of a converter-written file - the shape in which the reference's models/efficientdet_lite0_whole.tflite (absent from the
tree: reference .MISSING_LARGE_BLOBS, track.py:67,93) would arrive:

  * every constant tensor (weights, biases, reshape shapes, anchors) ahead of the activations in the tensor table, the activations
    in the REVERSE of their production order, TF-style tensor names: nothing may depend on tensor numbering;
  * operators in a topological order that is NOT the native container's: among the ready operators the one with the highest native
    index goes first, so lateral convs, resamples and head layers interleave with the blocks they do not belong to (a converter
    emits some valid order of the TF graph; which one is not known here);
  * per-channel weight quantisation with quantized_dimension (0 conv / 3 depthwise), int32 biases with scale s_x * s_w, fused RELU6
    as fused_activation_function, three-input BiFPN sums as two binary ADDs with a partial-sum tensor of its own quantisation, the
    RESHAPE -> CONCATENATION -> LOGISTIC -> DEQUANTIZE tail and TFLite_Detection_PostProcess with FlexBuffers custom options.

The numbers (weights, quantisation parameters, anchors) come from a native container, so that the imported file can be compared with
the oracle on that container: same detections, bit for bit."""
import os
import sys

import numpy as np

from tflite_minienc import (ACT_RELU6, BO_ADD, BO_CONCATENATION, BO_CONV_2D, BO_CUSTOM, BO_DEPTHWISE_CONV_2D, BO_DEQUANTIZE, BO_LOGISTIC,
                            BO_MAX_POOL_2D, BO_QUANTIZE, BO_RESHAPE, TT_FLOAT32, TT_INT8, TT_INT32, TT_UINT8, S, TinyModel, flexbuffer_map)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BO_RESIZE_NEAREST_NEIGHBOR = 97          # schema.fbs BuiltinOperator; ResizeNearestNeighborOptions = union tag 74


class ConverterStyleModel(TinyModel):
    def __init__(self, container_path, order="max"):
        """order: "max" = the ready operator with the highest native index first; an int = a seeded RANDOM topological order (chains of
        one block end up interleaved with operators of other blocks, levels and heads)."""
        super().__init__(S=0, seed=0)
        self.order_mode = order
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from vbt_amd.container import Container
        self.c = Container(container_path)

    def build(self):
        from export_tflite import weight_scales
        from vbt_amd import spec
        c = self.c
        T, f32 = c.tensors, np.float32
        size = int(c.header["image_size"])
        A = int(c.header["num_anchors"])
        # ---- 1. the graph as abstract nodes: (kind, input activation ids, output activation id, constants, options) ----
        nodes = []              # native order
        acts = {}               # activation id -> (shape, scale, zp, name)
        nxt = [len(T)]          # ids of the partial-sum tensors follow the container's

        def act(i, name):
            t = T[i]
            acts[i] = ([1, int(t["h"]), int(t["w"]), int(t["c"])], f32(t["scale"]), int(t["zero_point"]), name)

        acts["img"] = ([1, size, size, 3], f32(T[0]["scale"]), int(T[0]["zero_point"]) + 128, "serving_default_images:0")
        act(0, "tfl.quantize")
        nodes.append(dict(kind="quantize", ins=["img"], out=0))
        post = None
        for oi, r in enumerate(c.ops):
            typ = int(r["type"])
            ins = [int(v) for v in r["inputs"][:int(r["n_inputs"])]]
            out = int(r["output"])
            relu6 = (int(r["act_min"]), int(r["act_max"])) != (-128, 127)
            if typ in (spec.OP_STEM, spec.OP_PW, spec.OP_DW):
                ti, to = T[ins[0]], T[out]
                cin, cout, k = int(ti["c"]), int(to["c"]), int(r["k"])
                sw, ok = weight_scales(np.array(c.f32(int(r["m_off"]), cout)), ti["scale"], to["scale"])
                assert ok.all(), "multipliers of the source container must be recoverable as per-channel weight scales"
                bias = np.array(c.i32(int(r["b_off"]), cout)).astype("<i4")
                dw = typ == spec.OP_DW
                w = np.array(c.i8(int(r["w_off"]), k * k * cout if dw else cout * k * k * cin)).reshape((1, k, k, cout) if dw else (cout, k, k, cin))
                act(out, f"efficientdet-lite/{'depthwise_conv2d' if dw else 'conv2d'}_{oi}/BiasAdd;Relu6" if relu6 else f"efficientdet-lite/{'depthwise_conv2d' if dw else 'conv2d'}_{oi}/BiasAdd")
                nodes.append(dict(kind="dw" if dw else "conv", ins=[ins[0]], out=out, w=w, sw=sw, bias=bias, bscale=(f32(ti["scale"]) * sw).astype(np.float32),
                                  stride=int(r["stride"]), relu6=relu6, oi=oi))
            elif typ == spec.OP_ADD:
                to = T[out]
                cur = ins[0]
                lo = float(T[ins[0]]["scale"]) * (-128 - int(T[ins[0]]["zero_point"]))
                hi = float(T[ins[0]]["scale"]) * (127 - int(T[ins[0]]["zero_point"]))
                for j in range(1, len(ins) - 1):              # sum(nodes) of the Keras model: ((n0 + n1) + n2), the partial sum with its own range
                    tj = T[ins[j]]
                    lo += float(tj["scale"]) * (-128 - int(tj["zero_point"]))
                    hi += float(tj["scale"]) * (127 - int(tj["zero_point"]))
                    sc = f32((hi - lo) / 255.0)
                    zp = int(np.clip(np.rint(-128 - lo / float(sc)), -128, 127))
                    pid = nxt[0]
                    nxt[0] += 1
                    acts[pid] = ([1, int(to["h"]), int(to["w"]), int(to["c"])], sc, zp, f"efficientdet-lite/fpn_cells/add_{oi}_{j}")
                    nodes.append(dict(kind="add", ins=[cur, ins[j]], out=pid, relu6=False))
                    cur = pid
                act(out, f"efficientdet-lite/add_{oi}")
                nodes.append(dict(kind="add", ins=[cur, ins[-1]], out=out, relu6=relu6))
            elif typ == spec.OP_MAXPOOL:
                act(out, f"efficientdet-lite/max_pooling2d_{oi}/MaxPool")
                nodes.append(dict(kind="pool", ins=[ins[0]], out=out))
            elif typ == spec.OP_RESIZE_NN:
                act(out, f"efficientdet-lite/resize_{oi}/ResizeNearestNeighbor")
                nodes.append(dict(kind="resize", ins=[ins[0]], out=out, size=np.array([T[out]["h"], T[out]["w"]], "<i4")))
            elif typ == spec.OP_POSTPROCESS:
                post = (r, ins)
            else:
                raise ValueError(f"op type {typ}")
        # ---- 2. emission order: topological, highest native index among the ready operators first ----
        produced, order, left = {"img"}, [], list(range(len(nodes)))
        rng = None if self.order_mode == "max" else np.random.default_rng(int(self.order_mode))
        while left:
            ready = [i for i in left if all(t in produced for t in nodes[i]["ins"])]
            i = max(ready) if rng is None else int(rng.choice(ready))
            order.append(i)
            produced.add(nodes[i]["out"])
            left.remove(i)
        assert order != sorted(order)
        # ---- 3. tensor table: constants first (emission order), then activations in reverse production order ----
        for i in order:
            n = nodes[i]
            if n["kind"] in ("conv", "dw"):
                cout = n["bias"].size
                base = f"efficientdet-lite/{'depthwise_conv2d' if n['kind'] == 'dw' else 'conv2d'}_{n['oi']}"
                n["wt"] = self.tensor(base + ("/depthwise_kernel" if n["kind"] == "dw" else "/kernel"), n["w"].shape, TT_INT8, list(n["sw"]), [0] * cout,
                                      buffer=self.buf(n["w"]), qdim=3 if n["kind"] == "dw" else 0)
                n["bt"] = self.tensor(base + "/bias", [cout], TT_INT32, list(n["bscale"]), [0] * cout, buffer=self.buf(n["bias"]))
            elif n["kind"] == "resize":
                n["st"] = self.tensor(f"efficientdet-lite/resize_{i}/size", [2], TT_INT32, buffer=self.buf(n["size"]))
        r, pins = post
        NC = max(int(c.header["num_classes"]), 1)          # class columns per anchor
        levels = {"class": (pins[:5], NC), "box": (pins[5:], 4)}
        shape_t = {}
        for name, (lv, width) in levels.items():
            for li, t in enumerate(lv):
                nel = int(T[t]["h"]) * int(T[t]["w"]) * int(T[t]["c"]) // width
                shape_t[(name, li)] = (self.tensor(f"{name}_net/reshape_{li}/shape", [3], TT_INT32, buffer=self.buf(np.array([1, nel, width], "<i4"))), nel)
        anchors = np.array(c.f32(int(r["aux_off"]), A * 4)).reshape(A, 4).astype("<f4")
        at = self.tensor("anchors", [A, 4], TT_FLOAT32, buffer=self.buf(anchors))
        tid = {}
        prod_order = ["img"] + [nodes[i]["out"] for i in order]
        for a in reversed(prod_order):
            shp, sc, zp, name = acts[a]
            tid[a] = self.tensor(name, shp, TT_UINT8 if a == "img" else TT_INT8, [sc], [zp])
        # ---- 4. operators ----
        for i in order:
            n = nodes[i]
            ins, out = [tid[t] for t in n["ins"]], tid[n["out"]]
            act_ = ACT_RELU6 if n.get("relu6") else 0
            if n["kind"] == "quantize":
                self.op(BO_QUANTIZE, ins, [out])
            elif n["kind"] == "conv":    # Conv2DOptions: padding:0 (SAME = 0) stride_w:1 stride_h:2 fused_activation_function:3
                self.op(BO_CONV_2D, ins + [n["wt"], n["bt"]], [out], 1, {1: S("i", n["stride"]), 2: S("i", n["stride"]), 3: S("b", act_)})
            elif n["kind"] == "dw":      # DepthwiseConv2DOptions: stride_w:1 stride_h:2 depth_multiplier:3 fused_activation_function:4
                self.op(BO_DEPTHWISE_CONV_2D, ins + [n["wt"], n["bt"]], [out], 2, {1: S("i", n["stride"]), 2: S("i", n["stride"]), 3: S("i", 1), 4: S("b", act_)})
            elif n["kind"] == "add":     # AddOptions.fused_activation_function:0
                self.op(BO_ADD, ins, [out], 11, {0: S("b", act_)})
            elif n["kind"] == "pool":    # Pool2DOptions: stride_w:1 stride_h:2 filter_width:3 filter_height:4
                self.op(BO_MAX_POOL_2D, ins, [out], 5, {1: S("i", 2), 2: S("i", 2), 3: S("i", 3), 4: S("i", 3)})
            elif n["kind"] == "resize":  # ResizeNearestNeighborOptions (union tag 74): align_corners:0 half_pixel_centers:1 both false
                self.op(BO_RESIZE_NEAREST_NEIGHBOR, ins + [n["st"]], [out], 74, {})
        cats = {}
        for name, (lv, width) in levels.items():
            q = T[lv[0]]
            parts = []
            for li, t in enumerate(lv):
                st, nel = shape_t[(name, li)]
                rs = self.tensor(f"{name}_net/reshape_{li}", [1, nel, width], TT_INT8, [f32(T[t]["scale"])], [int(T[t]["zero_point"])])
                self.op(BO_RESHAPE, [tid[t], st], [rs], 17, {})
                parts.append(rs)
            cat = self.tensor(f"{name}_net/concat", [1, A, width], TT_INT8, [f32(q["scale"])], [int(q["zero_point"])])
            self.op(BO_CONCATENATION, parts, [cat], 10, {0: S("i", 1)})
            if name == "class":
                lg = self.tensor("class_net/Sigmoid", [1, A, NC], TT_INT8, [f32(1 / 256)], [-128])
                self.op(BO_LOGISTIC, [cat], [lg])
                cat = lg
            dq = self.tensor(f"{name}_net/dequantize", [1, A, width], TT_FLOAT32)
            self.op(BO_DEQUANTIZE, [cat], [dq])
            cats[name] = dq
        outs = [self.tensor(n_, s_, TT_FLOAT32) for n_, s_ in (("StatefulPartitionedCall:3", [1, 25, 4]), ("StatefulPartitionedCall:2", [1, 25]),
                                                               ("StatefulPartitionedCall:1", [1, 25]), ("StatefulPartitionedCall:0", [1]))]
        self.options = {"max_detections": int(c.header["max_detections"]), "max_classes_per_detection": 1, "detections_per_class": 100,
                        "use_regular_nms": False, "nms_score_threshold": float(c.header["nms_score_threshold"]),
                        "nms_iou_threshold": float(c.header["nms_iou_threshold"]), "num_classes": NC,
                        "y_scale": 1.0, "x_scale": 1.0, "h_scale": 1.0, "w_scale": 1.0}
        self.op(BO_CUSTOM, [cats["box"], cats["class"], at], outs, custom="TFLite_Detection_PostProcess", custom_options=flexbuffer_map(self.options))
        self.image, self.outputs, self.n_ops_native, self.order = tid["img"], outs, len(nodes), order
        return self
