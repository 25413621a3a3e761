"""Test infrastructure: a second, independently written TFLite file encoder.

vbt_amd/tflite_import.py is otherwise only ever fed files written by tools/export_tflite.py through the product's own
front-to-back flatbuffer Writer.  This module shares no code with either: it is a BACK-TO-FRONT FlatBuffers builder in the
style of the official libraries (objects are prepended to a growing buffer, children before parents, vtables
de-duplicated, offsets computed from the end of the buffer) and a FlexBuffers map writer, both written from the public
format descriptions (FlatBuffers "internals", FlexBuffers "internals") and the field ids of the published TFLite
schema.fbs, which are spelled out next to every use.
"""
import struct

import numpy as np


class Builder:
    """Minimal back-to-front FlatBuffers builder.  Offsets handed out are distances from the END of the buffer."""

    def __init__(self):
        self.buf = bytearray()          # grows at the FRONT: self.buf = new + self.buf
        self.minalign = 1
        self.vtables = {}

    def _off(self):
        return len(self.buf)

    def _pad(self, n):
        self.buf = bytearray(n) + self.buf

    def prep(self, size, additional):
        """make sure that after writing `additional` bytes, a `size`-aligned value can be written"""
        self.minalign = max(self.minalign, size)
        pad = (-(len(self.buf) + additional)) % size
        self._pad(pad)

    def _push(self, raw):
        self.buf = bytearray(raw) + self.buf

    def string(self, s):
        b = s.encode("utf-8")
        self.prep(4, len(b) + 1)
        self._push(b + b"\0")
        self._push(struct.pack("<I", len(b)))
        return self._off()

    def bytes_vector(self, raw, elem_size, align=None):
        raw = bytes(raw)
        n = len(raw) // elem_size
        self.prep(4, len(raw))
        self.prep(max(elem_size, align or 1), len(raw))
        self._push(raw)
        self._push(struct.pack("<I", n))
        return self._off()

    def np_vector(self, arr, align=None):
        arr = np.ascontiguousarray(arr)
        return self.bytes_vector(arr.astype(arr.dtype.newbyteorder("<")).tobytes(), arr.dtype.itemsize, align)

    def offset_vector(self, offs):
        self.prep(4, 4 * len(offs))
        for o in reversed(offs):            # element i sits at buffer position p_i; value = p_target - p_i
            self.prep(4, 0)
            here = self._off() + 4
            self._push(struct.pack("<I", here - o))
        self._push(struct.pack("<I", len(offs)))
        return self._off()

    def table(self, fields):
        """fields: {field_id: ('scalar', fmt, value) | ('offset', off)}; returns the table's offset"""
        if not fields:
            slots = []
        else:
            slots = [None] * (max(fields) + 1)
            for k, v in fields.items():
                slots[k] = v
        # write the fields back to front, largest first to keep padding small; remember where each lands
        order = sorted([i for i, v in enumerate(slots) if v is not None],
                       key=lambda i: (4 if slots[i][0] == "offset" else struct.calcsize("<" + slots[i][1])), reverse=False)
        end_before = self._off()
        pos = {}
        for i in order:
            v = slots[i]
            if v[0] == "offset":
                self.prep(4, 0)
                here = self._off() + 4
                self._push(struct.pack("<I", here - v[1]))
            else:
                size = struct.calcsize("<" + v[1])
                self.prep(size, 0)
                self._push(struct.pack("<" + v[1], v[2]))
            pos[i] = self._off()
        self.prep(4, 0)
        # soffset placeholder
        self._push(b"\0\0\0\0")
        table_off = self._off()
        table_size = table_off - end_before
        vt = [4 + 2 * len(slots), table_size] + [(table_off - pos[i]) if i in pos else 0 for i in range(len(slots))]
        vt_raw = struct.pack("<%dH" % len(vt), *vt)
        if vt_raw in self.vtables:
            vt_off = self.vtables[vt_raw]
        else:
            self.prep(2, len(vt_raw))
            self._push(vt_raw)
            vt_off = self._off()
            self.vtables[vt_raw] = vt_off
        # soffset = table_pos - vtable_pos (positions from the buffer START): = vt_off - table_off in end-distances
        idx = len(self.buf) - table_off
        self.buf[idx:idx + 4] = struct.pack("<i", vt_off - table_off)
        return table_off

    def finish(self, root_off, ident):
        self.prep(self.minalign, 8)
        self._push(ident)
        here = self._off() + 4
        self._push(struct.pack("<I", here - root_off))
        return bytes(self.buf)


def S(fmt, v):
    return ("scalar", fmt, v)


def O(off):
    return ("offset", off)


# ---- FlexBuffers: a flat map of scalars (what TFLite_Detection_PostProcess carries) -------------------------------
FBT_INT, FBT_FLOAT, FBT_KEY, FBT_MAP, FBT_VECTOR_KEY, FBT_BOOL = 1, 3, 4, 9, 14, 26


def flexbuffer_map(d):
    """Written like flexbuffers::Builder does for fbb.Map(...) of Int / Float / Bool entries: key strings first, then the
    typed keys vector, then the map (keys offset, keys byte width, length, values, packed types), then the root."""
    keys = sorted(d, key=lambda k: k.encode())
    out = bytearray()
    key_pos = {}
    for k in keys:
        key_pos[k] = len(out)
        out += k.encode() + b"\0"
    W = 4

    def align(w):
        while len(out) % w:
            out.append(0)
    # keys vector, byte width 4 (offsets are distances back from the offset's own location)
    align(W)
    out += struct.pack("<I", len(keys))
    kv_loc = len(out)
    for k in keys:
        out += struct.pack("<I", len(out) - key_pos[k])
    # the map itself
    align(W)
    out += struct.pack("<I", len(out) - kv_loc)      # offset to the keys vector
    out += struct.pack("<I", W)                       # its byte width
    out += struct.pack("<I", len(keys))
    map_loc = len(out)
    types = bytearray()
    for k in keys:
        v = d[k]
        if isinstance(v, bool):
            out += struct.pack("<I", int(v))
            types.append((FBT_BOOL << 2) | 2)
        elif isinstance(v, int):
            out += struct.pack("<i", v)
            types.append((FBT_INT << 2) | 2)
        else:
            out += struct.pack("<f", v)
            types.append((FBT_FLOAT << 2) | 2)
    out += types
    # root: offset to the map (width 1 suffices only for tiny maps: use the smallest width that fits), packed type, byte width
    align(1)
    dist = len(out) - map_loc
    if dist < 256:
        out += struct.pack("<B", dist)
        rw = 1
    else:
        align(2)
        dist = len(out) - map_loc
        out += struct.pack("<H", dist)
        rw = 2
    out.append((FBT_MAP << 2) | 2)                    # child vector byte width 4 -> width code 2
    out.append(rw)
    return bytes(out)


# ---- a tiny EfficientDet-shaped full-integer model ---------------------------------------------------------------
TT_FLOAT32, TT_INT32, TT_UINT8, TT_INT8 = 0, 2, 3, 9
BO_ADD, BO_CONCATENATION, BO_CONV_2D, BO_DEPTHWISE_CONV_2D, BO_DEQUANTIZE, BO_LOGISTIC, BO_MAX_POOL_2D, BO_RESHAPE, BO_CUSTOM, BO_QUANTIZE = 0, 2, 3, 4, 6, 14, 17, 22, 32, 114
# BuiltinOptions union tags: Conv2DOptions 1, DepthwiseConv2DOptions 2, Pool2DOptions 5, ConcatenationOptions 10, AddOptions 11, ReshapeOptions 17
ACT_RELU6 = 3


class TinyModel:
    """uint8 image [1,S,S,3] -> QUANTIZE -> CONV 3x3/2 (3->8, RELU6) -> DW 3x3 -> CONV 1x1 (8->8) -> ADD(skip, RELU6) = P3;
    P4..P7 = MAX_POOL 3x3/2 chain; per level class conv 1x1 8->9 and box conv 1x1 8->36 with weights / bias SHARED across
    levels (one buffer, five tensors); RESHAPE, CONCATENATION, LOGISTIC, DEQUANTIZE, TFLite_Detection_PostProcess."""

    def __init__(self, S=32, seed=5, num_classes=2):
        rng = np.random.default_rng(seed)
        self.S = S
        self.num_classes = int(num_classes)       # class columns per anchor (the reference's models: 2, vbt_amd/spec.py)
        self.rng = rng
        self.tensors, self.buffers, self.ops, self.codes = [], [np.zeros(0, np.uint8)], [], []
        self.meta = {}

    def buf(self, arr):
        self.buffers.append(np.frombuffer(np.ascontiguousarray(arr).tobytes(), np.uint8))
        return len(self.buffers) - 1

    def tensor(self, name, shape, ttype, scale=None, zp=None, buffer=0, qdim=0):
        self.tensors.append(dict(name=name, shape=list(shape), type=ttype, scale=scale, zp=zp, buffer=buffer, qdim=qdim))
        return len(self.tensors) - 1

    def op(self, code, ins, outs, opt_type=0, opts=None, custom=None, custom_options=None):
        key = (code, custom)
        if key not in self.codes:
            self.codes.append(key)
        self.ops.append(dict(code=self.codes.index(key), ins=ins, outs=outs, opt_type=opt_type, opts=opts, custom_options=custom_options))

    def act(self, name, h, c, scale, zp):
        return self.tensor(name, [1, h, h, c], TT_INT8, [scale], [zp])

    def conv(self, name, x, xs, cin, cout, k, stride, out_h, s_out, z_out, act=0, wbuf=None, dw=False):
        rng = self.rng
        if wbuf is None:
            sw = (rng.uniform(0.002, 0.02, cout)).astype(np.float32)
            shape = [1, k, k, cout] if dw else [cout, k, k, cin]
            w = rng.integers(-127, 128, shape).astype(np.int8)
            b = rng.integers(-2000, 2000, cout).astype(np.int32)
            wbuf = (self.buf(w), self.buf(b), sw, w, b, shape)
        wb, bb, sw, w, b, shape = wbuf
        wt = self.tensor(name + "/w", shape, TT_INT8, list(sw), [0] * cout, buffer=wb, qdim=3 if dw else 0)   # per-channel: quantized_dimension 3 (dw) / 0
        bt = self.tensor(name + "/b", [cout], TT_INT32, list((np.float32(xs) * sw).astype(np.float32)), [0] * cout, buffer=bb)
        o = self.act(name, out_h, cout, s_out, z_out)
        if dw:   # DepthwiseConv2DOptions: padding:0 stride_w:1 stride_h:2 depth_multiplier:3 fused_activation_function:4
            self.op(BO_DEPTHWISE_CONV_2D, [x, wt, bt], [o], 2, {1: S("i", stride), 2: S("i", stride), 3: S("i", 1), 4: S("b", act)})
        else:    # Conv2DOptions: padding:0 stride_w:1 stride_h:2 fused_activation_function:3
            self.op(BO_CONV_2D, [x, wt, bt], [o], 1, {1: S("i", stride), 2: S("i", stride), 3: S("b", act)})
        self.meta[name] = dict(w=w, b=b, sw=sw, xs=np.float32(xs), so=np.float32(s_out), zo=z_out)
        return o, wbuf

    def build(self):
        S_ = self.S
        f32 = np.float32
        img = self.tensor("serving_default_images:0", [1, S_, S_, 3], TT_UINT8, [f32(1 / 128)], [127])
        q = self.tensor("tfl.quantize", [1, S_, S_, 3], TT_INT8, [f32(1 / 128)], [-1])
        self.op(BO_QUANTIZE, [img], [q])
        h = S_ // 2
        stem, _ = self.conv("stem", q, 1 / 128, 3, 8, 3, 2, h, f32(0.0235), -128, act=ACT_RELU6)
        dw, _ = self.conv("dw", stem, 0.0235, 8, 8, 3, 1, h, f32(0.05), -3, dw=True)
        pw, _ = self.conv("pw", dw, 0.05, 8, 8, 1, 1, h, f32(0.031), 4)
        p3 = self.act("skip", h, 8, f32(0.0235), -128)
        self.op(BO_ADD, [pw, stem], [p3], 11, {0: S("b", ACT_RELU6)})                    # AddOptions.fused_activation_function:0
        levels = [p3]
        for i in range(4):
            h = (h + 1) // 2
            o = self.act(f"p{4 + i}", h, 8, f32(0.0235), -128)
            # Pool2DOptions: padding:0 stride_w:1 stride_h:2 filter_width:3 filter_height:4 fused_activation_function:5
            self.op(BO_MAX_POOL_2D, [levels[-1]], [o], 5, {1: S("i", 2), 2: S("i", 2), 3: S("i", 3), 4: S("i", 3)})
            levels.append(o)
        self.level_h = [self.tensors[t]["shape"][1] for t in levels]
        cats = []
        n_anchor = sum(x * x for x in self.level_h) * 9
        NC = self.num_classes
        for name, cout, width, s_out, z_out in (("class", 9 * NC, NC, f32(0.09), 12), ("box", 36, 4, f32(0.021), -7)):
            parts, shared = [], None
            for li, t in enumerate(levels):
                hh = self.level_h[li]
                o, shared = self.conv(f"{name}{li}", t, 0.0235, 8, cout, 1, 1, hh, s_out, z_out, wbuf=shared)
                shp = self.tensor(f"{name}{li}/shape", [3], TT_INT32, buffer=self.buf(np.asarray([1, hh * hh * 9, width], np.int32)))
                rs = self.tensor(f"{name}{li}/reshape", [1, hh * hh * 9, width], TT_INT8, [s_out], [z_out])
                self.op(BO_RESHAPE, [o, shp], [rs], 17, {})
                parts.append(rs)
            cat = self.tensor(f"{name}/concat", [1, n_anchor, width], TT_INT8, [s_out], [z_out])
            self.op(BO_CONCATENATION, parts, [cat], 10, {0: S("i", 1)})                  # ConcatenationOptions.axis:0
            if name == "class":
                lg = self.tensor("class/logistic", [1, n_anchor, NC], TT_INT8, [f32(1 / 256)], [-128])
                self.op(BO_LOGISTIC, [cat], [lg])
                cat = lg
            dq = self.tensor(f"{name}/dequantize", [1, n_anchor, width], TT_FLOAT32)
            self.op(BO_DEQUANTIZE, [cat], [dq])
            cats.append(dq)
        anchors = self.rng.uniform(0.05, 0.95, (n_anchor, 4)).astype(np.float32)
        at = self.tensor("anchors", [n_anchor, 4], TT_FLOAT32, buffer=self.buf(anchors))
        outs = [self.tensor(n, s, TT_FLOAT32) for n, s in (("StatefulPartitionedCall:3", [1, 25, 4]), ("StatefulPartitionedCall:2", [1, 25]),
                                                           ("StatefulPartitionedCall:1", [1, 25]), ("StatefulPartitionedCall:0", [1]))]
        self.options = {"max_detections": 25, "max_classes_per_detection": 1, "detections_per_class": 100, "use_regular_nms": False,
                        "nms_score_threshold": 0.0625, "nms_iou_threshold": 0.45, "num_classes": NC,
                        "y_scale": 10.0, "x_scale": 10.0, "h_scale": 5.0, "w_scale": 5.0}
        self.op(BO_CUSTOM, [cats[1], cats[0], at], outs, custom="TFLite_Detection_PostProcess", custom_options=flexbuffer_map(self.options))
        self.anchors, self.n_anchor, self.image, self.outputs = anchors, n_anchor, img, outs
        return self

    def serialize(self):
        b = Builder()
        # children first (back to front): buffers, operator codes, tensors, operators, subgraph, model
        buf_offs = []
        for data in self.buffers:
            if data.size:
                v = b.np_vector(data, align=16)
                buf_offs.append(b.table({0: O(v)}))                                       # Buffer.data:0
            else:
                buf_offs.append(b.table({}))
        code_offs = []
        for code, custom in self.codes:
            f = {0: S("b", min(code, 127)), 3: S("i", code), 2: S("i", 1)}               # deprecated_builtin_code:0 version:2 builtin_code:3
            if custom:
                f[1] = O(b.string(custom))                                                # custom_code:1
            code_offs.append(b.table(f))
        tens_offs = []
        for t in self.tensors:
            f = {0: O(b.np_vector(np.asarray(t["shape"], np.int32))), 1: S("b", t["type"]), 2: S("I", t["buffer"]), 3: O(b.string(t["name"]))}
            if t["scale"] is not None:                                                    # QuantizationParameters: scale:2 zero_point:3 quantized_dimension:6
                qf = {2: O(b.np_vector(np.asarray(t["scale"], np.float32))), 3: O(b.np_vector(np.asarray(t["zp"], np.int64)))}
                if t["qdim"]:
                    qf[6] = S("i", t["qdim"])
                f[4] = O(b.table(qf))                                                     # Tensor.quantization:4
            tens_offs.append(b.table(f))
        op_offs = []
        for o in self.ops:
            f = {0: S("I", o["code"]), 1: O(b.np_vector(np.asarray(o["ins"], np.int32))), 2: O(b.np_vector(np.asarray(o["outs"], np.int32)))}
            if o["opts"] is not None:
                f[3] = S("B", o["opt_type"])                                              # builtin_options_type:3
                f[4] = O(b.table(o["opts"]))                                              # builtin_options:4
            if o["custom_options"] is not None:
                f[5] = O(b.bytes_vector(o["custom_options"], 1))                          # custom_options:5
            op_offs.append(b.table(f))
        sub = b.table({0: O(b.offset_vector(tens_offs)), 1: O(b.np_vector(np.asarray([self.image], np.int32))),
                       2: O(b.np_vector(np.asarray(self.outputs, np.int32))), 3: O(b.offset_vector(op_offs)), 4: O(b.string("main"))})
        model = b.table({0: S("I", 3), 1: O(b.offset_vector(code_offs)), 2: O(b.offset_vector([sub])),
                         3: O(b.string("independent test encoder")), 4: O(b.offset_vector(buf_offs))})
        return b.finish(model, b"TFL3")
