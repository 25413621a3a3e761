"""VBTM model container: the on-disk form of a fully-integer-quantised EfficientDet-Lite.

It plays the role of the reference's ``models/efficientdet_lite*.tflite`` flatbuffers
(reference track.py:67,93; all missing from the tree, .MISSING_LARGE_BLOBS): a graph of
int8 tensors (per-tensor scale / zero-point), conv ops with per-output-channel int8 weights,
int32 biases and float32 requantisation multipliers, binary ADDs with the integer parameters of
XNNPACK's qs8-vadd (vbt_amd/quant.py), plus the anchors and look-up tables of the detection post-process.  The C-ABI library (csrc/vbt_model.cpp) and the C oracle
(oracle/detector.c) each parse this file on their own; this Python module is the writer
(tools/make_model.py) and a reader for tests.

Layout (little endian):
  header   128 B   magic "VBTM0002", ints, floats, blob offset/size
  tensors  nT x 32 B   {h, w, c, zero_point, scale(f32), pad[3]}
  ops      nO x 160 B  see OP_DTYPE
  blob     raw bytes (int8 weights, int32 biases, f32 multipliers, anchors, LUTs), 16-B aligned items
"""
from __future__ import annotations

import numpy as np

MAGIC = b"VBTM0002"   # 0002: binary integer ADD (XNNPACK qs8-vadd parameters), double-precision decode tables

HEADER_DTYPE = np.dtype([
    ("magic", "S8"), ("arch", "<i4"), ("image_size", "<i4"), ("num_tensors", "<i4"), ("num_ops", "<i4"),
    ("num_anchors", "<i4"), ("max_detections", "<i4"), ("nms_iou_threshold", "<f4"),
    ("nms_score_threshold", "<f4"), ("blob_offset", "<i8"), ("blob_bytes", "<i8"),
    ("input_tensor", "<i4"),
    ("num_classes", "<i4"),        # class columns per anchor in the head's class tensors (0 in files written before the field existed = 1)
    ("reserved", "<i4", (16,)),
])
assert HEADER_DTYPE.itemsize == 128

TENSOR_DTYPE = np.dtype([("h", "<i4"), ("w", "<i4"), ("c", "<i4"), ("zero_point", "<i4"),
                         ("scale", "<f4"), ("pad", "<i4", (3,))])
assert TENSOR_DTYPE.itemsize == 32

OP_DTYPE = np.dtype([
    ("type", "<i4"), ("n_inputs", "<i4"), ("inputs", "<i4", (12,)), ("output", "<i4"),
    ("k", "<i4"), ("stride", "<i4"), ("pad_t", "<i4"), ("pad_l", "<i4"),
    ("act_min", "<i4"), ("act_max", "<i4"), ("level", "<i4"),
    ("w_off", "<i8"), ("b_off", "<i8"), ("m_off", "<i8"),       # conv: weights / bias / multipliers
    ("aux_off", "<i8"), ("aux2_off", "<i8"),                   # postprocess: anchors / LUTs
    ("in_mult", "<f4", (3,)),                                  # ADD: s_a/s_out, s_b/s_out in float32 (informational)
    ("add_q", "<i4", (4,)),                                    # ADD: bias, a_multiplier, b_multiplier, shift (quant.xnn_qs8_add_params)
    ("reserved", "<i4", (1,)),
])
assert OP_DTYPE.itemsize == 160, OP_DTYPE.itemsize


class BlobWriter:
    def __init__(self):
        self.parts = []
        self.size = 0

    def add(self, arr: np.ndarray) -> int:
        pad = (-self.size) % 16
        if pad:
            self.parts.append(b"\0" * pad)
            self.size += pad
        off = self.size
        b = np.ascontiguousarray(arr).tobytes()
        self.parts.append(b)
        self.size += len(b)
        return off

    def bytes(self) -> bytes:
        return b"".join(self.parts)


def write_container(path, header: dict, tensors: np.ndarray, ops: np.ndarray, blob: bytes):
    h = np.zeros(1, HEADER_DTYPE)
    h["magic"] = MAGIC
    for k, v in header.items():
        h[k] = v
    h["num_tensors"] = len(tensors)
    h["num_ops"] = len(ops)
    off = HEADER_DTYPE.itemsize + tensors.nbytes + ops.nbytes
    off += (-off) % 16
    h["blob_offset"] = off
    h["blob_bytes"] = len(blob)
    with open(path, "wb") as f:
        f.write(h.tobytes())
        f.write(tensors.tobytes())
        f.write(ops.tobytes())
        f.write(b"\0" * (off - HEADER_DTYPE.itemsize - tensors.nbytes - ops.nbytes))
        f.write(blob)


class Container:
    def __init__(self, path):
        raw = np.fromfile(path, dtype=np.uint8)
        self.header = raw[:128].view(HEADER_DTYPE)[0]
        if bytes(self.header["magic"]) != MAGIC:
            raise ValueError(f"{path}: not a VBTM container")
        nt, no = int(self.header["num_tensors"]), int(self.header["num_ops"])
        o = 128
        self.tensors = raw[o:o + nt * 32].view(TENSOR_DTYPE)
        o += nt * 32
        self.ops = raw[o:o + no * 160].view(OP_DTYPE)
        bo = int(self.header["blob_offset"])
        self.blob = raw[bo:bo + int(self.header["blob_bytes"])]

    def i8(self, off, n):
        return self.blob[off:off + n].view(np.int8)

    def i32(self, off, n):
        return self.blob[off:off + 4 * n].view("<i4")

    def f32(self, off, n):
        return self.blob[off:off + 4 * n].view("<f4")
