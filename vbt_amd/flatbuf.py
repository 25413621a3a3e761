"""Minimal FlatBuffers / FlexBuffers codec (the `flatbuffers` package is not installed).

Only what a TFLite model file needs: reading tables through their vtables (scalars, strings, vectors of
scalars / tables, unions), reading a FlexBuffers map (the `custom_options` of TFLite_Detection_PostProcess)
and a small front-to-back writer used by tools/export_tflite.py to produce test files.

Format facts restated from the public FlatBuffers "internals" documentation:
  * the file starts with a uoffset32 to the root table; bytes 4..8 may hold a 4-char file identifier;
  * a table starts with an soffset32 `table_pos - vtable_pos`; the vtable is
    {u16 vtable_bytes, u16 table_bytes, u16 field_offset[...]}, offset 0 = field absent (default value);
  * strings / vectors / sub-tables are referenced by a uoffset32 relative to the location of the offset itself;
    a vector is {u32 length, elements...}, a string is a byte vector followed by a NUL.
"""
from __future__ import annotations

import struct

import numpy as np


class Table:
    __slots__ = ("buf", "pos", "_vt", "_vtlen")

    def __init__(self, buf, pos):
        self.buf = buf
        self.pos = pos
        if pos < 0 or pos + 4 > len(buf):
            raise ValueError("flatbuffer: table position out of range")
        self._vt = pos - struct.unpack_from("<i", buf, pos)[0]
        if self._vt < 0 or self._vt + 4 > len(buf):
            raise ValueError("flatbuffer: vtable position out of range")
        self._vtlen = struct.unpack_from("<H", buf, self._vt)[0]

    @classmethod
    def root(cls, buf):
        if len(buf) < 8:
            raise ValueError("flatbuffer: file too short")
        return cls(buf, struct.unpack_from("<I", buf, 0)[0])

    def _field(self, idx):
        slot = 4 + 2 * idx
        if slot + 2 > self._vtlen:
            return 0
        off = struct.unpack_from("<H", self.buf, self._vt + slot)[0]
        return self.pos + off if off else 0

    def has(self, idx):
        return self._field(idx) != 0

    def scalar(self, idx, fmt, default=0):
        p = self._field(idx)
        return struct.unpack_from("<" + fmt, self.buf, p)[0] if p else default

    def _indirect(self, idx):
        p = self._field(idx)
        if not p:
            return 0
        t = p + struct.unpack_from("<I", self.buf, p)[0]
        if t + 4 > len(self.buf):
            raise ValueError("flatbuffer: offset out of range")
        return t

    def table(self, idx):
        p = self._indirect(idx)
        return Table(self.buf, p) if p else None

    def string(self, idx, default=""):
        p = self._indirect(idx)
        if not p:
            return default
        n = struct.unpack_from("<I", self.buf, p)[0]
        return bytes(self.buf[p + 4:p + 4 + n]).decode("utf-8", "replace")

    def vector_len(self, idx):
        p = self._indirect(idx)
        return struct.unpack_from("<I", self.buf, p)[0] if p else 0

    def vector(self, idx, dtype):
        """Vector of scalars as a numpy array (a view into the file)."""
        p = self._indirect(idx)
        dt = np.dtype(dtype).newbyteorder("<")
        if not p:
            return np.empty(0, dt)
        n = struct.unpack_from("<I", self.buf, p)[0]
        if p + 4 + n * dt.itemsize > len(self.buf):
            raise ValueError("flatbuffer: vector exceeds the file")
        return np.frombuffer(self.buf, dtype=dt, count=n, offset=p + 4)

    def tables(self, idx):
        p = self._indirect(idx)
        if not p:
            return []
        n = struct.unpack_from("<I", self.buf, p)[0]
        out = []
        for i in range(n):
            q = p + 4 + 4 * i
            out.append(Table(self.buf, q + struct.unpack_from("<I", self.buf, q)[0]))
        return out


# ---------------------------------------------------------------------------------------------------
# FlexBuffers (schema-less) reader: enough for a root map of scalars.
FBT_NULL, FBT_INT, FBT_UINT, FBT_FLOAT, FBT_KEY, FBT_STRING = 0, 1, 2, 3, 4, 5
FBT_INDIRECT_INT, FBT_INDIRECT_UINT, FBT_INDIRECT_FLOAT, FBT_MAP, FBT_VECTOR, FBT_BOOL = 6, 7, 8, 9, 10, 26


def _uint(buf, p, w):
    return int.from_bytes(bytes(buf[p:p + w]), "little", signed=False)


def _sint(buf, p, w):
    return int.from_bytes(bytes(buf[p:p + w]), "little", signed=True)


def _flt(buf, p, w):
    if w == 4:
        return struct.unpack_from("<f", buf, p)[0]
    if w == 8:
        return struct.unpack_from("<d", buf, p)[0]
    raise ValueError("flexbuffer: float of width %d" % w)


def _flex_value(buf, p, parent_w, packed):
    typ, w = packed >> 2, 1 << (packed & 3)
    if typ == FBT_NULL:
        return None
    if typ == FBT_INT:
        return _sint(buf, p, parent_w)
    if typ == FBT_UINT:
        return _uint(buf, p, parent_w)
    if typ == FBT_BOOL:
        return bool(_uint(buf, p, parent_w))
    if typ == FBT_FLOAT:
        return _flt(buf, p, parent_w)
    q = p - _uint(buf, p, parent_w)          # everything else is stored behind an offset
    if typ == FBT_INDIRECT_INT:
        return _sint(buf, q, w)
    if typ == FBT_INDIRECT_UINT:
        return _uint(buf, q, w)
    if typ == FBT_INDIRECT_FLOAT:
        return _flt(buf, q, w)
    if typ in (FBT_KEY, FBT_STRING):
        if typ == FBT_STRING:
            n = _uint(buf, q - w, w)
            return bytes(buf[q:q + n]).decode("utf-8", "replace")
        e = q
        while buf[e] != 0:
            e += 1
        return bytes(buf[q:e]).decode("utf-8", "replace")
    if typ == FBT_MAP:
        n = _uint(buf, q - w, w)
        kp = q - 3 * w
        keys = kp - _uint(buf, kp, w)
        kw = _uint(buf, q - 2 * w, w)
        out = {}
        for i in range(n):
            ke = keys + i * kw
            ks = ke - _uint(buf, ke, kw)
            e = ks
            while buf[e] != 0:
                e += 1
            out[bytes(buf[ks:e]).decode()] = _flex_value(buf, q + i * w, w, buf[q + n * w + i])
        return out
    if typ == FBT_VECTOR:
        n = _uint(buf, q - w, w)
        return [_flex_value(buf, q + i * w, w, buf[q + n * w + i]) for i in range(n)]
    raise ValueError("flexbuffer: unsupported type %d" % typ)


def flex_root(buf):
    buf = bytes(buf)
    if len(buf) < 3:
        raise ValueError("flexbuffer: too short")
    w = buf[-1]
    return _flex_value(buf, len(buf) - 2 - w, w, buf[-2])


def flex_map(d):
    """Serialise a flat {str: int|float|bool} dict as a FlexBuffers root map (all widths 4)."""
    keys = sorted(d, key=lambda k: k.encode())
    out = bytearray()
    kpos = []
    for k in keys:
        kpos.append(len(out))
        out += k.encode() + b"\0"
    out += b"\0" * ((-len(out)) % 4)
    out += struct.pack("<I", len(keys))
    kvec = len(out)
    for i, kp in enumerate(kpos):
        here = len(out)
        out += struct.pack("<I", here - kp)
    here = len(out)
    out += struct.pack("<I", here - kvec)      # offset to the keys vector
    out += struct.pack("<I", 4)                # its byte width
    out += struct.pack("<I", len(keys))
    mpos = len(out)
    types = bytearray()
    for k in keys:
        v = d[k]
        if isinstance(v, bool):
            out += struct.pack("<I", int(v)); types.append((FBT_BOOL << 2) | 2)
        elif isinstance(v, int):
            out += struct.pack("<i", v); types.append((FBT_INT << 2) | 2)
        else:
            out += struct.pack("<f", float(v)); types.append((FBT_FLOAT << 2) | 2)
    out += types
    out += b"\0" * ((-len(out)) % 4)
    here = len(out)
    out += struct.pack("<I", here - mpos)
    out += bytes([(FBT_MAP << 2) | 2, 4])
    return bytes(out)


# ---------------------------------------------------------------------------------------------------
# Writer: objects are described as Python values and laid out front to back (parents before children, so that
# every uoffset points forward); each table is preceded by its own vtable.
class Scalar:
    def __init__(self, fmt, value):
        self.fmt, self.value = fmt, value


class Vec:
    """Vector of scalars (numpy array), of tables (list of TableSpec) or a string (str)."""
    def __init__(self, items, align=4):
        self.items, self.align = items, align


class TableSpec:
    def __init__(self, fields):
        """fields: {field_index: Scalar | Vec | TableSpec | str | None}"""
        self.fields = {k: v for k, v in fields.items() if v is not None}


class Writer:
    def __init__(self):
        self.out = bytearray()

    def _align(self, a, bias=0):
        while (len(self.out) + bias) % a:
            self.out.append(0)

    def finish(self, root: TableSpec, ident=b"TFL3"):
        self.out += b"\0" * 4 + ident
        pos = self._table(root)
        struct.pack_into("<I", self.out, 0, pos)
        return bytes(self.out)

    def _patch(self, at, target):
        struct.pack_into("<I", self.out, at, target - at)

    def _table(self, t: TableSpec):
        nf = (max(t.fields) + 1) if t.fields else 0
        # inline layout: soffset, then fields by decreasing size
        items = sorted(t.fields.items(), key=lambda kv: -self._isize(kv[1]))
        offs, cur = {}, 4
        for k, v in items:
            s = self._isize(v)
            cur += (-cur) % s
            offs[k] = cur
            cur += s
        tbytes = cur + ((-cur) % 4)
        maxal = max([self._isize(v) for v in t.fields.values()] + [4])
        vtbytes = 4 + 2 * nf
        self._align(2)
        # table start must be aligned to maxal: pad before the vtable
        while (len(self.out) + vtbytes) % maxal:
            self.out.append(0)
        vt = len(self.out)
        self.out += struct.pack("<HH", vtbytes, tbytes)
        for i in range(nf):
            self.out += struct.pack("<H", offs.get(i, 0))
        pos = len(self.out)
        self.out += b"\0" * tbytes
        struct.pack_into("<i", self.out, pos, pos - vt)
        pending = []
        for k, v in t.fields.items():
            at = pos + offs[k]
            if isinstance(v, Scalar):
                struct.pack_into("<" + v.fmt, self.out, at, v.value)
            else:
                pending.append((at, v))
        for at, v in pending:
            self._patch(at, self._object(v))
        return pos

    @staticmethod
    def _isize(v):
        return struct.calcsize("<" + v.fmt) if isinstance(v, Scalar) else 4

    def _object(self, v):
        if isinstance(v, TableSpec):
            return self._table(v)
        if isinstance(v, str):
            b = v.encode()
            self._align(4)
            pos = len(self.out)
            self.out += struct.pack("<I", len(b)) + b + b"\0"
            return pos
        assert isinstance(v, Vec)
        if isinstance(v.items, np.ndarray):
            a = np.ascontiguousarray(v.items)
            al = max(v.align, a.dtype.itemsize, 4)
            self._align(al, bias=4)
            pos = len(self.out)
            self.out += struct.pack("<I", a.size) + a.astype(a.dtype.newbyteorder("<")).tobytes()
            return pos
        self._align(4)
        pos = len(self.out)
        n = len(v.items)
        self.out += struct.pack("<I", n) + b"\0" * (4 * n)
        for i, it in enumerate(v.items):
            self._patch(pos + 4 + 4 * i, self._object(it))
        return pos
