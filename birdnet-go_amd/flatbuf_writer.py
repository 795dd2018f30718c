"""Minimal FlatBuffers writer (no third-party dependency; `flatbuffers` is not installed here).

Lays objects out parent-before-child so every uoffset is a positive forward reference, which
is all a conforming reader requires.  Used only to author synthetic ``.tflite`` model files
(`synth_model.py`); the engine's readers (C++ `csrc/tflite_model.cpp`, oracle
`oracle/tflite_reader.py`) are independent implementations of the read side.
"""
import struct

import numpy as np

_SCALAR = {"i8": ("<b", 1), "u8": ("<B", 1), "bool": ("<B", 1), "i16": ("<h", 2),
           "u16": ("<H", 2), "i32": ("<i", 4), "u32": ("<I", 4), "f32": ("<f", 4),
           "i64": ("<q", 8)}


class Table:
    """fields: {slot: (kind, value)}; kind 'offset' => value is Table|Vec|Str."""

    def __init__(self, fields=None):
        self.fields = dict(fields or {})

    def add(self, slot, kind, value):
        self.fields[slot] = (kind, value)
        return self


class Vec:
    """Vector of scalars (numpy array or list) or of offsets (list of objects)."""

    def __init__(self, kind, items, align=None):
        self.kind = kind
        if kind == "offset":
            self.items = list(items)
            self.elem = 4
        else:
            fmt, size = _SCALAR[kind]
            dt = {"i8": np.int8, "u8": np.uint8, "bool": np.uint8, "i32": np.int32,
                  "u32": np.uint32, "f32": np.float32, "i64": np.int64, "i16": np.int16,
                  "u16": np.uint16}[kind]
            self.items = np.ascontiguousarray(np.asarray(items, dtype=dt)).reshape(-1)
            self.elem = size
        self.align = align or max(4, self.elem)


class Str:
    def __init__(self, s):
        self.data = s.encode("utf-8") if isinstance(s, str) else bytes(s)


def _align(p, a):
    return (p + a - 1) // a * a


def build(root, file_identifier=None):
    """Serialise `root` (a Table) into a flatbuffer; returns bytes."""
    order = []  # (obj, pos, extra)
    pos = 8 if file_identifier else 4

    def place(obj):
        nonlocal pos
        if isinstance(obj, Table):
            slots = sorted(obj.fields)
            nslots = (slots[-1] + 1) if slots else 0
            vt_size = 4 + 2 * nslots
            # inline layout: soffset, then 8/4-byte fields, then 2, then 1-byte fields
            off = 4
            layout = {}
            def fsize(kind):
                return 4 if kind == "offset" else _SCALAR[kind][1]
            for want in (8, 4, 2, 1):
                for s in slots:
                    kind, _ = obj.fields[s]
                    if fsize(kind) == want:
                        off = _align(off, want)
                        layout[s] = off
                        off += want
            tbl_size = _align(off, 4)
            talign = 8 if any(fsize(k) == 8 for k, _ in obj.fields.values()) else 4
            vt_pos = pos
            tpos = _align(vt_pos + vt_size, talign)
            vt_pos = tpos - vt_size
            pos = tpos + tbl_size
            order.append((obj, tpos, (vt_pos, vt_size, tbl_size, nslots, layout)))
            obj._pos = tpos
            for s in slots:
                kind, val = obj.fields[s]
                if kind == "offset":
                    place(val)
        elif isinstance(obj, Vec):
            start = _align(pos + 4, obj.align) - 4
            start = max(start, _align(pos, 4))
            while (start + 4) % obj.align or start % 4:
                start += 4
            n = len(obj.items)
            pos = start + 4 + n * obj.elem
            order.append((obj, start, None))
            obj._pos = start
            if obj.kind == "offset":
                for it in obj.items:
                    place(it)
        elif isinstance(obj, Str):
            start = _align(pos, 4)
            pos = start + 4 + len(obj.data) + 1
            order.append((obj, start, None))
            obj._pos = start
        else:
            raise TypeError(type(obj))

    place(root)
    total = _align(pos, 16)
    buf = bytearray(total)
    struct.pack_into("<I", buf, 0, root._pos)
    if file_identifier:
        buf[4:8] = file_identifier
    mv = memoryview(buf)
    for obj, p, extra in order:
        if isinstance(obj, Table):
            vt_pos, vt_size, tbl_size, nslots, layout = extra
            struct.pack_into("<HH", buf, vt_pos, vt_size, tbl_size)
            for s in range(nslots):
                struct.pack_into("<H", buf, vt_pos + 4 + 2 * s, layout.get(s, 0))
            struct.pack_into("<i", buf, p, p - vt_pos)
            for s, foff in layout.items():
                kind, val = obj.fields[s]
                if kind == "offset":
                    struct.pack_into("<I", buf, p + foff, val._pos - (p + foff))
                else:
                    struct.pack_into(_SCALAR[kind][0], buf, p + foff, val)
        elif isinstance(obj, Vec):
            n = len(obj.items)
            struct.pack_into("<I", buf, p, n)
            if obj.kind == "offset":
                for i, it in enumerate(obj.items):
                    fp = p + 4 + 4 * i
                    struct.pack_into("<I", buf, fp, it._pos - fp)
            elif n:
                raw = obj.items.view(np.uint8)
                mv[p + 4:p + 4 + raw.size] = raw
        else:
            struct.pack_into("<I", buf, p, len(obj.data))
            buf[p + 4:p + 4 + len(obj.data)] = obj.data
    return bytes(buf)
