"""ORACLE (test infrastructure only — never imported by the product path).

Independent pure-Python reader for TFLite flatbuffers: walks vtables directly per the FlatBuffers
binary spec, so it validates both the model writer (`birdnet-go_amd/flatbuf_writer.py`) and the
engine's C++ reader against a third implementation.  Restates the container the reference passes
to its TFLite backend (`internal/inference/tflite/classifier.go:38-41`, `tflitelib.NewModel(modelData)`;
TensorFlow Lite 2.17.1 schema, third-party, pinned in reference `Taskfile.yml:6`).
"""
import struct

import numpy as np

# slot numbers restated from TFLite schema.fbs (kept literal here on purpose: the oracle must not
# share constants with the product package)
_OPNAMES = {0: "ADD", 1: "AVERAGE_POOL_2D", 2: "CONCATENATION", 3: "CONV_2D", 4: "DEPTHWISE_CONV_2D", 6: "DEQUANTIZE",
            8: "FLOOR", 9: "FULLY_CONNECTED", 14: "LOGISTIC", 17: "MAX_POOL_2D", 18: "MUL", 19: "RELU", 20: "RELU_N1_TO_1",
            21: "RELU6", 22: "RESHAPE", 25: "SOFTMAX", 28: "TANH", 34: "PAD", 36: "GATHER", 39: "TRANSPOSE", 40: "MEAN",
            41: "SUB", 42: "DIV", 43: "SQUEEZE", 45: "STRIDED_SLICE", 47: "EXP", 49: "SPLIT", 53: "CAST", 55: "MAXIMUM",
            57: "MINIMUM", 59: "NEG", 60: "PADV2", 65: "SLICE", 66: "SIN", 70: "EXPAND_DIMS", 73: "LOG",
            74: "SUM", 75: "SQRT", 76: "RSQRT", 78: "POW", 81: "REDUCE_PROD", 82: "REDUCE_MAX", 89: "REDUCE_MIN",
            92: "SQUARE", 98: "LEAKY_RELU", 99: "SQUARED_DIFFERENCE", 101: "ABS", 104: "CEIL", 105: "REVERSE_V2",
            108: "COS", 111: "ELU", 116: "ROUND", 117: "HARD_SWISH", 126: "BATCH_MATMUL", 131: "RFFT2D", 133: "IMAG",
            134: "REAL", 135: "COMPLEX_ABS", 150: "GELU"}
_DTYPES = {0: np.float32, 2: np.int32, 4: np.int64, 8: np.complex64, 3: np.uint8, 9: np.int8,
           1: np.float16, 6: np.bool_, 10: np.float64}

# option layouts: op -> [(field, kind)]
_CONV = [("padding", "b"), ("stride_w", "i"), ("stride_h", "i"), ("act", "b"), ("dil_w", "i"), ("dil_h", "i")]
_DW = [("padding", "b"), ("stride_w", "i"), ("stride_h", "i"), ("depth_multiplier", "i"), ("act", "b"),
       ("dil_w", "i"), ("dil_h", "i")]
_POOL = [("padding", "b"), ("stride_w", "i"), ("stride_h", "i"), ("filter_w", "i"), ("filter_h", "i"), ("act", "b")]
_ACT = [("act", "b")]
_OPTS = {"CONV_2D": _CONV, "DEPTHWISE_CONV_2D": _DW, "AVERAGE_POOL_2D": _POOL, "MAX_POOL_2D": _POOL,
         "FULLY_CONNECTED": [("act", "b"), ("weights_format", "b"), ("keep_num_dims", "?")],
         "ADD": _ACT, "MUL": _ACT, "SUB": _ACT, "DIV": _ACT,
         "CONCATENATION": [("axis", "i"), ("act", "b")], "SOFTMAX": [("beta", "f")],
         "MEAN": [("keep_dims", "?")], "SUM": [("keep_dims", "?")], "REDUCE_MAX": [("keep_dims", "?")],
         "REDUCE_MIN": [("keep_dims", "?")], "GATHER": [("axis", "i"), ("batch_dims", "i")],
         "RESHAPE": [("new_shape", "vi")], "SQUEEZE": [("squeeze_dims", "vi")],
         "STRIDED_SLICE": [("begin_mask", "i"), ("end_mask", "i"), ("ellipsis_mask", "i"),
                           ("new_axis_mask", "i"), ("shrink_axis_mask", "i")],
         "CAST": [("in_type", "b"), ("out_type", "b")],
         "BATCH_MATMUL": [("adj_x", "?"), ("adj_y", "?")], "REDUCE_PROD": [("keep_dims", "?")],
         "SPLIT": [("num_splits", "i")], "LEAKY_RELU": [("alpha", "f")], "GELU": [("approximate", "?")]}


class FB:
    """Bare flatbuffer accessor."""

    def __init__(self, buf):
        self.b = memoryview(buf)

    def u32(self, p):
        return struct.unpack_from("<I", self.b, p)[0]

    def i32(self, p):
        return struct.unpack_from("<i", self.b, p)[0]

    def root(self):
        return self.u32(0)

    def field(self, tpos, slot):
        """absolute position of inline field `slot` of table at tpos, or None."""
        vt = tpos - self.i32(tpos)
        vt_size = struct.unpack_from("<H", self.b, vt)[0]
        off_pos = 4 + 2 * slot
        if off_pos >= vt_size:
            return None
        off = struct.unpack_from("<H", self.b, vt + off_pos)[0]
        return tpos + off if off else None

    def scalar(self, tpos, slot, fmt, default=0):
        p = self.field(tpos, slot)
        return default if p is None else struct.unpack_from("<" + fmt, self.b, p)[0]

    def indirect(self, tpos, slot):
        p = self.field(tpos, slot)
        return None if p is None else p + self.u32(p)

    def vec(self, tpos, slot):
        """(data_pos, length) of a vector field, or (None, 0)."""
        p = self.indirect(tpos, slot)
        if p is None:
            return None, 0
        return p + 4, self.u32(p)

    def vec_np(self, tpos, slot, dtype):
        p, n = self.vec(tpos, slot)
        if p is None:
            return np.zeros(0, dtype)
        return np.frombuffer(self.b, dtype=dtype, count=n, offset=p)

    def vec_tables(self, tpos, slot):
        p, n = self.vec(tpos, slot)
        return [p + 4 * i + self.u32(p + 4 * i) for i in range(n)]

    def string(self, tpos, slot):
        p = self.indirect(tpos, slot)
        if p is None:
            return ""
        n = self.u32(p)
        return bytes(self.b[p + 4:p + 4 + n]).decode("utf-8", "replace")


class Tensor:
    __slots__ = ("name", "shape", "dtype", "data")


class Op:
    __slots__ = ("name", "inputs", "outputs", "opts")


class Model:
    pass


def read_model(buf) -> Model:
    if len(buf) < 8 or bytes(buf[4:8]) != b"TFL3":
        raise ValueError("not a TFLite flatbuffer (missing TFL3 identifier)")
    fb = FB(buf)
    root = fb.root()
    m = Model()
    m.version = fb.scalar(root, 0, "I")
    m.description = fb.string(root, 3)
    codes = []
    for ct in fb.vec_tables(root, 1):
        dep = fb.scalar(ct, 0, "b")
        new = fb.scalar(ct, 3, "i")
        codes.append(max(dep, new))
    buffers = []
    for bt in fb.vec_tables(root, 4):
        p, n = fb.vec(bt, 0)
        buffers.append(None if p is None or n == 0 else (p, n))
    sgs = fb.vec_tables(root, 2)
    if len(sgs) != 1:
        raise ValueError("expected exactly one subgraph")
    sg = sgs[0]
    m.tensors = []
    for tt in fb.vec_tables(sg, 0):
        t = Tensor()
        t.name = fb.string(tt, 3)
        t.shape = [int(v) for v in fb.vec_np(tt, 0, np.int32)]
        ty = fb.scalar(tt, 1, "b")
        if ty not in _DTYPES:
            raise ValueError(f"tensor {t.name}: unsupported type {ty}")
        t.dtype = _DTYPES[ty]
        bi = fb.scalar(tt, 2, "I")
        t.data = None
        if bi and buffers[bi] is not None:
            p, n = buffers[bi]
            t.data = np.frombuffer(fb.b, dtype=t.dtype, count=n // np.dtype(t.dtype).itemsize,
                                   offset=p).reshape(t.shape)
        m.tensors.append(t)
    m.inputs = [int(v) for v in fb.vec_np(sg, 1, np.int32)]
    m.outputs = [int(v) for v in fb.vec_np(sg, 2, np.int32)]
    m.ops = []
    for ot in fb.vec_tables(sg, 3):
        o = Op()
        code = codes[fb.scalar(ot, 0, "I")]
        if code not in _OPNAMES:
            raise ValueError(f"unsupported builtin operator code {code}")
        o.name = _OPNAMES[code]
        o.inputs = [int(v) for v in fb.vec_np(ot, 1, np.int32)]
        o.outputs = [int(v) for v in fb.vec_np(ot, 2, np.int32)]
        o.opts = {}
        optpos = fb.indirect(ot, 4)
        if optpos is not None:
            for slot, (fname, kind) in enumerate(_OPTS.get(o.name, [])):
                if kind == "vi":
                    o.opts[fname] = [int(v) for v in fb.vec_np(optpos, slot, np.int32)]
                else:
                    o.opts[fname] = fb.scalar(optpos, slot, kind)
        m.ops.append(o)
    return m
