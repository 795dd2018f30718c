"""Programmatic construction of TFLite flatbuffer models (op-level graph -> bytes)."""
import numpy as np

from . import tflite_schema as S
from .flatbuf_writer import Str, Table, Vec, build

_DT = {np.dtype(np.float32): S.FLOAT32, np.dtype(np.float16): S.FLOAT16, np.dtype(np.int32): S.INT32,
       np.dtype(np.int64): S.INT64, np.dtype(np.complex64): S.COMPLEX64}


class GraphBuilder:
    def __init__(self, description="birdnet-go_amd synthetic model"):
        self.tensors = []      # dict(name, shape, type, buffer)
        self.buffers = [None]  # buffer 0 = empty sentinel
        self.ops = []          # dict(op, inputs, outputs, options)
        self.opcodes = []      # list of builtin codes in first-use order
        self.description = description
        self._n = 0

    def tensor(self, shape, dtype=S.FLOAT32, name=None):
        self._n += 1
        self.tensors.append(dict(name=name or f"t{self._n}", shape=list(shape), type=dtype, buffer=0))
        return len(self.tensors) - 1

    def const(self, array, name=None):
        a = np.ascontiguousarray(array)
        self._n += 1
        self.buffers.append(a)
        self.tensors.append(dict(name=name or f"c{self._n}", shape=list(a.shape),
                                 type=_DT[a.dtype], buffer=len(self.buffers) - 1))
        return len(self.tensors) - 1

    def shape(self, t):
        return self.tensors[t]["shape"]

    def op(self, opname, inputs, out_shape, options=None, out_dtype=S.FLOAT32, name=None):
        code = S.OP[opname]
        if code not in self.opcodes:
            self.opcodes.append(code)
        if out_shape and isinstance(out_shape[0], (list, tuple)):      # several outputs (SPLIT): list of shapes
            outs = [self.tensor(sh, out_dtype, (name + f":{i}") if name else None) for i, sh in enumerate(out_shape)]
            self.ops.append(dict(op=opname, inputs=list(inputs), outputs=outs, options=options or {}))
            return outs
        out = self.tensor(out_shape, out_dtype, name)
        self.ops.append(dict(op=opname, inputs=list(inputs), outputs=[out], options=options or {}))
        return out

    # ------------------------------------------------------------------ serialise
    def finish(self, inputs, outputs):
        bufs = []
        for b in self.buffers:
            t = Table()
            if b is not None and b.size:
                t.add(S.BUFFER["data"], "offset", Vec("u8", b.view(np.uint8).reshape(-1), align=16))
            bufs.append(t)
        tens = []
        for t in self.tensors:
            tt = Table()
            tt.add(S.TENSOR["shape"], "offset", Vec("i32", t["shape"]))
            tt.add(S.TENSOR["type"], "i8", t["type"])
            tt.add(S.TENSOR["buffer"], "u32", t["buffer"])
            tt.add(S.TENSOR["name"], "offset", Str(t["name"]))
            tens.append(tt)
        ops = []
        for o in self.ops:
            ot = Table()
            ot.add(S.OPERATOR["opcode_index"], "u32", self.opcodes.index(S.OP[o["op"]]))
            ot.add(S.OPERATOR["inputs"], "offset", Vec("i32", o["inputs"]))
            ot.add(S.OPERATOR["outputs"], "offset", Vec("i32", o["outputs"]))
            optname = S.OP_OPTIONS[o["op"]]
            if optname is not None:
                opt = Table()
                for slot, (fname, kind) in enumerate(S.OPTION_FIELDS[optname]):
                    if fname not in o["options"]:
                        continue
                    v = o["options"][fname]
                    if kind == "vec_i32":
                        opt.add(slot, "offset", Vec("i32", v))
                    else:
                        opt.add(slot, kind, float(v) if kind == "f32" else int(v))
                ot.add(S.OPERATOR["builtin_options_type"], "u8", S.OPT[optname])
                ot.add(S.OPERATOR["builtin_options"], "offset", opt)
            ops.append(ot)
        codes = []
        for c in self.opcodes:
            ct = Table()
            ct.add(S.OPERATOR_CODE["deprecated_builtin_code"], "i8", min(c, 127))
            ct.add(S.OPERATOR_CODE["version"], "i32", 1)
            ct.add(S.OPERATOR_CODE["builtin_code"], "i32", c)
            codes.append(ct)
        sg = Table()
        sg.add(S.SUBGRAPH["tensors"], "offset", Vec("offset", tens))
        sg.add(S.SUBGRAPH["inputs"], "offset", Vec("i32", inputs))
        sg.add(S.SUBGRAPH["outputs"], "offset", Vec("i32", outputs))
        sg.add(S.SUBGRAPH["operators"], "offset", Vec("offset", ops))
        sg.add(S.SUBGRAPH["name"], "offset", Str("main"))
        model = Table()
        model.add(S.MODEL["version"], "u32", 3)
        model.add(S.MODEL["operator_codes"], "offset", Vec("offset", codes))
        model.add(S.MODEL["subgraphs"], "offset", Vec("offset", [sg]))
        model.add(S.MODEL["description"], "offset", Str(self.description))
        model.add(S.MODEL["buffers"], "offset", Vec("offset", bufs))
        return build(model, S.FILE_IDENTIFIER)
