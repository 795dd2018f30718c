"""Programmatic construction of ONNX model files (protobuf wire format written by hand: no `onnx` package exists in this
environment).  Authoring tool for tests and examples: the reference ships its secondary heads as ONNX
(`internal/classifier/bat_onnx.go:252-282`, `internal/inference/onnx/custom_classifier.go:148-174`), and the real files are
absent from the snapshot, so models with the published I/O contract (`[batch, dim] -> [batch, classes]`) are written here
and read back by the engine's C++ reader and by the oracle's independent reader.

Field numbers restate onnx.proto3 (ONNX IR, third-party): ModelProto{1 ir_version, 7 graph, 8 opset_import},
GraphProto{1 node, 2 name, 5 initializer, 11 input, 12 output}, NodeProto{1 input, 2 output, 3 name, 4 op_type, 5 attribute},
AttributeProto{1 name, 2 f, 3 i, 4 s, 5 t, 7 floats, 8 ints, 20 type}, TensorProto{1 dims, 2 data_type, 8 name, 9 raw_data},
ValueInfoProto{1 name, 2 type}, TypeProto{1 tensor_type{1 elem_type, 2 shape{1 dim{1 dim_value, 2 dim_param}}}}.
"""
import struct

import numpy as np

FLOAT, INT64, FLOAT16 = 1, 7, 10


def _varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _key(field, wire):
    return _varint((field << 3) | wire)


def _ld(field, payload):
    return _key(field, 2) + _varint(len(payload)) + payload


def _vi(field, v):
    return _key(field, 0) + _varint(v)


def _str(field, s):
    return _ld(field, s.encode())


def tensor(name, array):
    a = np.ascontiguousarray(array)
    dt = {np.dtype(np.float32): FLOAT, np.dtype(np.int64): INT64, np.dtype(np.float16): FLOAT16}[a.dtype]
    out = b"".join(_vi(1, int(d)) for d in a.shape) + _vi(2, dt) + _str(8, name) + _ld(9, a.tobytes())
    return out


def _attr(name, value):
    out = _str(1, name)
    if isinstance(value, float):
        out += _key(2, 5) + struct.pack("<f", value) + _vi(20, 1)
    elif isinstance(value, (int, np.integer)):
        out += _vi(3, int(value)) + _vi(20, 2)
    elif isinstance(value, str):
        out += _ld(4, value.encode()) + _vi(20, 3)
    elif isinstance(value, np.ndarray):
        out += _ld(5, tensor("", value)) + _vi(20, 4)
    elif isinstance(value, (list, tuple)) and value and isinstance(value[0], float):
        out += _ld(7, b"".join(struct.pack("<f", v) for v in value)) + _vi(20, 6)
    elif isinstance(value, (list, tuple)):
        out += _ld(8, b"".join(_varint(int(v)) for v in value)) + _vi(20, 7)
    else:
        raise TypeError(f"attribute {name}: {type(value)}")
    return out


def _value_info(name, shape, elem=FLOAT):
    dims = b""
    for d in shape:
        dims += _ld(1, _str(2, d) if isinstance(d, str) else _vi(1, int(d)))
    ttype = _vi(1, elem) + _ld(2, dims)
    return _str(1, name) + _ld(2, _ld(1, ttype))


class OnnxBuilder:
    def __init__(self, name="graph", opset=17):
        self.nodes, self.inits, self.inputs, self.outputs = [], [], [], []
        self.name, self.opset = name, opset
        self._n = 0

    def input(self, name, shape):
        self.inputs.append(_value_info(name, shape))
        return name

    def output(self, name, shape):
        self.outputs.append(_value_info(name, shape))

    def init(self, array, name=None, as_graph_input=False):
        self._n += 1
        name = name or f"init{self._n}"
        self.inits.append(tensor(name, array))
        if as_graph_input:                        # old exporters list initializers among the graph inputs too
            self.inputs.append(_value_info(name, np.asarray(array).shape))
        return name

    def node(self, op, inputs, n_out=1, name=None, **attrs):
        self._n += 1
        outs = [f"{op.lower()}_{self._n}_{i}" for i in range(n_out)]
        body = b"".join(_str(1, i) for i in inputs) + b"".join(_str(2, o) for o in outs)
        body += _str(3, name or f"{op}_{self._n}") + _str(4, op)
        body += b"".join(_ld(5, _attr(k, v)) for k, v in attrs.items())
        self.nodes.append(body)
        return outs[0] if n_out == 1 else outs

    def finish(self):
        graph = b"".join(_ld(1, n) for n in self.nodes) + _str(2, self.name)
        graph += b"".join(_ld(5, t) for t in self.inits)
        graph += b"".join(_ld(11, i) for i in self.inputs) + b"".join(_ld(12, o) for o in self.outputs)
        opset = _str(1, "") + _vi(2, self.opset)
        return _vi(1, 8) + _str(2, "birdnet-go_amd onnx_build") + _ld(7, graph) + _ld(8, opset)


def build_dense_head(dims, style="gemm", hidden_act="Relu", final=None, seed=11, batch_dim="N", fp16_weights=False):
    """Embedding -> class head in the forms exporters emit: `gemm` (torch.nn.Linear: Gemm with transB=1), `matmul`
    (tf2onnx / keras: MatMul [K,N] + Add), `bn` (MatMul + BatchNormalization).  -> (bytes, weights list)"""
    rng = np.random.default_rng(seed)
    b = OnnxBuilder()
    t = b.input("embedding", [batch_dim, dims[0]])
    ws = []
    for li in range(len(dims) - 1):
        cin, cout = dims[li], dims[li + 1]
        last = li == len(dims) - 2
        w = (rng.standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)
        bias = (rng.standard_normal(cout) * 0.1).astype(np.float32)
        if fp16_weights:
            w = w.astype(np.float16).astype(np.float32)
        ws.append((w, bias))
        if style == "gemm":
            wn = b.init(w.astype(np.float16) if fp16_weights else w, f"fc{li}.weight", as_graph_input=li == 0)
            t = b.node("Gemm", [t, wn, b.init(bias, f"fc{li}.bias")], alpha=1.0, beta=1.0, transB=1)
        else:
            t = b.node("MatMul", [t, b.init(np.ascontiguousarray(w.T), f"dense{li}/kernel")])
            if style == "bn" and not last:
                g_, be, mu, va = (rng.uniform(0.5, 1.5, cout), rng.standard_normal(cout) * 0.1, rng.standard_normal(cout) * 0.1,
                                  rng.uniform(0.5, 1.5, cout))
                t = b.node("BatchNormalization", [t] + [b.init(v.astype(np.float32)) for v in (g_, be, mu, va)], epsilon=1e-3)
                ws[-1] = (w, bias, tuple(v.astype(np.float32) for v in (g_, be, mu, va)))
            else:
                t = b.node("Add", [t, b.init(bias, f"dense{li}/bias")])
        if not last:
            t = b.node(hidden_act, [t]) if hidden_act != "LeakyRelu" else b.node("LeakyRelu", [t], alpha=0.1)
            if li == 0:
                t = b.node("Dropout", [t])          # inference-time identity
    if final:
        t = b.node(final, [t]) if final != "Softmax" else b.node("Softmax", [t], axis=-1)
    t = b.node("Identity", [t])
    b.output(t, [batch_dim, dims[-1]])
    return b.finish(), ws


def build_cnn(in_shape=(1, 40, 56), stem=16, blocks=((1, 3, 1, 8), (4, 3, 2, 16), (4, 5, 1, 16), (6, 3, 2, 24)), top=48, n_classes=21,
              style="torch", seed=5, emit_embedding=False, nhwc_input=False):
    """Convolutional classifier on an image-like input (a spectrogram), in the NCHW vocabulary exporters emit:
    Conv (+bias), Sigmoid * Mul (swish), depthwise Conv (group = C), squeeze-excite as GlobalAveragePool -> Conv 1x1 ->
    swish -> Conv 1x1 -> Sigmoid -> Mul, residual Add, then GlobalAveragePool -> Flatten -> Gemm.
    style "torch": explicit symmetric pads, Flatten + Gemm;  "tf": auto_pad SAME_UPPER, ReduceMean(axes 2,3, keepdims 0) +
    MatMul + Add, BatchNormalization left unfolded after the stem, HardSigmoid in the last block's gate, an AveragePool.
    nhwc_input: the graph input is [N, H, W, C] followed by the Transpose tf2onnx puts in front of the first Conv.
    -> bytes"""
    rng = np.random.default_rng(seed)
    b = OnnxBuilder()
    C, H, W = in_shape
    if nhwc_input:
        t = b.input("spectrogram", ["N", H, W, C])
        t = b.node("Transpose", [t], perm=[0, 3, 1, 2])
    else:
        t = b.input("spectrogram", ["N", C, H, W])
    tf = style == "tf"

    def conv(t, cin, cout, k, s, groups=1, gain=1.4, bias=True):
        w = (rng.standard_normal((cout, cin // groups, k, k)) * gain / np.sqrt(k * k * cin / groups)).astype(np.float32)
        ins = [t, b.init(w)]
        if bias:
            ins.append(b.init((rng.standard_normal(cout) * 0.1).astype(np.float32)))
        if tf:
            return b.node("Conv", ins, kernel_shape=[k, k], strides=[s, s], group=groups, auto_pad="SAME_UPPER")
        return b.node("Conv", ins, kernel_shape=[k, k], strides=[s, s], group=groups, pads=[k // 2] * 4, dilations=[1, 1])

    def swish(t):
        return b.node("Mul", [t, b.node("Sigmoid", [t])])

    t = conv(t, C, stem, 3, 2, bias=not tf)
    if tf:                                               # unfolded batch norm after the stem
        p = [rng.uniform(0.5, 1.5, stem), rng.standard_normal(stem) * 0.1, rng.standard_normal(stem) * 0.1, rng.uniform(0.5, 1.5, stem)]
        t = b.node("BatchNormalization", [t] + [b.init(v.astype(np.float32)) for v in p], epsilon=1e-3)
    t = swish(t)
    cin = stem
    for bi, (er, k, s, cout) in enumerate(blocks):
        inp, mid = t, cin * er
        if er != 1:
            t = swish(conv(t, cin, mid, 1, 1, gain=1.6))
        t = swish(conv(t, mid, mid, k, s, groups=mid, gain=1.6))
        cse = max(1, cin // 4)
        g_ = b.node("GlobalAveragePool", [t])
        g_ = swish(conv(g_, mid, cse, 1, 1, gain=1.0))
        g_ = conv(g_, cse, mid, 1, 1, gain=1.0)
        g_ = b.node("HardSigmoid", [g_], alpha=0.2, beta=0.5) if (tf and bi == len(blocks) - 1) else b.node("Sigmoid", [g_])
        t = b.node("Mul", [t, g_])
        t = conv(t, mid, cout, 1, 1)
        if s == 1 and cin == cout:
            t = b.node("Add", [t, inp])
        cin = cout
    t = swish(conv(t, cin, top, 1, 1, gain=1.6))
    if tf:
        t = b.node("AveragePool", [t], kernel_shape=[2, 2], strides=[2, 2])
        emb = b.node("ReduceMean", [t], axes=[2, 3], keepdims=0)
        wh = (rng.standard_normal((top, n_classes)) * 2.0 / np.sqrt(top)).astype(np.float32)
        t = b.node("Add", [b.node("MatMul", [emb, b.init(wh)]), b.init((rng.standard_normal(n_classes) * 0.5).astype(np.float32))])
    else:
        emb = b.node("Flatten", [b.node("GlobalAveragePool", [t])], axis=1)
        wh = (rng.standard_normal((n_classes, top)) * 2.0 / np.sqrt(top)).astype(np.float32)
        t = b.node("Gemm", [emb, b.init(wh), b.init((rng.standard_normal(n_classes) * 0.5).astype(np.float32))], alpha=1.0, beta=1.0, transB=1)
    b.output(t, ["N", n_classes])
    if emit_embedding:
        b.output(emb, ["N", top])
    return b.finish()
