"""ORACLE (test infrastructure only — never imported by the product path).

Op-by-op numpy execution of a TFLite graph: the CPU restatement of what the reference's hot call
`interpreter.Invoke()` (`internal/inference/tflite/classifier.go:107`) computes for a float model.
The arithmetic lives in TensorFlow Lite 2.17.1 (third-party, absent from /root/reference; pinned
in reference `Taskfile.yml:6`, `go.mod:42`), so each op below restates TFLite's *published*
builtin-op semantics (tensorflow/lite/kernels/*.cc reference kernels): NHWC activations, OHWI conv
filters, TF "SAME"/"VALID" padding, RFFT2D evaluated in double precision then narrowed to
complex64, CAST complex64->float32 keeping the real part, etc.

PARITY UNPINNED for network forward values: neither the real v2.4 weights nor a runnable TFLite
exist in this environment, and the reference's CI never executes a
real forward pass (section 4).  What IS pinned: the in-tree Go arithmetic either side of Invoke
(see postproc.py / pcm.py / ultrasonic.py, checked against the reference tests' known answers).

`precision="f32"` mirrors TFLite float kernels (float32 storage and accumulation via BLAS);
`precision="f64"` promotes everything to float64 and serves as the arbiter when the f32 oracle and
the HIP path differ by rounding.
"""
import numpy as np

from .tflite_reader import Model, read_model

_ACT = {0: None, 1: "relu", 2: "relu_n1_to_1", 3: "relu6", 4: "tanh"}


def _act(x, code):
    a = _ACT.get(code)
    if a is None:
        return x
    if a == "relu":
        return np.maximum(x, 0)
    if a == "relu6":
        return np.clip(x, 0, 6)
    if a == "relu_n1_to_1":
        return np.clip(x, -1, 1)
    if a == "tanh":
        return np.tanh(x)
    raise ValueError(a)


def _same_pad(n, k, s, d):
    out = (n + s - 1) // s
    total = max((out - 1) * s + (k - 1) * d + 1 - n, 0)
    return out, total // 2, total - total // 2


def _valid_out(n, k, s, d):
    return (n - ((k - 1) * d + 1)) // s + 1


def _geom(n, k, s, d, padding):
    if padding == 0:
        return _same_pad(n, k, s, d)
    return _valid_out(n, k, s, d), 0, 0


def conv2d(x, w, b, o):
    """TFLite CONV_2D: x NHWC, w [O,kh,kw,I], b [O]."""
    N, H, W, C = x.shape
    O, kh, kw, I = w.shape
    assert I == C
    sh, sw = o.get("stride_h") or 1, o.get("stride_w") or 1
    dh, dw = o.get("dil_h") or 1, o.get("dil_w") or 1
    Ho, pt, pb = _geom(H, kh, sh, dh, o.get("padding", 0))
    Wo, pl, pr = _geom(W, kw, sw, dw, o.get("padding", 0))
    if kh == 1 and kw == 1 and sh == 1 and sw == 1:
        y = x.reshape(-1, C) @ w.reshape(O, C).T
        y = y.reshape(N, H, W, O)
    else:
        xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
        y = np.zeros((N, Ho, Wo, O), x.dtype)
        for i in range(kh):
            for j in range(kw):
                sl = xp[:, i * dh:i * dh + (Ho - 1) * sh + 1:sh, j * dw:j * dw + (Wo - 1) * sw + 1:sw, :]
                y += (sl.reshape(-1, C) @ w[:, i, j, :].T).reshape(N, Ho, Wo, O)
    if b is not None:
        y = y + b
    return _act(y, o.get("act", 0))


def depthwise_conv2d(x, w, b, o):
    """TFLite DEPTHWISE_CONV_2D: w [1,kh,kw,C*mult]; output channel c*mult+q reads input channel c
    (tflite reference depthwiseconv_float.h)."""
    N, H, W, C = x.shape
    _, kh, kw, CO = w.shape
    if CO % C:
        raise ValueError("depthwise filter channels are not a multiple of the input channels")
    mult = CO // C
    sh, sw = o.get("stride_h") or 1, o.get("stride_w") or 1
    dh, dw = o.get("dil_h") or 1, o.get("dil_w") or 1
    Ho, pt, pb = _geom(H, kh, sh, dh, o.get("padding", 0))
    Wo, pl, pr = _geom(W, kw, sw, dw, o.get("padding", 0))
    xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
    y = np.zeros((N, Ho, Wo, CO), x.dtype)
    for i in range(kh):
        for j in range(kw):
            sl = xp[:, i * dh:i * dh + (Ho - 1) * sh + 1:sh, j * dw:j * dw + (Wo - 1) * sw + 1:sw, :]
            if mult != 1:
                sl = np.repeat(sl, mult, axis=3)
            y += sl * w[0, i, j, :]
    if b is not None:
        y = y + b
    return _act(y, o.get("act", 0))


def pool2d(x, o, mode):
    N, H, W, C = x.shape
    kh, kw = o["filter_h"], o["filter_w"]
    sh, sw = o.get("stride_h") or 1, o.get("stride_w") or 1
    Ho, pt, pb = _geom(H, kh, sh, 1, o.get("padding", 0))
    Wo, pl, pr = _geom(W, kw, sw, 1, o.get("padding", 0))
    y = np.empty((N, Ho, Wo, C), x.dtype)
    for a in range(Ho):
        for c in range(Wo):
            h0, w0 = a * sh - pt, c * sw - pl
            win = x[:, max(h0, 0):min(h0 + kh, H), max(w0, 0):min(w0 + kw, W), :]
            y[:, a, c, :] = win.mean(axis=(1, 2)) if mode == "avg" else win.max(axis=(1, 2))
    return _act(y, o.get("act", 0))


def _batch_reshape(x, shape, batch):
    shape = [int(s) for s in shape]
    n = int(np.prod(shape)) if -1 not in shape else None
    if n is None or n == x.size:
        return x.reshape(shape)
    if shape and shape[0] == 1 and x.size == batch * n:
        return x.reshape([batch] + shape[1:])
    if x.size == batch * n:
        return x.reshape([batch * shape[0]] + shape[1:])
    raise ValueError(f"RESHAPE {x.shape} -> {shape} (batch {batch})")


def _torch_conv(x, w, b, o, depthwise):
    """Same op semantics as conv2d/depthwise_conv2d above, evaluated by torch's CPU convolution (oneDNN): used only for
    the timed CPU baseline (bench.py cpu_baseline) where the numpy tap loops would understate what a CPU can do."""
    import torch
    import torch.nn.functional as F
    N, H, W, C = x.shape
    sh, sw = o.get("stride_h") or 1, o.get("stride_w") or 1
    dh, dw = o.get("dil_h") or 1, o.get("dil_w") or 1
    if depthwise:
        kh, kw = w.shape[1], w.shape[2]
        wt = torch.from_numpy(np.ascontiguousarray(np.transpose(w, (3, 0, 1, 2))))      # [C,1,kh,kw]
        groups = C
    else:
        kh, kw = w.shape[1], w.shape[2]
        wt = torch.from_numpy(np.ascontiguousarray(np.transpose(w, (0, 3, 1, 2))))      # [O,I,kh,kw]
        groups = 1
    Ho, pt, pb = _geom(H, kh, sh, dh, o.get("padding", 0))
    Wo, pl, pr = _geom(W, kw, sw, dw, o.get("padding", 0))
    xt = torch.from_numpy(np.ascontiguousarray(x)).permute(0, 3, 1, 2)
    if pt or pb or pl or pr:
        xt = F.pad(xt, (pl, pr, pt, pb))
    y = F.conv2d(xt, wt, None if b is None else torch.from_numpy(np.ascontiguousarray(b)), stride=(sh, sw),
                 dilation=(dh, dw), groups=groups)
    return _act(y.permute(0, 2, 3, 1).contiguous().numpy(), o.get("act", 0))


class Interpreter:
    """Executes the graph for a batch of inputs (the graph itself is authored with batch 1).
    conv_backend="torch" swaps the two convolution ops for torch's CPU kernels (same semantics; fp32 only)."""

    def __init__(self, model, precision="f32", conv_backend="numpy"):
        self.conv_backend = conv_backend if precision == "f32" else "numpy"
        self.m = model if isinstance(model, Model) else read_model(model)
        assert precision in ("f32", "f64")
        self.fdt = np.float32 if precision == "f32" else np.float64
        self.cdt = np.complex64 if precision == "f32" else np.complex128
        inp = self.m.tensors[self.m.inputs[0]]
        self.in_shape = [int(v) for v in inp.shape[1:]]      # per-clip input block ([n_samples] for the audio models)
        self.n_samples = int(np.prod(self.in_shape))        # clip length, or feature width for dense-only graphs
        self.out_dims = [int(self.m.tensors[o].shape[-1]) for o in self.m.outputs]

    def _const(self, idx):
        t = self.m.tensors[idx]
        if t.data is None:
            return None
        if t.dtype in (np.float32, np.float16):
            return t.data.astype(self.fdt)
        return t.data

    def invoke(self, samples, keep=None):
        """samples [B, n_samples] float32 -> list of output arrays (float32)."""
        x = np.asarray(samples, np.float32)
        if x.ndim == 1:
            x = x[None, :]
        x = x.reshape(x.shape[0], -1)
        if x.shape[1] != self.n_samples:
            raise ValueError(f"input size mismatch: expected {self.n_samples} samples, got {x.shape[1]}")
        B = x.shape[0]
        x = x.reshape([B] + self.in_shape)
        vals = {self.m.inputs[0]: x.astype(self.fdt)}

        def get(i):
            if i < 0:
                return None
            if i in vals:
                return vals[i]
            c = self._const(i)
            if c is None:
                raise ValueError(f"tensor {i} ({self.m.tensors[i].name}) has no value")
            return c

        fdt = self.fdt
        for op in self.m.ops:
            a = [get(i) for i in op.inputs]
            o = op.opts
            n = op.name
            if n == "CONV_2D":
                if self.conv_backend == "torch":
                    y = _torch_conv(a[0], a[1], a[2] if len(a) > 2 else None, o, False)
                else:
                    y = conv2d(a[0], a[1], a[2] if len(a) > 2 else None, o)
            elif n == "DEPTHWISE_CONV_2D":
                if self.conv_backend == "torch":
                    y = _torch_conv(a[0], a[1], a[2] if len(a) > 2 else None, o, True)
                else:
                    y = depthwise_conv2d(a[0], a[1], a[2] if len(a) > 2 else None, o)
            elif n == "DEQUANTIZE":
                y = np.asarray(a[0]).astype(fdt)      # float16 -> float32 (tflite dequantize.cc, kTfLiteFloat16 branch)
            elif n == "FULLY_CONNECTED":
                w = a[1]
                xin = a[0]
                lead = xin.shape[:-1] if o.get("keep_num_dims") else (-1,)
                y = xin.reshape(-1, w.shape[1]) @ w.T
                if len(a) > 2 and a[2] is not None:
                    y = y + a[2]
                y = _act(y, o.get("act", 0)).reshape(*lead, w.shape[0])
            elif n in ("ADD", "SUB", "MUL", "DIV"):
                f = {"ADD": np.add, "SUB": np.subtract, "MUL": np.multiply, "DIV": np.divide}[n]
                y = _act(f(a[0], a[1]), o.get("act", 0))
            elif n == "LOGISTIC":
                y = (1.0 / (1.0 + np.exp(-a[0]))).astype(a[0].dtype)
            elif n == "RELU":
                y = np.maximum(a[0], 0)
            elif n == "RELU6":
                y = np.clip(a[0], 0, 6)
            elif n == "HARD_SWISH":
                y = a[0] * np.clip(a[0] + 3, 0, 6) / 6
            elif n in ("MEAN", "SUM", "REDUCE_MAX", "REDUCE_MIN"):
                axes = tuple(int(v) % a[0].ndim for v in np.atleast_1d(a[1]))
                f = {"MEAN": np.mean, "SUM": np.sum, "REDUCE_MAX": np.max, "REDUCE_MIN": np.min}[n]
                y = f(a[0], axis=axes, keepdims=bool(o.get("keep_dims")))
                y = np.asarray(y, a[0].dtype)
            elif n == "RESHAPE":
                shape = o.get("new_shape") or [int(v) for v in a[1]]
                y = _batch_reshape(a[0], shape, B)
            elif n == "SQUEEZE":
                dims = tuple(o.get("squeeze_dims") or [i for i, s in enumerate(a[0].shape) if s == 1 and i > 0])
                y = np.squeeze(a[0], axis=dims)
            elif n == "EXPAND_DIMS":
                y = np.expand_dims(a[0], int(np.atleast_1d(a[1])[0]))
            elif n == "TRANSPOSE":
                y = np.transpose(a[0], [int(v) for v in a[1]])
            elif n == "CONCATENATION":
                # constants are authored for batch 1: repeat them along the batch like every per-clip tensor
                parts = [np.repeat(v, B, axis=0) if (v.ndim and v.shape[0] == 1 and B > 1 and any(u.shape[0] == B for u in a)) else v
                         for v in a]
                y = _act(np.concatenate(parts, axis=o.get("axis", 0)), o.get("act", 0))
            elif n == "PAD":
                y = np.pad(a[0], [(int(p[0]), int(p[1])) for p in a[1]])
            elif n == "GATHER":
                if o.get("batch_dims"):
                    raise ValueError("GATHER batch_dims != 0 unsupported")
                y = np.take(a[0], a[1], axis=o.get("axis", 0))
            elif n == "REVERSE_V2":
                y = np.flip(a[0], axis=tuple(int(v) for v in np.atleast_1d(a[1])))
            elif n == "POW":
                y = np.power(a[0], np.asarray(a[1], a[0].dtype))
            elif n == "RFFT2D":
                fl = [int(v) for v in a[1]]
                # TFLite rfft2d.cc runs Ooura fft2d on doubles, then narrows to complex64.
                y = np.fft.rfft2(a[0].astype(np.float64), s=fl, axes=(-2, -1)).astype(self.cdt)
            elif n == "CAST":
                src = a[0]
                out_t = {0: fdt, 2: np.int32, 4: np.int64, 8: self.cdt}[o.get("out_type", 0)]
                if np.iscomplexobj(src) and not np.issubdtype(out_t, np.complexfloating):
                    src = src.real           # TFLite cast.cc: complex64 -> real keeps std::real()
                y = src.astype(out_t)
            elif n == "REAL":
                y = a[0].real.astype(fdt)
            elif n == "IMAG":
                y = a[0].imag.astype(fdt)
            elif n == "COMPLEX_ABS":
                y = np.abs(a[0]).astype(fdt)
            elif n == "SOFTMAX":
                z = a[0] * (o.get("beta") or 1.0)
                z = z - z.max(axis=-1, keepdims=True)
                e = np.exp(z)
                y = (e / e.sum(axis=-1, keepdims=True)).astype(a[0].dtype)
            elif n == "AVERAGE_POOL_2D":
                y = pool2d(a[0], o, "avg")
            elif n == "MAX_POOL_2D":
                y = pool2d(a[0], o, "max")
            elif n == "BATCH_MATMUL":
                l = np.swapaxes(a[0], -1, -2) if o.get("adj_x") else a[0]
                r = np.swapaxes(a[1], -1, -2) if o.get("adj_y") else a[1]
                y = l @ r
            elif n == "STRIDED_SLICE":
                begin, end, strides = ([int(v) for v in t] for t in a[1:4])
                sl = []
                for d in range(len(begin)):
                    b0 = None if (o.get("begin_mask", 0) >> d) & 1 else begin[d]
                    e0 = None if (o.get("end_mask", 0) >> d) & 1 else end[d]
                    if (o.get("shrink_axis_mask", 0) >> d) & 1:
                        sl.append(begin[d])
                    else:
                        sl.append(slice(b0, e0, strides[d]))
                y = a[0][tuple(sl)]
            elif n in ("MAXIMUM", "MINIMUM"):
                y = (np.maximum if n == "MAXIMUM" else np.minimum)(a[0], a[1])
            elif n == "SQUARED_DIFFERENCE":
                d = np.subtract(a[0], a[1])
                y = d * d
            elif n in ("EXP", "LOG", "SQRT", "ABS", "NEG", "SQUARE", "TANH", "SIN", "COS", "FLOOR", "CEIL", "ROUND", "RSQRT"):
                f = {"EXP": np.exp, "LOG": np.log, "SQRT": np.sqrt, "ABS": np.abs, "NEG": np.negative, "SQUARE": np.square,
                     "TANH": np.tanh, "SIN": np.sin, "COS": np.cos, "FLOOR": np.floor, "CEIL": np.ceil,
                     "ROUND": np.rint,                       # TFLite round.h: round half to even
                     "RSQRT": lambda v: 1.0 / np.sqrt(v)}[n]
                with np.errstate(all="ignore"):
                    y = np.asarray(f(a[0]), a[0].dtype)
            elif n == "LEAKY_RELU":
                y = np.where(a[0] > 0, a[0], a[0] * np.asarray(o.get("alpha", 0.0), a[0].dtype))
            elif n == "ELU":
                y = np.where(a[0] > 0, a[0], np.expm1(np.minimum(a[0], 0))).astype(a[0].dtype)
            elif n == "RELU_N1_TO_1":
                y = np.clip(a[0], -1, 1)
            elif n == "GELU":
                x64 = a[0].astype(np.float64)
                if o.get("approximate"):
                    y = 0.5 * x64 * (1.0 + np.tanh(0.7978845608028654 * (x64 + 0.044715 * x64 ** 3)))
                else:
                    from scipy.special import erf
                    y = 0.5 * x64 * (1.0 + erf(x64 / np.sqrt(2.0)))
                y = y.astype(a[0].dtype)
            elif n == "REDUCE_PROD":
                axes = tuple(int(v) % a[0].ndim for v in np.atleast_1d(a[1]))
                y = np.asarray(np.prod(a[0], axis=axes, keepdims=bool(o.get("keep_dims"))), a[0].dtype)
            elif n == "PADV2":
                y = np.pad(a[0], [(int(p[0]), int(p[1])) for p in a[1]], constant_values=float(np.asarray(a[2]).reshape(-1)[0]))
            elif n == "SLICE":
                begin, size = [int(v) for v in a[1]], [int(v) for v in a[2]]
                y = a[0][tuple(slice(b, None if sz < 0 else b + sz) for b, sz in zip(begin, size))]
            elif n == "SPLIT":
                # inputs: axis, value (tflite split.cc); equal parts
                parts = np.split(a[1], len(op.outputs), axis=int(np.asarray(a[0]).reshape(-1)[0]))
                for t, part in zip(op.outputs, parts):
                    vals[t] = part
                    if keep is not None:
                        keep[t] = part
                continue
            else:
                raise ValueError(f"oracle: unsupported op {n}")
            vals[op.outputs[0]] = y
            if keep is not None:
                keep[op.outputs[0]] = y
        return [np.asarray(vals[o], np.float32).reshape(B, -1) for o in self.m.outputs]

    # reference-shaped convenience (inference.Classifier semantics, backend.go:8-19)
    def predict(self, samples):
        return self.invoke(samples)[0]
