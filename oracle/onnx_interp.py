"""ORACLE (test infrastructure only — never imported by the product path).

Independent reader + numpy executor for the ONNX graphs the reference runs through ONNX Runtime 1.25.1
(`github.com/yalue/onnxruntime_go v1.30.1`, third-party, absent from /root/reference): secondary dense heads
(`internal/inference/onnx/custom_classifier.go:148-174` CustomClassifier.PredictRaw; `internal/classifier/bat_onnx.go:252-282`)
and convolutional classifiers (`internal/inference/onnx/classifier.go:268-430`: NCHW Conv / pooling / reductions).
Each operator restates the ONNX operator specification (Gemm: Y = alpha*A'*B' + beta*C; MatMul; BatchNormalization
inference form; Softmax over `axis`; elementwise ops with numpy broadcasting).  PARITY UNPINNED against ONNX Runtime itself
(no runtime and no real head files exist here); what this pins is engine-vs-specification on files both readers parse.

The protobuf wire parser below shares no code with the product's C++ reader or with birdnet-go_amd/onnx_build.py.
"""
import struct

import numpy as np


def _fields(buf):
    """yield (field, wire, value) with value = int (wire 0), bytes (wire 1, 2, 5)."""
    mv = memoryview(buf)
    i, n = 0, len(mv)
    while i < n:
        key = 0
        shift = 0
        while True:
            b = mv[i]
            i += 1
            key |= (b & 0x7F) << shift
            shift += 7
            if not b & 0x80:
                break
        field, wire = key >> 3, key & 7
        if wire == 0:
            v = 0
            shift = 0
            while True:
                b = mv[i]
                i += 1
                v |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            yield field, wire, v
        elif wire == 1:
            yield field, wire, bytes(mv[i:i + 8])
            i += 8
        elif wire == 5:
            yield field, wire, bytes(mv[i:i + 4])
            i += 4
        elif wire == 2:
            ln = 0
            shift = 0
            while True:
                b = mv[i]
                i += 1
                ln |= (b & 0x7F) << shift
                shift += 7
                if not b & 0x80:
                    break
            yield field, wire, bytes(mv[i:i + ln])
            i += ln
        else:
            raise ValueError(f"unsupported wire type {wire}")


def _signed(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _packed_varints(b):
    out, v, shift = [], 0, 0
    for byte in b:
        v |= (byte & 0x7F) << shift
        shift += 7
        if not byte & 0x80:
            out.append(_signed(v))
            v, shift = 0, 0
    return out


_NP = {1: np.float32, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64}


def _tensor(buf):
    dims, dtype, name, raw, fdata, idata = [], 0, "", None, [], []
    for f, w, v in _fields(buf):
        if f == 1:
            dims += [_signed(v)] if w == 0 else _packed_varints(v)
        elif f == 2:
            dtype = v
        elif f == 8:
            name = v.decode()
        elif f == 9:
            raw = v
        elif f == 4:
            fdata += [struct.unpack("<f", v)[0]] if w == 5 else list(np.frombuffer(v, "<f4"))
        elif f == 7:
            idata += [_signed(v)] if w == 0 else _packed_varints(v)
    if raw is not None:
        a = np.frombuffer(raw, _NP[dtype]).reshape(dims)
    elif fdata:
        a = np.asarray(fdata, np.float32).reshape(dims)
    else:
        a = np.asarray(idata, np.int64).reshape(dims)
    return name, a


def _attr(buf):
    name, val = "", None
    for f, w, v in _fields(buf):
        if f == 1:
            name = v.decode()
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = _signed(v)
        elif f == 4:
            val = v.decode()
        elif f == 5:
            val = _tensor(v)[1]
        elif f == 7:
            val = (val or []) + ([struct.unpack("<f", v)[0]] if w == 5 else list(np.frombuffer(v, "<f4")))
        elif f == 8:
            val = (val or []) + ([_signed(v)] if w == 0 else _packed_varints(v))
    return name, val


def _value_name(buf):
    for f, w, v in _fields(buf):
        if f == 1:
            return v.decode()
    return ""


class OnnxModel:
    def __init__(self, blob):
        graph = None
        for f, w, v in _fields(blob):
            if f == 7:
                graph = v
        if graph is None:
            raise ValueError("not an ONNX ModelProto")
        self.nodes, self.inits, self.inputs, self.outputs = [], {}, [], []
        for f, w, v in _fields(graph):
            if f == 1:
                ins, outs, op, attrs = [], [], "", {}
                for f2, w2, v2 in _fields(v):
                    if f2 == 1:
                        ins.append(v2.decode())
                    elif f2 == 2:
                        outs.append(v2.decode())
                    elif f2 == 4:
                        op = v2.decode()
                    elif f2 == 5:
                        k, a = _attr(v2)
                        attrs[k] = a
                self.nodes.append((op, ins, outs, attrs))
            elif f == 5:
                name, a = _tensor(v)
                self.inits[name] = a
            elif f == 11:
                self.inputs.append(_value_name(v))
            elif f == 12:
                self.outputs.append(_value_name(v))
        self.runtime_inputs = [n for n in self.inputs if n not in self.inits]


def _window_pads(at, H, W, kh, kw, sh, sw, dh, dw):
    """ONNX padding attributes -> (top, left, bottom, right)."""
    ap = at.get("auto_pad") or "NOTSET"
    eh, ew = dh * (kh - 1) + 1, dw * (kw - 1) + 1
    if ap == "NOTSET":
        p = at.get("pads") or [0, 0, 0, 0]
        return p[0], p[1], p[2], p[3]
    if ap == "VALID":
        return 0, 0, 0, 0
    th = max((-(-H // sh) - 1) * sh + eh - H, 0)
    tw = max((-(-W // sw) - 1) * sw + ew - W, 0)
    if ap == "SAME_UPPER":
        return th // 2, tw // 2, th - th // 2, tw - tw // 2
    return th - th // 2, tw - tw // 2, th // 2, tw // 2


def _conv(x, w, b, at, fdt):
    """Conv (2-D, NCHW, OIHW weights, group 1 or depthwise-style groups): direct sum over the kernel window.
    A rank-3 input (Conv1d, NCW) is the same operator with a unit height."""
    if x.ndim == 3:
        at2 = dict(at)
        for k, d in (("strides", 1), ("dilations", 1), ("kernel_shape", 1)):
            if at.get(k):
                at2[k] = [d if k != "kernel_shape" else 1] + list(at[k])
        if at.get("pads"):
            at2["pads"] = [0, at["pads"][0], 0, at["pads"][1]]
        return _conv(x[:, :, None, :], w[:, :, None, :], b, at2, fdt)[:, :, 0, :]
    N, C, H, W = x.shape
    M, Cg, kh, kw = w.shape
    g = int(at.get("group", 1))
    sh, sw = at.get("strides") or [1, 1]
    dh, dw = at.get("dilations") or [1, 1]
    pt, pl, pb, pr = _window_pads(at, H, W, kh, kw, sh, sw, dh, dw)
    xp = np.pad(x, [(0, 0), (0, 0), (pt, pb), (pl, pr)])
    Ho = (H + pt + pb - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + pl + pr - (dw * (kw - 1) + 1)) // sw + 1
    y = np.zeros((N, M, Ho, Wo), fdt)
    mg = M // g
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, :, i * dh:i * dh + (Ho - 1) * sh + 1:sh, j * dw:j * dw + (Wo - 1) * sw + 1:sw]     # [N, C, Ho, Wo]
            if g == 1:
                y += np.einsum("nchw,mc->nmhw", patch, w[:, :, i, j], optimize=True)
            else:
                for q in range(g):
                    y[:, q * mg:(q + 1) * mg] += np.einsum("nchw,mc->nmhw", patch[:, q * Cg:(q + 1) * Cg], w[q * mg:(q + 1) * mg, :, i, j],
                                                          optimize=True)
    if b is not None:
        y += b.reshape(1, -1, 1, 1)
    return y


def _pool(x, at, is_max, fdt):
    N, C, H, W = x.shape
    kh, kw = at["kernel_shape"]
    sh, sw = at.get("strides") or [1, 1]
    pt, pl, pb, pr = _window_pads(at, H, W, kh, kw, sh, sw, 1, 1)
    fill = -np.inf if is_max else 0.0
    xp = np.pad(x, [(0, 0), (0, 0), (pt, pb), (pl, pr)], constant_values=fill)
    cnt = np.pad(np.ones((1, 1, H, W), fdt), [(0, 0), (0, 0), (pt, pb), (pl, pr)])
    Ho = (H + pt + pb - kh) // sh + 1
    Wo = (W + pl + pr - kw) // sw + 1
    y = np.full((N, C, Ho, Wo), fill, fdt)
    n = np.zeros((1, 1, Ho, Wo), fdt)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, :, i:i + (Ho - 1) * sh + 1:sh, j:j + (Wo - 1) * sw + 1:sw]
            if is_max:
                y = np.maximum(y, patch)
            else:
                y = y + patch
                n = n + cnt[:, :, i:i + (Ho - 1) * sh + 1:sh, j:j + (Wo - 1) * sw + 1:sw]
    return y if is_max else y / n           # count_include_pad = 0 (the default)


def run(blob, x, precision="f32"):
    """x [B, dim] -> list of outputs (float32)."""
    m = blob if isinstance(blob, OnnxModel) else OnnxModel(blob)
    fdt = np.float32 if precision == "f32" else np.float64
    vals = {k: (v.astype(fdt) if v.dtype.kind == "f" else v) for k, v in m.inits.items()}
    vals[m.runtime_inputs[0]] = np.asarray(x, np.float32).astype(fdt)
    for op, ins, outs, at in m.nodes:
        if op == "Constant":                  # (torch's exporter writes scalars and index vectors as Constant nodes)
            v = at["value"] if "value" in at else np.asarray(at["value_float"], np.float32)
            vals[outs[0]] = v.astype(fdt) if v.dtype.kind == "f" else v
            continue
        a = [vals[i] if i else None for i in ins]
        if op == "Gemm":
            A = a[0].T if at.get("transA") else a[0]
            B = a[1].T if at.get("transB") else a[1]
            y = np.asarray(at.get("alpha", 1.0), fdt) * (A @ B)
            if len(a) > 2 and a[2] is not None:
                y = y + np.asarray(at.get("beta", 1.0), fdt) * a[2]
        elif op == "MatMul":
            y = a[0] @ a[1]
        elif op in ("Add", "Sub", "Mul", "Div", "Pow", "Max", "Min"):
            y = {"Add": np.add, "Sub": np.subtract, "Mul": np.multiply, "Div": np.divide, "Pow": np.power, "Max": np.maximum,
                 "Min": np.minimum}[op](a[0], a[1])
        elif op == "Relu":
            y = np.maximum(a[0], 0)
        elif op == "Sigmoid":
            y = (1.0 / (1.0 + np.exp(-a[0]))).astype(fdt)
        elif op == "Tanh":
            y = np.tanh(a[0])
        elif op == "LeakyRelu":
            y = np.where(a[0] > 0, a[0], a[0] * np.asarray(at.get("alpha", 0.01), fdt))
        elif op == "Softmax":
            ax = at.get("axis", -1)
            z = a[0] - a[0].max(axis=ax, keepdims=True)
            e = np.exp(z)
            y = e / e.sum(axis=ax, keepdims=True)
        elif op == "BatchNormalization":
            x_, sc, bi, mu, va = a[:5]
            if x_.ndim == 4:                     # per-channel parameters of an NCHW image
                sc, bi, mu, va = (v.reshape(1, -1, 1, 1) for v in (sc, bi, mu, va))
            y = (x_ - mu) / np.sqrt(va + np.asarray(at.get("epsilon", 1e-5), fdt)) * sc + bi
        elif op in ("Identity", "Dropout", "Cast"):      # Cast: the oracle computes in one float type; index tensors are converted where they are used
            y = a[0]
        elif op == "ConstantOfShape":
            v = at.get("value")
            y = np.full([int(d) for d in np.asarray(a[0]).reshape(-1)], 0.0 if v is None else np.asarray(v).reshape(-1)[0])
        elif op == "Flatten":
            y = a[0].reshape(a[0].shape[0], -1)
        elif op == "Clip":
            lo = at.get("min", a[1] if len(a) > 1 and a[1] is not None else -np.inf)
            hi = at.get("max", a[2] if len(a) > 2 and a[2] is not None else np.inf)
            y = np.clip(a[0], lo, hi)
        elif op == "Concat":
            y = np.concatenate(a, axis=at.get("axis", 1))
        elif op == "Conv":
            y = _conv(a[0], a[1], a[2] if len(a) > 2 else None, at, fdt)
        elif op in ("MaxPool", "AveragePool"):
            y = _pool(a[0], at, op == "MaxPool", fdt)
        elif op == "GlobalAveragePool":
            y = a[0].mean(axis=(2, 3), keepdims=True, dtype=fdt)
        elif op == "ReduceMean":
            axes = at.get("axes")
            if axes is None and len(a) > 1 and a[1] is not None:
                axes = [int(v) for v in a[1]]
            y = a[0].mean(axis=tuple(axes), keepdims=bool(at.get("keepdims", 1)), dtype=fdt)
        elif op == "Transpose":
            y = np.transpose(a[0], at.get("perm") or list(range(a[0].ndim))[::-1])
        elif op == "Reshape":
            shp = [int(v) for v in a[1]]
            shp = [a[0].shape[i] if v == 0 else v for i, v in enumerate(shp)]
            shp[0] = a[0].shape[0] if shp[0] != -1 else -1       # the file says batch 1; the oracle runs any batch
            y = a[0].reshape(shp)
        elif op == "Squeeze":
            axes = at.get("axes")
            if axes is None and len(a) > 1 and a[1] is not None:
                axes = [int(v) for v in a[1]]
            y = np.squeeze(a[0], axis=tuple(axes) if axes is not None else tuple(i for i in range(1, a[0].ndim) if a[0].shape[i] == 1))
        elif op == "Unsqueeze":
            axes = at.get("axes")
            if axes is None:
                axes = [int(v) for v in a[1]]
            y = a[0]
            for ax in sorted(ax_ % (a[0].ndim + len(axes)) for ax_ in axes):
                y = np.expand_dims(y, ax)
        elif op == "HardSigmoid":
            y = np.clip(np.asarray(at.get("alpha", 0.2), fdt) * a[0] + np.asarray(at.get("beta", 0.5), fdt), 0, 1)
        elif op == "HardSwish":
            y = a[0] * np.clip(a[0] / np.asarray(6.0, fdt) + np.asarray(0.5, fdt), 0, 1)
        elif op == "Gather":                  # out = data.take(indices, axis): indices of any rank replace that axis
            y = np.take(a[0], a[1].astype(np.int64), axis=int(at.get("axis", 0)))
        elif op == "Slice":                   # opset >= 10: starts, ends, axes, steps as inputs
            starts, ends = [int(v) for v in a[1]], [int(v) for v in a[2]]
            axes = [int(v) for v in a[3]] if len(a) > 3 and a[3] is not None else list(range(len(starts)))
            steps = [int(v) for v in a[4]] if len(a) > 4 and a[4] is not None else [1] * len(starts)
            sl = [slice(None)] * a[0].ndim
            for st, en, ax, sp in zip(starts, ends, axes, steps):
                n = a[0].shape[ax]
                if sp > 0:
                    st = min(max(st + n if st < 0 else st, 0), n); en = min(max(en + n if en < 0 else en, 0), n)
                    sl[ax] = slice(st, en, sp)
                else:                         # negative step: clamp to [-1, n-1] (ONNX Slice specification)
                    st = min(max(st + n if st < 0 else st, -1), n - 1); en = min(max(en + n if en < 0 else en, -1), n - 1)
                    sl[ax] = slice(st, None if en < 0 else en, sp)
            y = a[0][tuple(sl)]
        elif op in ("ReduceSum", "ReduceMin", "ReduceMax"):
            axes = at.get("axes")
            if axes is None and len(a) > 1 and a[1] is not None:
                axes = [int(v) for v in a[1]]
            f = {"ReduceSum": np.sum, "ReduceMin": np.min, "ReduceMax": np.max}[op]
            y = f(a[0], axis=tuple(axes) if axes is not None else None, keepdims=bool(at.get("keepdims", 1)))
        elif op == "Sqrt":
            y = np.sqrt(a[0])
        elif op == "Log":
            y = np.log(a[0])
        elif op == "Exp":
            y = np.exp(a[0])
        elif op == "Abs":
            y = np.abs(a[0])
        elif op == "Neg":
            y = -a[0]
        elif op == "STFT":
            # ONNX STFT (opset 17): signal [B, T, 1] (real), frame_step, window [frame_length], frame_length; onesided ->
            # [B, frames, frame_length / 2 + 1, 2] with frames = (T - frame_length) / frame_step + 1; evaluated in double, narrowed
            sig = a[0][..., 0] if a[0].ndim == 3 else a[0]
            step = int(np.asarray(a[1]).reshape(-1)[0])
            win = a[2] if len(a) > 2 and a[2] is not None else None
            flen = int(np.asarray(a[3]).reshape(-1)[0]) if len(a) > 3 and a[3] is not None else len(win)
            nfr = (sig.shape[1] - flen) // step + 1
            idx = np.arange(nfr)[:, None] * step + np.arange(flen)[None, :]
            fr = sig[:, idx].astype(np.float64)
            if win is not None:
                fr = fr * win.astype(np.float64)
            sp = np.fft.rfft(fr, axis=-1) if at.get("onesided", 1) else np.fft.fft(fr, axis=-1)
            y = np.stack([sp.real, sp.imag], axis=-1)
        elif op == "DFT":
            # ONNX DFT (opset 17): input [..., n, 1] real or [..., n, 2] complex, attribute axis, optional dft_length
            ax = int(at.get("axis", 1))
            z = a[0][..., 0].astype(np.float64) if a[0].shape[-1] == 1 else a[0][..., 0].astype(np.float64) + 1j * a[0][..., 1].astype(np.float64)
            nfft = int(np.asarray(a[1]).reshape(-1)[0]) if len(a) > 1 and a[1] is not None else z.shape[ax]
            if at.get("inverse", 0):
                raise ValueError("oracle: inverse DFT unsupported")
            sp = np.fft.rfft(z.real, n=nfft, axis=ax) if at.get("onesided", 0) else np.fft.fft(z, n=nfft, axis=ax)
            y = np.stack([sp.real, sp.imag], axis=-1)
        elif op == "Pad":
            pads = at.get("pads")
            if pads is None:
                pads = [int(v) for v in a[1]]
            r = a[0].ndim
            y = np.pad(a[0], [(pads[k], pads[r + k]) for k in range(r)])
        else:
            raise ValueError(f"oracle: unsupported ONNX op {op}")
        vals[outs[0]] = np.asarray(y, fdt)
    return [np.asarray(vals[o], np.float32) for o in m.outputs]
