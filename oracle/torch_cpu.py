"""ORACLE (test infrastructure only - never imported by the product path).

Whole-model CPU execution of a TFLite float graph on torch-CPU (oneDNN convolutions, MKL/pocketfft FFT): the *timed* CPU
baseline of bench.py (`cpu_baseline`), i.e. the closest thing to "the reference's CPU path on this box's host cores" that
can exist here - the reference runs TFLite 2.17.1 + XNNPACK (`internal/inference/tflite/classifier.go:38-119`,
`threads.go:13-30`, loop shape `cmd/benchmark/benchmark.go:99-133`), neither of which is present.  Same op
semantics as oracle/interp.py (the parity oracle), which stays the arbiter: tests/test_torch_cpu.py pins this executor
against it.  Differences are purely about speed: activations stay NHWC in memory and are handed to the convolutions as
channels_last views (no layout copies), LOGISTIC + MUL pairs run as one SiLU, constants are converted once.

RESTATEMENT BASELINE - NOT TFLite.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .tflite_reader import Model, read_model


def _same_pad(n, k, s, d):
    out = (n + s - 1) // s
    total = max((out - 1) * s + (k - 1) * d + 1 - n, 0)
    return total // 2, total - total // 2


_ACT = {0: None, 1: "relu", 2: "relu_n1_to_1", 3: "relu6", 4: "tanh"}


def _act(y, code):
    a = _ACT.get(code)
    if a is None:
        return y
    if a == "relu":
        return F.relu(y)
    if a == "relu6":
        return y.clamp(0, 6)
    if a == "relu_n1_to_1":
        return y.clamp(-1, 1)
    return torch.tanh(y)


class TorchCPU:
    """invoke(samples [B, n]) -> list of float32 numpy outputs, like oracle.interp.Interpreter.invoke."""

    def __init__(self, model):
        self.m = model if isinstance(model, Model) else read_model(model)
        inp = self.m.tensors[self.m.inputs[0]]
        self.in_shape = [int(v) for v in inp.shape[1:]]
        self.n_samples = int(np.prod(self.in_shape))
        m = self.m
        # consumers per tensor (for the LOGISTIC + MUL -> SiLU peephole)
        cons = {}
        for oi, op in enumerate(m.ops):
            for t in op.inputs:
                cons.setdefault(t, []).append(oi)
        outs = set(m.outputs)
        self.silu = {}          # MUL op index -> x tensor (the LOGISTIC feeding it is skipped)
        self.skip = set()
        for oi, op in enumerate(m.ops):
            if op.name != "LOGISTIC":
                continue
            t, x = op.outputs[0], op.inputs[0]
            cs = cons.get(t, [])
            if len(cs) == 1 and t not in outs:
                mul = m.ops[cs[0]]
                if mul.name == "MUL" and sorted(mul.inputs) == sorted([x, t]) and not mul.opts.get("act"):
                    self.silu[cs[0]] = x
                    self.skip.add(oi)
        self.consts = {}
        self.convw = {}

    def _const(self, i):
        c = self.consts.get(i)
        if c is None:
            t = self.m.tensors[i]
            if t.data is None:
                raise ValueError(f"tensor {i} ({t.name}) has no value")
            d = t.data.astype(np.float32) if t.data.dtype in (np.float32, np.float16) else t.data
            c = torch.from_numpy(np.ascontiguousarray(d))
            self.consts[i] = c
        return c

    def _conv_weight(self, i, depthwise):
        w = self.convw.get(i)
        if w is None:
            a = self.m.tensors[i].data.astype(np.float32)
            a = np.transpose(a, (3, 0, 1, 2)) if depthwise else np.transpose(a, (0, 3, 1, 2))      # -> [O, I/g, kh, kw]
            w = torch.from_numpy(np.ascontiguousarray(a)).contiguous(memory_format=torch.channels_last)
            self.convw[i] = w
        return w

    @torch.inference_mode()
    def invoke(self, samples):
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(samples, np.float32)))
        if x.ndim == 1:
            x = x[None, :]
        B = x.shape[0]
        x = x.reshape([B] + self.in_shape)
        m = self.m
        vals = {m.inputs[0]: x}

        def get(i):
            if i < 0:
                return None
            v = vals.get(i)
            return v if v is not None else self._const(i)

        for oi, op in enumerate(m.ops):
            if oi in self.skip:
                continue
            o, n = op.opts, op.name
            if oi in self.silu:
                vals[op.outputs[0]] = F.silu(get(self.silu[oi]))
                continue
            a = [get(i) for i in op.inputs]
            if n in ("CONV_2D", "DEPTHWISE_CONV_2D"):
                dwc = n == "DEPTHWISE_CONV_2D"
                xin = a[0]
                N, H, W, C = xin.shape
                w = self._conv_weight(op.inputs[1], dwc)
                kh, kw = w.shape[2], w.shape[3]
                sh, sw = o.get("stride_h") or 1, o.get("stride_w") or 1
                dh, dw = o.get("dil_h") or 1, o.get("dil_w") or 1
                xt = xin.permute(0, 3, 1, 2)                       # NHWC memory seen as an NCHW channels_last tensor
                if o.get("padding", 0) == 0:
                    pt, pb = _same_pad(H, kh, sh, dh)
                    pl, pr = _same_pad(W, kw, sw, dw)
                    if pt == pb and pl == pr:
                        pad = (pt, pl)
                    else:
                        xt = F.pad(xt, (pl, pr, pt, pb))
                        pad = 0
                else:
                    pad = 0
                b = a[2] if len(a) > 2 else None
                if kh == 1 and kw == 1 and sh == 1 and sw == 1 and not dwc:
                    y = F.linear(xin.reshape(-1, C), w.reshape(w.shape[0], C), b).reshape(N, H, W, w.shape[0])
                else:
                    y = F.conv2d(xt, w, b, stride=(sh, sw), padding=pad, dilation=(dh, dw), groups=C if dwc else 1)
                    y = y.permute(0, 2, 3, 1)
                    if not y.is_contiguous():
                        y = y.contiguous()
                y = _act(y, o.get("act", 0))
            elif n == "FULLY_CONNECTED":
                w = a[1]
                xin = a[0]
                lead = tuple(xin.shape[:-1]) if o.get("keep_num_dims") else (-1,)
                y = F.linear(xin.reshape(-1, w.shape[1]), w, a[2].reshape(-1) if len(a) > 2 and a[2] is not None and a[2].numel() == w.shape[0] else None)
                y = _act(y, o.get("act", 0)).reshape(*lead, w.shape[0])
            elif n in ("ADD", "SUB", "MUL", "DIV"):
                f = {"ADD": torch.add, "SUB": torch.sub, "MUL": torch.mul, "DIV": torch.div}[n]
                y = _act(f(a[0], a[1]), o.get("act", 0))
            elif n == "LOGISTIC":
                y = torch.sigmoid(a[0])
            elif n in ("MEAN", "REDUCE_MAX", "REDUCE_MIN", "SUM"):
                axes = tuple(int(v) % a[0].ndim for v in np.atleast_1d(a[1].numpy()))
                keep = bool(o.get("keep_dims"))
                if n == "MEAN":
                    y = a[0].mean(dim=axes, keepdim=keep)
                elif n == "SUM":
                    y = a[0].sum(dim=axes, keepdim=keep)
                else:
                    y = a[0].amax(dim=axes, keepdim=keep) if n == "REDUCE_MAX" else a[0].amin(dim=axes, keepdim=keep)
            elif n == "RESHAPE":
                shape = [int(v) for v in (o.get("new_shape") or a[1].numpy())]
                tot = int(np.prod(shape)) if -1 not in shape else None
                if tot is None or tot == a[0].numel():
                    y = a[0].reshape(shape)
                elif shape and shape[0] == 1 and a[0].numel() == B * tot:
                    y = a[0].reshape([B] + shape[1:])
                else:
                    y = a[0].reshape([B * shape[0]] + shape[1:])
            elif n == "GATHER":
                idx = a[1].to(torch.int64)
                ax = o.get("axis", 0)
                y = a[0].index_select(ax, idx.reshape(-1)).reshape(list(a[0].shape[:ax]) + list(idx.shape) + list(a[0].shape[ax + 1:]))
            elif n == "RFFT2D":
                fl = [int(v) for v in a[1].numpy()]
                # TFLite rfft2d.cc: Ooura fft2d on doubles, then narrowed to complex64
                y = torch.fft.rfft2(a[0].to(torch.float64), s=fl, dim=(-2, -1)).to(torch.complex64)
            elif n == "CAST":
                src = a[0]
                if src.is_complex():
                    src = src.real                    # cast.cc: complex64 -> float keeps std::real()
                y = src.to({0: torch.float32, 2: torch.int32, 4: torch.int64}[o.get("out_type", 0)])
            elif n == "COMPLEX_ABS":
                y = a[0].abs().to(torch.float32)
            elif n == "POW":
                y = torch.pow(a[0], a[1])
            elif n == "REVERSE_V2":
                y = torch.flip(a[0], dims=tuple(int(v) for v in np.atleast_1d(a[1].numpy())))
            elif n == "TRANSPOSE":
                y = a[0].permute([int(v) for v in a[1].numpy()]).contiguous()
            elif n == "CONCATENATION":
                parts = [v.expand(B, *v.shape[1:]) if (v.ndim and v.shape[0] == 1 and B > 1 and any(u.shape[0] == B for u in a)) else v for v in a]
                y = _act(torch.cat(parts, dim=o.get("axis", 0)), o.get("act", 0))
            elif n == "PAD":
                p = a[1].numpy().reshape(-1, 2)
                flat = []
                for lo, hi in p[::-1]:
                    flat += [int(lo), int(hi)]
                y = F.pad(a[0], flat)
            elif n in ("MAXIMUM", "MINIMUM"):
                y = (torch.maximum if n == "MAXIMUM" else torch.minimum)(a[0], a[1])
            elif n == "LOG":
                y = torch.log(a[0])
            elif n == "RELU":
                y = F.relu(a[0])
            elif n == "RELU6":
                y = a[0].clamp(0, 6)
            elif n == "SOFTMAX":
                y = torch.softmax(a[0] * (o.get("beta") or 1.0), dim=-1)
            else:
                raise ValueError(f"torch-CPU baseline: unsupported op {n}")
            vals[op.outputs[0]] = y
        return [vals[t].to(torch.float32).reshape(B, -1).numpy() for t in m.outputs]

    def predict(self, samples):
        return self.invoke(samples)[0]
