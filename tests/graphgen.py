"""Seeded random TFLite graphs over the op vocabulary a TF -> TFLite export can contain (test infrastructure).

The reference call being replaced accepts any float graph (internal/inference/tflite/classifier.go:38-92), so the planner
is exercised with graphs it did not author patterns for: every graph here goes through the C++ reader, the graph passes
(PAD folding, unfolded batch norm, ...) and the generic kernel tier, and is compared with the numpy oracle.
"""
import numpy as np

import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import tflite_schema as S
from birdnet_go_amd.tflite_build import GraphBuilder

i32 = lambda v: np.asarray(v, np.int32)
f32 = lambda v: np.asarray(v, np.float32)


def _same(n, s):
    return (n + s - 1) // s


def _valid(n, k, s, d=1):
    return (n - ((k - 1) * d + 1)) // s + 1


class Gen:
    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.g = GraphBuilder(description=f"random graph seed={seed}")
        self.ops_used = []

    def pick(self, seq):
        return seq[int(self.rng.integers(0, len(seq)))]

    def w(self, *shape, fan=None):
        fan = fan or int(np.prod(shape[1:])) or 1
        return (self.rng.standard_normal(shape) / np.sqrt(fan)).astype(np.float32)

    # ---- ops on a rank-4 [1,H,W,C] tensor; each returns (tensor, shape)
    def conv(self, t, sh, force=None):
        g, rng = self.g, self.rng
        _, H, W, C = sh
        k = force or self.pick([1, 1, 3, 3, 5])
        s = self.pick([1, 1, 2])
        d = self.pick([1, 1, 1, 2]) if s == 1 and k > 1 else 1
        co = self.pick([4, 6, 8, 12, 16])
        pad = self.pick([S.PAD_SAME, S.PAD_VALID])
        if pad == S.PAD_VALID and (H < (k - 1) * d + 1 or W < (k - 1) * d + 1):
            pad = S.PAD_SAME
        Ho, Wo = (_same(H, s), _same(W, s)) if pad == S.PAD_SAME else (_valid(H, k, s, d), _valid(W, k, s, d))
        act = self.pick([S.ACT_NONE, S.ACT_NONE, S.ACT_RELU, S.ACT_RELU6, S.ACT_RELU_N1_TO_1, S.ACT_TANH])
        ins = [t, g.const(self.w(co, k, k, C))]
        ins.append(g.const((rng.standard_normal(co) * 0.1).astype(np.float32)) if rng.random() < 0.8 else -1)
        y = g.op("CONV_2D", ins, [1, Ho, Wo, co],
                 dict(padding=pad, stride_w=s, stride_h=s, fused_activation_function=act, dilation_w_factor=d, dilation_h_factor=d))
        self.ops_used.append(f"CONV_2D k{k} s{s} d{d} act{act}")
        return y, [1, Ho, Wo, co]

    def pad_conv(self, t, sh):
        """Keras ZeroPadding2D + VALID stride-2 conv (EfficientNet's down-sampling blocks)."""
        g = self.g
        _, H, W, C = sh
        k = self.pick([3, 5])
        pt, pb = (k - 1) // 2 - (1 if H % 2 == 0 else 0), (k - 1) // 2
        pl, pr = (k - 1) // 2 - (1 if W % 2 == 0 else 0), (k - 1) // 2
        Hp, Wp = H + pt + pb, W + pl + pr
        if Hp < k or Wp < k:
            return self.conv(t, sh)
        p = g.op("PAD", [t, g.const(i32([[0, 0], [pt, pb], [pl, pr], [0, 0]]))], [1, Hp, Wp, C], {})
        depthwise = self.rng.random() < 0.5
        Ho, Wo = _valid(Hp, k, 2), _valid(Wp, k, 2)
        if depthwise:
            y = g.op("DEPTHWISE_CONV_2D", [p, g.const(self.w(1, k, k, C, fan=k * k)), g.const(f32(self.rng.standard_normal(C) * 0.1))],
                     [1, Ho, Wo, C], dict(padding=S.PAD_VALID, stride_w=2, stride_h=2, depth_multiplier=1,
                                          fused_activation_function=S.ACT_NONE, dilation_w_factor=1, dilation_h_factor=1))
            self.ops_used.append(f"PAD+DEPTHWISE k{k}")
            return y, [1, Ho, Wo, C]
        co = self.pick([4, 8, 16])
        y = g.op("CONV_2D", [p, g.const(self.w(co, k, k, C)), g.const(f32(self.rng.standard_normal(co) * 0.1))], [1, Ho, Wo, co],
                 dict(padding=S.PAD_VALID, stride_w=2, stride_h=2, fused_activation_function=S.ACT_NONE,
                      dilation_w_factor=1, dilation_h_factor=1))
        self.ops_used.append(f"PAD+CONV_2D k{k}")
        return y, [1, Ho, Wo, co]

    def dwconv(self, t, sh):
        g = self.g
        _, H, W, C = sh
        k = self.pick([3, 5])
        s = self.pick([1, 2])
        mult = self.pick([1, 1, 2])
        d = self.pick([1, 1, 2]) if s == 1 else 1
        Ho, Wo = _same(H, s), _same(W, s)
        y = g.op("DEPTHWISE_CONV_2D", [t, g.const(self.w(1, k, k, C * mult, fan=k * k)), g.const(f32(self.rng.standard_normal(C * mult) * 0.1))],
                 [1, Ho, Wo, C * mult], dict(padding=S.PAD_SAME, stride_w=s, stride_h=s, depth_multiplier=mult,
                                             fused_activation_function=self.pick([S.ACT_NONE, S.ACT_RELU6]),
                                             dilation_w_factor=d, dilation_h_factor=d))
        self.ops_used.append(f"DEPTHWISE k{k} s{s} m{mult} d{d}")
        return y, [1, Ho, Wo, C * mult]

    def conv_bn(self, t, sh):
        """conv (no activation) followed by an unfolded batch norm: per-channel MUL then ADD, then RELU."""
        g = self.g
        y, ysh = self.conv(t, sh, force=self.pick([1, 3]))
        self.g.ops[-1]["options"]["fused_activation_function"] = S.ACT_NONE
        co = ysh[3]
        y = g.op("MUL", [y, g.const(f32(self.rng.uniform(0.5, 1.5, co)))], ysh, {})
        y = g.op("ADD", [y, g.const(f32(self.rng.standard_normal(co) * 0.2))], ysh, dict(fused_activation_function=S.ACT_RELU))
        self.ops_used.append("BN(MUL,ADD)")
        return y, ysh

    def pool(self, t, sh):
        _, H, W, C = sh
        k = self.pick([2, 3])
        s = self.pick([1, 2])
        name = self.pick(["AVERAGE_POOL_2D", "MAX_POOL_2D"])
        pad = self.pick([S.PAD_SAME, S.PAD_VALID]) if H >= k and W >= k else S.PAD_SAME
        Ho, Wo = (_same(H, s), _same(W, s)) if pad == S.PAD_SAME else (_valid(H, k, s), _valid(W, k, s))
        y = self.g.op(name, [t], [1, Ho, Wo, C], dict(padding=pad, stride_w=s, stride_h=s, filter_width=k, filter_height=k,
                                                      fused_activation_function=S.ACT_NONE))
        self.ops_used.append(f"{name} k{k} s{s}")
        return y, [1, Ho, Wo, C]

    def unary(self, t, sh):
        g = self.g
        name = self.pick(["TANH", "ABS", "NEG", "SQUARE", "LEAKY_RELU", "ELU", "SIN", "COS", "FLOOR", "CEIL", "ROUND",
                          "RELU_N1_TO_1", "GELU", "LOGISTIC", "RELU", "RELU6", "HARD_SWISH", "EXP", "LOG", "SQRT", "RSQRT"])
        if name in ("LOG", "SQRT", "RSQRT"):                     # keep the argument away from the singularities
            t = g.op("ABS", [t], sh)
            t = g.op("ADD", [t, g.const(f32(0.5))], sh, {})
        if name == "EXP":
            t = g.op("TANH", [t], sh)
        opts = dict(alpha=0.2) if name == "LEAKY_RELU" else (dict(approximate=int(self.rng.random() < 0.5)) if name == "GELU" else {})
        y = g.op(name, [t], sh, opts)
        self.ops_used.append(name)
        return y, sh

    def binary(self, t, sh, other=None):
        g = self.g
        name = self.pick(["ADD", "SUB", "MUL", "DIV", "MAXIMUM", "MINIMUM", "SQUARED_DIFFERENCE", "POW"])
        kind = self.pick(["scalar", "channel", "full", "hw1", "tensor" if other is not None else "channel"])
        if kind == "tensor":
            b = other
        else:
            bs = {"scalar": [], "channel": [sh[-1]], "full": sh, "hw1": sh[:-1] + [1]}[kind]
            v = self.rng.standard_normal(bs).astype(np.float32)
            if name in ("DIV", "POW"):
                v = (np.abs(v) + 0.5).astype(np.float32)
            b = g.const(v)
        a = t
        if name == "POW":
            a = g.op("ABS", [a], sh)
            a = g.op("ADD", [a, g.const(f32(0.5))], sh, {})
        elif name == "DIV" and kind == "tensor":
            b = g.op("ABS", [b], sh)
            b = g.op("ADD", [b, g.const(f32(1.0))], sh, {})
        ins = [a, b] if self.rng.random() < 0.7 or name in ("POW", "DIV") else [b, a]
        opts = dict(fused_activation_function=self.pick([S.ACT_NONE, S.ACT_RELU])) if name in ("ADD", "SUB", "MUL", "DIV") else {}
        y = g.op(name, ins, sh, opts)
        self.ops_used.append(f"{name}[{kind}]")
        return y, sh

    def concat(self, t, sh):
        g = self.g
        ax = self.pick([1, 2, 3, -1])
        other, osh = self.unary(t, sh)
        parts = [t, other]
        if self.rng.random() < 0.4:
            csh = list(sh)
            csh[ax] = 2
            parts.append(g.const(self.rng.standard_normal(csh).astype(np.float32)))
        out = list(sh)
        out[ax] = sum(g.shape(p)[ax] for p in parts)
        y = g.op("CONCATENATION", parts, out, dict(axis=ax, fused_activation_function=S.ACT_NONE))
        self.ops_used.append(f"CONCATENATION ax{ax} n{len(parts)}")
        return y, out

    def strided_slice(self, t, sh):
        g, rng = self.g, self.rng
        begin, end, stride, out = [0], [1], [1], [1]
        for d in sh[1:]:
            st = self.pick([1, 1, 2, -1]) if d > 2 else 1
            if st > 0:
                b = int(rng.integers(0, max(d - 1, 1)))
                e = int(rng.integers(b + 1, d + 1))
                n = (e - b + st - 1) // st
            else:
                b = int(rng.integers(1, d))
                e = int(rng.integers(-1, b))
                n = b - e
                e = e if e >= 0 else -d - 1               # "before the first element"
            begin.append(b); end.append(e); stride.append(st); out.append(n)
        use_slice = all(s == 1 for s in stride) and rng.random() < 0.5
        if use_slice:
            size = [-1] + [e - b for b, e in zip(begin[1:], end[1:])]
            y = g.op("SLICE", [t, g.const(i32(begin)), g.const(i32(size))], out, {})
            self.ops_used.append("SLICE")
        else:
            y = g.op("STRIDED_SLICE", [t, g.const(i32(begin)), g.const(i32(end)), g.const(i32(stride))], out,
                     dict(begin_mask=1, end_mask=1, ellipsis_mask=0, new_axis_mask=0, shrink_axis_mask=0))
            self.ops_used.append(f"STRIDED_SLICE {stride[1:]}")
        return y, out

    def transpose(self, t, sh):
        perm = [0] + [int(v) + 1 for v in self.rng.permutation(3)]
        out = [sh[p] for p in perm]
        y = self.g.op("TRANSPOSE", [t, self.g.const(i32(perm))], out, {})
        self.ops_used.append(f"TRANSPOSE {perm}")
        return y, out

    def reverse(self, t, sh):
        ax = self.pick([[1], [2], [3], [1, 2], [-1]])
        y = self.g.op("REVERSE_V2", [t, self.g.const(i32(ax))], sh, {})
        self.ops_used.append(f"REVERSE_V2 {ax}")
        return y, sh

    def pad(self, t, sh):
        pv = [[0, 0]] + [[int(self.rng.integers(0, 3)), int(self.rng.integers(0, 3))] for _ in range(3)]
        out = [d + a + b for d, (a, b) in zip(sh, pv)]
        if self.rng.random() < 0.5:
            y = self.g.op("PAD", [t, self.g.const(i32(pv))], out, {})
        else:
            y = self.g.op("PADV2", [t, self.g.const(i32(pv)), self.g.const(f32(-1.5))], out, {})
        self.ops_used.append("PAD")
        return y, out

    def reduce(self, t, sh):
        name = self.pick(["MEAN", "SUM", "REDUCE_MAX", "REDUCE_MIN", "REDUCE_PROD"])
        axes = self.pick([[1], [2], [3], [1, 3], [2, 3], [-1], [1, 2]])
        if name == "REDUCE_PROD":
            t, _ = self._bounded(t, sh)
        norm = sorted(a % 4 for a in axes)
        out = [1 if i in norm else d for i, d in enumerate(sh)]
        y = self.g.op(name, [t, self.g.const(i32(axes))], out, dict(keep_dims=1))
        self.ops_used.append(f"{name} {axes}")
        return y, out

    def _bounded(self, t, sh):
        t = self.g.op("TANH", [t], sh)
        return self.g.op("ADD", [t, self.g.const(f32(1.0))], sh, {}), sh

    def softmax(self, t, sh):
        y = self.g.op("SOFTMAX", [t], sh, dict(beta=float(self.pick([1.0, 0.5]))))
        self.ops_used.append("SOFTMAX")
        return y, sh

    def split(self, t, sh):
        g = self.g
        cands = [(ax, n) for ax in (1, 2, 3) for n in (2, 3) if sh[ax] % n == 0 and sh[ax] >= n]
        if not cands:
            return self.unary(t, sh)
        ax, n = self.pick(cands)
        part = list(sh)
        part[ax] = sh[ax] // n
        outs = g.op("SPLIT", [g.const(i32(ax)), t], [part] * n, dict(num_splits=n))
        a = g.op("MUL", [outs[0], g.const(f32(2.0))], part, {})
        y = g.op("ADD", [a, outs[-1]], part, {})
        self.ops_used.append(f"SPLIT ax{ax} n{n}")
        return y, part

    # ---- whole graphs
    def build(self, n_ops=None):
        g, rng = self.g, self.rng
        H, W, C = int(rng.integers(5, 13)), int(rng.integers(5, 13)), self.pick([2, 3, 4, 8])
        x = g.tensor([1, H, W, C], name="INPUT")
        t, sh = x, [1, H, W, C]
        saved = None
        steps = [self.conv, self.conv, self.pad_conv, self.dwconv, self.conv_bn, self.pool, self.unary, self.unary, self.binary,
                 self.binary, self.concat, self.strided_slice, self.transpose, self.reverse, self.pad, self.reduce, self.softmax,
                 self.split]
        for _ in range(n_ops or int(rng.integers(3, 8))):
            if min(sh[1:3]) < 2 and rng.random() < 0.7:
                break
            f = self.pick(steps)
            if f == self.binary and saved is not None and saved[1] == sh and rng.random() < 0.5:
                t, sh = self.binary(t, sh, other=saved[0])
            else:
                t, sh = f(t, sh)
            if rng.random() < 0.3:
                saved = (t, list(sh))
            if int(np.prod(sh)) > 20000:                              # keep the oracle fast
                t, sh = self.pool(t, sh)
        # head: global mean -> dense (leading dims through FULLY_CONNECTED when the tensor is still spatial)
        n_cls = self.pick([3, 5, 10])
        if rng.random() < 0.5:
            m = g.op("MEAN", [t, g.const(i32([1, 2]))], [1, sh[3]], dict(keep_dims=0))
            y = g.op("FULLY_CONNECTED", [m, g.const(self.w(n_cls, sh[3])), g.const(f32(rng.standard_normal(n_cls) * 0.1))], [1, n_cls],
                     dict(fused_activation_function=S.ACT_NONE))
        else:
            rows = sh[1] * sh[2]
            r = g.op("RESHAPE", [t, g.const(i32([1, rows, sh[3]]))], [1, rows, sh[3]], dict(new_shape=[1, rows, sh[3]]))
            y = g.op("FULLY_CONNECTED", [r, g.const(self.w(n_cls, sh[3])), -1], [1, rows, n_cls],
                     dict(fused_activation_function=self.pick([S.ACT_NONE, S.ACT_TANH]), keep_num_dims=1))
            y = g.op("RESHAPE", [y, g.const(i32([1, rows * n_cls]))], [1, rows * n_cls], dict(new_shape=[1, rows * n_cls]))
        self.io = (x, y)
        return g.finish([x], [y]), (H, W, C)


def random_graph(seed):
    """-> (tflite bytes, per-clip input shape (H, W, C), list of op descriptions)"""
    gen = Gen(seed)
    blob, shape = gen.build()
    return blob, shape, gen.ops_used


def random_input(seed, shape, batch):
    return np.random.default_rng(10_000 + seed).standard_normal((batch,) + tuple(shape)).astype(np.float32)


def mutated_graph(seed, mut_seed):
    """A random graph whose operator records were corrupted before serialisation (well-framed flatbuffer, malformed graph):
    operand indices dropped / set to -1 / redirected, constants re-typed, outputs removed, weight shapes changed."""
    gen = Gen(seed)
    gen.build()
    g, rng = gen.g, np.random.default_rng(mut_seed)
    nt = len(g.tensors)
    for _ in range(int(rng.integers(1, 4))):
        o = g.ops[int(rng.integers(0, len(g.ops)))]
        kind = int(rng.integers(0, 8))
        if kind == 0 and o["inputs"]:
            o["inputs"][int(rng.integers(0, len(o["inputs"])))] = -1
        elif kind == 1 and o["inputs"]:
            o["inputs"][int(rng.integers(0, len(o["inputs"])))] = int(rng.integers(0, nt))
        elif kind == 2 and o["inputs"]:
            o["inputs"].pop()
        elif kind == 3:
            o["outputs"] = []
        elif kind == 4:
            o["inputs"] = o["inputs"] + [int(rng.integers(0, nt))]
        elif kind == 5:                                   # re-type a constant (float weights become int32 / float16 bits)
            consts = [t for t in g.tensors if t["buffer"]]
            if consts:
                t = consts[int(rng.integers(0, len(consts)))]
                t["type"] = int(rng.choice([S.INT32, S.FLOAT16, S.INT8, S.FLOAT32]))
        elif kind == 6:                                   # change a declared shape without touching the data
            t = g.tensors[int(rng.integers(0, nt))]
            if t["shape"]:
                t["shape"][int(rng.integers(0, len(t["shape"])))] = int(rng.integers(0, 9))
        else:
            o["outputs"] = [int(rng.integers(0, nt))]
    return g.finish([gen.io[0]], [gen.io[1]])
