"""Seeded random convolutional ONNX graphs (NCHW vocabulary) for the reader's layout propagation: Conv with assorted kernels /
strides / pads / dilations / depthwise groups, pools, BatchNormalization, per-channel constants, residual adds, channel
Concat, Transpose round trips, Pad, Flatten / Reshape of real images (forces an NCHW materialisation), reductions with and
without keepdims, a dense head.  Every graph is executed by oracle/onnx_interp.py in NCHW numpy and by the engine."""
import numpy as np

from birdnet_go_amd import onnx_build as ob


def random_cnn(seed):
    rng = np.random.default_rng(seed)
    b = ob.OnnxBuilder(name=f"rand{seed}")
    C, H, W = int(rng.choice([1, 2, 3, 4])), int(rng.integers(9, 24)), int(rng.integers(9, 28))
    nhwc_in = bool(rng.integers(0, 2))
    if nhwc_in:
        x = b.input("x", ["N", H, W, C])
        t = b.node("Transpose", [x], perm=[0, 3, 1, 2])
        in_shape = (H, W, C)
    else:
        t = b.input("x", ["N", C, H, W])
        in_shape = (C, H, W)
    f32 = lambda a: np.asarray(a, np.float32)

    def conv(t, C, H, W):
        k = int(rng.choice([1, 3, 3, 5]))
        s = int(rng.choice([1, 1, 2]))
        d = int(rng.choice([1, 1, 1, 2])) if k > 1 else 1
        dw = bool(rng.integers(0, 3) == 0) and C > 1
        M = C * int(rng.choice([1, 2])) if dw else int(rng.choice([4, 6, 8, 12, 16]))
        w = f32(rng.standard_normal((M, 1 if dw else C, k, k)) / np.sqrt(k * k * (1 if dw else C)))
        ins = [t, b.init(w)]
        if rng.integers(0, 2):
            ins.append(b.init(f32(rng.standard_normal(M) * 0.1)))
        e = d * (k - 1) + 1
        mode = int(rng.integers(0, 4))
        at = dict(kernel_shape=[k, k], strides=[s, s], dilations=[d, d], group=C if dw else 1)
        if mode == 0:
            at["auto_pad"] = "SAME_UPPER"
            Ho, Wo = -(-H // s), -(-W // s)
        elif mode == 1:
            at["auto_pad"] = "SAME_LOWER"
            Ho, Wo = -(-H // s), -(-W // s)
        elif mode == 2 and H >= e and W >= e:
            at["auto_pad"] = "VALID"
            Ho, Wo = (H - e) // s + 1, (W - e) // s + 1
        else:
            p = [int(rng.integers(0, e)) for _ in range(4)]
            if H + p[0] + p[2] < e or W + p[1] + p[3] < e:
                p = [e // 2] * 4
            at["pads"] = p
            Ho, Wo = (H + p[0] + p[2] - e) // s + 1, (W + p[1] + p[3] - e) // s + 1
        return b.node("Conv", ins, **at), M, Ho, Wo

    n_ops = int(rng.integers(4, 10))
    saved = None
    for _ in range(n_ops):
        kind = int(rng.integers(0, 12))
        if kind <= 3 or H < 3 or W < 3:
            if min(H, W) < 2:
                break
            t, C, H, W = conv(t, C, H, W)
        elif kind == 4:
            t = b.node(str(rng.choice(["Relu", "Sigmoid", "Tanh", "HardSwish"])), [t])
        elif kind == 5:                                   # swish
            t = b.node("Mul", [t, b.node("Sigmoid", [t])])
        elif kind == 6:                                   # per-channel affine, constants in the shapes exporters use
            shp = [(1, C, 1, 1), (C, 1, 1)][int(rng.integers(0, 2))]
            t = b.node("Mul", [t, b.init(f32(rng.uniform(0.5, 1.5, shp)))])
            t = b.node("Add", [b.init(f32(rng.standard_normal(shp) * 0.1)), t])
        elif kind == 7:
            p = [rng.uniform(0.5, 1.5, C), rng.standard_normal(C) * 0.1, rng.standard_normal(C) * 0.1, rng.uniform(0.5, 1.5, C)]
            t = b.node("BatchNormalization", [t] + [b.init(f32(v)) for v in p], epsilon=1e-3)
        elif kind == 8 and H >= 4 and W >= 4:
            k = int(rng.choice([2, 3]))
            if rng.integers(0, 2):
                t = b.node(str(rng.choice(["MaxPool", "AveragePool"])), [t], kernel_shape=[k, k], strides=[2, 2])
                H, W = (H - k) // 2 + 1, (W - k) // 2 + 1
            else:
                t = b.node(str(rng.choice(["MaxPool", "AveragePool"])), [t], kernel_shape=[k, k], strides=[2, 2], auto_pad="SAME_UPPER")
                H, W = -(-H // 2), -(-W // 2)
        elif kind == 9:                                   # squeeze-excite gate
            g = b.node("GlobalAveragePool", [t]) if rng.integers(0, 2) else b.node("ReduceMean", [t], axes=[2, 3], keepdims=1)
            r = max(1, C // 2)
            g = b.node("Relu", [b.node("Conv", [g, b.init(f32(rng.standard_normal((r, C, 1, 1)) / np.sqrt(C))), b.init(f32(rng.standard_normal(r) * 0.1))], kernel_shape=[1, 1])])
            g = b.node("Conv", [g, b.init(f32(rng.standard_normal((C, r, 1, 1)) / np.sqrt(r)))], kernel_shape=[1, 1])
            g = b.node("HardSigmoid", [g], alpha=0.2, beta=0.5) if rng.integers(0, 2) else b.node("Sigmoid", [g])
            t = b.node("Mul", [t, g])
        elif kind == 10:
            if saved is not None and saved[1:] == (C, H, W):
                t = b.node("Add", [t, saved[0]])          # residual
            else:
                u = b.node("Relu", [t])
                t = b.node("Concat", [t, u], axis=1)      # channel concat
                C *= 2
            saved = None
        elif kind == 11:
            pads = [0, 0, int(rng.integers(0, 3)), int(rng.integers(0, 3)), 0, 0, int(rng.integers(0, 3)), int(rng.integers(0, 3))]
            t = b.node("Pad", [t, b.init(np.asarray(pads, np.int64))], mode="constant")
            H, W = H + pads[2] + pads[6], W + pads[3] + pads[7]
        if saved is None and rng.integers(0, 3) == 0:
            saved = (t, C, H, W)
    # head: one of the ways a graph leaves the image domain
    tail = int(rng.integers(0, 5))
    if tail == 0:
        t = b.node("Flatten", [b.node("GlobalAveragePool", [t])], axis=1)
        feat = C
    elif tail == 1:
        t = b.node("ReduceMean", [t], axes=[2, 3], keepdims=0)
        feat = C
    elif tail == 2:                                       # flatten a real image: element order matters (NCHW)
        t = b.node("Flatten", [t], axis=1)
        feat = C * H * W
    elif tail == 3:
        t = b.node("Reshape", [t, b.init(np.asarray([0, -1], np.int64))])
        feat = C * H * W
    else:                                                 # NCHW -> NHWC -> flatten: the tf2onnx tail
        t = b.node("Flatten", [b.node("Transpose", [t], perm=[0, 2, 3, 1])], axis=1)
        feat = C * H * W
    n_cls = int(rng.integers(3, 12))
    wh = f32(rng.standard_normal((n_cls, feat)) / np.sqrt(feat))
    t = b.node("Gemm", [t, b.init(wh), b.init(f32(rng.standard_normal(n_cls) * 0.1))], transB=1)
    b.output(t, ["N", n_cls])
    return b.finish(), in_shape, n_cls


def random_image(seed, in_shape, n):
    return np.random.default_rng(1000 + seed).standard_normal((n,) + tuple(in_shape)).astype(np.float32)
