"""The synthetic audio models of synth_model.py written as ONNX files - the container the reference's ONNX backend loads
(`internal/inference/onnx/classifier.go:268-430`) - in the forms exporters give the in-graph audio front-end:

  dft="matmul"   tf2onnx style: tf.signal.frame as Reshape + Gather, window Mul, the real FFT as a MatMul with a constant DFT
                 basis [frame, bins] (cos; and -sin when the graph takes the magnitude).  trunc=True keeps only the DFT columns
                 the mel matrix uses: the "dfttrunc" files the reference ships for BirdNET v2.4 and for the BattyBirdNET
                 backbone (`internal/classifier/model_catalog.go:412-426,490-501`: "DFT-truncated ... about 2x faster").
  dft="conv1d"   torch style: framing + window + DFT as ONE strided Conv over the clip [N, 1, T] ("its mel front-end is a Conv1d",
                 `internal/classifier/birdnet_v3_onnx.go:44-48`): bins-major [N, K, F] tensors, the mel matrix multiplied from the left.
  dft="stft"     the opset-17 `STFT` operator (what torch.stft exports to): real / imaginary parts in a trailing axis of 2.
  dft="dft"      the opset-17 `DFT` operator behind the Gather framing (Perch v2's "in-graph DFT", model_catalog.go:273-311).

The CNN body is written NCHW (Conv, depthwise Conv with group = C, GlobalAveragePool, ReduceMean, Gemm), which is what tf2onnx
and torch.onnx both produce.  The graph is transcribed from the SAME recorded op list the TFLite writer serialises
(tflite_build.GraphBuilder), so the two containers carry identical weights.  Authoring tool for tests, examples and bench
legs - the real files are absent from the reference snapshot; it knows the op vocabulary of synth_model.py, nothing more.
"""
import numpy as np

from .onnx_build import OnnxBuilder


def _i64(v):
    return np.asarray(v, np.int64)


class _Tx:
    def __init__(self, g, dft, trunc):
        self.g, self.dft, self.trunc = g, dft, trunc
        self.b = OnnxBuilder(name="audio_" + dft, opset=17)
        self.val = {}             # tflite tensor id -> ONNX value name
        self.nchw = set()         # tflite rank-4 tensors whose ONNX value is [N, C, H, W]
        self.done = set()         # op indices emitted as part of a front-end branch
        self.prod = {}
        for oi, o in enumerate(g.ops):
            for t in o["outputs"]:
                self.prod[t] = oi
        self.cons = {}
        for oi, o in enumerate(g.ops):
            for t in o["inputs"]:
                if t >= 0:
                    self.cons.setdefault(t, []).append(oi)

    # ---- helpers over the recorded graph
    def const(self, t):
        buf = self.g.tensors[t]["buffer"]
        return None if buf == 0 else self.g.buffers[buf]

    def shape(self, t):
        return self.g.tensors[t]["shape"]

    def operand(self, t):
        """ONNX name of an activation, or an initializer for a constant."""
        if t in self.val:
            return self.val[t]
        c = self.const(t)
        assert c is not None, f"tensor {t} has no value"
        a = c.astype(np.float32) if c.dtype.kind == "f" else c.astype(np.int64)
        self.val[t] = self.b.init(a, self.g.tensors[t]["name"].replace("/", "_") + f"_{t}")
        return self.val[t]

    def up(self, t, *names):
        o = self.g.ops[self.prod[t]]
        assert o["op"] in names, (o["op"], names)
        return self.prod[t], o

    def down(self, t):
        cs = self.cons.get(t, [])
        assert len(cs) == 1, f"tensor {t} has {len(cs)} consumers"
        return cs[0], self.g.ops[cs[0]]

    # ---- one mel branch around RFFT2D op `ri`
    def frontend(self, ri):
        g, b = self.g, self.b
        R = g.ops[ri]
        Lfft = int(self.const(R["inputs"][1])[1])
        ops = [ri]
        i, o = self.up(R["inputs"][0], "RESHAPE"); ops.append(i)
        t = o["inputs"][0]
        if g.ops[self.prod[t]]["op"] == "PAD":
            i, o = self.up(t, "PAD"); ops.append(i); t = o["inputs"][0]
        i, o = self.up(t, "MUL"); ops.append(i)
        window = self.const(o["inputs"][1]).astype(np.float32)
        i, o = self.up(o["inputs"][0], "RESHAPE"); ops.append(i)
        i, o = self.up(o["inputs"][0], "GATHER"); ops.append(i)
        sel = self.const(o["inputs"][1])
        i, o = self.up(o["inputs"][0], "RESHAPE"); ops.append(i)
        src = o["inputs"][0]                                  # [1, T] (normalised / padded clip)
        sub = self.shape(o["outputs"][0])[2]
        F, L = sel.shape[0], sel.shape[1] * sub
        hop = int(sel[1, 0] - sel[0, 0]) * sub if F > 1 else L
        T = self.shape(src)[1]
        # downstream
        i, o = self.down(R["outputs"][0]); ops.append(i)      # RESHAPE [1,F,nb]
        i, o = self.down(o["outputs"][0]); ops.append(i)
        magnitude = o["op"] == "COMPLEX_ABS"
        assert magnitude or o["op"] == "CAST"
        i, o = self.down(o["outputs"][0]); ops.append(i)      # RESHAPE [F,nb]
        i, o = self.down(o["outputs"][0]); ops.append(i)
        assert o["op"] == "FULLY_CONNECTED", "only the FULLY_CONNECTED mel form is transcribed"
        melT = self.const(o["inputs"][1]).astype(np.float32)  # [n_mels, nb]
        n_mels, nb = melT.shape
        i, o = self.down(o["outputs"][0]); ops.append(i)      # RESHAPE [1,F,n_mels]
        t = o["outputs"][0]
        chain = []                                            # elementwise compression ops, in graph order
        while True:
            i, o = self.down(t)
            if o["op"] in ("POW", "MAXIMUM", "LOG", "MUL"):
                chain.append(o); ops.append(i); t = o["outputs"][0]
            else:
                break
        reverse = False
        if o["op"] == "REVERSE_V2":
            reverse = True; ops.append(i); t = o["outputs"][0]; i, o = self.down(t)
        time_major = True
        if o["op"] == "TRANSPOSE":
            time_major = False; ops.append(i); t = o["outputs"][0]; i, o = self.down(t)
        assert o["op"] == "RESHAPE"; ops.append(i)
        out_t = o["outputs"][0]                               # [1, n_mels, F, 1] or [1, F, n_mels, 1]
        self.done.update(ops)

        x = self.val[src]
        bins = np.arange(nb)
        if self.trunc and self.dft in ("matmul", "conv1d"):
            bins = np.nonzero(np.abs(melT).sum(0) > 0)[0]
        n = np.arange(Lfft, dtype=np.float64)
        ang = 2.0 * np.pi * np.outer(n, bins.astype(np.float64)) / Lfft          # [Lfft, K]
        wpad = np.zeros(Lfft, np.float32); wpad[:L] = window
        mel_k = np.ascontiguousarray(melT[:, bins].T)         # [K, n_mels]
        K = len(bins)

        def magn(re, im):
            return b.node("Sqrt", [b.node("Add", [b.node("Mul", [re, re]), b.node("Mul", [im, im])])])

        bins_major = False
        if self.dft in ("matmul", "dft"):
            r1 = b.node("Reshape", [x, b.init(_i64([1, T // sub, sub]))])
            ga = b.node("Gather", [r1, b.init(sel.astype(np.int64))], axis=1)
            fr = b.node("Reshape", [ga, b.init(_i64([1, F, L]))])
            wn = b.node("Mul", [fr, b.init(window)])
            if Lfft != L:
                wn = b.node("Pad", [wn, b.init(_i64([0, 0, 0, 0, 0, Lfft - L]))])
            if self.dft == "matmul":
                re = b.node("MatMul", [wn, b.init(np.cos(ang).astype(np.float32))])                 # [1, F, K]
                spec = magn(re, b.node("MatMul", [wn, b.init((-np.sin(ang)).astype(np.float32))])) if magnitude else re
            else:
                d = b.node("DFT", [b.node("Unsqueeze", [wn, b.init(_i64([3]))])], axis=2, onesided=1)   # [1, F, nb, 2]
                re = b.node("Squeeze", [b.node("Slice", [d, b.init(_i64([0])), b.init(_i64([1])), b.init(_i64([3]))]), b.init(_i64([3]))])
                if magnitude:
                    im = b.node("Squeeze", [b.node("Slice", [d, b.init(_i64([1])), b.init(_i64([2])), b.init(_i64([3]))]), b.init(_i64([3]))])
                    spec = magn(re, im)
                else:
                    spec = re
            mel = b.node("MatMul", [spec, b.init(mel_k)])                                            # [1, F, n_mels]
        elif self.dft == "conv1d":
            # filters [rows, 1, Lfft -> L]: window * cos rows, then window * (-sin) rows (taps beyond the frame are zero: dropped)
            basis = [np.cos(ang).T[:, :L] * window[None, :]]
            if magnitude:
                basis.append((-np.sin(ang)).T[:, :L] * window[None, :])
            w = np.concatenate(basis, 0)[:, None, :].astype(np.float32)
            y = b.node("Conv", [b.node("Unsqueeze", [x, b.init(_i64([1]))]), b.init(w)], kernel_shape=[L], strides=[hop])   # [1, rows, F]
            if magnitude:
                re = b.node("Slice", [y, b.init(_i64([0])), b.init(_i64([K])), b.init(_i64([1]))])
                im = b.node("Slice", [y, b.init(_i64([K])), b.init(_i64([2 * K])), b.init(_i64([1]))])
                spec = magn(re, im)
            else:
                spec = y
            mel = b.node("MatMul", [b.init(np.ascontiguousarray(mel_k.T)), spec])                    # [n_mels, K] x [1, K, F] -> [1, n_mels, F]
            bins_major = True
        elif self.dft == "stft":
            sig = b.node("Unsqueeze", [x, b.init(_i64([2]))])                                        # [1, T, 1]
            d = b.node("STFT", [sig, b.init(_i64(hop)), b.init(wpad), b.init(_i64(Lfft))], onesided=1)   # [1, F', nb, 2]
            if (T - Lfft) // hop + 1 != F:                     # frames shorter than the transform: STFT steps over Lfft-long frames
                raise ValueError("dft='stft' needs frame_length == fft_length")
            re = b.node("Squeeze", [b.node("Slice", [d, b.init(_i64([0])), b.init(_i64([1])), b.init(_i64([3]))]), b.init(_i64([3]))])
            if magnitude:
                im = b.node("Squeeze", [b.node("Slice", [d, b.init(_i64([1])), b.init(_i64([2])), b.init(_i64([3]))]), b.init(_i64([3]))])
                spec = magn(re, im)
            else:
                spec = re
            mel = b.node("MatMul", [spec, b.init(mel_k)])
        else:
            raise ValueError(self.dft)
        t = mel
        for o in chain:
            if o["op"] == "POW":
                t = b.node("Pow", [t, b.init(np.asarray(self.const(o["inputs"][1]), np.float32).reshape(()))])
            elif o["op"] == "MAXIMUM":
                t = b.node("Max", [t, b.init(np.asarray(self.const(o["inputs"][1]), np.float32).reshape(()))])
            elif o["op"] == "LOG":
                t = b.node("Log", [t])
            else:
                t = b.node("Mul", [t, b.init(np.asarray(self.const(o["inputs"][1]), np.float32).reshape(()))])
        mel_axis = 1 if bins_major else 2
        if reverse:                                            # ReverseV2 arrives as a Slice with step -1 over the whole axis
            t = b.node("Slice", [t, b.init(_i64([-1])), b.init(_i64([-(2 ** 31)])), b.init(_i64([mel_axis])), b.init(_i64([-1]))])
        want_mel_first = not time_major
        if want_mel_first != bins_major:
            t = b.node("Transpose", [t], perm=[0, 2, 1])
        self.val[out_t] = b.node("Unsqueeze", [t, b.init(_i64([1]))])    # [1, 1, H, W]
        self.nchw.add(out_t)

    # ---- everything else, op by op
    def emit(self, oi):
        g, b = self.g, self.b
        o = g.ops[oi]
        op, ins, out, opt = o["op"], o["inputs"], o["outputs"][0], o["options"]
        img = len(self.shape(out)) == 4
        if op in ("REDUCE_MIN", "REDUCE_MAX"):
            self.val[out] = b.node("ReduceMin" if op == "REDUCE_MIN" else "ReduceMax", [self.operand(ins[0])],
                                   axes=[int(v) for v in self.const(ins[1])], keepdims=int(opt.get("keep_dims", 0)))
        elif op in ("ADD", "SUB", "MUL", "DIV"):
            a, c = ins
            if any(t in self.nchw for t in ins):
                self.nchw.add(out)
                na, nc = (self._img_operand(t) for t in (a, c))
            else:
                na, nc = self.operand(a), self.operand(c)
            self.val[out] = b.node({"ADD": "Add", "SUB": "Sub", "MUL": "Mul", "DIV": "Div"}[op], [na, nc])
        elif op == "LOGISTIC":
            self.val[out] = b.node("Sigmoid", [self.operand(ins[0])])
            if ins[0] in self.nchw:
                self.nchw.add(out)
        elif op == "PAD":
            p = self.const(ins[1]).reshape(-1, 2)
            self.val[out] = b.node("Pad", [self.operand(ins[0]), b.init(_i64(list(p[:, 0]) + list(p[:, 1])))])
        elif op == "CONCATENATION":
            assert all(t in self.nchw for t in ins) and opt.get("axis") == 3
            self.val[out] = b.node("Concat", [self.val[t] for t in ins], axis=1)
            self.nchw.add(out)
        elif op in ("CONV_2D", "DEPTHWISE_CONV_2D"):
            w = self.const(ins[1]).astype(np.float32)
            dw = op == "DEPTHWISE_CONV_2D"
            wn = np.transpose(w, (3, 0, 1, 2)) if dw else np.transpose(w, (0, 3, 1, 2))
            args = [self.val[ins[0]], b.init(np.ascontiguousarray(wn), g.tensors[ins[1]]["name"].replace("/", "_"))]
            if len(ins) > 2 and ins[2] >= 0:
                args.append(b.init(self.const(ins[2]).astype(np.float32)))
            s = int(opt.get("stride_h", 1))
            self.val[out] = b.node("Conv", args, kernel_shape=[int(w.shape[1]), int(w.shape[2])], strides=[s, int(opt.get("stride_w", 1))],
                                   group=int(w.shape[3]) if dw else 1, auto_pad="SAME_UPPER")
            self.nchw.add(out)
        elif op == "MEAN":
            axes = [int(v) for v in self.const(ins[1])]
            assert axes == [1, 2] and ins[0] in self.nchw
            if opt.get("keep_dims"):
                self.val[out] = b.node("GlobalAveragePool", [self.val[ins[0]]])
                self.nchw.add(out)
            else:
                self.val[out] = b.node("ReduceMean", [self.val[ins[0]]], axes=[2, 3], keepdims=0)
        elif op == "FULLY_CONNECTED":
            w = self.const(ins[1]).astype(np.float32)
            args = [self.operand(ins[0]), b.init(w, g.tensors[ins[1]]["name"].replace("/", "_"))]
            if len(ins) > 2 and ins[2] >= 0:
                args.append(b.init(self.const(ins[2]).astype(np.float32)))
            self.val[out] = b.node("Gemm", args, alpha=1.0, beta=1.0, transB=1)
        elif op == "RESHAPE":
            src = ins[0]
            if src in self.nchw and len(self.shape(out)) == 3 and self.shape(src)[3] == 1:       # [1,H,W,1] image -> [1,H,W]
                self.val[out] = b.node("Squeeze", [self.val[src], b.init(_i64([1]))])
            else:
                raise ValueError(f"RESHAPE {self.shape(src)} -> {self.shape(out)} outside the front-end is not transcribed")
        else:
            raise ValueError(f"op {op} is not transcribed")
        if img and op not in ("CONCATENATION", "CONV_2D", "DEPTHWISE_CONV_2D", "MEAN") and out not in self.nchw:
            raise ValueError(f"rank-4 result of {op} without an image operand")

    def _img_operand(self, t):
        if t in self.val:
            return self.val[t]
        c = self.const(t).astype(np.float32)                 # per-channel constant [.., C] -> [1, C, 1, 1]
        return self.b.init(c.reshape(1, -1, 1, 1) if c.size > 1 else c.reshape(()))


def transcribe(g, inputs, outputs, dft="matmul", trunc=True):
    """g: a tflite_build.GraphBuilder holding a synth_model graph; -> ONNX bytes with the same weights."""
    tx = _Tx(g, dft, trunc)
    b = tx.b
    (x,) = inputs
    tx.val[x] = b.input("INPUT", ["N", g.tensors[x]["shape"][1]])
    rffts = [oi for oi, o in enumerate(g.ops) if o["op"] == "RFFT2D"]
    fe_at = {}                                                # emit each branch where its first op stands
    for ri in rffts:
        # the branch is emitted when its source tensor exists: at the position of its first RESHAPE (sub-frame split)
        t = g.ops[ri]["inputs"][0]
        first = ri
        while True:
            oi = tx.prod.get(t)
            if oi is None:
                break
            o = g.ops[oi]
            if o["op"] in ("RESHAPE", "PAD", "MUL", "GATHER") and (o["op"] != "PAD" or len(tx.shape(o["outputs"][0])) == 3):
                first = oi
                t = o["inputs"][0]
                if o["op"] == "RESHAPE" and len(tx.shape(o["outputs"][0])) == 3 and g.ops[tx.cons[o["outputs"][0]][0]]["op"] == "GATHER":
                    break
            else:
                break
        fe_at[first] = ri
    for oi in range(len(g.ops)):
        if oi in fe_at:
            tx.frontend(fe_at[oi])
        if oi in tx.done:
            continue
        tx.emit(oi)
    for t in outputs:
        sh = g.tensors[t]["shape"]
        name = tx.val[t]
        if t in tx.nchw and len(sh) == 4:                     # image outputs leave channels-last, as the reference lists them
            name = b.node("Transpose", [name], perm=[0, 2, 3, 1])
        name = b.node("Identity", [name])
        b.output(name, ["N"] + [int(v) for v in sh[1:]])
    return b.finish()
