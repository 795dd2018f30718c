"""The C++ model readers, operand validation and graph passes under AddressSanitizer + UBSan ("a well-framed
but malformed .tflite causes out-of-bounds reads"): mutated TFLite and ONNX files - byte flips, truncations, corrupted
operand lists and constant dtypes - must be accepted or rejected, never read out of bounds.  Builds with plain g++."""
import os
import struct
import subprocess

import numpy as np
import pytest

import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import onnx_build as ob, synth_model as sm

from graphgen import random_graph

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "birdnet-go_amd", "csrc")


@pytest.fixture(scope="module")
def fuzz_bin(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("asan") / "reader_fuzz")
    srcs = [os.path.join(HERE, "native", "reader_fuzz_main.cpp")] + [os.path.join(CSRC, f) for f in
                                                                     ("model_onnx.cpp", "tflite_model.cpp", "graph_passes.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer"] + srcs + ["-o", out])
    return out


def _mutations(blob, rng, n, window=None):
    out = []
    hi = min(len(blob), window or len(blob))
    for _ in range(n):
        b = bytearray(blob)
        for p in rng.integers(0, hi, int(rng.integers(1, 6))):
            b[p] ^= int(rng.integers(1, 256))
        out.append(bytes(b))
    for cut in rng.integers(1, len(blob), max(n // 8, 2)):
        out.append(blob[:int(cut)])
    return out


def test_readers_never_read_out_of_bounds(fuzz_bin, tmp_path, tiny_blob):
    rng = np.random.default_rng(2024)
    corpus = []
    seeds = [tiny_blob, sm.build_dense_model([3, 16, 8], final_sigmoid=True, fp16_weights=True)]
    seeds += [random_graph(s)[0] for s in (1, 5, 9, 12)]
    for b in seeds:
        corpus.append(b)
        corpus += _mutations(b, rng, 120)
        # the flatbuffer's tables (operators, tensors, vtables) sit at the end of the file, the weights at the start
        tail = b[-4096:]
        for m in _mutations(tail, rng, 120):
            corpus.append(b[:-len(tail)] + m[:len(tail)].ljust(len(tail), b"\0"))
    for style in ("gemm", "matmul", "bn"):
        ox, _ = ob.build_dense_head([16, 8, 4], style=style, final="Softmax")
        corpus.append(ox)
        corpus += _mutations(ox, rng, 300)
    for style, nhwc in (("torch", False), ("tf", True)):           # convolutional ONNX graphs: Conv / pools / layout propagation
        ox = ob.build_cnn(in_shape=(1, 12, 16), stem=8, blocks=((1, 3, 1, 8), (4, 5, 2, 8)), top=16, n_classes=5, style=style,
                          nhwc_input=nhwc, emit_embedding=True)
        corpus.append(ox)
        corpus += _mutations(ox, rng, 400)
    path = tmp_path / "corpus.bin"
    with open(path, "wb") as f:
        for b in corpus:
            f.write(struct.pack("<I", len(b)) + b)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([fuzz_bin, str(path)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    n, acc = [int(v) for v in r.stdout.split()[1::2]]
    assert n == len(corpus) and acc >= len(seeds) + 5          # the unmutated files are all accepted


def _crafted_onnx():
    """The files ADVICE r2 reproduced heap overflows with: initializer dims that are negative / wrap the element count while
    the payload is empty, an Unsqueeze with a repeated axis, a Transpose whose perm is not a permutation."""
    out = []
    from birdnet_go_amd.onnx_build import OnnxBuilder, _ld, _str, _vi, _varint, _key

    def raw_tensor(name, dims, payload=b""):
        return b"".join(_key(1, 0) + _varint(d) for d in dims) + _vi(2, 1) + _str(8, name) + _ld(9, payload)

    big = -(2 ** 32) + 4096
    b = OnnxBuilder()
    x = b.input("x", ["N", 4096])
    b.inits.append(raw_tensor("w", [big, big]))                                   # numel wraps, no data
    b.output(b.node("MatMul", [x, "w"]), ["N", 4096])
    out.append(b.finish())
    b = OnnxBuilder()
    x = b.input("x", ["N", 3, 4, 4])
    w = b.init(np.ones((3, 3, 1, 1), np.float32))
    b.inits.append(raw_tensor("c", [3, (2 ** 64 + 2) // 3, 1, 1], b"\0" * 8))      # dims multiply to 2 (mod 2^64): 8 bytes "match"
    b.output(b.node("Add", [b.node("Conv", [x, w], kernel_shape=[1, 1]), "c"]), ["N", 3, 4, 4])
    out.append(b.finish())
    b = OnnxBuilder()
    x = b.input("x", ["N", 8])
    b.output(b.node("Unsqueeze", [x, b.init(np.asarray([1, 1, 1, 1], np.int64))]), ["N", 1, 1, 1, 1, 8])
    out.append(b.finish())
    b = OnnxBuilder()
    x = b.input("x", ["N", 4, 6])
    b.output(b.node("Transpose", [x], perm=[0, 1, 1]), ["N", 4, 4])
    out.append(b.finish())
    # STFT whose frame_step is 2^32 (narrowed to hop 0 -> SIGFPE in the framing), a negative frame_length without
    # a window (-> a huge window allocation), and a conv-DFT-sized stride of 2^32
    for step, flen, win in ((2 ** 32, 256, True), (64, -8, False), (2 ** 32 + 64, 256, True), (64, 2 ** 40, False)):
        b = OnnxBuilder()
        x = b.input("x", ["N", 4096])
        sig = b.node("Unsqueeze", [x, b.init(np.asarray([2], np.int64))])
        ins = [sig, b.init(np.asarray(step, np.int64).reshape(())), b.init(np.hanning(256).astype(np.float32)) if win else "",
               b.init(np.asarray(flen, np.int64).reshape(()))]
        b.output(b.node("STFT", ins, onesided=1), ["N", 61, 129, 2])
        out.append(b.finish())
    return out


def _crafted_fold_arithmetic():
    """constant folding on values taken from the file - Slice steps of INT64_MAX / INT64_MIN (signed overflow in the
    count arithmetic), an opset-15 Shape with `start` (the full shape is not its value), Cast of NaN / inf / 1e30 to int64, and
    integer Div / Mul at the ends of the range.  Valid or not, none may trip UBSan."""
    from birdnet_go_amd.onnx_build import OnnxBuilder
    out = []
    i64 = lambda *v: np.asarray(v, np.int64)
    for step, a, e in ((2 ** 63 - 1, 0, 8), (-(2 ** 63), 7, -(2 ** 62)), (-(2 ** 63) + 1, 7, -9)):
        b = OnnxBuilder()
        x = b.input("x", ["N", 8])
        c = b.init(np.arange(8, dtype=np.float32))
        sl = b.node("Slice", [c, b.init(i64(a)), b.init(i64(e)), b.init(i64(0)), b.init(i64(step))])
        b.output(b.node("Add", [x, sl]), ["N", 8])
        out.append(b.finish())
    b = OnnxBuilder()
    x = b.input("x", ["N", 8])
    sh = b.node("Shape", [b.init(np.zeros((2, 4), np.float32))], start=1)
    b.output(b.node("Reshape", [x, b.node("Concat", [b.init(i64(-1)), sh], axis=0)]), ["N", 4])
    out.append(b.finish())
    b = OnnxBuilder()
    x = b.input("x", ["N", 8])
    k = b.node("Cast", [b.init(np.asarray([np.nan, np.inf, 1e30], np.float32))], to=7)
    b.output(b.node("Reshape", [x, k]), ["N", 8])
    out.append(b.finish())
    for op, u, v in (("Div", -(2 ** 63), -1), ("Mul", 2 ** 62, 4), ("Add", 2 ** 63 - 1, 2 ** 63 - 1), ("Sub", -(2 ** 63), 1)):
        b = OnnxBuilder()
        x = b.input("x", ["N", 8])
        b.output(b.node("Reshape", [x, b.node(op, [b.init(i64(u)), b.init(i64(v))])]), ["N", 8])
        out.append(b.finish())
    return out


def _crafted_empty_slice_axes():
    """the reverse `Slice` behind the mel MatMul with an EMPTY `axes` initializer (dims=[0]): tail_ok() indexed
    ax[0] of an empty vector.  Written by the audio transcriber with the Slice's axes operand swapped for an empty one."""
    from birdnet_go_amd import onnx_audio
    Builder = onnx_audio.OnnxBuilder          # (the class object the transcriber uses, whatever name the module was imported under)
    orig = Builder.node
    made = []

    def node(self, op, inputs, *a, **k):
        if op == "Slice" and len(inputs) == 5:
            inputs = list(inputs)
            inputs[3] = self.init(np.zeros((0,), np.int64))
            made.append(1)
        return orig(self, op, inputs, *a, **k)

    Builder.node = node
    try:
        blob = sm.build_model(sm.tiny_config(), container="onnx", dft="matmul")
    finally:
        Builder.node = orig
    assert made, "the transcriber no longer writes the reverse Slice this file is about"
    return blob


def test_onnx_audio_front_ends_and_crafted_files_under_asan(fuzz_bin, tmp_path):
    """The ONNX audio front-end recogniser (DFT MatMul / Conv1d / STFT / DFT forms, symbolic spectra, lazy transposition) and
    the crafted files of ADVICE r2 under ASan + UBSan: accept or reject, never read or write out of bounds."""
    rng = np.random.default_rng(77)
    corpus = []
    n_seed = 0
    for cfg in (sm.tiny_config(), sm.tiny_config(complex_mode="abs"), sm.tiny_perch_config()):
        for form in ("matmul", "conv1d", "stft", "dft"):
            try:
                ox = sm.build_model(cfg, container="onnx", dft=form)
            except ValueError:
                continue
            n_seed += 1
            corpus.append(ox)
            # the nodes and small initializers sit at the front of the graph, the big weights behind them: mutate both ends
            corpus += _mutations(ox, rng, 32, window=6000)
            tail = ox[-3000:]
            for mt in _mutations(tail, rng, 16):
                corpus.append(ox[:-len(tail)] + mt[:len(tail)].ljust(len(tail), b"\0"))
    crafted = _crafted_onnx() + [_crafted_empty_slice_axes()]
    corpus += crafted
    corpus += _crafted_fold_arithmetic()          # (may be accepted - some are valid ONNX; UBSan must stay silent)
    path = tmp_path / "corpus_audio.bin"
    with open(path, "wb") as f:
        for b in corpus:
            f.write(struct.pack("<I", len(b)) + b)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([fuzz_bin, str(path)], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-6000:])
    n, acc = [int(v) for v in r.stdout.split()[1::2]]
    assert n == len(corpus) and acc >= n_seed
    # and none of the crafted files is accepted
    path2 = tmp_path / "crafted.bin"
    with open(path2, "wb") as f:
        for b in crafted:
            f.write(struct.pack("<I", len(b)) + b)
    r = subprocess.run([fuzz_bin, str(path2)], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr[-4000:]
    n, acc = [int(v) for v in r.stdout.split()[1::2]]
    assert (n, acc) == (len(crafted), 0)
