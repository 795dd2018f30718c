"""ORACLE (test infrastructure only) — ctypes front for `oracle/c/oracle.c`, the plain-C restatement
of the reference's in-tree Go arithmetic around the native call (PCM convert, sigmoid, softmax,
top-K, ultrasonic frame-CV).  Pinned against the reference tests' known answers in
`tests/test_oracle_kat.py` (analyze_test.go, postprocess_test.go, pcm_test.go, filter_test.go).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "c", "liboracle.so")
_SRC = os.path.join(_HERE, "c", "oracle.c")


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-fPIC", "-shared", "-ffp-contract=off",
                               "-o", _SO, _SRC, "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_topk.restype = C.c_int
        _lib.orc_us_frame_cv.restype = C.c_int
        _lib.orc_cv.restype = C.c_double
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def pcm_to_f32(raw: bytes, bit_depth: int) -> np.ndarray:
    """convert/pcm.go:206-268 ConvertToFloat32 (one channel)."""
    if bit_depth not in (16, 24, 32):
        raise ValueError(f"unsupported audio bit depth: {bit_depth}")
    b = np.frombuffer(raw, np.uint8)
    n = len(b) // (bit_depth // 8)
    out = np.empty(n, np.float32)
    fn = {16: lib().orc_pcm16_to_f32, 24: lib().orc_pcm24_to_f32, 32: lib().orc_pcm32_to_f32}[bit_depth]
    fn(_p(b, C.c_uint8), C.c_int(n), _p(out, C.c_float))
    return out


def sigmoid_sensitivity(logits, sensitivity=1.0) -> np.ndarray:
    x = np.ascontiguousarray(logits, np.float32).reshape(-1)
    out = np.empty_like(x)
    lib().orc_sigmoid_sens(_p(x, C.c_float), C.c_int(x.size), C.c_double(sensitivity), _p(out, C.c_float))
    return out.reshape(np.shape(logits))


def sigmoid_f32div(logits) -> np.ndarray:
    x = np.ascontiguousarray(logits, np.float32).reshape(-1)
    out = np.empty_like(x)
    lib().orc_sigmoid_f32div(_p(x, C.c_float), C.c_int(x.size), _p(out, C.c_float))
    return out.reshape(np.shape(logits))


def softmax(logits) -> np.ndarray:
    x = np.ascontiguousarray(logits, np.float32).reshape(-1)
    out = np.empty_like(x)
    lib().orc_softmax(_p(x, C.c_float), C.c_int(x.size), _p(out, C.c_float))
    return out


def topk(conf, k):
    """-> (conf[k'], idx[k']) like getTopKResults (analyze.go:220-253)."""
    x = np.ascontiguousarray(conf, np.float32).reshape(-1)
    m = max(0, min(k, x.size))
    co = np.empty(max(m, 1), np.float32)
    io = np.empty(max(m, 1), np.int32)
    got = lib().orc_topk(_p(x, C.c_float), C.c_int(x.size), C.c_int(k), _p(co, C.c_float), _p(io, C.c_int))
    return co[:got].copy(), io[:got].copy()


def fft_c128(data) -> np.ndarray:
    d = np.ascontiguousarray(data, np.complex128).copy()
    lib().orc_fft(d.ctypes.data_as(C.c_void_p), C.c_int(d.size))
    return d


def hanning(n) -> np.ndarray:
    w = np.empty(n, np.float64)
    lib().orc_hanning(_p(w, C.c_double), C.c_int(n))
    return w


def coefficient_of_variation(v) -> float:
    a = np.ascontiguousarray(v, np.float64).reshape(-1)
    if a.size == 0:
        return 0.0
    return float(lib().orc_cv(_p(a, C.c_double), C.c_int(a.size)))


def us_frame_cv(samples, sample_rate, fft_size=8192, hop=4096, split_hz=20000, want_powers=False):
    """ultrasonic/filter.go:20-66 ComputeUSFrameCV -> (cv, ok[, frame_powers])."""
    s = np.ascontiguousarray(samples, np.float64).reshape(-1)
    cv = C.c_double(0.0)
    frames = 1 + (s.size - fft_size) // hop if (hop > 0 and s.size >= fft_size) else 0
    pw = np.zeros(max(frames, 1), np.float64)
    ok = lib().orc_us_frame_cv(_p(s, C.c_double), C.c_int(s.size), C.c_int(sample_rate), C.c_int(fft_size),
                               C.c_int(hop), C.c_int(split_hz), C.byref(cv), _p(pw, C.c_double))
    if want_powers:
        return cv.value, bool(ok), pw[:frames]
    return cv.value, bool(ok)


def is_unlikely(cv, threshold=0.15) -> bool:
    """filter.go:71-73: strict less-than."""
    return cv < threshold


def resample_edge_out(f) -> np.ndarray:
    x = np.ascontiguousarray(f, np.float32).reshape(-1)
    out = np.empty(x.size, np.int16)
    lib().orc_resample_edge_out(_p(x, C.c_float), C.c_int(x.size), _p(out, C.c_int16))
    return out
