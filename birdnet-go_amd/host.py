"""Host-side mirror of the reference's backend interface over the C ABI (ctypes).

Mirrors, name for name, the Go seam this engine drops in behind:
  inference.Classifier          internal/inference/backend.go:8-19   -> HipClassifier.predict/num_species/close
  inference.EmbeddingExtractor  internal/inference/backend.go:21-29  -> HipClassifier.predict_with_embeddings
  onnx.Classifier.PredictBatch  internal/inference/onnx/classifier.go:372-430 -> HipClassifier.predict_batch
  (*BirdNET).Predict post-proc  internal/classifier/analyze.go:25-110 -> BirdNET.predict (sigmoid(sens) + top-10)
Error behaviour follows the reference: size mismatch is an error (tflite/classifier.go:102-104), the
"backend unavailable" condition is a distinct sentinel so callers can fall back
(openvino/openvino.go:26-31 ErrOpenVINOUnavailable), and nothing ever silently falls back to a CPU path.

This file is plumbing for tests/bench in this repo (Go is not installed here); the production
binding is the cgo file under go/ (see INTEGRATION.md).
"""
import ctypes as C
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TUNE_DIR = os.path.join(_HERE, "tune")             # recorded create-time tunings shipped with the package (bnhip.h "tune_dir")
LIB_PATH = os.environ.get("BNHIP_LIB") or os.path.join(_HERE, "lib", "libbnhip.so")   # BNHIP_LIB: A/B a second build

BNHIP_OK, E_INVALID, E_NO_DEVICE, E_MODEL, E_UNSUPPORTED, E_RUNTIME, E_NOMEM = 0, -1, -2, -3, -4, -5, -6

SYMBOLS = ["bnhip_init", "bnhip_shutdown", "bnhip_model_create", "bnhip_model_info", "bnhip_predict",
           "bnhip_predict_pcm16", "bnhip_predict_pcm", "bnhip_predict_device", "bnhip_postprocess_topk", "bnhip_predict_topk",
           "bnhip_us_frame_cv", "bnhip_set_stream", "bnhip_synchronize", "bnhip_profile_enable",
           "bnhip_profile_read", "bnhip_model_describe", "bnhip_model_destroy", "bnhip_last_error",
           "bnhip_version", "bnhip_debug_fetch", "bnhip_profile_filter", "bnhip_resample_length",
           "bnhip_resample_f32", "bnhip_resample_pcm16", "bnhip_model_devices", "bnhip_last_error_copy",
           "bnhip_resampler_create", "bnhip_resampler_estimate", "bnhip_resampler_process_pcm16",
           "bnhip_resampler_process_f32", "bnhip_resampler_flush_pcm16", "bnhip_resampler_flush_f32",
           "bnhip_resampler_destroy", "bnhip_us_frame_cv_device", "bnhip_profile_steps", "bnhip_profile_steps_read",
           "bnhip_host_alloc", "bnhip_host_free", "bnhip_windows_create", "bnhip_windows_info", "bnhip_windows_add_source",
           "bnhip_windows_remove_source", "bnhip_windows_write", "bnhip_windows_collect", "bnhip_windows_ready",
           "bnhip_windows_stats", "bnhip_windows_reset", "bnhip_windows_destroy", "bnhip_predict_pcm_topk",
           "bnhip_windows_predict_topk"]


class HipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"bnhip error {code}: {msg}")
        self.code = code


class ErrHIPUnavailable(HipError):
    """Sentinel like openvino.ErrOpenVINOUnavailable: library missing / no gfx950 device."""


_lib = None


def load_library(path=None):
    """Loads libbnhip.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ErrHIPUnavailable(E_NO_DEVICE, f"{p} not built (run `python -c 'import __graft_entry__ as g; g.build()'`)")
    lib = C.CDLL(p)
    lib.bnhip_last_error.restype = C.c_char_p
    lib.bnhip_version.restype = C.c_char_p
    lib.bnhip_model_create.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.POINTER(C.c_void_p)]
    lib.bnhip_model_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.bnhip_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.bnhip_predict_pcm16.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.bnhip_predict_pcm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.bnhip_predict_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.bnhip_postprocess_topk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                           C.c_void_p, C.c_void_p]
    lib.bnhip_predict_topk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                       C.c_void_p]
    lib.bnhip_predict_pcm_topk.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_void_p,
                                           C.c_void_p]
    lib.bnhip_us_frame_cv.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p]
    lib.bnhip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.bnhip_synchronize.argtypes = [C.c_void_p]
    lib.bnhip_profile_enable.argtypes = [C.c_void_p, C.c_int]
    lib.bnhip_profile_read.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.bnhip_profile_filter.argtypes = [C.c_void_p, C.c_char_p]
    lib.bnhip_model_describe.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
    lib.bnhip_model_destroy.argtypes = [C.c_void_p]
    lib.bnhip_init.argtypes = [C.POINTER(C.c_int)]
    if path is None:
        _lib = lib
    return lib


def _check(lib, rc):
    if rc != BNHIP_OK:
        msg = (lib.bnhip_last_error() or b"").decode("utf-8", "replace")
        raise (ErrHIPUnavailable if rc == E_NO_DEVICE else HipError)(rc, msg)


def init():
    """-> number of devices (InitOV analogue, backend_openvino.go:477)."""
    lib = load_library()
    n = C.c_int(0)
    _check(lib, lib.bnhip_init(C.byref(n)))
    return n.value


class PinnedArray:
    """A numpy array over page-locked memory from bnhip_host_alloc (the reference's shim keeps a C-allocated input buffer per
    classifier, backend_openvino.go:673-680): bnhip_predict* read / write such buffers by DMA, without the staging copy.
    `.array` is the view; free() (or the with-statement) releases it.  The memory is C-owned: views or slices of `.array` taken
    before free() dangle afterwards (numpy cannot know) - copy what must outlive it, and never free() during a predict call."""

    def __init__(self, shape, dtype=np.float32):
        self._lib = load_library()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = C.c_void_p()
        self._lib.bnhip_host_alloc.argtypes = [C.c_size_t, C.POINTER(C.c_void_p)]
        _check(self._lib, self._lib.bnhip_host_alloc(n, C.byref(p)))          # (zero bytes: BNHIP_E_INVALID)
        self._p = p
        buf = (C.c_char * n).from_address(p.value)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self._p:
            self.array = None
            self._lib.bnhip_host_free.argtypes = [C.c_void_p]
            self._lib.bnhip_host_free(self._p)
            self._p = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.free()


class HipClassifier:
    """inference.Classifier + EmbeddingExtractor over libbnhip.so.  NOT thread-safe (backend.go:7)."""

    def __init__(self, model_bytes: bytes, device=0, max_batch=256, plan_only=False, debug_no_reuse=False,
                 graphs=None, frontend_fft=None, depth=None, lanes=None, autotune=None, devices=None, replicate=None,
                 bf16x3=None, precision=None, logits_output=None, embedding_output=None, host_depth=None, tune_dir=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        o = {"device": device, "max_batch": max_batch, "plan_only": int(plan_only), "debug_no_reuse": int(debug_no_reuse)}
        if devices is not None:          # one handle sharding every call over several GPUs (bnhip.h "devices")
            o["devices"] = [int(d) for d in devices]
        if replicate is not None:
            o["replicate"] = str(replicate)
        if bf16x3 is not None:
            o["bf16x3"] = int(bf16x3)
        if logits_output is not None:    # graph output indices (default: the reference's per-family rule, bnhip.h)
            o["logits_output"] = int(logits_output)
        if embedding_output is not None:
            o["embedding_output"] = int(embedding_output)
        if precision is not None:        # "f32" (default) | "bf16": MFMA operands rounded to bf16, fp32 accumulate (Perch-style)
            o["precision"] = str(precision)
        if graphs is not None:
            o["graphs"] = int(graphs)
        if frontend_fft is not None:
            o["frontend_fft"] = int(frontend_fft)
        if depth is not None:
            o["depth"] = int(depth)
        if host_depth is not None:       # contexts the blocking host-pointer entries pipeline their chunks over (default 2)
            o["host_depth"] = int(host_depth)
        if lanes is not None:
            o["lanes"] = int(lanes)
        if autotune is not None:
            o["autotune"] = int(autotune)
        # recorded tunings (bnhip.h "tune_dir"): the package's own directory unless the caller or BNHIP_TUNE_DIR says otherwise
        # ("" = none: always time the candidates)
        if tune_dir is None and "BNHIP_TUNE_DIR" not in os.environ and os.path.isdir(TUNE_DIR):
            tune_dir = TUNE_DIR
        if tune_dir:
            o["tune_dir"] = str(tune_dir)
        opts = json.dumps(o).encode()
        buf = (C.c_char * len(model_bytes)).from_buffer_copy(model_bytes)
        _check(self._lib, self._lib.bnhip_model_create(C.cast(buf, C.c_void_p), len(model_bytes), opts, C.byref(self._h)))
        ns, nc, ed = C.c_int(), C.c_int(), C.c_int()
        _check(self._lib, self._lib.bnhip_model_info(self._h, C.byref(ns), C.byref(nc), C.byref(ed)))
        self.n_samples, self._n_classes, self.emb_dim = ns.value, nc.value, ed.value
        self.max_batch = max_batch

    # ---- inference.Classifier
    def predict(self, samples):
        """Predict(samples []float32) ([]float32, error): raw logits for ONE clip."""
        x = np.ascontiguousarray(samples, np.float32).reshape(-1)
        if x.size != self.n_samples:
            raise HipError(E_INVALID, f"input size mismatch: expected {self.n_samples} samples, got {x.size}")
        return self.predict_batch(x, 1)[0]

    def num_species(self):
        return self._n_classes

    def close(self):
        if self._h:
            self._lib.bnhip_model_destroy(self._h)
            self._h = C.c_void_p()

    # ---- inference.EmbeddingExtractor
    def predict_with_embeddings(self, samples):
        x = np.ascontiguousarray(samples, np.float32).reshape(-1)
        if x.size != self.n_samples:
            raise HipError(E_INVALID, f"input size mismatch: expected {self.n_samples} samples, got {x.size}")
        if not self.emb_dim:
            return self.predict(x), None
        lg, em = self.predict_batch(x, 1, want_embeddings=True)
        return lg[0], em[0]

    # ---- onnx.Classifier.PredictBatch shape: flat [B*N] in, [B, classes] out
    def predict_batch(self, flat, batch_size, want_embeddings=False, out=None):
        """`out`: optional preallocated float32 [batch_size, classes] result array (a serving loop reuses it; a fresh
        numpy array costs a page fault per 4 KB on its first write)."""
        self._alive()
        x = np.ascontiguousarray(flat, np.float32).reshape(-1)
        if batch_size <= 0 or x.size != batch_size * self.n_samples:
            raise HipError(E_INVALID, f"input size mismatch: expected {batch_size * self.n_samples} samples, got {x.size}")
        logits = self._out(out, batch_size)
        emb = np.empty((batch_size, self.emb_dim), np.float32) if (want_embeddings and self.emb_dim) else None
        _check(self._lib, self._lib.bnhip_predict(self._h, x.ctypes.data, batch_size, logits.ctypes.data,
                                                  emb.ctypes.data if emb is not None else None))
        return (logits, emb) if want_embeddings else logits

    def _out(self, out, batch_size):
        if out is None:
            return np.empty((batch_size, self._n_classes), np.float32)
        if out.dtype != np.float32 or not out.flags.c_contiguous or out.size != batch_size * self._n_classes:
            raise HipError(E_INVALID, "out must be a C-contiguous float32 array of batch_size * classes elements")
        return out.reshape(batch_size, self._n_classes)

    def predict_pcm16(self, pcm, batch_size, out=None):
        self._alive()
        x = np.ascontiguousarray(pcm, np.int16).reshape(-1)
        if x.size != batch_size * self.n_samples:
            raise HipError(E_INVALID, f"input size mismatch: expected {batch_size * self.n_samples} samples, got {x.size}")
        logits = self._out(out, batch_size)
        _check(self._lib, self._lib.bnhip_predict_pcm16(self._h, x.ctypes.data, batch_size, logits.ctypes.data, None))
        return logits

    def predict_pcm(self, raw: bytes, bit_depth: int, batch_size: int):
        """Little-endian PCM bytes of the three depths ConvertToFloat32 accepts (convert/pcm.go:206-268), converted on the
        device.  An unsupported depth is a validation error, as in the reference (pcm.go:215-222)."""
        self._alive()
        if bit_depth not in (16, 24, 32):
            raise HipError(E_INVALID, f"unsupported bit depth: {bit_depth} (supported: 16, 24, 32)")
        x = np.frombuffer(raw, np.uint8)
        if x.size != batch_size * self.n_samples * (bit_depth // 8):
            raise HipError(E_INVALID, f"input size mismatch: expected {batch_size * self.n_samples} samples of "
                                      f"{bit_depth // 8} bytes, got {x.size} bytes")
        logits = np.empty((batch_size, self._n_classes), np.float32)
        _check(self._lib, self._lib.bnhip_predict_pcm(self._h, x.ctypes.data, bit_depth, batch_size, logits.ctypes.data, None))
        return logits

    def predict_device(self, d_samples_ptr, n_clips, d_logits_ptr, d_emb_ptr=None):
        self._alive()
        _check(self._lib, self._lib.bnhip_predict_device(self._h, d_samples_ptr, n_clips, d_logits_ptr, d_emb_ptr))

    def postprocess_topk(self, logits, k=10, activation=0, sensitivity=1.0):
        self._alive()
        lg = np.ascontiguousarray(logits, np.float32).reshape(-1, self._n_classes)
        kk = min(k, self._n_classes)
        conf = np.empty((lg.shape[0], kk), np.float32)
        idx = np.empty((lg.shape[0], kk), np.int32)
        _check(self._lib, self._lib.bnhip_postprocess_topk(self._h, lg.ctypes.data, lg.shape[0], self._n_classes,
                                                           activation, sensitivity, k, conf.ctypes.data, idx.ctypes.data))
        return conf, idx

    def predict_topk(self, flat, batch_size, k=10, activation=0, sensitivity=1.0):
        self._alive()
        x = np.ascontiguousarray(flat, np.float32).reshape(-1)
        if x.size != batch_size * self.n_samples:
            raise HipError(E_INVALID, f"input size mismatch: expected {batch_size * self.n_samples} samples, got {x.size}")
        kk = min(k, self._n_classes)
        conf = np.empty((batch_size, kk), np.float32)
        idx = np.empty((batch_size, kk), np.int32)
        _check(self._lib, self._lib.bnhip_predict_topk(self._h, x.ctypes.data, batch_size, activation, sensitivity, k,
                                                       conf.ctypes.data, idx.ctypes.data))
        return conf, idx

    def predict_pcm_topk(self, raw, bit_depth, batch_size, k=10, activation=0, sensitivity=1.0):
        """predict_pcm + postprocess_topk in one call (bnhip_predict_pcm_topk): the windows' PCM bytes in, top-k out; neither the
        float samples nor the logits exist on the host."""
        self._alive()
        if bit_depth not in (16, 24, 32):
            raise HipError(E_INVALID, f"unsupported bit depth: {bit_depth} (supported: 16, 24, 32)")
        x = np.frombuffer(raw, np.uint8)
        if x.size != batch_size * self.n_samples * (bit_depth // 8):
            raise HipError(E_INVALID, f"input size mismatch: expected {batch_size * self.n_samples} samples of "
                                      f"{bit_depth // 8} bytes, got {x.size} bytes")
        kk = min(k, self._n_classes)
        conf = np.empty((batch_size, kk), np.float32)
        idx = np.empty((batch_size, kk), np.int32)
        _check(self._lib, self._lib.bnhip_predict_pcm_topk(self._h, x.ctypes.data, bit_depth, batch_size, activation, sensitivity, k,
                                                           conf.ctypes.data, idx.ctypes.data))
        return conf, idx

    # ---- plumbing
    def set_stream(self, hip_stream_ptr):
        _check(self._lib, self._lib.bnhip_set_stream(self._h, hip_stream_ptr))

    def synchronize(self):
        _check(self._lib, self._lib.bnhip_synchronize(self._h))

    def profile_enable(self, on=True):
        _check(self._lib, self._lib.bnhip_profile_enable(self._h, int(on)))

    def profile_filter(self, kernel_class=None):
        _check(self._lib, self._lib.bnhip_profile_filter(self._h, kernel_class.encode() if kernel_class else None))

    def profile_read(self, per_step=False):
        """Per-kernel-class timing rows; with per_step=True returns (classes, steps)."""
        buf = C.create_string_buffer(1 << 18)
        rc = self._lib.bnhip_profile_read(self._h, buf, len(buf))
        if rc < 0:
            _check(self._lib, rc)
        rows = json.loads(buf.value.decode())
        classes = [r for r in rows if "step" not in r]
        return (classes, [r for r in rows if "step" in r]) if per_step else classes

    def profile_steps(self, on=True):
        _check(self._lib, self._lib.bnhip_profile_steps(self._h, int(on)))

    def profile_steps_read(self, cap=4096):
        """(start_ms, end_ms) of every call since the last read, relative to the first call's start."""
        a, b = np.zeros(cap, np.float64), np.zeros(cap, np.float64)
        lib = self._lib
        lib.bnhip_profile_steps_read.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        n = lib.bnhip_profile_steps_read(self._h, a.ctypes.data, b.ctypes.data, cap)
        if n < 0:
            _check(lib, n)
        n = min(n, cap)
        return a[:n].copy(), b[:n].copy()

    def describe(self):
        need = self._lib.bnhip_model_describe(self._h, None, 0)
        if need < 0:
            _check(self._lib, need)
        buf = C.create_string_buffer(need + 16)
        self._lib.bnhip_model_describe(self._h, buf, len(buf))
        return json.loads(buf.value.decode())

    def debug_fetch(self, tensor_index, n_clips, max_floats_per_clip):
        out = np.empty(n_clips * max_floats_per_clip, np.float32)
        lib = self._lib
        lib.bnhip_debug_fetch.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        rc = lib.bnhip_debug_fetch(self._h, tensor_index, n_clips, out.ctypes.data, out.size)
        if rc < 0:
            _check(lib, rc)
        return out[:rc * n_clips].reshape(n_clips, rc)

    def _alive(self):
        if not self._h:
            raise HipError(E_INVALID, "classifier is closed")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def us_frame_cv(samples, sample_rate, fft_size=8192, hop=4096, split_hz=20000, device=0):
    """ultrasonic.ComputeUSFrameCV (filter.go:20) for a batch: samples [B, n] float64 -> (cv[B], ok[B])."""
    lib = load_library()
    s = np.ascontiguousarray(samples, np.float64)
    if s.ndim == 1:
        s = s[None, :]
    cv = np.zeros(s.shape[0], np.float64)
    ok = np.zeros(s.shape[0], np.int32)
    _check(lib, lib.bnhip_us_frame_cv(device, s.ctypes.data, s.shape[0], s.shape[1], sample_rate, fft_size, hop,
                                      split_hz, cv.ctypes.data, ok.ctypes.data))
    return cv, ok.astype(bool)


def _sigmoid_f32div(x):
    """onnx/postprocess.go:8-10: 1.0 / (1.0 + float32(exp(float64(-x)))), the division in float32."""
    x = np.asarray(x, np.float32)
    e = np.exp(-x.astype(np.float64)).astype(np.float32)
    return (np.float32(1.0) / (np.float32(1.0) + e)).astype(np.float32)


class CustomClassifier:
    """inference.CustomClassifier (backend.go:31-52): secondary head on embedding vectors, e.g. a BattyBirdNET
    regional classifier.  PredictEmbedding returns sigmoid-applied scores (custom_classifier.go:148-174)."""

    def __init__(self, head_bytes: bytes, labels, device=0, max_batch=256):
        self._clf = HipClassifier(head_bytes, device=device, max_batch=max_batch)
        if len(labels) != self._clf.num_species():
            raise HipError(E_INVALID, f"label count {len(labels)} != head outputs {self._clf.num_species()}")
        self._labels = list(labels)

    def predict_embedding(self, embeddings):
        e = np.ascontiguousarray(embeddings, np.float32).reshape(-1)
        if e.size != self.input_dim():
            raise HipError(E_INVALID, f"input size mismatch: expected {self.input_dim()} values, got {e.size}")
        return _sigmoid_f32div(self._clf.predict_batch(e, 1)[0])

    def predict_embedding_batch(self, embeddings, batch_size):
        return _sigmoid_f32div(self._clf.predict_batch(embeddings, batch_size))

    def num_classes(self):
        return self._clf.num_species()

    def input_dim(self):
        return self._clf.n_samples

    def labels(self):
        return list(self._labels)

    def close(self):
        self._clf.close()


class RangeFilter:
    """inference.RangeFilter / BatchRangeFilter (backend.go:55-76): [lat, lon, week] -> per-species occurrence."""

    def __init__(self, model_bytes: bytes, device=0, max_batch=1024):
        self._clf = HipClassifier(model_bytes, device=device, max_batch=max_batch)
        if self._clf.n_samples != 3:
            raise HipError(E_INVALID, f"range filter model must take 3 inputs, takes {self._clf.n_samples}")

    def predict(self, latitude, longitude, week):
        return self._clf.predict_batch(np.asarray([latitude, longitude, week], np.float32), 1)[0]

    def predict_batch(self, inputs, batch_size):
        x = np.ascontiguousarray(inputs, np.float32).reshape(-1)
        if x.size != batch_size * 3:
            raise HipError(E_INVALID, f"input size mismatch: expected {batch_size * 3} values, got {x.size}")
        return self._clf.predict_batch(x, batch_size).reshape(-1)

    def num_species(self):
        return self._clf.num_species()

    def close(self):
        self._clf.close()


class Bat:
    """Bat.Predict (classifier/bat_onnx.go:220-342): v2.4 backbone -> 1024-d embedding -> regional head ->
    plain sigmoid -> confidence threshold -> top-10.  The audio is 256 kHz material fed as if 48 kHz."""

    TOP_K = 10

    def __init__(self, backbone: HipClassifier, head: CustomClassifier, threshold=0.0):
        if not backbone.emb_dim:
            raise HipError(E_INVALID, "backbone model exposes no embedding output")
        if head.input_dim() != backbone.emb_dim:
            raise HipError(E_INVALID, f"head expects {head.input_dim()}-d embeddings, backbone yields {backbone.emb_dim}")
        self.backbone, self.head, self.threshold = backbone, head, float(threshold)

    def predict(self, samples):
        _, emb = self.backbone.predict_with_embeddings(samples)
        scores = self.head.predict_embedding(emb)
        order = np.argsort(-scores, kind="stable")
        labels = self.head.labels()
        return [(labels[i], float(scores[i])) for i in order if scores[i] >= self.threshold][:self.TOP_K]


class Resampler:
    """Resampler (internal/audiocore/resample/resample.go:57-172) on the GPU, stateless per clip.
    `resample_to(pcm16)` keeps the reference's int16 edges; `resample_f32` is the float path."""

    def __init__(self, from_rate, to_rate, device=0):
        if from_rate <= 0 or to_rate <= 0:
            raise HipError(E_INVALID, f"failed to create resampler from {from_rate} Hz to {to_rate} Hz")
        self.from_rate, self.to_rate, self.device = int(from_rate), int(to_rate), device
        self._lib = load_library()
        self._lib.bnhip_resample_f32.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        self._lib.bnhip_resample_pcm16.argtypes = self._lib.bnhip_resample_f32.argtypes

    def estimate_output(self, n_in):
        return int(self._lib.bnhip_resample_length(int(n_in), self.from_rate, self.to_rate))

    def _run(self, fn, x, dtype):
        x = np.ascontiguousarray(x, dtype)
        flat = x.ndim == 1
        if flat:
            x = x[None, :]
        no = self.estimate_output(x.shape[1])
        out = np.empty((x.shape[0], no), dtype)
        n = C.c_int(0)
        _check(self._lib, fn(self.device, x.ctypes.data, x.shape[0], x.shape[1], self.from_rate, self.to_rate,
                             out.ctypes.data, no, C.byref(n)))
        return out[0] if flat else out

    def resample_f32(self, samples):
        return self._run(self._lib.bnhip_resample_f32, samples, np.float32)

    def resample_to(self, pcm16):
        """int16 in -> int16 out (ResampleTo); accepts raw little-endian bytes or an int16 array."""
        if isinstance(pcm16, (bytes, bytearray)):
            if len(pcm16) % 2:
                raise HipError(E_INVALID, f"input length {len(pcm16)} is not a multiple of 2 (16-bit PCM requires even byte count)")
            return self._run(self._lib.bnhip_resample_pcm16, np.frombuffer(pcm16, "<i2"), np.int16).tobytes()
        return self._run(self._lib.bnhip_resample_pcm16, pcm16, np.int16)


class StreamResampler:
    """The reference's stateful Resampler (internal/audiocore/resample/resample.go:44-224), method for method:
    NewResampler(from, to) returns None for equal rates; resample_to / resample_into consume 16-bit PCM frames of any size
    and return what the stream so far determines; the FIR history stays on the GPU between calls, so the concatenated
    output of any chunking equals one one-shot call over the whole stream, bit for bit (after `flush`)."""

    def __init__(self, from_rate, to_rate, device=0):
        self._lib = load_library()
        L = self._lib
        L.bnhip_resampler_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.bnhip_resampler_estimate.argtypes = [C.c_void_p, C.c_int]
        for fn in (L.bnhip_resampler_process_pcm16, L.bnhip_resampler_process_f32):
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        for fn in (L.bnhip_resampler_flush_pcm16, L.bnhip_resampler_flush_f32):
            fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.bnhip_resampler_destroy.argtypes = [C.c_void_p]
        self.from_rate, self.to_rate = int(from_rate), int(to_rate)
        self._h = C.c_void_p()
        _check(L, L.bnhip_resampler_create(device, self.from_rate, self.to_rate, C.byref(self._h)))

    @classmethod
    def new(cls, from_rate, to_rate, device=0):
        """NewResampler: None when no resampling is required (resample.go:58-60)."""
        return None if from_rate == to_rate else cls(from_rate, to_rate, device)

    def estimate_output_bytes(self, input_bytes):
        if input_bytes <= 0:
            return 0
        return int(self._lib.bnhip_resampler_estimate(self._h, input_bytes // 2)) * 2

    def resample_to(self, pcm: bytes, dst: bytearray):
        """ResampleTo(input, dst) -> bytes written; errors leave the stream state untouched."""
        self._alive()
        if len(pcm) == 0:
            return 0
        if len(pcm) % 2:
            raise HipError(E_INVALID, f"input length {len(pcm)} is not a multiple of 2 (16-bit PCM requires even byte count)")
        x = np.frombuffer(pcm, "<i2")
        buf = (C.c_char * len(dst)).from_buffer(dst)
        n = C.c_int(0)
        _check(self._lib, self._lib.bnhip_resampler_process_pcm16(self._h, x.ctypes.data, x.size, C.addressof(buf), len(dst) // 2, C.byref(n)))
        return n.value * 2

    def resample_into(self, pcm: bytes) -> bytes:
        dst = bytearray(self.estimate_output_bytes(len(pcm)))
        n = self.resample_to(pcm, dst)
        return bytes(dst[:n])

    def process_f32(self, samples):
        self._alive()
        x = np.ascontiguousarray(samples, np.float32).reshape(-1)
        out = np.empty(int(self._lib.bnhip_resampler_estimate(self._h, x.size)) if x.size else 0, np.float32)
        if not x.size:
            return out
        n = C.c_int(0)
        _check(self._lib, self._lib.bnhip_resampler_process_f32(self._h, x.ctypes.data, x.size, out.ctypes.data, out.size, C.byref(n)))
        return out[:n.value]

    def flush(self, pcm16=True, cap=1 << 16):
        """End of stream: the tail that needed future (zero) input; resets the state."""
        self._alive()
        out = np.empty(cap, np.int16 if pcm16 else np.float32)
        n = C.c_int(0)
        fn = self._lib.bnhip_resampler_flush_pcm16 if pcm16 else self._lib.bnhip_resampler_flush_f32
        _check(self._lib, fn(self._h, out.ctypes.data, out.size, C.byref(n)))
        return out[:n.value].tobytes() if pcm16 else out[:n.value]

    def close(self):
        if self._h:
            self._lib.bnhip_resampler_destroy(self._h)
            self._h = C.c_void_p()

    def _alive(self):
        if not self._h:
            raise HipError(E_INVALID, "resampler is closed")

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Perch:
    """Perch.Predict (classifier/perch_onnx.go:216-255): 160000 samples (32 kHz x 5 s) -> logits (graph output 3 of the
    reference's ONNX artefact, `:28`) -> perchSoftmax (`:315-335`: max-subtract, exp in float64, float32 running sum) ->
    label pairing -> top-10, with softmax + top-k on device.  `predict_with_embeddings` is the EmbeddingExtractor face
    (embedding = graph output 0, [1536])."""

    TOP_K = 10

    def __init__(self, classifier: HipClassifier, labels):
        if len(labels) != classifier.num_species():
            raise HipError(E_INVALID, f"label count {len(labels)} != model outputs {classifier.num_species()}")
        self.classifier, self.labels = classifier, list(labels)

    def predict(self, samples):
        conf, idx = self.classifier.predict_topk(np.asarray(samples, np.float32), 1, self.TOP_K, 1)
        return [(self.labels[i], float(c)) for c, i in zip(conf[0], idx[0])]

    def predict_batch(self, flat, batch_size):
        conf, idx = self.classifier.predict_topk(flat, batch_size, self.TOP_K, 1)
        return [[(self.labels[i], float(c)) for c, i in zip(cr, ir)] for cr, ir in zip(conf, idx)]

    def predict_with_embeddings(self, samples):
        return self.classifier.predict_with_embeddings(samples)


class BirdNET:
    """(*BirdNET).Predict (classifier/analyze.go:25-110): backend logits -> sigmoid(sensitivity) ->
    label pairing -> top-10, with the post-processing on device."""

    TOP_K = 10  # defaultTopKResults

    def __init__(self, classifier: HipClassifier, labels, sensitivity=1.0):
        if len(labels) != classifier.num_species():
            # validateModelAndLabels, classifier/birdnet.go:1248-1256
            raise HipError(E_INVALID, f"label count {len(labels)} != model outputs {classifier.num_species()}")
        self.classifier, self.labels, self.sensitivity = classifier, list(labels), float(sensitivity)

    def predict(self, samples):
        conf, idx = self.classifier.predict_topk(np.asarray(samples, np.float32), 1, self.TOP_K, 0, self.sensitivity)
        return [(self.labels[i], float(c)) for c, i in zip(conf[0], idx[0])]

    def predict_batch(self, flat, batch_size):
        conf, idx = self.classifier.predict_topk(flat, batch_size, self.TOP_K, 0, self.sensitivity)
        return [[(self.labels[i], float(c)) for c, i in zip(cr, ir)] for cr, ir in zip(conf, idx)]

    def predict_windows(self, win, bit_depth=16):
        """One tick for this model: every ready window of the assembler `win` (stream.NativeWindows) through one device call, the
        rows assembled under the device's work on the previous chunk (bnhip_windows_predict_topk).
        -> (source indices, uint8 rows view, per-window top-10 lists); a source index of -1 marks a row to skip."""
        idxs, rows, conf, idx = win.predict_topk(self.classifier, bit_depth, self.TOP_K, 0, self.sensitivity)
        return idxs, rows, [[(self.labels[i], float(c)) for c, i in zip(cr, ir)] for cr, ir in zip(conf, idx)]

    def predict_pcm_batch(self, raw, bit_depth, batch_size):
        """The windows' little-endian PCM bytes as captured (a1 runs in the kernel, process.go:479-497) -> per-window top-10."""
        conf, idx = self.classifier.predict_pcm_topk(raw, bit_depth, batch_size, self.TOP_K, 0, self.sensitivity)
        return [[(self.labels[i], float(c)) for c, i in zip(cr, ir)] for cr, ir in zip(conf, idx)]
