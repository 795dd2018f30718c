"""Real-time window path either side of the device call, host side.

  a3  `AnalysisBuffer`      internal/audiocore/buffer/analysis.go:30-273 (+ overwrite.go)     ring -> `overlap ‖ fresh` windows
  a4  `process_data`        internal/analysis/process.go:253-422 (+ :44-213 overrun tracker)   convert -> predict -> overrun -> enqueue
  a5  `Orchestrator`        internal/classifier/orchestrator.go:63-68,514-572                  lock protocol + counters
      `ModelSpec`           internal/classifier/model.go:33-56                                  window geometry (50 % overlap)
      `BufferMonitor.tick`  internal/analysis/buffer_manager.go:388-496                         one poll of one (source, model)

The reference runs one window at a time: every (source, model) pair has its own 100 ms poll loop and all of them queue on
`inferenceMu` for a batch-1 native call.  `WindowBatcher` is the same contract turned the way the GPU wants it: one tick reads
every buffer that has a window ready and hands all of them to ONE device call (PCM bytes go to the device as they are - the
16/24/32-bit conversion of a1 happens in the kernel), then builds one `Results` message per window.  Per-window semantics
(copy of the PCM bytes, overrun accounting against the model's buffer interval, drop-on-full queue) are the reference's.

Two holders of the rings: `AnalysisBuffer` below mirrors the reference's type one to one (its tests are the reference's
table); `NativeWindows` is the library's window assembler behind the C ABI (`bnhip_windows_*`, csrc/windows.cpp) - the form a
Go host binds - which reads every ready source into consecutive rows of one page-locked batch buffer, so that the device call
copies the windows from where they were assembled.  `WindowBatcher` uses the native one unless told otherwise; both are held
to the same oracle.

Byte work here is exact by construction (numpy slices); tests/test_stream.py replays the reference's own table of cases
(analysis_test.go:23-243) and compares against the line-by-line restatement in oracle/gostream.py.
"""
import ctypes as C
import threading
import time
from dataclasses import dataclass

import numpy as np

from . import host as _host
from . import results as _results

# analysis.go:13-18
OVERWRITE_WINDOW_S = 5 * 60.0
OVERWRITE_RATE_THRESHOLD = 10
OVERWRITE_MIN_WRITES = 50
OVERWRITE_NOTIFY_COOLDOWN_S = 3600.0
# process.go:34-40
OVERRUN_REPORT_COOLDOWN_S = 3600.0
OVERRUN_MIN_COUNT = 10
NUM_CHANNELS, BYTES_PER_SAMPLE = 1, 2        # conf.NumChannels, conf.BytesPerSample (16-bit capture)


class StreamError(ValueError):
    """Validation failure (the reference's errors.CategoryValidation)."""


def _as_bytes(data):
    """bytes-like or ndarray of any dtype / layout -> flat uint8 view (a copy only when the array is not contiguous)."""
    if isinstance(data, np.ndarray):
        return np.ascontiguousarray(data).reshape(-1).view(np.uint8)
    return np.frombuffer(data, np.uint8)


class ByteRing:
    """Byte ring in overwrite mode.  The reference uses github.com/smallnest/ringbuffer v0.1.1 (go.mod:31, absent from the
    snapshot) with `SetOverwrite(true)` (analysis.go:120): a write never fails, the oldest unread bytes are dropped when the
    data does not fit, and of a write longer than the ring only the last `capacity` bytes survive."""

    def __init__(self, capacity):
        self.buf = np.zeros(capacity, np.uint8)
        self.cap, self.r, self.n = capacity, 0, 0          # read position, unread byte count

    def length(self):
        return self.n

    def free(self):
        return self.cap - self.n

    def write(self, data):
        d = _as_bytes(data)
        total = d.size
        if total > self.cap:                                # only the newest `cap` bytes can survive
            drop = total - self.cap
            d = d[drop:]
            self.r, self.n = (self.r + self.n) % self.cap, 0   # everything unread is older than what is kept
        over = d.size - self.free()
        if over > 0:                                        # advance the read position past the bytes being overwritten
            self.r = (self.r + over) % self.cap
            self.n -= over
        w = (self.r + self.n) % self.cap
        first = min(d.size, self.cap - w)
        self.buf[w:w + first] = d[:first]
        self.buf[:d.size - first] = d[first:]
        self.n += d.size
        return total

    def read_into(self, out):
        """Fills `out` with up to len(out) unread bytes; returns the count (ring.Read)."""
        n = min(out.size, self.n)
        first = min(n, self.cap - self.r)
        out[:first] = self.buf[self.r:self.r + first]
        out[first:n] = self.buf[:n - first]
        self.r = (self.r + n) % self.cap
        self.n -= n
        return n

    def reset(self):
        self.r = self.n = 0


class OverwriteTracker:
    """overwrite.go: writes / overwrites within a sliding window, a warning when the rate passes the threshold."""

    def __init__(self, window_s=OVERWRITE_WINDOW_S, rate_threshold=OVERWRITE_RATE_THRESHOLD, min_writes=OVERWRITE_MIN_WRITES,
                 notify_cooldown_s=OVERWRITE_NOTIFY_COOLDOWN_S, on_warn=None, clock=time.monotonic):
        self.mu = threading.Lock()
        self.total_writes = self.overwrite_count = 0
        self.clock = clock
        self.window_start, self.last_notified = clock(), None
        self.window_s, self.rate_threshold, self.min_writes, self.cooldown = window_s, rate_threshold, min_writes, notify_cooldown_s
        self.on_warn = on_warn

    def record_write(self):
        with self.mu:
            if self.window_s > 0 and self.clock() - self.window_start > self.window_s:
                self.total_writes = self.overwrite_count = 0
                self.window_start = self.clock()
            self.total_writes += 1

    def record_overwrite(self):
        with self.mu:
            self.overwrite_count += 1

    def check_and_notify(self, source_id):
        with self.mu:
            if self.total_writes == 0 or self.total_writes < self.min_writes:
                return False
            rate = self.overwrite_count / self.total_writes * 100
            if int(rate) < self.rate_threshold:
                return False
            now = self.clock()
            if self.cooldown > 0 and self.last_notified is not None and now - self.last_notified < self.cooldown:
                return False
            self.last_notified = now
            writes, overwrites = self.total_writes, self.overwrite_count
        if self.on_warn:                                  # outside the lock: the callback may look at the tracker
            self.on_warn(source_id, rate, writes, overwrites)
        return True

    def overwrite_rate(self):
        with self.mu:
            return 0.0 if self.total_writes == 0 else self.overwrite_count / self.total_writes * 100

    def reset(self):
        with self.mu:
            self.total_writes = self.overwrite_count = 0
            self.window_start = self.clock()


class AnalysisBuffer:
    """analysis.go: each `read` returns `overlap_size` bytes kept from the end of the previous window followed by
    `read_size` fresh bytes; the very first window's prefix is zeros; `None` = try again later."""

    def __init__(self, capacity, overlap_size, read_size, source_id, on_warn=None):
        if capacity <= 0:
            raise StreamError(f"invalid analysis buffer capacity: {capacity}, must be greater than 0")
        if overlap_size < 0:
            raise StreamError(f"invalid overlap size: {overlap_size}, must be >= 0")
        if read_size <= 0:
            raise StreamError(f"invalid read size: {read_size}, must be greater than 0")
        if read_size < overlap_size:
            raise StreamError(f"read size {read_size} must be >= overlap size {overlap_size}")
        if capacity < read_size:
            raise StreamError(f"capacity {capacity} must be >= read size {read_size}")
        if not source_id:
            raise StreamError("source ID must not be empty")
        self.ring = ByteRing(capacity)
        self.prev = None
        self.overlap_size, self.read_size, self.window_size = overlap_size, read_size, overlap_size + read_size
        self.source_id = source_id
        self.tracker = OverwriteTracker(on_warn=on_warn)
        self.mu = threading.Lock()

    def write(self, data):
        d = _as_bytes(data)
        with self.mu:
            will_overwrite = d.size > self.ring.free()
            self.ring.write(d)
        self.tracker.record_write()
        if will_overwrite:
            self.tracker.record_overwrite()
        self.tracker.check_and_notify(self.source_id)

    def ready(self):
        with self.mu:
            return self.ring.length() >= self.read_size

    def read(self, out=None):
        """-> uint8[window_size] (a fresh array, or `out` filled), or None when fewer than read_size bytes are buffered."""
        with self.mu:
            if self.ring.length() < self.read_size:
                return None
            win = out if out is not None else np.empty(self.window_size, np.uint8)
            ov = self.overlap_size
            if ov > 0:
                if self.prev is not None and self.prev.size == ov:
                    win[:ov] = self.prev
                else:
                    win[:ov] = 0
            n = self.ring.read_into(win[ov:ov + self.read_size])
            if n < self.read_size:
                win[ov + n:] = 0
            if ov > 0:
                if self.prev is None:
                    self.prev = np.zeros(ov, np.uint8)
                fresh_end = ov + n
                if n >= ov:
                    self.prev[:] = win[fresh_end - ov:fresh_end]
                else:
                    self.prev[:] = 0
                    self.prev[ov - n:] = win[ov:fresh_end]
            return win

    def overwrite_count(self):
        with self.tracker.mu:
            return self.tracker.overwrite_count

    def reset(self):
        with self.mu:
            self.ring.reset()
            self.prev = None
        self.tracker.reset()


class NativeWindows:
    """`bnhip_windows` (include/bnhip.h): per-source rings + overlap tails in the library, every ready window collected into one
    batch buffer (page-locked when a device is present).  One assembler per window geometry, i.e. per model."""

    _proto_done = False

    def __init__(self, overlap_bytes, read_bytes, max_batch=256):
        lib = self._lib = _host.load_library()
        if not NativeWindows._proto_done:
            vp, sz, ci = C.c_void_p, C.c_size_t, C.c_int
            lib.bnhip_windows_create.argtypes = [sz, sz, ci, C.POINTER(vp)]
            lib.bnhip_windows_info.argtypes = [vp, C.POINTER(sz), C.POINTER(ci), C.POINTER(ci), C.POINTER(ci)]
            lib.bnhip_windows_add_source.argtypes = [vp, C.c_char_p, sz, C.POINTER(ci)]
            lib.bnhip_windows_remove_source.argtypes = [vp, ci]
            lib.bnhip_windows_write.argtypes = [vp, ci, vp, sz]
            lib.bnhip_windows_collect.argtypes = [vp, ci, C.POINTER(ci), C.POINTER(ci), C.POINTER(vp)]
            lib.bnhip_windows_ready.argtypes = [vp, C.POINTER(ci)]
            lib.bnhip_windows_stats.argtypes = [vp, ci, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(sz)]
            lib.bnhip_windows_reset.argtypes = [vp, ci]
            lib.bnhip_windows_destroy.argtypes = [vp]
            lib.bnhip_windows_predict_topk.argtypes = [vp, vp, ci, ci, C.c_double, ci, C.POINTER(ci), C.POINTER(ci), vp, vp,
                                                       C.POINTER(vp)]
            lib.bnhip_windows_destroy.restype = None
            NativeWindows._proto_done = True
        if overlap_bytes < 0 or read_bytes < 0:
            raise StreamError(f"invalid window geometry: overlap {overlap_bytes}, read {read_bytes}")
        h = C.c_void_p()
        self._h = None
        self._check(lib.bnhip_windows_create(overlap_bytes, read_bytes, max_batch, C.byref(h)))
        self._h = h
        wb, mb, pin = C.c_size_t(), C.c_int(), C.c_int()
        self._check(lib.bnhip_windows_info(h, C.byref(wb), C.byref(mb), C.byref(pin), None))
        self.window_bytes, self.max_batch, self.pinned = wb.value, mb.value, bool(pin.value)
        self.overlap_size, self.read_size = overlap_bytes, read_bytes
        self._sources = (C.c_int * self.max_batch)()
        self._rows = None                               # uint8 [max_batch, window_bytes] over the library's batch buffer

    def _check(self, rc):
        if rc != _host.BNHIP_OK:
            msg = (self._lib.bnhip_last_error() or b"").decode("utf-8", "replace")
            if rc == _host.E_INVALID:
                raise StreamError(msg)
            raise _host.HipError(rc, msg)

    def _alive(self):
        if self._h is None:
            raise StreamError("window assembler is closed")
        return self._h

    def add_source(self, source_id, capacity):
        idx = C.c_int(-1)
        self._check(self._lib.bnhip_windows_add_source(self._alive(), source_id.encode(), max(0, capacity), C.byref(idx)))
        return idx.value

    def remove_source(self, source):
        self._check(self._lib.bnhip_windows_remove_source(self._alive(), source))

    def write(self, source, data):
        d = _as_bytes(data)
        self._check(self._lib.bnhip_windows_write(self._alive(), source, d.ctypes.data, d.size))

    def collect(self, cap=None):
        """-> (source indices, uint8 [n, window_bytes] VIEW of the batch buffer, valid until the next collect)."""
        n, p = C.c_int(0), C.c_void_p()
        cap = self.max_batch if cap is None else min(cap, self.max_batch)
        self._check(self._lib.bnhip_windows_collect(self._alive(), cap, self._sources, C.byref(n), C.byref(p)))
        return list(self._sources[:n.value]), self._view(p)[:n.value]

    def _view(self, p):
        if self._rows is None:
            buf = (C.c_uint8 * (self.max_batch * self.window_bytes)).from_address(p.value)
            self._rows = np.frombuffer(buf, np.uint8).reshape(self.max_batch, self.window_bytes)
        return self._rows

    def predict_topk(self, clf, bit_depth=16, k=10, activation=0, sensitivity=1.0):
        """collect + device call in one (bnhip_windows_predict_topk): the rows of chunk c + 1 are assembled while chunk c is on
        the device.  clf: host.HipClassifier.  -> (source indices, rows view, conf [n, kk], idx [n, kk]); on a device error the
        exception carries `.sources` - those windows are consumed all the same."""
        clf._alive()
        kk = min(k, clf.num_species())
        if getattr(self, "_tk", None) is None or self._tk[0].shape[1] != kk:
            self._tk = (np.empty((self.max_batch, kk), np.float32), np.empty((self.max_batch, kk), np.int32))
        conf, idx = self._tk
        n, p = C.c_int(0), C.c_void_p()
        rc = self._lib.bnhip_windows_predict_topk(self._alive(), clf._h, bit_depth, activation, sensitivity, k, self._sources,
                                                  C.byref(n), conf.ctypes.data, idx.ctypes.data, C.byref(p))
        srcs = list(self._sources[:n.value])
        if rc != _host.BNHIP_OK:
            try:
                self._check(rc)
            except Exception as e:
                e.sources = srcs
                raise
        return srcs, self._view(p)[:n.value], conf[:n.value].copy(), idx[:n.value].copy()

    def ready(self):
        n = C.c_int(0)
        self._check(self._lib.bnhip_windows_ready(self._alive(), C.byref(n)))
        return n.value

    def stats(self, source):
        w, o, b = C.c_uint64(), C.c_uint64(), C.c_size_t()
        self._check(self._lib.bnhip_windows_stats(self._alive(), source, C.byref(w), C.byref(o), C.byref(b)))
        return w.value, o.value, b.value

    def reset(self, source):
        self._check(self._lib.bnhip_windows_reset(self._alive(), source))

    def close(self):
        if self._h is not None:
            self._rows = None
            self._lib.bnhip_windows_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _NativeSource:
    """One source of a NativeWindows with the AnalysisBuffer's surface (write / ready / overwrite_count / reset)."""

    def __init__(self, win: NativeWindows, index, source_id):
        self.win, self.index, self.source_id = win, index, source_id
        self.read_size, self.overlap_size, self.window_size = win.read_size, win.overlap_size, win.window_bytes

    def write(self, data):
        self.win.write(self.index, data)

    def ready(self):
        return self.win.stats(self.index)[2] >= self.read_size

    def overwrite_count(self):
        return self.win.stats(self.index)[1]

    def reset(self):
        self.win.reset(self.index)


@dataclass(frozen=True)
class ModelSpec:
    """model.go:33-56.  `sample_rate` is what the window is sized by; `raw_sample_rate` what the model is fed at."""
    sample_rate: int
    clip_length_s: float
    raw_sample_rate: int = 0
    clip_bytes: int = 0            # test geometries whose clip is not a whole number of seconds; 0 = the reference's formula

    def clip_size_bytes(self):
        if self.clip_bytes:
            return self.clip_bytes
        return self.sample_rate * int(self.clip_length_s) * NUM_CHANNELS * BYTES_PER_SAMPLE

    def buffer_dimensions(self):
        clip = self.clip_size_bytes()
        overlap = clip // 2
        return clip, overlap, clip - overlap

    def buffer_interval_s(self):
        return self.clip_length_s / 2

    def effective_sample_rate(self):
        return self.raw_sample_rate if self.raw_sample_rate > 0 else self.sample_rate


class OverrunTrackers:
    """process.go:44-213: per (source, model) tumbling one-hour window of "inference took longer than the buffer interval"."""

    class _T:
        __slots__ = ("mu", "source", "model_id", "count", "window_start", "max_elapsed", "buffer_len")

        def __init__(self, source, model_id):
            self.mu = threading.Lock()
            self.source, self.model_id = source, model_id
            self.count, self.window_start, self.max_elapsed, self.buffer_len = 0, None, 0.0, 0.0

    def __init__(self, on_report=None, clock=time.monotonic):
        self.mu = threading.Lock()
        self.m = {}
        self.on_report, self.clock = on_report, clock

    def get(self, source, model_id):
        key = source + ":" + model_id
        with self.mu:
            t = self.m.get(key)
            if t is None:
                t = self.m[key] = self._T(source, model_id)
            return t

    def record(self, source, model_id, elapsed_s, buffer_len_s):
        t = self.get(source, model_id)
        with t.mu:
            now = self.clock()
            if t.window_start is None:
                t.window_start = now
            if now - t.window_start >= OVERRUN_REPORT_COOLDOWN_S:
                if t.count >= OVERRUN_MIN_COUNT and self.on_report:
                    self.on_report(t.source, t.model_id, t.count, t.max_elapsed, t.buffer_len, now - t.window_start)
                t.count, t.max_elapsed, t.window_start = 0, 0.0, now
            t.count += 1
            if elapsed_s > t.max_elapsed:
                t.max_elapsed, t.buffer_len = elapsed_s, buffer_len_s

    def count(self, source, model_id):
        t = self.get(source, model_id)
        with t.mu:
            return t.count

    def remove_source(self, source):
        with self.mu:
            for k in [k for k in self.m if k.startswith(source + ":")]:
                del self.m[k]

    def cleanup(self, max_age_s):
        now = self.clock()
        with self.mu:
            for k, t in list(self.m.items()):
                with t.mu:
                    idle = t.window_start is not None and now - t.window_start > max_age_s
                    nothing = t.window_start is None and t.count == 0
                if idle or nothing:
                    del self.m[k]

    def reset(self):
        with self.mu:
            self.m.clear()


class OrchestratorError(RuntimeError):
    pass


class Orchestrator:
    """orchestrator.go:63-68,514-572.  Lock order: models map -> `inference_mu` (one native call at a time, whatever the model)
    -> the entry's own lock (instance lifecycle).  The map lock is dropped before the other two are taken so that unloading a
    model never waits behind an inference.  An instance is anything with `predict_batch(flat, n)` (host.BirdNET / Perch)."""

    class _Entry:
        def __init__(self, instance, spec):
            self.mu = threading.Lock()
            self.instance, self.spec, self.active = instance, spec, True

    def __init__(self, counters: _results.CounterMap = None):
        self.mu = threading.Lock()
        self.inference_mu = threading.Lock()
        self.models = {}
        self.counters = counters if counters is not None else _results.CounterMap()

    def register(self, model_id, instance, spec: ModelSpec):
        with self.mu:
            self.models[model_id] = self._Entry(instance, spec)

    def unload(self, model_id):
        """Takes the map lock and the entry lock but NOT `inference_mu` (orchestrator.go:66)."""
        with self.mu:
            e = self.models.pop(model_id, None)
        if e is not None:
            with e.mu:
                inst, e.instance = e.instance, None
            if inst is not None and hasattr(inst, "close"):
                inst.close()
            self.counters.delete(model_id)

    def model_spec_for(self, model_id):
        with self.mu:
            e = self.models.get(model_id)
        return e.spec if e is not None else None

    def set_active(self, model_id, on):
        with self.mu:
            e = self.models.get(model_id)
        if e is not None:
            e.active = bool(on)

    def is_model_active(self, model_id):
        with self.mu:
            e = self.models.get(model_id)
        return e is not None and e.active

    def predict_model(self, model_id, call):
        """`call(instance)` runs under the three locks; its duration goes to the per-model counters, an exception to the
        error counter (PredictModel :553-569)."""
        with self.mu:
            e = self.models.get(model_id)
        if e is None:
            raise OrchestratorError(f"unknown model: {model_id}")
        with self.inference_mu:
            with e.mu:
                if e.instance is None:
                    raise OrchestratorError(f"model {model_id} has been closed")
                t0 = time.perf_counter()
                try:
                    out = call(e.instance)
                except Exception:
                    self.counters.record_error(model_id)
                    raise
                self.counters.record_invoke(model_id, (time.perf_counter() - t0) * 1e6)
                return out


def _pcm_windows_predict(instance, windows, bit_depth):
    """windows uint8 [n, window_bytes] -> per-window top-K lists.  The bytes go to the device as they are when the instance
    can take them (`predict_pcm_batch`); otherwise they are widened here with the reference's own arithmetic (a1)."""
    n = windows.shape[0]
    if hasattr(instance, "predict_pcm_batch"):
        return instance.predict_pcm_batch(windows.reshape(-1), bit_depth, n)
    if bit_depth == 16:
        x = windows.reshape(-1).view("<i2").astype(np.float32) / np.float32(32768.0)
    elif bit_depth == 32:
        x = windows.reshape(-1).view("<i4").astype(np.float32) / np.float32(2147483648.0)
    elif bit_depth == 24:
        b = windows.reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v & 0x800000, v - (1 << 24), v)
        x = v.astype(np.float32) / np.float32(8388608.0)
    else:
        raise StreamError(f"unsupported audio bit depth: {bit_depth}")
    return instance.predict_batch(x, n)


def process_windows(orch: Orchestrator, windows, start_times, captured_at, sources, model_id, queue: _results.ResultsQueue,
                    overruns: OverrunTrackers, bit_depth=16, threshold=0.0):
    """`ProcessData` for n windows of one model at once.  Returns the number of messages enqueued.
    Elapsed time, the overrun check and the queue message are per window as in process.go:327-420; the one difference is
    that the n windows share one device call, so each window's elapsed time is the whole call's (what a consumer waiting
    for that window actually saw)."""
    windows = np.ascontiguousarray(windows, np.uint8)
    if windows.ndim == 1:
        windows = windows[None]
    t0 = time.perf_counter()
    lists = orch.predict_model(model_id, lambda inst: _pcm_windows_predict(inst, windows, bit_depth))
    elapsed = time.perf_counter() - t0
    return _emit_results(orch, windows, lists, elapsed, start_times, captured_at, sources, model_id, queue, overruns, threshold)


def _emit_results(orch, windows, lists, elapsed, start_times, captured_at, sources, model_id, queue, overruns, threshold=0.0):
    """ProcessData's tail per window (process.go:327-420): overrun check, PCM copy, enqueue-or-drop.  sources[i] None = skip."""
    spec = orch.model_spec_for(model_id)
    interval = spec.buffer_interval_s() if spec is not None else 1.5       # fallback: BirdNET v2.4 (process.go:353)
    sent = 0
    for i in range(windows.shape[0]):
        if sources[i] is None:
            continue
        if elapsed > interval:
            overruns.record(sources[i], model_id, elapsed, interval)
        dets = [_results.Detection(lbl, conf) for lbl, conf in lists[i] if conf >= threshold]
        msg = _results.Results(start_time=float(start_times[i]), audio_captured_at=float(captured_at[i]),
                               pcm_data=windows[i].tobytes(),            # independent copy: the window goes back to its pool
                               results=dets, elapsed_time=elapsed, source=sources[i], model_id=model_id)
        sent += queue.offer(msg)
    return sent


def process_data(orch, data, start_time, captured_at, source, model_id, queue, overruns, bit_depth=16):
    """The reference's signature: one window."""
    return process_windows(orch, np.frombuffer(data, np.uint8)[None], [start_time], [captured_at], [source], model_id, queue,
                           overruns, bit_depth)


class WindowBatcher:
    """All (source, model) monitors of buffer_manager.go:388-496 in one object.  `tick()` is one poll: every buffer with a
    window ready is read, windows of the same model form one batch (at most `max_batch` per device call), inactive models
    are read and discarded exactly as the reference does (:478-480: the audio is consumed, not analysed)."""

    def __init__(self, orch: Orchestrator, queue: _results.ResultsQueue = None, overruns: OverrunTrackers = None,
                 max_batch=256, pre_capture_s=0.0, bit_depth=16, clock=time.time, on_error=None, native=True):
        self.orch = orch
        self.queue = queue if queue is not None else _results.ResultsQueue()
        self.overruns = overruns if overruns is not None else OverrunTrackers()
        self.max_batch, self.pre_capture_s, self.bit_depth, self.clock = max_batch, pre_capture_s, bit_depth, clock
        self.buffers = {}                    # (source, model_id) -> AnalysisBuffer | _NativeSource
        self.native = native                 # rings in the library, windows collected into its page-locked batch buffer
        self.assemblers = {}                 # native: model_id -> NativeWindows
        self.mu = threading.Lock()
        self.on_error, self.errors = on_error, 0   # a failed device call costs its own windows only (the reference logs and polls on)

    def allocate(self, source, model_id, capacity=None):
        spec = self.orch.model_spec_for(model_id)
        if spec is None:
            raise OrchestratorError(f"unknown model: {model_id}")
        clip, overlap, read = spec.buffer_dimensions()
        capacity = capacity if capacity is not None else 2 * clip
        if not self.native:
            ab = AnalysisBuffer(capacity, overlap, read, source)
            with self.mu:
                self.buffers[(source, model_id)] = ab
            return ab
        with self.mu:
            win = self.assemblers.get(model_id)
            if win is None or (win.overlap_size, win.read_size) != (overlap, read):
                if win is not None:                       # the model was re-registered with another geometry
                    for k in [k for k in self.buffers if k[1] == model_id]:
                        del self.buffers[k]
                    win.close()
                win = self.assemblers[model_id] = NativeWindows(overlap, read, self.max_batch)
            old = self.buffers.pop((source, model_id), None)
            if old is not None:
                win.remove_source(old.index)
            ab = self.buffers[(source, model_id)] = _NativeSource(win, win.add_source(source, capacity), source)
        return ab

    def remove(self, source, model_id=None):
        with self.mu:
            for k in [k for k in self.buffers if k[0] == source and (model_id is None or k[1] == model_id)]:
                ab = self.buffers.pop(k)
                if isinstance(ab, _NativeSource):
                    ab.win.remove_source(ab.index)
        self.overruns.remove_source(source)

    def write(self, source, data):
        """Capture side: the same bytes go to every model's buffer of that source."""
        with self.mu:
            abs_ = [ab for (s, _), ab in self.buffers.items() if s == source]
        for ab in abs_:
            ab.write(data)

    def _tick_native(self):
        with self.mu:
            wins = list(self.assemblers.items())
            names = {(id(ab.win), ab.index): src for (src, _), ab in self.buffers.items() if isinstance(ab, _NativeSource)}
        sent = 0
        def name(win, i):
            return None if i < 0 else names.get((id(win), i), f"source#{i}")

        def one_tick(inst, win):
            # an instance that can run the whole tick in the library does (rows assembled under the device's work on the
            # previous chunk); anything else gets the collected rows
            if hasattr(inst, "predict_windows"):
                return inst.predict_windows(win, self.bit_depth)
            idxs, rows = win.collect(self.max_batch)
            try:
                return idxs, rows, (_pcm_windows_predict(inst, rows, self.bit_depth) if idxs else [])
            except Exception as e:
                e.sources = idxs                              # (consumed all the same)
                raise

        for model_id, win in wins:
            while True:
                spec = self.orch.model_spec_for(model_id)
                if spec is None or not self.orch.is_model_active(model_id):   # consumed, not analysed (buffer_manager.go:478-481)
                    idxs, _ = win.collect(self.max_batch)
                    if len(idxs) < min(self.max_batch, win.max_batch):
                        break
                    continue
                if not win.ready():                               # nothing to do: no inference lock, no invoke counted
                    break
                now = self.clock()
                start = now - (self.pre_capture_s + spec.clip_length_s)
                t0 = time.perf_counter()
                try:
                    idxs, rows, lists = self.orch.predict_model(model_id, lambda inst: one_tick(inst, win))
                except Exception as e:
                    lost = [name(win, i) for i in getattr(e, "sources", [])]
                    self.errors += max(1, len(lost))
                    if self.on_error:
                        self.on_error(model_id, lost, e)
                    break
                if not idxs:
                    break
                sources = [name(win, i) for i in idxs]
                sent += _emit_results(self.orch, rows, lists, time.perf_counter() - t0, [start] * len(idxs), [now] * len(idxs), sources,
                                      model_id, self.queue, self.overruns)
                if len(idxs) < min(self.max_batch, win.max_batch):
                    break
        return sent

    def close(self):
        with self.mu:
            for win in self.assemblers.values():
                win.close()
            self.assemblers.clear()
            self.buffers.clear()

    def tick(self):
        if self.native:
            return self._tick_native()
        with self.mu:
            items = list(self.buffers.items())
        per_model = {}
        for (source, model_id), ab in items:
            win = ab.read()
            if win is None:
                continue
            spec = self.orch.model_spec_for(model_id)
            if spec is None or not self.orch.is_model_active(model_id):       # (unloaded since the last tick: consumed, not analysed)
                continue
            now = self.clock()
            start = now - (self.pre_capture_s + spec.clip_length_s)       # beginTimeOffset, buffer_manager.go:490-491
            per_model.setdefault(model_id, []).append((source, win, start, now))
        sent = 0
        for model_id, ws in per_model.items():
            for lo in range(0, len(ws), self.max_batch):
                part = ws[lo:lo + self.max_batch]
                try:
                    sent += process_windows(self.orch, np.stack([w for _, w, _, _ in part]), [s for _, _, s, _ in part],
                                            [c for _, _, _, c in part], [src for src, _, _, _ in part], model_id, self.queue,
                                            self.overruns, self.bit_depth)
                except Exception as e:                    # buffer_manager.go:494-499: the monitor logs the error and keeps polling
                    self.errors += len(part)
                    if self.on_error:
                        self.on_error(model_id, [src for src, _, _, _ in part], e)
        return sent
