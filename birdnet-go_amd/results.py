"""Results hand-off: what happens to a batch of device top-K lists after the hot path.

Host-side mirror of the reference's contract, batched:
  * `Results` message + bounded `ResultsQueue` with drop-on-full accounting
    (`internal/classifier/queue.go:10-28`, enqueue/drop in `internal/analysis/process.go:391-420`);
  * per-model inference counters (`internal/classifier/inferencestats/counters.go:173-246`): count, total and
    windowed-max latency (reset on read), lifetime max, rolling p95 over the last 1024 invocations, error count;
  * the timed dispatch wrapper of `Orchestrator.PredictModel` (`internal/classifier/orchestrator.go:553-569`).
One `predict_topk` call on the device serves a whole batch of clips; this module turns its two arrays into the
per-clip messages the Go consumer expects without per-clip device work.
"""
import math
import queue
import threading
import time
from dataclasses import dataclass, field

import numpy as np

DEFAULT_QUEUE_SIZE = 100          # queue.go: DefaultQueueSize
LATENCY_WINDOW = 1024             # counters.go: latencyWindowSize
HEALTH_PERCENTILE = 0.95          # counters.go: healthLatencyPercentile


@dataclass
class Detection:                  # datastore.Results as far as the hot path fills it
    species: str
    confidence: float


@dataclass
class Results:                    # classifier.Results (queue.go:10-19)
    start_time: float
    audio_captured_at: float
    pcm_data: bytes
    results: list
    elapsed_time: float
    source: str
    model_id: str
    clip_name: str = ""


class ResultsQueue:
    """Bounded hand-off queue.  Ownership moves to the receiver on put (queue.go:24-28: the sender must not touch the
    message afterwards); a full queue drops the message and counts it (process.go:405-419) instead of blocking the
    analysis loop."""

    def __init__(self, size=DEFAULT_QUEUE_SIZE):
        self._q = queue.Queue(maxsize=size)
        self._drops = {}
        self._lock = threading.Lock()

    def offer(self, msg: Results) -> bool:
        try:
            self._q.put_nowait(msg)
            return True
        except queue.Full:
            with self._lock:
                key = (msg.source, msg.model_id)
                self._drops[key] = self._drops.get(key, 0) + 1
            return False

    def get(self, timeout=None) -> Results:
        return self._q.get(timeout=timeout)

    def qsize(self):
        return self._q.qsize()

    def drops(self):
        with self._lock:
            return dict(self._drops)


class _Counters:
    def __init__(self):
        self.count = self.total_us = self.max_us = self.max_us_lifetime = self.errors = 0
        self.ring = [0] * LATENCY_WINDOW
        self.pos = self.len = 0
        self.lock = threading.Lock()

    def record_invoke(self, us):
        with self.lock:
            self.count += 1
            self.total_us += us
            self.max_us = max(self.max_us, us)
            self.max_us_lifetime = max(self.max_us_lifetime, us)
            self.ring[self.pos] = us
            self.pos = (self.pos + 1) % LATENCY_WINDOW
            self.len = min(self.len + 1, LATENCY_WINDOW)

    def recent_percentile(self, p):
        with self.lock:
            n = self.len
            samples = sorted(self.ring[:n])
        if n == 0:
            return 0
        idx = min(max(int(math.ceil(p * n)) - 1, 0), n - 1)     # counters.go recentPercentileUs
        return samples[idx]


class CounterMap:
    """Per-model inference counters (counters.go:173-246)."""

    def __init__(self):
        self._m = {}
        self._lock = threading.Lock()

    def _get(self, model_id):
        with self._lock:
            c = self._m.get(model_id)
            if c is None:
                c = self._m[model_id] = _Counters()
            return c

    def record_invoke(self, model_id, duration_us):
        self._get(model_id).record_invoke(int(duration_us))

    def record_error(self, model_id):
        c = self._get(model_id)
        with c.lock:
            c.errors += 1

    def snapshot_all(self):
        """Windowed max is reset on read (Counters.Snapshot: InvokeMaxUs.Swap(0))."""
        out = {}
        with self._lock:
            items = list(self._m.items())
        for k, c in items:
            with c.lock:
                out[k] = dict(invoke_count=c.count, invoke_total_us=c.total_us, invoke_max_us=c.max_us,
                              invoke_errors=c.errors, collected_at=time.time())
                c.max_us = 0
        return out

    def peek_all(self):
        """Never resets anything; reports the lifetime max and the rolling p95 (PeekAll)."""
        out = {}
        with self._lock:
            items = list(self._m.items())
        for k, c in items:
            p95 = c.recent_percentile(HEALTH_PERCENTILE)
            with c.lock:
                out[k] = dict(invoke_count=c.count, invoke_total_us=c.total_us, invoke_max_us_lifetime=c.max_us_lifetime,
                              recent_p95_us=p95, invoke_errors=c.errors)
        return out

    def delete(self, model_id):
        with self._lock:
            self._m.pop(model_id, None)


def sanitize_model_id(model_id: str) -> str:      # counters.go SanitizeModelID
    return "".join(ch if (ch.isascii() and (ch.isalnum() or ch == "_")) else "_" for ch in model_id)


def metric_key(model_id: str) -> str:
    return "inference." + sanitize_model_id(model_id) + ".avg_ms"


class BatchDispatcher:
    """PredictModel for a batch: one timed device call, counters updated per invocation, one `Results` message per clip
    offered to the queue.  `birdnet` is a `host.BirdNET` (device sigmoid + top-K)."""

    def __init__(self, birdnet, model_id, results_queue: ResultsQueue = None, counters: CounterMap = None,
                 confidence_threshold=0.0):
        self.birdnet, self.model_id = birdnet, model_id
        self.queue = results_queue if results_queue is not None else ResultsQueue()
        self.counters = counters if counters is not None else CounterMap()
        self.threshold = float(confidence_threshold)

    def dispatch(self, clips, start_times, source, pcm_chunks=None, captured_at=None):
        """clips [n, n_samples] float32; start_times [n] seconds.  Returns the number of messages enqueued."""
        clips = np.ascontiguousarray(clips, np.float32)
        n = clips.shape[0]
        t0 = time.perf_counter()
        try:
            lists = self.birdnet.predict_batch(clips.reshape(-1), n)
        except Exception:
            self.counters.record_error(self.model_id)
            raise
        elapsed = time.perf_counter() - t0
        self.counters.record_invoke(self.model_id, elapsed * 1e6)
        sent = 0
        now = time.time()
        for i in range(n):
            dets = [Detection(lbl, conf) for lbl, conf in lists[i] if conf >= self.threshold]
            msg = Results(start_time=float(start_times[i]), audio_captured_at=float(captured_at[i]) if captured_at is not None else now,
                          pcm_data=bytes(pcm_chunks[i]) if pcm_chunks is not None else b"", results=dets,
                          elapsed_time=elapsed / n, source=source, model_id=self.model_id)
            sent += self.queue.offer(msg)
        return sent
