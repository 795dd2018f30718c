"""analysis-window assembly, ProcessData, Orchestrator.

The first block replays the reference's own cases (internal/audiocore/buffer/analysis_test.go:23-243) against
birdnet_go_amd.stream, against the library's window assembler behind the C ABI (bnhip_windows_*, one source) AND against the
byte-at-a-time oracle (oracle/gostream.py), so the oracle is pinned by the same table.
"""
import threading
import time

import numpy as np
import pytest

import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import results as R
from birdnet_go_amd import stream as S
from oracle.gostream import GoAnalysisBuffer


# ------------------------------------------------------------------ a3: the reference's table, both implementations
class _OracleAB:
    """oracle/gostream.py behind the product's method names."""

    def __init__(self, capacity, overlap, read, source="s"):
        self.g = GoAnalysisBuffer(capacity, overlap, read)

    def write(self, data):
        self.g.Write(bytes(data))

    def read(self):
        w = self.g.Read()
        return None if w is None else np.frombuffer(w, np.uint8)

    def overwrite_count(self):
        return self.g.overwrites

    def reset(self):
        self.g.Reset()


def _product(capacity, overlap, read, source="test-source"):
    return S.AnalysisBuffer(capacity, overlap, read, source)


class _NativeAB:
    """One source of a `bnhip_windows` assembler behind the product's method names."""

    def __init__(self, capacity, overlap, read, source="test-source"):
        self.w = S.NativeWindows(overlap, read, max_batch=4)
        self.i = self.w.add_source(source, capacity)

    def write(self, data):
        self.w.write(self.i, data)

    def read(self):
        idxs, rows = self.w.collect()
        assert idxs in ([], [self.i])
        return rows[0].copy() if idxs else None

    def overwrite_count(self):
        return self.w.stats(self.i)[1]

    def reset(self):
        self.w.reset(self.i)


IMPLS = [pytest.param(_product, id="product"), pytest.param(_NativeAB, id="native"), pytest.param(_OracleAB, id="oracle")]


@pytest.mark.parametrize("make", IMPLS)
def test_write_read(make):                                   # TestAnalysisBuffer_WriteRead :23-61
    ab = make(64 * 1024, 512, 1024)
    payload = bytes(i % 256 for i in range(1024))
    ab.write(payload)
    got = None
    for _ in range(10):
        got = ab.read()
        if got is not None:
            break
        ab.write(payload)
    assert got is not None and len(got) == 512 + 1024


@pytest.mark.parametrize("make", IMPLS)
def test_overwrite_recorded(make):                           # TestAnalysisBuffer_Overwrite :64-93
    ab = make(4 * 1024, 128, 256)
    chunk = bytes(i % 256 for i in range(1024))
    seen = False
    for _ in range(20):
        ab.write(chunk)
        if ab.overwrite_count() > 0:
            seen = True
            break
    assert seen


def test_read_size_less_than_overlap_rejected():             # TestAnalysisBuffer_ReadSizeLessThanOverlapSize :97-103
    with pytest.raises(S.StreamError, match="read size 512 must be >= overlap size 1024"):
        S.AnalysisBuffer(4096, 1024, 512, "test-source")


def test_constructor_validation_messages():                  # analysis.go:55-108
    for args, msg in [((0, 0, 1, "s"), "invalid analysis buffer capacity: 0"), ((8, -1, 1, "s"), "invalid overlap size: -1"),
                      ((8, 0, 0, "s"), "invalid read size: 0"), ((8, 0, 16, "s"), "capacity 8 must be >= read size 16"),
                      ((8, 0, 4, ""), "source ID must not be empty")]:
        with pytest.raises(S.StreamError, match=msg):
            S.AnalysisBuffer(*args)


@pytest.mark.parametrize("make", IMPLS)
def test_overlap_read(make):                                 # TestAnalysisBuffer_OverlapRead :107-165
    ov, rd = 512, 1024
    ab = make(64 * 1024, ov, rd)
    first = None
    for _ in range(20):
        ab.write(bytes([0xAA]) * rd)
        first = ab.read()
        if first is not None:
            break
    assert first is not None and len(first) == ov + rd
    first = bytes(first)
    for _ in range(5):
        ab.write(bytes([0xBB]) * rd)
        second = ab.read()
        if second is not None:
            assert bytes(second[:ov]) == first[rd:]
            return
    pytest.fail("second read did not return data in time")


def test_native_validation_messages():                       # analysis.go:55-108 through the C ABI
    with pytest.raises(S.StreamError, match="read size must be >= overlap size"):
        S.NativeWindows(1024, 512)
    with pytest.raises(S.StreamError, match="invalid read size: 0"):
        S.NativeWindows(0, 0)
    with pytest.raises(S.StreamError, match="max_batch must be positive"):
        S.NativeWindows(0, 8, max_batch=0)
    w = S.NativeWindows(4, 16, max_batch=2)
    with pytest.raises(S.StreamError, match="capacity must be >= read size"):
        w.add_source("mic", 8)
    with pytest.raises(S.StreamError, match="invalid analysis buffer capacity: 0"):
        w.add_source("mic", 0)
    with pytest.raises(S.StreamError, match="source ID must not be empty"):
        w.add_source("", 64)
    with pytest.raises(S.StreamError, match="no such source"):
        w.write(3, b"abcd")
    i = w.add_source("mic", 64)
    w.remove_source(i)
    with pytest.raises(S.StreamError, match="no such source"):
        w.write(i, b"abcd")
    assert w.add_source("mic2", 64) == i                     # the slot is reused
    w.close()
    with pytest.raises(S.StreamError, match="closed"):
        w.ready()


def test_overwrite_tracker_rate():                           # TestOverwriteTracker_RateCalculation :169-197
    t = S.OverwriteTracker(window_s=300.0, rate_threshold=10, min_writes=50, notify_cooldown_s=3600.0)
    for _ in range(80):
        t.record_write()
    for _ in range(20):
        t.record_write()
        t.record_overwrite()
    assert abs(t.overwrite_rate() - 20.0) < 0.01
    t.reset()
    assert abs(t.overwrite_rate()) < 0.01


@pytest.mark.parametrize("make", IMPLS)
def test_read_content_parity(make):                          # TestAnalysisBuffer_Read_ContentParity :203-243
    ov, rd = 32, 128
    ab = make(4096, ov, rd)
    stream = bytes(i & 0xFF for i in range(rd * 4))
    ab.write(stream)
    w1 = bytes(ab.read())
    assert len(w1) == ov + rd and w1[:ov] == bytes(ov) and w1[ov:] == stream[:rd]
    w2 = bytes(ab.read())
    assert w2[:ov] == stream[rd - ov:rd] and w2[ov:] == stream[rd:2 * rd]


def test_try_again_later_is_none():                          # TestAnalysisBuffer_Read_TryAgainLaterReleaseIsNoop :277-296
    ab = S.AnalysisBuffer(4096, 32, 128, "s")
    ab.write(bytes(100))
    assert ab.read() is None and not ab.ready()
    ab.write(bytes(28))
    assert ab.ready() and ab.read() is not None


def test_read_into_caller_buffer_overwrites_every_byte():
    """The pooled-window path (analysis.go:196-200): a recycled slice holds stale bytes, so Read must set all of them."""
    ab = S.AnalysisBuffer(4096, 16, 64, "s")
    ab.write(bytes(range(64)))
    out = np.full(80, 0xEE, np.uint8)
    assert ab.read(out) is out
    assert bytes(out[:16]) == bytes(16) and bytes(out[16:]) == bytes(range(64))


# ------------------------------------------------------------------ a3: product == oracle on random traffic
@pytest.mark.parametrize("seed", range(6))
def test_product_equals_oracle_on_random_traffic(seed):
    rng = np.random.default_rng(seed)
    ov = int(rng.integers(0, 40))
    rd = int(rng.integers(max(ov, 1), 90))
    cap = int(rng.integers(rd, 4 * rd + 7))
    p, o = S.AnalysisBuffer(cap, ov, rd, "s"), GoAnalysisBuffer(cap, ov, rd)
    n_windows = 0
    for step in range(400):
        op = rng.random()
        if op < 0.55:
            n = int(rng.integers(0, 2 * cap + 3)) if rng.random() < 0.1 else int(rng.integers(0, rd + 5))
            data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            p.write(data)
            o.Write(data)
        elif op < 0.97:
            a, b = p.read(), o.Read()
            assert (a is None) == (b is None), step
            if a is not None:
                assert bytes(a) == b, step
                n_windows += 1
        else:
            p.reset()
            o.Reset()
        assert p.ring.length() == o.ring.Length()
    assert p.overwrite_count() <= o.overwrites               # a tracker reset (Reset()) zeroes the product's count only
    assert n_windows > 20


@pytest.mark.parametrize("seed", range(6))
def test_native_assembler_equals_oracle_on_random_traffic(seed):
    """Several sources behind one `bnhip_windows`: random writes (some longer than the ring), collects with random caps,
    resets, a source removed and another added mid-stream.  Every collected row must be the window the byte-at-a-time oracle
    of that source returns next, sources served round-robin from behind the last one looked at, nothing ready left behind
    when the cap was not hit."""
    rng = np.random.default_rng(100 + seed)
    ov = int(rng.integers(0, 40))
    rd = int(rng.integers(max(ov, 1), 90))
    nsrc = int(rng.integers(2, 7))
    mb = int(rng.integers(1, nsrc + 2))
    w = S.NativeWindows(ov, rd, max_batch=mb)
    assert w.window_bytes == ov + rd and w.max_batch == mb
    caps = [int(rng.integers(rd, 4 * rd + 7)) for _ in range(nsrc)]
    idx = [w.add_source(f"s{k}", caps[k]) for k in range(nsrc)]
    assert idx == list(range(nsrc))
    ora = {i: GoAnalysisBuffer(caps[i], ov, rd) for i in idx}
    n_windows, nxt = 0, 0
    for step in range(600):
        op = rng.random()
        live = sorted(ora)
        if op < 0.6:
            i = int(rng.choice(live))
            cap = caps[i]
            n = int(rng.integers(0, 2 * cap + 3)) if rng.random() < 0.1 else int(rng.integers(0, rd + 5))
            data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            w.write(i, data)
            ora[i].Write(data)
        elif op < 0.93:
            cap = int(rng.integers(1, mb + 2))
            n_ready = sum(o.ring.Length() >= rd for o in ora.values())
            assert w.ready() == n_ready
            got, rows = w.collect(cap)
            assert len(got) == min(n_ready, cap, mb) and len(set(got)) == len(got), step
            # round-robin: the table is walked from `nxt`, wrapping once
            nslots = len(caps)
            order = [(nxt + k) % nslots for k in range(nslots)]
            want = [i for i in order if i in ora and ora[i].ring.Length() >= rd][:min(cap, mb)]
            assert got == want, step
            if len(got) == min(cap, mb):                     # stopped at the cap: the next pass starts behind the last source
                nxt = (nxt + order.index(got[-1]) + 1) % nslots      # served (otherwise the whole table was looked at)
            for k, i in enumerate(got):
                assert bytes(rows[k]) == ora[i].Read(), (step, i)
                n_windows += 1
        elif op < 0.96:
            i = int(rng.choice(live))
            w.reset(i)
            ora[i].Reset()
        elif op < 0.98 and len(live) > 1:
            i = int(rng.choice(live))
            w.remove_source(i)
            del ora[i]
        else:
            c = int(rng.integers(rd, 4 * rd + 7))
            i = w.add_source(f"n{step}", c)
            assert i not in ora
            if i == len(caps):
                caps.append(c)
            else:
                caps[i] = c
            ora[i] = GoAnalysisBuffer(c, ov, rd)
        for i, o in ora.items():
            assert w.stats(i)[2] == o.ring.Length()
    assert n_windows > 30
    w.close()


def test_native_assembler_concurrent_writers_lose_nothing():
    """Capture threads write while the consumer collects (analysis.go: "Write is safe for concurrent use"): rings large enough
    never to overwrite, so the fresh halves of each source's windows, concatenated, must be exactly what its thread wrote."""
    ov, rd, nsrc = 16, 64, 4
    w = S.NativeWindows(ov, rd, max_batch=3)
    total = 64 * 200
    idx = [w.add_source(f"t{k}", total) for k in range(nsrc)]
    data = [np.random.default_rng(k).integers(0, 256, total, dtype=np.uint8).tobytes() for k in range(nsrc)]

    def writer(k):
        rng = np.random.default_rng(50 + k)
        pos = 0
        while pos < total:
            n = int(rng.integers(1, 200))
            w.write(idx[k], data[k][pos:pos + n])
            pos += n

    ts = [threading.Thread(target=writer, args=(k,)) for k in range(nsrc)]
    for t in ts:
        t.start()
    fresh = {i: bytearray() for i in idx}
    tails = {i: bytes(ov) for i in idx}
    while any(t.is_alive() for t in ts) or w.ready():
        got, rows = w.collect()
        for k, i in enumerate(got):
            row = bytes(rows[k])
            assert row[:ov] == tails[i]
            fresh[i] += row[ov:]
            tails[i] = row[-ov:]
    for t in ts:
        t.join()
    for k, i in enumerate(idx):
        assert bytes(fresh[i]) == data[k] and w.stats(i)[1] == 0
    w.close()


def test_native_assembler_large_windows_take_the_row_pool():
    """Batches of >= 4 MB are filled by the library's helper threads, one row each (csrc/windows.cpp RowPool): two assemblers
    (two models) collecting at the same time from two threads, several ticks, every row == the hand-cut window
    `previous tail || fresh bytes`."""
    ov = rd = 96 * 1024
    nsrc, ticks = 24, 3                                       # 24 x 192 KB = 4.5 MB per collect
    errs = []

    def one(seed):
        try:
            rng = np.random.default_rng(seed)
            w = S.NativeWindows(ov, rd, max_batch=32)
            streams = rng.integers(0, 256, (nsrc, ticks * rd), dtype=np.uint8)
            ids = [w.add_source(f"s{seed}-{k}", 2 * (ov + rd)) for k in range(nsrc)]
            for t in range(ticks):
                for k in ids:
                    w.write(k, streams[k, t * rd:(t + 1) * rd])
                got, rows = w.collect()
                assert got == ids[t % nsrc:] + ids[:t % nsrc] or sorted(got) == ids
                for r, k in enumerate(got):
                    prefix = np.zeros(ov, np.uint8) if t == 0 else streams[k, t * rd - ov:t * rd]
                    assert np.array_equal(rows[r, :ov], prefix) and np.array_equal(rows[r, ov:], streams[k, t * rd:(t + 1) * rd]), (t, k)
            assert w.collect()[0] == []
            w.close()
        except Exception as e:                               # pragma: no cover
            errs.append(repr(e))

    ts = [threading.Thread(target=one, args=(s,)) for s in (1, 2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_native_assembler_under_sanitizers(tmp_path, san):
    """csrc/windows.cpp compiled with g++ -fsanitize=thread (then address + undefined) and driven by
    tests/native/windows_stress.cpp: a writer thread per source, two assemblers collecting through the shared row pool at once,
    a thread adding / resetting / removing a spare source and reading stats beside them.  No data race, no out-of-bounds ring
    arithmetic reported, every row equal to the stream it was cut from."""
    import os
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "windows_stress")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + san, "-fno-sanitize-recover=all", "-I", os.path.join(root, "birdnet-go_amd", "csrc"),
                           os.path.join(root, "tests", "native", "windows_stress.cpp"),
                           os.path.join(root, "birdnet-go_amd", "csrc", "windows.cpp"), "-o", exe, "-lpthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    if "unexpected memory mapping" in r.stderr:              # (a kernel whose ASLR layout this libtsan does not know)
        pytest.skip("ThreadSanitizer cannot run on this kernel")
    assert r.returncode == 0 and "Sanitizer" not in r.stderr and "runtime error" not in r.stderr and "0 mismatches" in r.stdout, \
        r.stderr[-3000:] + r.stdout


def test_ring_wraps_and_keeps_newest_bytes():
    r = S.ByteRing(10)
    r.write(bytes(range(8)))
    out = np.empty(5, np.uint8)
    assert r.read_into(out) == 5 and bytes(out) == bytes(range(5))
    r.write(bytes(range(8, 18)))                            # 3 unread + 10 new > 10: the three oldest are dropped
    assert r.length() == 10
    out = np.empty(10, np.uint8)
    assert r.read_into(out) == 10 and bytes(out) == bytes(range(8, 18))
    r.write(bytes(range(100, 125)))                         # longer than the ring: last 10 survive
    assert r.read_into(out) == 10 and bytes(out) == bytes(range(115, 125))


def test_model_spec_geometry():                              # model.go:33-56
    v24 = S.ModelSpec(48000, 3.0)
    assert v24.clip_size_bytes() == 288000 and v24.buffer_dimensions() == (288000, 144000, 144000)
    assert v24.buffer_interval_s() == 1.5
    perch = S.ModelSpec(32000, 5.0)
    assert perch.buffer_dimensions() == (320000, 160000, 160000) and perch.buffer_interval_s() == 2.5
    bat = S.ModelSpec(48000, 3.0, raw_sample_rate=256000)
    assert bat.effective_sample_rate() == 256000 and bat.clip_size_bytes() == 288000


# ------------------------------------------------------------------ a5: Orchestrator
class _Fake:
    def __init__(self, delay=0.0, fail=False):
        self.delay, self.fail, self.calls, self.closed, self.inside = delay, fail, [], False, 0
        self.max_inside = 0

    def predict_batch(self, flat, n):
        self.inside += 1
        self.max_inside = max(self.max_inside, self.inside)
        try:
            if self.delay:
                time.sleep(self.delay)
            if self.fail:
                raise RuntimeError("device lost")
            x = np.asarray(flat, np.float32).reshape(n, -1)
            self.calls.append(x.copy())
            return [[("Strix aluco_Tawny Owl", float(np.float32(0.5) + x[i, 0])), ("noise", 0.01)] for i in range(n)]
        finally:
            self.inside -= 1

    def close(self):
        self.closed = True


def test_orchestrator_unknown_and_closed_model():
    o = S.Orchestrator()
    with pytest.raises(S.OrchestratorError, match="unknown model: nope"):
        o.predict_model("nope", lambda inst: None)
    f = _Fake()
    o.register("m", f, S.ModelSpec(48000, 3.0))
    with o.mu:
        e = o.models["m"]
    with e.mu:
        e.instance = None                                    # closed under the entry lock, still in the map
    with pytest.raises(S.OrchestratorError, match="model m has been closed"):
        o.predict_model("m", lambda inst: None)


def test_orchestrator_counters_and_errors():
    o = S.Orchestrator()
    o.register("ok", _Fake(), S.ModelSpec(48000, 3.0))
    o.register("bad", _Fake(fail=True), S.ModelSpec(48000, 3.0))
    o.predict_model("ok", lambda inst: inst.predict_batch(np.zeros(4, np.float32), 1))
    with pytest.raises(RuntimeError, match="device lost"):
        o.predict_model("bad", lambda inst: inst.predict_batch(np.zeros(4, np.float32), 1))
    p = o.counters.peek_all()
    assert p["ok"]["invoke_count"] == 1 and p["ok"]["invoke_errors"] == 0
    assert p["bad"]["invoke_count"] == 0 and p["bad"]["invoke_errors"] == 1


def test_orchestrator_serialises_inference_across_models():
    """inferenceMu (orchestrator.go:531): two models never run at once, whatever thread asks."""
    o = S.Orchestrator()
    inside, peak = [0], [0]
    lock = threading.Lock()

    def call(inst):
        with lock:
            inside[0] += 1
            peak[0] = max(peak[0], inside[0])
        time.sleep(0.01)
        with lock:
            inside[0] -= 1

    for m in ("a", "b", "c"):
        o.register(m, _Fake(), S.ModelSpec(48000, 3.0))
    ts = [threading.Thread(target=lambda m=m: [o.predict_model(m, call) for _ in range(4)]) for m in ("a", "b", "c")]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert peak[0] == 1


def test_unload_does_not_wait_for_inference_lock_and_closes():
    o = S.Orchestrator()
    f = _Fake()
    o.register("m", f, S.ModelSpec(48000, 3.0))
    o.register("other", _Fake(), S.ModelSpec(48000, 3.0))
    started, release = threading.Event(), threading.Event()
    t = threading.Thread(target=lambda: o.predict_model("other", lambda inst: (started.set(), release.wait(2))))
    t.start()
    started.wait(2)
    t0 = time.perf_counter()
    o.unload("m")                                            # takes mu + entry.mu only (orchestrator.go:66)
    assert time.perf_counter() - t0 < 0.5 and f.closed
    release.set()
    t.join()
    with pytest.raises(S.OrchestratorError, match="unknown model: m"):
        o.predict_model("m", lambda inst: None)


# ------------------------------------------------------------------ a4: ProcessData
def _pcm16(vals):
    return np.asarray(vals, "<i2").tobytes()


def test_process_data_converts_copies_and_enqueues():
    o = S.Orchestrator()
    f = _Fake()
    o.register("BirdNET_V2.4", f, S.ModelSpec(48000, 3.0))
    q, ov = R.ResultsQueue(size=4), S.OverrunTrackers()
    data = bytearray(_pcm16([16384, -32768, 32767, 0]))
    assert S.process_data(o, data, 10.0, 13.0, "mic", "BirdNET_V2.4", q, ov) == 1
    np.testing.assert_array_equal(f.calls[0][0], np.float32([0.5, -1.0, 32767 / 32768, 0.0]))     # process.go:491-495
    msg = q.get()
    data[:] = bytes(len(data))                               # the window goes back to its pool: the message holds its own copy
    assert msg.pcm_data == _pcm16([16384, -32768, 32767, 0])
    assert (msg.start_time, msg.audio_captured_at, msg.source, msg.model_id) == (10.0, 13.0, "mic", "BirdNET_V2.4")
    assert msg.results[0].species == "Strix aluco_Tawny Owl" and msg.results[0].confidence == 1.0
    assert ov.count("mic", "BirdNET_V2.4") == 0


def test_process_data_24_and_32_bit():
    o = S.Orchestrator()
    f = _Fake()
    o.register("m", f, S.ModelSpec(48000, 3.0))
    q, ov = R.ResultsQueue(), S.OverrunTrackers()
    S.process_data(o, np.asarray([1 << 30, -(1 << 31)], "<i4").tobytes(), 0, 0, "s", "m", q, ov, bit_depth=32)
    np.testing.assert_array_equal(f.calls[-1][0], np.float32([0.5, -1.0]))
    b24 = bytes([0x00, 0x00, 0x40, 0x00, 0x00, 0x80, 0xFF, 0xFF, 0xFF])      # 0x400000, -0x800000, -1
    S.process_data(o, b24, 0, 0, "s", "m", q, ov, bit_depth=24)
    np.testing.assert_array_equal(f.calls[-1][0], np.float32([0.5, -1.0, -1 / 8388608]))
    with pytest.raises(S.StreamError, match="unsupported audio bit depth: 8"):
        S.process_data(o, bytes(4), 0, 0, "s", "m", q, ov, bit_depth=8)


def test_process_data_overrun_and_drop_accounting():
    o = S.Orchestrator()
    o.register("slow", _Fake(delay=0.03), S.ModelSpec(48000, 3.0, clip_bytes=8))
    o.models["slow"].spec = S.ModelSpec(48000, 0.04, clip_bytes=8)          # buffer interval 20 ms < 30 ms inference
    q, ov = R.ResultsQueue(size=1), S.OverrunTrackers()
    assert S.process_data(o, bytes(8), 0, 0, "mic", "slow", q, ov) == 1
    assert S.process_data(o, bytes(8), 0, 0, "mic", "slow", q, ov) == 0     # queue full: dropped and counted (process.go:405-419)
    assert ov.count("mic", "slow") == 2
    assert q.drops() == {("mic", "slow"): 1}
    ov.remove_source("mic")
    assert ov.count("mic", "slow") == 0


def test_overrun_tracker_tumbling_window_reports_once():
    now = [0.0]
    reports = []
    ov = S.OverrunTrackers(on_report=lambda *a: reports.append(a), clock=lambda: now[0])
    for _ in range(12):
        ov.record("mic", "m", 2.0, 1.5)
    ov.record("mic", "m", 4.0, 1.5)
    assert reports == [] and ov.count("mic", "m") == 13
    now[0] = 3601.0
    ov.record("mic", "m", 1.6, 1.5)                          # window expired with >= 10 overruns: one report, then reset
    assert len(reports) == 1 and reports[0][:5] == ("mic", "m", 13, 4.0, 1.5)
    assert ov.count("mic", "m") == 1
    now[0] = 7300.0
    ov.record("mic", "m", 1.6, 1.5)                          # only 1 in the expired window: reset without a report
    assert len(reports) == 1 and ov.count("mic", "m") == 1


# ------------------------------------------------------------------ WindowBatcher: the monitors of all sources in one tick
def _reference_windows(stream_bytes, overlap, read):
    """What a reference monitor would have analysed for that byte stream: zero prefix, then 50 % overlapped windows."""
    out, prev = [], bytes(overlap)
    for lo in range(0, len(stream_bytes) - read + 1, read):
        fresh = stream_bytes[lo:lo + read]
        out.append(prev + fresh)
        prev = fresh[read - overlap:]
    return out


@pytest.mark.parametrize("native", [True, False], ids=["native", "python"])
def test_batcher_equals_per_source_monitors(native):
    o = S.Orchestrator()
    f = _Fake()
    spec = S.ModelSpec(48000, 3.0, clip_bytes=64)            # 32-sample windows, 16 overlap
    o.register("m", f, spec)
    wb = S.WindowBatcher(o, R.ResultsQueue(size=1000), max_batch=3, clock=lambda: 100.0, native=native)
    rng = np.random.default_rng(5)
    streams = {f"src{i}": rng.integers(-32768, 32767, 16 * 37, dtype=np.int16).tobytes() for i in range(5)}
    for s in streams:
        wb.allocate(s, "m", capacity=4096)
    pos = {s: 0 for s in streams}
    sent = 0
    while any(pos[s] < len(streams[s]) for s in streams):
        for s, b in streams.items():
            n = int(rng.integers(1, 25)) * 2
            wb.write(s, b[pos[s]:pos[s] + n])
            pos[s] += n
        sent += wb.tick()
    while True:
        k = wb.tick()
        if k == 0:
            break
        sent += k
    got = {}
    while wb.queue.qsize():
        m = wb.queue.get()
        got.setdefault(m.source, []).append(m)
        assert m.start_time == 100.0 - 3.0 and m.audio_captured_at == 100.0       # buffer_manager.go:489-491
    for s, b in streams.items():
        want = _reference_windows(b, 32, 32)
        assert [m.pcm_data for m in got[s]] == want
    assert sent == sum(len(v) for v in got.values())
    assert all(c.shape[0] <= 3 for c in f.calls)             # max_batch respected
    assert o.counters.peek_all()["m"]["invoke_count"] == len(f.calls)


@pytest.mark.parametrize("native", [True, False], ids=["native", "python"])
def test_batcher_inactive_model_consumes_without_analysing(native):
    o = S.Orchestrator()
    f = _Fake()
    o.register("bat", f, S.ModelSpec(48000, 3.0, clip_bytes=16))
    wb = S.WindowBatcher(o, native=native)
    wb.allocate("mic", "bat", capacity=256)
    o.set_active("bat", False)
    wb.write("mic", bytes(range(8)))
    assert wb.tick() == 0 and f.calls == [] and not wb.buffers[("mic", "bat")].ready()      # buffer_manager.go:478-480
    o.set_active("bat", True)
    wb.write("mic", bytes(range(8, 16)))
    assert wb.tick() == 1
    assert wb.queue.get().pcm_data == bytes(range(8)) + bytes(range(8, 16))                 # overlap kept across the skipped window


@pytest.mark.parametrize("native", [True, False], ids=["native", "python"])
def test_batcher_failed_model_costs_only_its_own_windows(native):
    """buffer_manager.go:494-499: a ProcessData error is logged by that monitor and polling goes on - one model's failing
    device call must not cost the other models the windows this tick already consumed from their rings."""
    class _Boom:
        def predict_batch(self, flat, n):
            raise RuntimeError("device lost")
    o = S.Orchestrator()
    good = _Fake()
    o.register("bad", _Boom(), S.ModelSpec(48000, 3.0, clip_bytes=16))
    o.register("good", good, S.ModelSpec(48000, 3.0, clip_bytes=16))
    seen = []
    wb = S.WindowBatcher(o, on_error=lambda model, sources, e: seen.append((model, sources, str(e))), native=native)
    for m in ("bad", "good"):
        wb.allocate("mic", m, capacity=256)
    wb.write("mic", bytes(range(8)))
    assert wb.tick() == 1 and wb.errors == 1
    assert seen == [("bad", ["mic"], "device lost")]
    assert wb.queue.get().model_id == "good" and len(good.calls) == 1
    assert o.counters.peek_all()["bad"]["invoke_errors"] == 1


@pytest.mark.parametrize("native", [True, False], ids=["native", "python"])
def test_batcher_model_unloaded_between_ticks(native):
    o = S.Orchestrator()
    f = _Fake()
    o.register("m", f, S.ModelSpec(48000, 3.0, clip_bytes=16))
    wb = S.WindowBatcher(o, native=native)
    wb.allocate("mic", "m", capacity=256)
    wb.write("mic", bytes(range(8)))
    o.unload("m")                                            # the buffer is still allocated: its audio is consumed, nothing raised
    assert wb.tick() == 0 and wb.errors == 0 and not wb.buffers[("mic", "m")].ready()


def test_analysis_buffer_takes_arrays_by_their_bytes():
    """A capture callback that hands over int16 arrays (or strided views): the ring counts bytes, not elements."""
    ab = S.AnalysisBuffer(16, 4, 4, "mic")
    ab.write(np.arange(6, dtype="<i2"))                      # 12 bytes
    ab.write(np.arange(12, dtype="<i2")[::2][:3])            # 6 bytes, not contiguous: 18 > 16 -> an overwrite
    assert ab.overwrite_count() == 1
    ref = GoAnalysisBuffer(16, 4, 4)
    ref.Write(np.arange(6, dtype="<i2").tobytes())
    ref.Write(np.ascontiguousarray(np.arange(12, dtype="<i2")[::2][:3]).tobytes())
    assert ref.overwrites == 1
    for _ in range(4):
        a, b = ab.read(), ref.Read()
        assert (a is None) == (b is None)
        if a is not None:
            assert a.tobytes() == bytes(b)


def test_overwrite_warning_callback_may_query_the_tracker():
    """The callback runs outside the tracker's lock (it used to deadlock a callback that reads the rate back)."""
    rates = []
    t = S.OverwriteTracker(min_writes=1, rate_threshold=1, notify_cooldown_s=0, on_warn=lambda *a: rates.append(t.overwrite_rate()))
    t.record_write(); t.record_overwrite()
    assert t.check_and_notify("mic") and rates == [100.0]


# ------------------------------------------------------------------ through the device
@pytest.mark.gpu
@pytest.mark.parametrize("native", [True, False], ids=["native", "python"])
def test_batcher_streams_through_the_device(tiny_cfg, tiny_blob, native):
    """Five sources streaming 16-bit PCM in ragged chunks through WindowBatcher + host.BirdNET == the same windows cut by hand
    and sent through predict_pcm16 one call at a time (the reference's one-window-per-call pattern).  native: the rings live in
    the library and the device call reads the windows out of its page-locked batch buffer."""
    from birdnet_go_amd import host
    clf = host.HipClassifier(tiny_blob, device=0, max_batch=8)
    labels = [f"sp{i}" for i in range(clf.num_species())]
    bn = host.BirdNET(clf, labels, sensitivity=1.0)
    clip_bytes = tiny_cfg.n_samples * 2
    spec = S.ModelSpec(tiny_cfg.sample_rate, 3.0, clip_bytes=clip_bytes)
    o = S.Orchestrator()
    o.register("tiny", bn, spec)
    wb = S.WindowBatcher(o, R.ResultsQueue(size=1000), max_batch=8, native=native)
    rng = np.random.default_rng(11)
    n_src, n_win = 5, 4
    t = np.arange(tiny_cfg.n_samples // 2 * (n_win + 1)) / tiny_cfg.sample_rate
    streams = {}
    for i in range(n_src):
        x = 0.4 * np.sin(2 * np.pi * (700 + 310 * i) * t) + rng.normal(0, 0.05, t.size)
        streams[f"src{i}"] = (np.clip(x, -1, 1) * 32767).astype("<i2").tobytes()
        wb.allocate(f"src{i}", "tiny")
    pos = {s: 0 for s in streams}
    while any(pos[s] < len(streams[s]) for s in streams):
        for s, b in streams.items():
            n = int(rng.integers(200, 4000)) * 2
            wb.write(s, b[pos[s]:pos[s] + n])
            pos[s] += n
        wb.tick()
    while wb.tick():
        pass
    got = {}
    while wb.queue.qsize():
        m = wb.queue.get()
        got.setdefault(m.source, []).append(m)
    _, overlap, read = spec.buffer_dimensions()
    for s, b in streams.items():
        want = _reference_windows(b, overlap, read)
        assert len(want) >= n_win and [m.pcm_data for m in got[s]] == want
        for m, w in zip(got[s], want):
            one = bn.predict_pcm_batch(w, 16, 1)[0]
            assert [d.species for d in m.results] == [lbl for lbl, _ in one]
            np.testing.assert_allclose([d.confidence for d in m.results], [c for _, c in one], rtol=0, atol=2e-6)
    assert o.counters.peek_all()["tiny"]["invoke_count"] >= n_win
    if native:
        assert wb.assemblers["tiny"].pinned and wb.assemblers["tiny"].window_bytes == clip_bytes
    wb.close()
    clf.close()


@pytest.mark.gpu
def test_native_windows_feed_the_host_pipeline_in_place(tiny_cfg, tiny_blob):
    """140 sources ready in one tick: the call is large enough for the chunked host pipeline (>= 128 clips), which detects that
    the library's batch buffer is page-locked and lets the copy engines read the windows where they were assembled
    (csrc/hostpipe.cpp is_pinned).  Same logits, bit for bit, as the same windows passed from pageable memory."""
    from birdnet_go_amd import host
    clf = host.HipClassifier(tiny_blob, device=0, max_batch=256)
    clip_bytes = tiny_cfg.n_samples * 2
    overlap, read = clip_bytes // 2, clip_bytes - clip_bytes // 2
    w = S.NativeWindows(overlap, read, max_batch=256)
    assert w.pinned
    rng = np.random.default_rng(3)
    n_src = 140
    pcm = (rng.normal(0, 0.2, (n_src, 2 * read // 2)).clip(-1, 1) * 32767).astype("<i2")
    for k in range(n_src):
        assert w.add_source(f"mic{k}", 2 * clip_bytes) == k
        w.write(k, pcm[k])                                    # two reads' worth: two windows per source
    for rnd in range(2):
        idxs, rows = w.collect()
        assert idxs == list(range(n_src))
        got = clf.predict_pcm(rows.reshape(-1), 16, n_src)                 # the library's buffer, in place
        want = clf.predict_pcm(rows.copy().reshape(-1), 16, n_src)         # pageable copy of the same bytes
        assert np.array_equal(got, want)
        b = pcm.view(np.uint8).reshape(n_src, -1)
        prefix = np.zeros((n_src, overlap), np.uint8) if rnd == 0 else b[:, read - overlap:read]
        assert np.array_equal(rows, np.concatenate([prefix, b[:, rnd * read:(rnd + 1) * read]], axis=1))
    assert w.collect()[0] == []
    # the whole tick in one call: the pipeline asks the assembler for chunk c + 1's rows while chunk c is on the device
    # (bnhip_windows_predict_topk; 140 windows = three chunks on two contexts).  Same rows, same top-k as the two-step path.
    for k in range(n_src):
        w.reset(k)
        w.write(k, pcm[k])
    for rnd in range(2):
        idxs, rows, conf, idx = w.predict_topk(clf, 16, 10, 0, 1.25)
        assert idxs == list(range(n_src))
        prefix = np.zeros((n_src, overlap), np.uint8) if rnd == 0 else b[:, read - overlap:read]
        want_rows = np.concatenate([prefix, b[:, rnd * read:(rnd + 1) * read]], axis=1)
        assert np.array_equal(rows, want_rows)
        c2, i2 = clf.predict_pcm_topk(want_rows.reshape(-1), 16, n_src, 10, 0, 1.25)
        assert np.array_equal(conf, c2) and np.array_equal(idx, i2)
    assert w.predict_topk(clf)[0] == []
    with pytest.raises(S.StreamError, match="window size mismatch"):
        w.predict_topk(clf, 32)
    w.close()
    clf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("bits", [24, 32])
def test_tick_at_the_other_capture_depths(tiny_cfg, tiny_blob, bits):
    """The reference captures 16-bit (conf.BytesPerSample); the boundary takes the three depths of ConvertToFloat32
    (convert/pcm.go:206-268).  A tick over 24- / 32-bit windows == bnhip_predict_pcm_topk on the same windows cut by hand."""
    from birdnet_go_amd import host
    clf = host.HipClassifier(tiny_blob, device=0, max_batch=8)
    bps = bits // 8
    clip_b = tiny_cfg.n_samples * bps
    ov = (tiny_cfg.n_samples // 2) * bps                      # whole samples on either side of the cut
    rd = clip_b - ov
    w = S.NativeWindows(ov, rd, max_batch=8)
    rng = np.random.default_rng(bits)
    x = rng.normal(0, 0.2, (3, 2 * rd // bps)).clip(-1, 1)
    i32 = np.round(x * (2 ** 31 - 1)).astype("<i4")
    raw = i32.view(np.uint8).reshape(3, -1, 4)[:, :, 4 - bps:].reshape(3, -1)     # top `bps` bytes of each sample, little-endian
    for k in range(3):
        assert w.add_source(f"mic{k}", 2 * clip_b) == k
        w.write(k, raw[k])
    for rnd in range(2):
        idxs, rows, conf, idx = w.predict_topk(clf, bits, 5, 0, 1.0)
        assert idxs == [0, 1, 2]
        prefix = np.zeros((3, ov), np.uint8) if rnd == 0 else raw[:, rd - ov:rd]
        want = np.concatenate([prefix, raw[:, rnd * rd:(rnd + 1) * rd]], axis=1)
        assert np.array_equal(rows, want)
        c2, i2 = clf.predict_pcm_topk(want.reshape(-1), bits, 3, 5, 0, 1.0)
        assert np.array_equal(conf, c2) and np.array_equal(idx, i2)
    with pytest.raises(S.StreamError, match="window size mismatch"):
        w.predict_topk(clf, 16)
    w.close()
    clf.close()
