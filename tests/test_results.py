"""Results hand-off: queue contract, counters semantics (reference counters.go / queue.go / process.go)."""
import numpy as np
import pytest

import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import results as R


class _FakeBirdNET:
    def __init__(self, fail=False):
        self.fail = fail

    def predict_batch(self, flat, n):
        if self.fail:
            raise RuntimeError("device lost")
        return [[("Strix aluco_Tawny Owl", 0.9 - 0.1 * i), ("noise", 0.05)] for i in range(n)]


def test_queue_drops_when_full_and_counts():
    q = R.ResultsQueue(size=2)
    msgs = [R.Results(0.0, 0.0, b"", [], 0.0, "mic", "BirdNET_V2.4") for _ in range(3)]
    assert [q.offer(m) for m in msgs] == [True, True, False]
    assert q.drops() == {("mic", "BirdNET_V2.4"): 1}
    assert q.get() is msgs[0]            # ownership moves, no copy (queue.go:24-28)


def test_counters_window_max_resets_lifetime_does_not():
    c = R.CounterMap()
    for us in (100, 900, 300):
        c.record_invoke("m", us)
    c.record_error("m")
    s1 = c.snapshot_all()["m"]
    assert (s1["invoke_count"], s1["invoke_total_us"], s1["invoke_max_us"], s1["invoke_errors"]) == (3, 1300, 900, 1)
    c.record_invoke("m", 200)
    s2 = c.snapshot_all()["m"]
    assert s2["invoke_max_us"] == 200 and s2["invoke_count"] == 4      # max was reset on the previous read
    p = c.peek_all()["m"]
    assert p["invoke_max_us_lifetime"] == 900 and p["invoke_errors"] == 1
    c.delete("m")
    assert c.peek_all() == {}


def test_rolling_p95_matches_reference_index_rule():
    c = R.CounterMap()
    for us in range(1, 101):
        c.record_invoke("m", us)
    # idx = ceil(0.95 * n) - 1 over the sorted window
    assert c.peek_all()["m"]["recent_p95_us"] == 95
    for us in range(R.LATENCY_WINDOW):         # window holds only the last 1024
        c.record_invoke("m", 7)
    assert c.peek_all()["m"]["recent_p95_us"] == 7


def test_metric_key_sanitised():
    assert R.metric_key("BirdNET V2.4/fp32") == "inference.BirdNET_V2_4_fp32.avg_ms"


def test_dispatcher_one_message_per_clip_with_threshold():
    d = R.BatchDispatcher(_FakeBirdNET(), "BirdNET_V2.4", confidence_threshold=0.1)
    clips = np.zeros((3, 16), np.float32)
    sent = d.dispatch(clips, start_times=[0.0, 1.5, 3.0], source="file.wav", pcm_chunks=[b"ab", b"cd", b"ef"])
    assert sent == 3
    m = d.queue.get()
    assert m.start_time == 0.0 and m.pcm_data == b"ab" and m.model_id == "BirdNET_V2.4"
    assert [x.species for x in m.results] == ["Strix aluco_Tawny Owl"]     # 0.05 filtered by the threshold
    snap = d.counters.snapshot_all()["BirdNET_V2.4"]
    assert snap["invoke_count"] == 1 and snap["invoke_errors"] == 0


def test_dispatcher_records_errors():
    d = R.BatchDispatcher(_FakeBirdNET(fail=True), "m")
    with pytest.raises(RuntimeError):
        d.dispatch(np.zeros((1, 4), np.float32), [0.0], "s")
    assert d.counters.peek_all()["m"]["invoke_errors"] == 1
