"""bench.py quotes PMC counters (roofline.traffic, mfma_util) from committed files - only when they were collected on the library
that is loaded (a kernel change without re-profiling must not ship stale counters under a fresh headline)."""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    sys.modules["bench_mod"] = m
    spec.loader.exec_module(m)
    return m


def test_counter_files_are_quoted_only_beside_their_library(tmp_path, monkeypatch):
    b = _bench()
    monkeypatch.setattr(b, "ROOT", str(tmp_path))
    prof = tmp_path / "profiles"
    prof.mkdir()
    traffic = {"k_expand_dw<1>": {"dispatches": 4, "hbm_bytes_per_launch": 100.0}, "k_expand_dw<2>": {"dispatches": 12, "hbm_bytes_per_launch": 200.0}}
    mfma = {"classes": {"pw_gemm": {"mfma_util": 0.3, "valu_per_mfma": 5.0}, "expand_dw": {"mfma_util": 0.33, "valu_per_mfma": 9.0}}, "numerator": "n", "denominator": "d"}
    # a file from before the binding existed: not quoted
    (prof / "r04_traffic.json").write_text(json.dumps(traffic))
    (prof / "r04_mfma_util.json").write_text(json.dumps(mfma))
    t = b.pmc_traffic("expand_dw", "birdnet", "aaaa", "p1")
    assert t["bytes_per_launch"] is None and "no library digest" in t["binding"]["dropped"]
    assert b.pmc_mfma_util("birdnet", "aaaa", "p1")["pointwise_and_dense"] is None
    # a newer file collected on another library: not quoted either (the newest file decides, never an older match)
    (prof / "r05_traffic.json").write_text(json.dumps({**traffic, "_binding": {"lib_digest": "bbbb", "plan_signature": "p1", "tune_sha256": "t"}}))
    (prof / "r05_mfma_util.json").write_text(json.dumps({**mfma, "_binding": {"lib_digest": "bbbb", "plan_signature": "p1", "tune_sha256": "t"}}))
    t = b.pmc_traffic("expand_dw", "birdnet", "aaaa", "p1")
    assert t["bytes_per_launch"] is None and "collected on library bbbb" in t["binding"]["dropped"]
    m = b.pmc_mfma_util("birdnet", "aaaa", "p1")
    assert m["pointwise_and_dense"] is None and m["expand_dw"] is None and "dropped" in m["binding"]
    # the loaded library is the one the counters were collected on: quoted, with the plan comparison
    t = b.pmc_traffic("expand_dw", "birdnet", "bbbb", "p1")
    assert abs(t["bytes_per_launch"] - (4 * 100.0 + 12 * 200.0) / 16) < 1e-9 and t["binding"]["plan_matches_this_run"] is True
    m = b.pmc_mfma_util("birdnet", "bbbb", "p2")
    assert m["pointwise_and_dense"] == 0.3 and m["expand_dw"] == 0.33 and m["binding"]["plan_matches_this_run"] is False
    # Perch files answer the Perch workload only
    assert b.pmc_traffic("expand_dw", "perch", "bbbb", "p1")["bytes_per_launch"] is None


def test_plan_signature_follows_the_tuners_decisions():
    b = _bench()
    steps = [{"name": "b1/project", "nt": 1, "wm": 1, "nt_full": 1, "wm_full": 1, "shape": -1, "dw_lds": 0, "bx": 0},
             {"name": "b13/expand", "nt": 6, "wm": 12, "nt_full": 6, "wm_full": 12, "shape": -1, "dw_lds": 0, "bx": 1}]
    s0 = b.plan_signature({"steps": steps})
    assert s0 == b.plan_signature({"steps": [dict(s) for s in steps]})
    steps[1]["wm_full"] = 6
    assert s0 != b.plan_signature({"steps": steps})


def test_baseline_batch_per_gpu():
    """bench.py: the per-GPU batch of the BASELINE configuration a run stands for - configs[1] at N = 1, the 1 024-clip shard of
    configs[2] at any N > 1 (the N-rank run used to measure 256 per rank), Perch's 512 at every N."""
    import bench
    assert [bench.baseline_batch("birdnet", n) for n in (1, 2, 4, 8)] == [256, 1024, 1024, 1024]
    assert [bench.baseline_batch("perch", n) for n in (1, 8)] == [512, 512]
