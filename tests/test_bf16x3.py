"""Split-bf16 MFMA path (VERDICT r1 "Next" #5): fp32 operands as three exact bf16 pieces, six products on
v_mfma_f32_16x16x32_bf16, fp32 accumulation.  This is arithmetic that EMULATES fp32, so the gate is an error comparison
against the float64 arbiter next to the plain-fp32 engine, not just the probability tolerance."""
import numpy as np
import pytest

from birdnet_go_amd import host, synth_model as sm
from oracle.interp import Interpreter


def test_bf16x3_plans_weight_images(built_lib, full_blob):
    a = host.HipClassifier(full_blob, plan_only=True, bf16x3=0)
    b = host.HipClassifier(full_blob, plan_only=True, bf16x3=2)
    da, db = a.describe(), b.describe()
    pw = [s for s in db["steps"] if s["kernel"] == "pw_gemm"]
    forced = [s for s in pw if s["wm_full"] == 6]
    # every layer with K >= 16, K % 4 == 0 (a K tail inside the last 32-wide slab is zero weights x zero-filled columns)
    assert len(forced) >= 16 and all(s["C"] % 4 == 0 and s["C"] >= 16 for s in forced)
    assert any(s["C"] % 32 for s in forced)
    assert all(s["wm_full"] != 6 for s in pw if s["C"] % 4 or s["C"] < 16)
    assert db["weight_bytes"] > 2.0 * da["weight_bytes"]                     # + 1.5x image of every eligible layer
    a.close(); b.close()


def _bf16_rne(x):
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def _split3(x):
    """the kernel's decomposition, restated in numpy: x = hi + mid + lo, bf16 pieces by round-to-nearest-even"""
    x = np.asarray(x, np.float32)
    hi = _bf16_rne(x)
    r = x - hi
    mid = _bf16_rne(r)
    lo = r - mid
    return hi, mid, lo


def test_three_way_split_is_exact_and_dropped_terms_are_below_one_ulp():
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * 10.0 ** rng.uniform(-30, 30, 200000)).astype(np.float32)
    x[:3] = [0.0, -0.0, np.float32(1.0) + np.float32(2.0) ** -23]
    hi, mid, lo = _split3(x)
    assert np.array_equal((hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64)).astype(np.float32), x)
    for piece in (hi, mid, lo):                                              # each piece IS a bf16: low 16 bits clear
        assert not (piece.view(np.uint32) & 0xFFFF).any()
    assert (np.abs(mid) <= 2.0 ** -8 * np.abs(x)).all() and (np.abs(lo) <= 2.0 ** -16 * np.abs(x)).all()
    # the three dropped cross terms together stay below 2^-23 |x w| (an fp32 product's own rounding is <= 2^-24 |x w|)
    w = rng.standard_normal(200000).astype(np.float32)
    wh, wm, wl = _split3(w)
    f = lambda v: v.astype(np.float64)
    kept = f(hi) * f(wh) + f(hi) * f(wm) + f(mid) * f(wh) + f(hi) * f(wl) + f(lo) * f(wh) + f(mid) * f(wm)
    exact = f(x) * f(w)
    ok = (np.abs(exact) > 1e-30) & (np.abs(exact) < 1e30)
    assert (np.abs(kept - exact)[ok] <= 2.0 ** -23 * np.abs(exact)[ok] * 1.01).all()


@pytest.mark.gpu
def test_bf16x3_gemm_error_vs_f64_next_to_fp32(gpu):
    """Dense stack, K = 1024 / 1536, inputs spanning six decades: the split-bf16 engine's error against float64 must be of
    the fp32 engine's order (both are dominated by fp32 accumulation), and far below a plain bf16 GEMM's (~4e-3)."""
    dims = [1024, 1536, 640]
    blob = sm.build_dense_model(dims, hidden_act="none", seed=3)
    rng = np.random.default_rng(5)
    x = (rng.standard_normal((96, 1024)) * 10.0 ** rng.uniform(-3, 3, (96, 1))).astype(np.float32)
    ref = Interpreter(blob, "f64").invoke(x)[0].astype(np.float64)
    scale = np.abs(ref).max(axis=1, keepdims=True)
    errs = {}
    for mode in (0, 2):
        clf = host.HipClassifier(blob, max_batch=128, bf16x3=mode)
        kinds = [(s["kernel"], s["wm_full"]) for s in clf.describe()["steps"]]
        assert all(k == "pw_gemm" for k, _ in kinds) and all((w >= 5) == (mode == 2) for _, w in kinds), kinds
        got = clf.predict_batch(x.reshape(-1), 96).astype(np.float64)
        clf.close()
        errs[mode] = float((np.abs(got - ref) / scale).max())
    print(f"max relative error vs f64: fp32 MFMA {errs[0]:.3e}, split-bf16 {errs[2]:.3e}")
    assert errs[2] <= 3.0 * errs[0] + 1e-7 and errs[2] < 5e-6


@pytest.mark.gpu
def test_bf16x3_full_model_parity_and_error(gpu, full_blob):
    from test_parity_gpu import assert_parity
    x = sm.synth_clips(3, 144000, 48000)
    ref32 = Interpreter(full_blob).invoke(x)[0]
    ref64 = Interpreter(full_blob, "f64").invoke(x[:1])[0]
    out = {}
    for mode in (0, 2):
        clf = host.HipClassifier(full_blob, max_batch=4, bf16x3=mode)
        out[mode] = clf.predict_batch(x.reshape(-1), 3)
        clf.close()
        assert_parity(out[mode], ref32)
    e0, e2 = np.abs(out[0][:1] - ref64).max(), np.abs(out[2][:1] - ref64).max()
    print(f"max |logit - f64 arbiter|: fp32 engine {e0:.3e}, split-bf16 engine {e2:.3e}, oracle f32 {np.abs(ref32[:1] - ref64).max():.3e}")
    assert e2 <= 3.0 * e0 + 2e-6
    assert (out[2].argmax(1) == out[0].argmax(1)).all()
    # batch through the autotuned choice (bf16x3 = 1) and SE-scaled + residual layers at batch size
    xb = sm.synth_clips(40, 144000, 48000)
    a = host.HipClassifier(full_blob, max_batch=40, bf16x3=0)
    b = host.HipClassifier(full_blob, max_batch=40, bf16x3=1)
    ya, yb = a.predict_batch(xb.reshape(-1), 40), b.predict_batch(xb.reshape(-1), 40)
    picked = sum(1 for s in b.describe()["steps"] if s["kernel"] == "pw_gemm" and (s["wm"] >= 5 or s["wm_full"] >= 5))
    print("layers the autotuner moved to the split-bf16 kernel:", picked)
    a.close(); b.close()
    assert np.abs(ya - yb).max() < 1e-4 and (ya.argmax(1) == yb.argmax(1)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("i", [0, 1, 2, 4, 6])
def test_bf16x3_forced_on_geometry_sweep(gpu, i):
    """Every eligible layer forced onto the split-bf16 kernels - pointwise GEMMs (K % 32 == 0) and the expand GEMM inside the
    fused expand+depthwise kernel (Cin % 8 == 0, K tails zero-padded to 32) - on the geometry-sweep models: tile shapes,
    strides, ragged edges and K tails the v2.4 topology does not have, vs the oracle."""
    from test_parity_gpu import _geo_cfg, assert_parity
    cfg = _geo_cfg(i)
    blob = sm.build_model(cfg)
    x = sm.synth_clips(5, cfg.n_samples, cfg.sample_rate, first=3 * i)
    ref = Interpreter(blob).invoke(x)[0]
    for opts in (dict(), dict(lanes=1, autotune=False)):
        c = host.HipClassifier(blob, max_batch=64 if not opts else 5, bf16x3=2, **opts)
        try:
            got = c.predict_batch(x.reshape(-1), 5)
            steps = c.describe()["steps"]
        finally:
            c.close()
        assert_parity(got, ref)
        assert steps


@pytest.mark.gpu
def test_bf16x3_expand_dw_variant_is_exercised(gpu, full_blob):
    c = host.HipClassifier(full_blob, max_batch=8, bf16x3=2)
    try:
        ed = [s for s in c.describe()["steps"] if s["kernel"] == "expand_dw"]
        assert sum(s["bx"] for s in ed) >= 10, [(s["name"], s["bx"]) for s in ed]      # b2..b12 (the fused stem stays f32)
    finally:
        c.close()


@pytest.mark.gpu
def test_bf16x3_range_edges_huge_and_near_denormal_operands(gpu):
    """The two ends DESIGN.md section 5 only described.  A dense layer (K = 64) on the split-bf16 kernel:
    * operands up to 3.3e38 (just under the largest bf16, 3.39e38): finite, relative error vs float64 <= 4e-6, like fp32;
    * operands whose mid / lo pieces are bf16 subnormals (|x| ~ 1e-37) or that are fp32 denormals themselves (1e-39): finite,
      no NaN; whatever the matrix pipe does with subnormal pieces, at worst only the hi piece survives, so the ABSOLUTE error
      is <= 2^-7 of the row's largest product - below 1e-37, i.e. nothing;
    * an operand ABOVE the largest bf16 (still finite in fp32): its hi piece rounds to infinity and the row is not finite.
      Pinned as the documented limit of the split (audio activations are O(1..1e3); the f32-MFMA path, bf16x3 = 0, has none)."""
    K, N = 64, 32
    rng = np.random.default_rng(3)
    blob = sm.build_dense_model([K, N], seed=3)
    x = np.zeros((6, K), np.float32)
    x[0] = rng.uniform(0.5, 1.0, K) * 3.3e38 * rng.choice([-1, 1], K) / K          # huge, the sums stay finite
    x[1, :4] = [3.3e38 / 8, -3.2e38 / 8, 1.0, -1.0]
    x[2] = rng.uniform(1.0, 9.0, K) * 1e-37 * rng.choice([-1, 1], K)               # mid / lo pieces are subnormal bf16
    x[3] = rng.uniform(1.0, 9.0, K) * 1e-39 * rng.choice([-1, 1], K)               # fp32 denormals
    x[4] = rng.standard_normal(K)
    x[5, 0] = 3.4e38                                                                 # > largest bf16
    with np.errstate(over="ignore", invalid="ignore"):
        ref64 = Interpreter(blob, precision="f64").invoke(x)[0].astype(np.float64)
    out = {}
    for bx in (2, 0):
        c = host.HipClassifier(blob, max_batch=8, bf16x3=bx, autotune=False)
        try:
            out[bx] = c.predict_batch(x.reshape(-1), 6)
        finally:
            c.close()
    g = out[2].astype(np.float64)
    assert np.isfinite(out[2][:5]).all() and np.isfinite(out[0][:5]).all()
    for r in (0, 1, 4):
        assert np.abs(g[r] - ref64[r]).max() <= 4e-6 * np.abs(ref64[r]).max(), r
    for r in (2, 3):
        big = float(np.abs(x[r]).max()) * 1.0 * K           # (|w| < 1 for these weights)
        bias_free = np.abs(g[r] - ref64[r]).max()
        assert bias_free <= 2.0 ** -7 * big + 1e-6 * np.abs(ref64[r]).max(), r      # (the bias, O(0.1), dominates the row: relative part)
    assert not np.isfinite(out[2][5]).all()                # documented limit: hi = bf16(3.4e38) = inf
    assert np.isfinite(out[0][5]).all()                    # the f32-MFMA kernel has no such limit
