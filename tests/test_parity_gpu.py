"""Parity tests proper: HIP path (through the C ABI) vs the oracle on the same seeded inputs, against
the committed golden fixtures, and through size-independent properties at BASELINE.json's full
sizes.  Run on the GPU box with `-m gpu`.

Stated fp32 tolerance (north_star: "match the reference ... to a stated fp32 tolerance"):
  top-1 identical AND max-abs probability diff <= 1e-4 vs the fp32 oracle restatement
(the reference's own cross-backend gate is top-1 + 0.05: openvino_parity_functional_test.go:56,112-116;
its measured f32-vs-f32 drift is ~6e-6: model_openvino.go:103)."""
import os

import numpy as np
import pytest

from birdnet_go_amd import host, synth_model as sm
from oracle import gofuncs as G
from oracle.interp import Interpreter

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
PROB_TOL = 1e-4


def sig(v):
    return 1.0 / (1.0 + np.exp(-np.asarray(v, np.float64)))


def assert_parity(got, ref, tol=PROB_TOL):
    assert np.isfinite(got).all()
    assert (got.argmax(1) == ref.argmax(1)).all(), "top-1 differs"
    d = np.abs(sig(got) - sig(ref)).max()
    assert d <= tol, f"max-abs prob diff {d}"


@pytest.fixture(scope="module")
def tiny_clf(built_lib, tiny_blob):
    c = host.HipClassifier(tiny_blob, max_batch=8)
    yield c
    c.close()


@pytest.fixture(scope="module")
def full_clf(built_lib, full_blob):
    c = host.HipClassifier(full_blob, max_batch=64)
    yield c
    c.close()


def test_library_is_the_native_one(built_lib):
    assert host.init() >= 1
    assert os.path.realpath(built_lib).startswith(os.path.realpath(os.path.dirname(os.path.dirname(__file__))))


def test_tiny_vs_oracle_and_golden(tiny_clf, tiny_blob, tiny_cfg):
    x = sm.synth_clips(3, tiny_cfg.n_samples, tiny_cfg.sample_rate)
    got = tiny_clf.predict_batch(x.reshape(-1), 3)
    assert_parity(got, Interpreter(tiny_blob).invoke(x)[0])
    g = np.load(os.path.join(GOLD, "tiny_logits.npz"))
    assert_parity(got, g["logits_f32"])
    # neither fp32 path should be further from the fp64 arbiter than a few fp32 ulps of the logit scale
    assert np.abs(got - g["logits_f64"]).max() < 2e-5


def test_full_vs_oracle_and_golden(full_clf, full_blob):
    x = sm.synth_clips(4, 144000, 48000)
    got = full_clf.predict_batch(x.reshape(-1), 4)
    ref = Interpreter(full_blob).invoke(x)[0]
    assert_parity(got, ref)
    assert np.abs(got - ref).max() < 1e-3          # logits "EQUIVALENT" band, cmd/perch-benchmark/main.go:455-462
    g = np.load(os.path.join(GOLD, "full_top32.npz"))
    sel = np.take_along_axis(got[:2], g["idx"], 1)
    assert np.abs(sel - g["logits"]).max() < 1e-3
    assert (got[:2].argmax(1) == g["idx"][:, 0]).all()


class _DevBuf:
    """Device memory through the HIP runtime the library itself uses (torch bundles a second runtime; mixing the two in
    one process after libbnhip has initialised the first one finds no GPU)."""
    _hip = None

    def __init__(self, nbytes):
        import ctypes as C
        if _DevBuf._hip is None:
            _DevBuf._hip = C.CDLL("libamdhip64.so")
        self.C, self.n = C, nbytes
        self.ptr = C.c_void_p()
        assert _DevBuf._hip.hipMalloc(C.byref(self.ptr), C.c_size_t(nbytes)) == 0

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert _DevBuf._hip.hipMemcpy(self.ptr, a.ctypes.data_as(self.C.c_void_p), self.C.c_size_t(a.nbytes), 1) == 0

    def download(self, shape, dtype=np.float32):
        out = np.empty(shape, dtype)
        assert _DevBuf._hip.hipMemcpy(out.ctypes.data_as(self.C.c_void_p), self.ptr, self.C.c_size_t(out.nbytes), 2) == 0
        return out

    def at(self, byte_off):
        return self.ptr.value + byte_off

    def free(self):
        _DevBuf._hip.hipFree(self.ptr)


def test_pipeline_depth_gives_identical_results(full_blob):
    """"depth" > 1: successive device calls run on alternating contexts; every call's output must equal the serial one."""
    xh = sm.synth_clips(12, 144000, 48000)
    a = host.HipClassifier(full_blob, max_batch=4)
    b = host.HipClassifier(full_blob, max_batch=4, depth=2, lanes=1)
    x, ref, out = _DevBuf(xh.nbytes), _DevBuf(12 * 6522 * 4), _DevBuf(12 * 6522 * 4)
    try:
        x.upload(xh)
        for i in range(3):                                        # reference 1: another engine (own autotuned tiles)
            a.predict_device(x.at(4 * i * 144000 * 4), 4, ref.at(4 * i * 6522 * 4))
        a.synchronize()
        r = ref.download((12, 6522))
        for i in range(3):                                        # reference 2: the same engine, one call at a time
            b.predict_device(x.at(4 * i * 144000 * 4), 4, ref.at(4 * i * 6522 * 4))
            b.synchronize()
        serial = ref.download((12, 6522))
        for i in range(3):                                        # three calls in flight over two contexts
            b.predict_device(x.at(4 * i * 144000 * 4), 4, out.at(4 * i * 6522 * 4))
        b.synchronize()
        o = out.download((12, 6522))
        assert np.array_equal(serial, o)                          # overlapping changes nothing, bit for bit
        assert np.abs(o - r).max() < 1e-4                         # other tile shapes = another fp32 summation order
        # an unsplit (host-pointer) call right after pipelined ones is ordered behind them
        b.predict_device(x.at(0), 4, out.at(0))
        h = b.predict_batch(xh[:4].reshape(-1), 4)
        # (the blocking entry launches the squeeze-excite kernel with 1024-thread blocks, a pipelined call with 256-thread
        # ones - another fixed summation order: equal to rounding, each entry bit-reproducible on its own)
        assert np.abs(h - o[:4]).max() < 1e-4 and np.array_equal(h, b.predict_batch(xh[:4].reshape(-1), 4))
    finally:
        a.close(); b.close()
        x.free(); ref.free(); out.free()


def test_pipelined_engine_tiles_at_batch_size(full_blob):
    """A depth-2 engine tunes its pointwise tiles by least padded work (exact 80- / 112-column tiles ...), a choice no
    serial engine makes and one that only materialises at a batch large enough to fill the GPU: 64 clips through such an
    engine must match a serial engine's logits (other tiles = another summation order) and the oracle.  (Which tiles win
    depends on the box; test_pw_gemm_kernel_variants_agree pins each of them explicitly.)"""
    xh = sm.synth_clips(64, 144000, 48000)
    a = host.HipClassifier(full_blob, max_batch=64)
    b = host.HipClassifier(full_blob, max_batch=64, depth=2, lanes=1)
    x, out = _DevBuf(xh.nbytes), _DevBuf(64 * 6522 * 4)
    try:
        x.upload(xh)
        a.predict_device(x.at(0), 64, out.at(0)); a.synchronize()
        ra = out.download((64, 6522)).copy()
        b.predict_device(x.at(0), 64, out.at(0)); b.synchronize()
        rb = out.download((64, 6522)).copy()
        assert np.isfinite(rb).all() and np.abs(ra - rb).max() < 1e-4
        assert_parity(rb[:3], Interpreter(full_blob).invoke(xh[:3])[0])
    finally:
        a.close(); b.close()
        x.free(); out.free()


# ---- geometry sweep: spectrogram sizes, kernel sizes, strides and widths the v2.4 topology does not have, so that every
# tile-shape / halo / padding-row / ragged-edge branch of the fused kernels meets the oracle at least once
def _geo_cfg(i):
    rng = np.random.default_rng(1000 + i)
    n_mels = int(rng.choice([20, 24, 33, 40, 48]))
    hop = int(rng.choice([94, 96, 120]))
    frames = int(rng.choice([37, 50, 64, 91, 123]))
    n_samples = 512 + hop * (frames - 1)
    specs = (sm.SpecConfig(512, hop, 0.0, 3000.0), sm.SpecConfig(512, hop, 500.0, 15000.0))
    blocks, c = [(1, 3, 1, int(rng.choice([8, 12])), 1)], None
    for _ in range(int(rng.integers(3, 6))):
        blocks.append((int(rng.choice([4, 6])), int(rng.choice([3, 5])), int(rng.choice([1, 1, 2])),
                       int(rng.choice([12, 16, 20, 24, 36])), int(rng.integers(1, 3))))
    return sm.tiny_config(n_samples=n_samples, n_mels=n_mels, specs=specs, blocks=tuple(blocks), stem=int(rng.choice([8, 16, 32, 32])),
                          top=int(rng.choice([48, 64])), n_classes=int(rng.choice([17, 50, 101])), seed=77 + i)


@pytest.mark.parametrize("i", range(8))
def test_geometry_sweep_vs_oracle(built_lib, i):
    cfg = _geo_cfg(i)
    blob = sm.build_model(cfg)
    x = sm.synth_clips(5, cfg.n_samples, cfg.sample_rate, first=10 * i)
    ref = Interpreter(blob).invoke(x)[0]
    for opts in (dict(), dict(lanes=1, autotune=False)):      # autotuned shapes, and the cost-model / heuristic defaults
        c = host.HipClassifier(blob, max_batch=5 if "lanes" in opts else 64, **opts)
        try:
            got = c.predict_batch(x.reshape(-1), 5)
        finally:
            c.close()
        assert_parity(got, ref)


@pytest.mark.parametrize("i", [0, 3, 5])
def test_blocks_without_squeeze_excite_vs_oracle(built_lib, i):
    """Inverted-residual blocks with no squeeze-excite (se_form "none"): the fused expand + depthwise kernels then run without
    their per-tile channel sums (partial == nullptr: no sums pass, and in the chunk-loop form no sums hand-over between the
    barriers) and the projection without a fused scale - on the geometry sweep's layer mix (3x3 / 5x5, stride 1 / 2, channel
    counts on both sides of the small-K limit, tail chunks, border tiles)."""
    import dataclasses
    cfg = dataclasses.replace(_geo_cfg(i), se_form="none")
    blob = sm.build_model(cfg)
    x = sm.synth_clips(5, cfg.n_samples, cfg.sample_rate, first=7 * i)
    ref = Interpreter(blob).invoke(x)[0]
    for opts in (dict(), dict(lanes=1, autotune=False)):
        c = host.HipClassifier(blob, max_batch=8, **opts)
        try:
            kinds = [s["kernel"] for s in c.describe()["steps"]]
            assert "se" not in kinds and "expand_dw" in kinds and not any(k.startswith("generic") for k in kinds), kinds
            got = c.predict_batch(x.reshape(-1), 5)
        finally:
            c.close()
        assert_parity(got, ref)


# ---- FFT front-end (stft.hip): serves the magnitude (COMPLEX_ABS) graph and, on request, the real-part graph
FFT_TINY_SPECS = (sm.SpecConfig(512, 94, 0.0, 3000.0), sm.SpecConfig(512, 94, 500.0, 15000.0))


@pytest.mark.parametrize("mode", ["abs", "real"])
def test_fft_frontend_tiny_vs_oracle(built_lib, mode):
    cfg = sm.tiny_config(complex_mode=mode, specs=FFT_TINY_SPECS)
    blob = sm.build_model(cfg)
    x = sm.synth_clips(3, cfg.n_samples, cfg.sample_rate)
    x[2] = 0.0                                                  # digital silence: the bins that cancel to ~0
    ref = Interpreter(blob).invoke(x)[0]
    c = host.HipClassifier(blob, max_batch=4, frontend_fft=1)
    try:
        assert "stft" in [s["kernel"] for s in c.describe()["steps"]]
        assert_parity(c.predict_batch(x.reshape(-1), 3), ref)
    finally:
        c.close()


def test_fft_frontend_full_magnitude_graph_vs_oracle(built_lib):
    blob = sm.build_model(sm.SynthConfig(complex_mode="abs"))
    x = sm.synth_clips(3, 144000, 48000)
    x[2] = 0.0
    ref = Interpreter(blob).invoke(x)[0]
    c = host.HipClassifier(blob, max_batch=8)                   # magnitude graphs select the FFT path themselves
    try:
        got = c.predict_batch(x.reshape(-1), 3)
        assert_parity(got, ref)
        # ragged batch through the lanes / chunking
        x9 = sm.synth_clips(9, 144000, 48000)
        g9 = c.predict_batch(x9.reshape(-1), 9)
        assert np.abs(g9[:2] - got[:2]).max() == 0.0
    finally:
        c.close()


def test_fft_frontend_matches_folded_gemm_on_the_real_part_graph(full_clf, full_blob):
    x = sm.synth_clips(4, 144000, 48000)
    x[3] = 0.0
    a = full_clf.predict_batch(x.reshape(-1), 4)                 # default: FFT front-end
    assert "stft" in [s["kernel"] for s in full_clf.describe()["steps"]]
    c = host.HipClassifier(full_blob, max_batch=8, frontend_fft=0)   # folded-GEMM front-end
    assert "stft" not in [s["kernel"] for s in c.describe()["steps"]]
    try:
        b = c.predict_batch(x.reshape(-1), 4)
    finally:
        c.close()
    assert (a.argmax(1) == b.argmax(1)).all()
    assert np.abs(sig(a) - sig(b)).max() <= 1e-4


def test_reference_benchmark_input_silence(full_clf, full_blob):
    """cmd/benchmark/benchmark.go:99-101 feeds 144000 zeros; the graph's min/max normalisation maps a
    constant clip to -1 everywhere (0/(0+1e-6) - 0.5)*2, which must not produce NaN/Inf."""
    z = np.zeros(144000, np.float32)
    got = full_clf.predict(z)
    ref = Interpreter(full_blob).invoke(z)[0][0]
    assert np.isfinite(got).all()
    assert_parity(got[None], ref[None])


def test_single_clip_predict_equals_batch_row(full_clf):
    x = sm.synth_clips(5, 144000, 48000)
    b = full_clf.predict_batch(x.reshape(-1), 5)
    for i in (0, 4):
        assert np.array_equal(full_clf.predict(x[i]), b[i]), "per-clip result must not depend on batch composition"


def test_chunking_over_max_batch_and_ragged_tail(tiny_clf, tiny_blob, tiny_cfg):
    n = 19                                    # max_batch 8 -> chunks 8, 8, 3
    x = sm.synth_clips(n, tiny_cfg.n_samples, tiny_cfg.sample_rate)
    got = tiny_clf.predict_batch(x.reshape(-1), n)
    assert_parity(got, Interpreter(tiny_blob).invoke(x)[0])
    perm = np.random.default_rng(0).permutation(n)
    assert np.array_equal(tiny_clf.predict_batch(x[perm].reshape(-1), n), got[perm])   # order independence, bit-exact


def test_host_batch_split_and_ragged_sizes(built_lib):
    """Host-pointer calls of >= 128 clips are split in two (copy/compute overlap), larger ones chunked by max_batch: every
    size must give the rows a small call gives, embeddings included."""
    cfg = sm.tiny_config(emit_embeddings=True)
    blob = sm.build_model(cfg)
    x = sm.synth_clips(300, cfg.n_samples, cfg.sample_rate)
    c = host.HipClassifier(blob, max_batch=256)
    try:
        ref_l, ref_e = c.predict_with_embeddings(x[:7].reshape(-1), 7) if False else (None, None)
        small = np.concatenate([c.predict_batch(x[i:i + 20].reshape(-1), 20) for i in range(0, 300, 20)])
        for n in (127, 128, 131, 256, 259, 300):
            got = c.predict_batch(x[:n].reshape(-1), n)
            assert got.shape == (n, cfg.n_classes)
            assert np.abs(got - small[:n]).max() < 1e-5, n       # other batch sizes may pick other tiles: fp32 reorder only
            assert (got.argmax(1) == small[:n].argmax(1)).all()
        # embeddings through the lane / split paths (the embedding tensor lives in a lane's own arena region)
        se = np.concatenate([c.predict_batch(x[i:i + 20].reshape(-1), 20, want_embeddings=True)[1] for i in range(0, 140, 20)])
        for n in (40, 131):
            lg, em = c.predict_batch(x[:n].reshape(-1), n, want_embeddings=True)
            assert em.shape == (n, c.emb_dim) and np.abs(em - se[:n]).max() < 1e-5, n
            assert np.abs(lg - small[:n]).max() < 1e-5, n
        # device top-k on a lane-sized batch
        conf, idx = c.predict_topk(x[:50].reshape(-1), 50, 5, 0, 1.0)
        c2, i2 = c.postprocess_topk(small[:50], k=5)
        assert np.array_equal(idx, i2) and np.abs(conf - c2).max() < 1e-6
        pcm = np.clip(np.round(x[:131] * 32767.0), -32768, 32767).astype(np.int16)
        a = c.predict_pcm16(pcm.reshape(-1), 131)
        b = c.predict_batch((pcm.astype(np.float32) / 32768.0).reshape(-1), 131)
        assert np.abs(a - b).max() < 1e-5
    finally:
        c.close()


def test_lanes_full_model_equal_small_batches(full_blob):
    """Batches of >= 32 clips run as two concurrent lanes (each with its own arena region): every row must match what a
    small, single-lane call computes.  (Lanes once shared one liveness-reused layout: out-of-phase lanes overwrote each
    other's live tensors - only caught by comparing against small batches, the outputs stayed finite.)"""
    x = sm.synth_clips(70, 144000, 48000)
    c = host.HipClassifier(full_blob, max_batch=128)
    try:
        small = np.concatenate([c.predict_batch(x[i:i + 10].reshape(-1), 10) for i in range(0, 70, 10)])
        for n in (33, 64, 70):
            for rep in range(2):
                got = c.predict_batch(x[:n].reshape(-1), n)
                assert (got.argmax(1) == small[:n].argmax(1)).all(), (n, rep)
                assert np.abs(sig(got) - sig(small[:n])).max() <= 1e-4, (n, rep)
                assert np.abs(got - small[:n]).max() < 1e-3, (n, rep)
    finally:
        c.close()


def test_eight_wave_blocks_of_the_small_k_form_match(full_blob, monkeypatch):
    """k_expand_dw_sk<..., NW = 8> (two waves per SIMD sharing one expanded footprint: shape indices 14-21) is an autotune
    candidate of the early layers; forced onto b2 / b3 / b4 - 3 x 3 stride 2, 3 x 3 stride 1 and 5 x 5 stride 2, K widths 16 and
    24, three and five channel chunks - it must reproduce the four-wave kernels' results to rounding and the oracle's top-1."""
    x = sm.synth_clips(12, 144000, 48000)
    ref_c = host.HipClassifier(full_blob, max_batch=16, autotune=False)
    try:
        ref = ref_c.predict_batch(x.reshape(-1), 12)
    finally:
        ref_c.close()
    monkeypatch.setenv("BNHIP_EXPDW_FORCE", "b2/expand+dw=20,b3/expand+dw=16,b4/expand+dw=21")
    c = host.HipClassifier(full_blob, max_batch=16)
    try:
        shapes = {s["name"]: s["shape"] for s in c.describe()["steps"] if s["kernel"] == "expand_dw"}
        assert (shapes["b2/expand+dw"], shapes["b3/expand+dw"], shapes["b4/expand+dw"]) == (20, 16, 21), shapes
        got = c.predict_batch(x.reshape(-1), 12)
    finally:
        c.close()
    assert (got.argmax(1) == ref.argmax(1)).all()
    assert np.abs(got - ref).max() < 2e-4 and np.abs(sig(got) - sig(ref)).max() <= 1e-5


def test_opt_in_graph_replay_matches_eager(tiny_blob, tiny_cfg):
    """"graphs":1 replays the plan as a hipGraph from the third identical call on; results must not change (single lane and
    two lanes)."""
    x = sm.synth_clips(40, tiny_cfg.n_samples, tiny_cfg.sample_rate)
    e = host.HipClassifier(tiny_blob, max_batch=64)
    g = host.HipClassifier(tiny_blob, max_batch=64, graphs=True)
    try:
        for n in (3, 40):
            ref = e.predict_batch(x[:n].reshape(-1), n)
            for rep in range(4):
                got = g.predict_batch(x[:n].reshape(-1), n)
                assert np.abs(got - ref).max() < 1e-5, (n, rep)
    finally:
        e.close(); g.close()


def test_determinism(full_clf):
    x = sm.synth_clips(3, 144000, 48000, first=100)
    a = full_clf.predict_batch(x.reshape(-1), 3)
    b = full_clf.predict_batch(x.reshape(-1), 3)
    assert np.array_equal(a, b)


def test_input_size_mismatch_is_an_error(tiny_clf, tiny_cfg):
    with pytest.raises(host.HipError, match="input size mismatch"):
        tiny_clf.predict(np.zeros(tiny_cfg.n_samples - 1, np.float32))
    with pytest.raises(host.HipError, match="input size mismatch"):
        tiny_clf.predict_batch(np.zeros(tiny_cfg.n_samples * 2 + 3, np.float32), 2)


def test_affine_invariance_property(full_clf):
    """The per-clip min/max normalisation makes logits invariant to gain and DC offset (up to rounding):
    a size-independent property checked at the full clip size."""
    x = sm.synth_clips(2, 144000, 48000, first=7)
    a = full_clf.predict_batch(x.reshape(-1), 2)
    b = full_clf.predict_batch((0.25 * x + 0.125).reshape(-1), 2)   # exact in fp32 (powers of two)
    assert_parity(b, a, tol=1e-4)


def test_pcm16_path_matches_float_path(full_clf):
    x = sm.synth_clips(2, 144000, 48000)
    pcm = np.clip(np.round(x * 32767), -32768, 32767).astype(np.int16)
    got = full_clf.predict_pcm16(pcm.reshape(-1), 2)
    f = G.pcm_to_f32(pcm.tobytes(), 16).reshape(2, -1)       # process.go:491-495 restatement
    want = full_clf.predict_batch(f.reshape(-1), 2)
    assert np.array_equal(got, want)


def test_pcm24_pcm32_paths_match_float_path(full_clf):
    """a1: the 24- and 32-bit branches of ConvertToFloat32 (convert/pcm.go:242-268) on the device, bit-exact against the
    oracle's restatement fed through the float entry; an unsupported depth is rejected like the reference does."""
    x = sm.synth_clips(2, 144000, 48000)
    i24 = np.clip(np.round(x.astype(np.float64) * 8388607), -8388608, 8388607).astype(np.int32)
    i24[0, :4] = [-8388608, 8388607, -1, 0]                               # sign-extension edges
    b = i24.astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3].tobytes()   # packed 3-byte little-endian
    got = full_clf.predict_pcm(b, 24, 2)
    want = full_clf.predict_batch(G.pcm_to_f32(b, 24), 2)
    assert np.array_equal(got, want)
    i32 = np.clip(np.round(x.astype(np.float64) * 2147483647), -2147483648, 2147483647).astype("<i4")
    i32[1, :3] = [-2147483648, 2147483647, 16777217]                      # float32(int32) rounding edge
    got = full_clf.predict_pcm(i32.tobytes(), 32, 2)
    want = full_clf.predict_batch(G.pcm_to_f32(i32.tobytes(), 32), 2)
    assert np.array_equal(got, want)
    with pytest.raises(host.HipError):
        full_clf.predict_pcm(bytes(2 * 144000), 8, 2)


def test_pcm_topk_in_one_call_equals_the_two_step_path(full_clf):
    """bnhip_predict_pcm_topk (one analysis window's whole (*BirdNET).Predict: convert -> classifier -> sigmoid(sensitivity) ->
    top-10, analyze.go:25-110) == bnhip_predict_pcm followed by bnhip_postprocess_topk, bit for bit, at every bit depth and for
    the softmax / plain-sigmoid activations; an unsupported depth is rejected."""
    x = sm.synth_clips(3, 144000, 48000, first=3)
    i16 = np.clip(np.round(x * 32767), -32768, 32767).astype("<i2")
    i32 = np.clip(np.round(x.astype(np.float64) * 2147483647), -2147483648, 2147483647).astype("<i4")
    i24 = (i32 >> 8).astype("<i4").view(np.uint8).reshape(-1, 4)[:, :3].tobytes()
    for raw, bits in ((i16.tobytes(), 16), (i24, 24), (i32.tobytes(), 32)):
        logits = full_clf.predict_pcm(raw, bits, 3)
        for act, sens in ((0, 1.0), (0, 1.35), (1, 1.0), (2, 1.0)):
            conf, idx = full_clf.predict_pcm_topk(raw, bits, 3, 10, act, sens)
            c2, i2 = full_clf.postprocess_topk(logits, 10, act, sens)
            assert np.array_equal(conf, c2) and np.array_equal(idx, i2), (bits, act, sens)
            assert (idx[:, 0] == logits.argmax(1)).all() and (np.diff(conf, axis=1) <= 0).all()
    with pytest.raises(host.HipError, match="unsupported bit depth"):
        full_clf.predict_pcm_topk(bytes(3 * 144000), 8, 3)
    with pytest.raises(host.HipError, match="size mismatch"):
        full_clf.predict_pcm_topk(bytes(10), 16, 3)


def test_banded_mel_matches_gemm_mel(built_lib, full_blob):
    """The fused banded mel + pow + store kernel against the dense mel GEMM + finish pair it replaces (same products, other
    summation order), and both against the oracle."""
    x = sm.synth_clips(3, 144000, 48000)
    x[2] = 0.0                                                            # digital silence: the near-empty-bin case
    a = host.HipClassifier(full_blob, max_batch=4)
    os.environ["BNHIP_NO_MEL_BANDED"] = "1"
    try:
        b = host.HipClassifier(full_blob, max_batch=4)
    finally:
        del os.environ["BNHIP_NO_MEL_BANDED"]
    assert "melband0+1" in [s["name"] for s in a.describe()["steps"]]
    assert "melband0+1" not in [s["name"] for s in b.describe()["steps"]]
    ya, yb = a.predict_batch(x.reshape(-1), 3), b.predict_batch(x.reshape(-1), 3)
    a.close(); b.close()
    assert_parity(ya, yb, tol=2e-5)
    assert_parity(ya, Interpreter(full_blob).invoke(x)[0])


@pytest.mark.parametrize("dense_mel", [False, True])
def test_three_channel_front_end_vs_oracle(built_lib, dense_mel):
    """An odd channel count: channels 0+1 leave the mel stage as float2 pixels, channel 2 alone (the single-channel
    instantiations of k_mel_banded / k_mel_finish, stride-3 NHWC stores), and the stem sees Cin = 3 (k_conv_direct)."""
    specs = (sm.SpecConfig(512, 94, 0.0, 3000.0), sm.SpecConfig(512, 94, 500.0, 15000.0), sm.SpecConfig(512, 94, 200.0, 8000.0))
    cfg = sm.tiny_config(specs=specs)
    blob = sm.build_model(cfg)
    if dense_mel:
        os.environ["BNHIP_NO_MEL_BANDED"] = "1"
    try:
        c = host.HipClassifier(blob, max_batch=4)
    finally:
        os.environ.pop("BNHIP_NO_MEL_BANDED", None)
    names = [s["name"] for s in c.describe()["steps"]]
    assert ("melband2" in names) != dense_mel and ("melspec2" in names) == dense_mel
    x = sm.synth_clips(3, cfg.n_samples, cfg.sample_rate)
    x[1] = 0.0
    got = c.predict_batch(x.reshape(-1), 3)
    c.close()
    assert_parity(got, Interpreter(blob).invoke(x)[0])


def test_pw_gemm_kernel_variants_agree(built_lib, full_blob):
    """Every pointwise layer forced onto each k_pw_gemm flavour in turn (64/128-row tiles; 3/4 = the software-pipelined
    k_pw_pipe, which falls back per layer where K is not a whole number of slabs) must give the same logits up to the
    summation order, and match the oracle.  Runs in a subprocess per setting: the switch is read once per process."""
    import subprocess, sys, json
    code = (
        "import sys, json, numpy as np; sys.path.insert(0, %r)\n"
        "import birdnet_go_amd\n"
        "from birdnet_go_amd import host, synth_model as sm\n"
        "blob = open(%r, 'rb').read()\n"
        "c = host.HipClassifier(blob, max_batch=8, autotune=False)\n"
        "x = sm.synth_clips(5, 144000, 48000)\n"
        "np.save(sys.argv[1], c.predict_batch(x.reshape(-1), 5))\n"
    )
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        bp = os.path.join(td, "m.tflite")
        open(bp, "wb").write(full_blob)
        outs = {}
        # (wm, nt): nt 0 = the built-in width rule; 5..8 = the wide tiles a pipelined engine's work-based tuner may pick
        # (64-row tiles with nt >= 7 once overflowed the epilogue staging area)
        settings = [(1, 0), (2, 0), (3, 0), (4, 0), (1, 5), (1, 7), (1, 8), (2, 6), (2, 7), (2, 8)]
        for wm, nt in settings:
            env = dict(os.environ, BNHIP_PW_WM=str(wm))
            if nt:
                env["BNHIP_PW_NT"] = str(nt)
            op = os.path.join(td, f"o{wm}_{nt}.npy")
            subprocess.run([sys.executable, "-c", code % (root, bp), op], check=True, env=env, timeout=300)
            outs[(wm, nt)] = np.load(op)
    ref = Interpreter(full_blob).invoke(sm.synth_clips(5, 144000, 48000))[0]
    for k in settings:
        assert_parity(outs[k], ref)
        assert_parity(outs[k], outs[(1, 0)], tol=2e-5)


def test_embeddings_output(built_lib):
    cfg = sm.tiny_config(emit_embeddings=True)
    blob = sm.build_model(cfg)
    clf = host.HipClassifier(blob, max_batch=4)
    x = sm.synth_clips(3, cfg.n_samples, cfg.sample_rate)
    lg, em = clf.predict_batch(x.reshape(-1), 3, want_embeddings=True)
    ref_l, ref_e = Interpreter(blob).invoke(x)
    assert_parity(lg, ref_l)
    assert em.shape == (3, 64) and np.abs(em - ref_e).max() < 5e-3     # bat f32-vs-f32 drift bound, parity test :263-265
    assert np.abs(em - ref_e).max() < 1e-4
    l1, e1 = clf.predict_with_embeddings(x[1])
    assert np.array_equal(l1, lg[1]) and np.array_equal(e1, em[1])
    clf.close()
    clf.close()     # idempotent Close()


def test_postprocess_sigmoid_topk_matches_go_restatement(full_clf):
    rng = np.random.default_rng(5)
    lg = (rng.standard_normal((7, 6522)) * 4 - 3).astype(np.float32)
    for sens in (1.0, 1.5, 0.5):
        conf, idx = full_clf.postprocess_topk(lg, k=10, activation=0, sensitivity=sens)
        for r in range(7):
            want_c, want_i = G.topk(G.sigmoid_sensitivity(lg[r], sens), 10)
            assert np.array_equal(conf[r], want_c)                     # bit-exact confidences
            assert set(idx[r]) == set(want_i) or len(set(want_c)) < 10  # ties: order unspecified in the reference
    conf, idx = full_clf.postprocess_topk(lg, k=10, activation=2)
    for r in range(7):
        want_c, _ = G.topk(G.sigmoid_f32div(lg[r]), 10)
        assert np.array_equal(conf[r], want_c)


def test_postprocess_softmax_bit_exact(full_clf):
    rng = np.random.default_rng(6)
    lg = (rng.standard_normal((3, 6522)) * 3).astype(np.float32)
    conf, idx = full_clf.postprocess_topk(lg, k=5, activation=1)
    for r in range(3):
        sm_ref = G.softmax(lg[r])
        want_c, want_i = G.topk(sm_ref, 5)
        assert np.array_equal(conf[r], want_c) and np.array_equal(idx[r], want_i)


def test_predict_topk_fused(full_clf):
    x = sm.synth_clips(3, 144000, 48000)
    lg = full_clf.predict_batch(x.reshape(-1), 3)
    conf, idx = full_clf.predict_topk(x.reshape(-1), 3, k=10, sensitivity=1.0)
    c2, i2 = full_clf.postprocess_topk(lg, k=10)
    assert np.array_equal(conf, c2) and np.array_equal(idx, i2)
    bn = host.BirdNET(full_clf, [f"Species{i}_Common{i}" for i in range(6522)], sensitivity=1.0)
    res = bn.predict(x[0])
    assert len(res) == 10 and res[0][0] == f"Species{idx[0][0]}_Common{idx[0][0]}"
    assert all(res[i][1] >= res[i + 1][1] for i in range(9))
    with pytest.raises(host.HipError, match="label count"):
        host.BirdNET(full_clf, ["a"] * 10)


def test_results_handoff_batch(full_clf):
    """one device call -> one Results message per clip, labels paired from the device top-K."""
    from birdnet_go_amd import results as R
    labels = [f"Species{i}_Common{i}" for i in range(6522)]
    bn = host.BirdNET(full_clf, labels, sensitivity=1.0)
    x = sm.synth_clips(5, 144000, 48000)
    d = R.BatchDispatcher(bn, "BirdNET_GLOBAL_6K_V2.4", R.ResultsQueue(size=4))
    sent = d.dispatch(x, start_times=[1.5 * i for i in range(5)], source="soundscape.wav")
    assert sent == 4 and d.queue.drops() == {("soundscape.wav", "BirdNET_GLOBAL_6K_V2.4"): 1}   # bounded queue: 5th dropped
    lg = full_clf.predict_batch(x.reshape(-1), 5)
    for i in range(4):
        m = d.queue.get()
        assert m.start_time == 1.5 * i and len(m.results) == 10
        assert m.results[0].species == labels[int(np.argmax(lg[i]))]
        want = G.sigmoid_sensitivity(lg[i][np.argmax(lg[i])][None], 1.0)[0]
        assert m.results[0].confidence == float(want)
    snap = d.counters.peek_all()["BirdNET_GLOBAL_6K_V2.4"]
    assert snap["invoke_count"] == 1 and snap["recent_p95_us"] > 0


# ---- ultrasonic frame-CV (filter_test.go signals), GPU float64 vs the Go restatement
SR, N = 256000, 144000


def _us_signals():
    t = np.arange(N) / SR
    tone = 0.01 * np.sin(2 * np.pi * 40000.0 * t)
    burst = np.zeros(N)
    i = np.arange(N // 3, 2 * N // 3)
    burst[i] = 0.5 * np.sin(2 * np.pi * 45000.0 * i / SR)
    rng = np.random.default_rng(4321)
    chirp = rng.normal(0, 0.01, N)
    for k in range(0, N, SR // 10):                       # FM sweep 80->25 kHz, 5 ms, 10 Hz repetition (SURVEY 8d cfg 4)
        m = np.arange(min(int(0.005 * SR), N - k))
        ph = 2 * np.pi * (80000.0 * m / SR + 0.5 * (25000.0 - 80000.0) / 0.005 * (m / SR) ** 2)
        chirp[k:k + m.size] += 0.3 * np.sin(ph)
    return np.stack([tone, burst, chirp, np.zeros(N)])


def test_us_frame_cv_matches_go_restatement():
    s = _us_signals()
    cv, ok = host.us_frame_cv(s, SR)
    assert ok.all()
    for i in range(s.shape[0]):
        want, wok = G.us_frame_cv(s[i], SR)
        assert wok and abs(cv[i] - want) <= 1e-9 * max(1.0, abs(want)), (i, cv[i], want)
    assert cv[0] < 0.15 and cv[1] > 0.15 and cv[3] == 0.0      # filter_test.go:23-65 verdicts; silence -> mean<=0 -> 0
    cv10, _ = host.us_frame_cv(s[1:2] * 10.0, SR)
    assert abs(cv10[0] - cv[1]) < 0.01                          # scale invariance, filter_test.go:180-207


def test_us_other_geometries():
    s = _us_signals()[:2, :40000]
    for fft, hop in ((1024, 512), (4096, 1024), (256, 256)):
        cv, ok = host.us_frame_cv(s, SR, fft_size=fft, hop=hop, split_hz=30000)
        for i in range(2):
            want, wok = G.us_frame_cv(s[i], SR, fft, hop, 30000)
            assert ok[i] == wok and abs(cv[i] - want) <= 1e-9 * max(1.0, abs(want))


# ---- secondary models: bat head (a12) and range filter (8f row 3) as dense-only graphs
def test_bat_two_stage_pipeline(built_lib):
    cfg = sm.tiny_config(emit_embeddings=True)
    backbone_blob = sm.build_model(cfg)
    head_blob = sm.build_dense_model([64, 30], seed=11)
    backbone = host.HipClassifier(backbone_blob, max_batch=4)
    labels = [f"Bat{i}" for i in range(30)]
    head = host.CustomClassifier(head_blob, labels, max_batch=4)
    assert (head.input_dim(), head.num_classes(), head.labels()[3]) == (64, 30, "Bat3")
    bat = host.Bat(backbone, head, threshold=0.3)
    x = sm.synth_clips(2, cfg.n_samples, cfg.sample_rate)
    got = bat.predict(x[1])
    # oracle pipeline: backbone embedding -> head logits -> float32-division sigmoid (onnx/postprocess.go:8-10)
    emb = Interpreter(backbone_blob).invoke(x[1])[1][0]
    scores = G.sigmoid_f32div(Interpreter(head_blob).invoke(emb[None, :])[0][0])
    want = [(labels[i], float(scores[i])) for i in np.argsort(-scores, kind="stable") if scores[i] >= 0.3][:10]
    assert [g[0] for g in got] == [w[0] for w in want]
    assert np.allclose([g[1] for g in got], [w[1] for w in want], atol=1e-5)
    with pytest.raises(host.HipError, match="input size mismatch"):
        head.predict_embedding(np.zeros(63, np.float32))
    with pytest.raises(host.HipError, match="label count"):
        host.CustomClassifier(head_blob, labels[:5])
    plain = host.HipClassifier(sm.build_model(sm.tiny_config()), max_batch=2)
    with pytest.raises(host.HipError, match="no embedding output"):
        host.Bat(plain, head)
    for c in (backbone, plain):
        c.close()
    head.close()


def test_range_filter_fp16_batch(built_lib):
    blob = sm.build_dense_model([3, 64, 128, 6522], final_sigmoid=True, fp16_weights=True, input_scale=[90.0, 180.0, 48.0])
    rf = host.RangeFilter(blob, max_batch=64)
    assert rf.num_species() == 6522
    rng = np.random.default_rng(3)
    pts = np.stack([rng.uniform(-90, 90, 150), rng.uniform(-180, 180, 150), rng.integers(1, 49, 150)], 1).astype(np.float32)
    got = rf.predict_batch(pts.reshape(-1), 150).reshape(150, 6522)       # 150 > max_batch: chunked
    ref = Interpreter(blob).invoke(pts)[0]
    assert np.abs(got - ref).max() < 2e-6
    one = rf.predict(*pts[7])
    assert np.array_equal(one, got[7])
    with pytest.raises(host.HipError, match="input size mismatch"):
        rf.predict_batch(np.zeros(7, np.float32), 2)
    rf.close()


def test_streamed_split_gemm_equals_the_staged_one(full_blob, monkeypatch):
    """k_pw_b16 in its six-product form (csrc/pw_b16.hip: A fragments straight from global memory two slabs ahead, the three
    weight planes of a slab through LDS) computes what k_pw_bx3 computes - same split, same product order - so forcing every
    split-bf16 layer onto it changes no bit: 64-row tiles (12 clips, tile shrinking) and 128-row tiles (64 clips)."""
    import ctypes
    lib = host.load_library()
    lib.bnhip_debug_pw_b16_launches.restype = ctypes.c_long
    for n in (12, 64):
        x = sm.synth_clips(n, 144000, 48000)
        out = {}
        for mode in ("2", "0"):
            monkeypatch.setenv("BNHIP_PW_B16", mode)
            c = host.HipClassifier(full_blob, max_batch=n, autotune=False, lanes=1)
            before = lib.bnhip_debug_pw_b16_launches()
            out[mode] = c.predict_batch(x.reshape(-1), n).copy()
            used = lib.bnhip_debug_pw_b16_launches() - before
            c.close()
            assert (used >= 15) if mode == "2" else (used == 0), (n, mode, used)
        assert np.array_equal(out["2"], out["0"]), (n, np.abs(out["2"] - out["0"]).max())


def test_small_calls_take_the_latency_kernels_and_keep_the_bits(full_blob, monkeypatch):
    """One clip per Predict is the product's call pattern.  Calls of a few clips route their long-K layers (projections, dense head)
    to k_pw_lat (one barrier per four K slabs, weight fragments from a register ring, the operand split shared by a block's waves)
    and their min / max to the 16-blocks-per-clip form with a self-resetting arrival counter.  Both are bit-identical to what a
    large call runs: the same clips inside a 40-clip call give the same logits, twice in a row (the counter resets), and with
    the latency kernel switched off."""
    import ctypes
    lib = host.load_library()
    lib.bnhip_debug_pw_lat_launches.restype = ctypes.c_long
    x = sm.synth_clips(40, 144000, 48000, first=77)
    x[3] = 0.0                                               # (silence: min == max)
    c = host.HipClassifier(full_blob, max_batch=64, autotune=False)
    try:
        big = c.predict_batch(x.reshape(-1), 40).copy()
        for n in (1, 3, 8, 16):
            for rep in range(2):
                before = lib.bnhip_debug_pw_lat_launches()
                got = c.predict_batch(x[:n].reshape(-1), n)
                assert lib.bnhip_debug_pw_lat_launches() - before >= 10, n
                assert np.array_equal(got, big[:n]), (n, rep, np.abs(got - big[:n]).max())
            got = c.predict_batch(x[5:5 + n].reshape(-1), n)            # other clips through the same scratch
            assert np.array_equal(got, big[5:5 + n]), n
    finally:
        c.close()
    monkeypatch.setenv("BNHIP_PW_LAT", "0")
    c = host.HipClassifier(full_blob, max_batch=64, autotune=False)
    try:
        before = lib.bnhip_debug_pw_lat_launches()
        assert np.array_equal(c.predict_batch(x[:3].reshape(-1), 3), big[:3]) and lib.bnhip_debug_pw_lat_launches() == before
        assert np.array_equal(c.predict_batch(x[:16].reshape(-1), 16), big[:16])      # (16 clips: the two-column-tiles-per-wave form was behind `big`)
    finally:
        c.close()
