"""BASELINE configs[4] (Google Perch v2): the log-mel front-end variant and an EfficientNet-B3-shaped stack at Perch's published
dimensions (synth_model.perch_config: 160000 samples in, 500 x 128 time-major log-mel image, 16 x 4 x 1536 spatial
embedding, 1536-d embedding, 14795 logits - the shapes the reference lists in internal/inference/onnx/classifier.go:495-505).
The real artefact is ONNX and absent from the snapshot, so these pin the engine against the oracle on a stand-in, not
Perch's numerics.  Softmax / top-k follow perchSoftmax (internal/classifier/perch_onnx.go:315-335)."""
import numpy as np
import pytest

from birdnet_go_amd import host, synth_model as sm
from oracle import gofuncs as G
from oracle.interp import Interpreter


def oracle_logits_emb(blob, x):
    """(logits, embedding) of the oracle in the graph's own output order: Perch v2 lists embedding first and logits fourth
    (internal/inference/onnx/classifier.go:495-505, perch_onnx.go:28), two-output graphs logits first."""
    outs = Interpreter(blob).invoke(x)
    return (outs[3], outs[0]) if len(outs) == 4 else (outs[0], outs[1] if len(outs) > 1 else None)


def softmax64(v):
    z = np.asarray(v, np.float64)
    e = np.exp(z - z.max(axis=-1, keepdims=True))
    return e / e.sum(axis=-1, keepdims=True)


def test_perch_like_plans_with_log_mel_front_end(built_lib):
    """CPU: the matcher accepts the variant (clip PAD, frames shorter than the FFT, no normalisation, MAXIMUM/LOG/MUL
    compression, time-major image) and the Perch-size graph plans with every op on a fused kernel."""
    for cfg, shape in ((sm.tiny_perch_config(), (8000, 50, 64)), (sm.perch_config(), (160000, 14795, 1536))):
        c = host.HipClassifier(sm.build_model(cfg), plan_only=True)
        try:
            assert (c.n_samples, c.num_species(), c.emb_dim) == shape
            if cfg.perch_outputs:                                    # detection.go:107-111: logits are output 3, the embedding output 0
                d = c.describe()
                assert (d["logits_output"], d["embedding_output"]) == (3, 0)
            kinds = [s["kernel"] for s in c.describe()["steps"]]
            assert kinds[:2] == ["stft", "frontend"] and "clip_minmax" not in kinds      # raw samples: no min/max pass
            assert "elementwise" not in kinds and not any(k.startswith("generic") for k in kinds)
        finally:
            c.close()


def test_perch_like_oracle_shapes():
    """CPU: the stand-in's tensors have the shapes the reference lists for Perch v2's outputs."""
    from oracle.tflite_reader import read_model
    m = read_model(sm.build_model(sm.perch_config()))
    # the four outputs, in Perch v2's order (internal/inference/onnx/classifier.go:495-505)
    assert [tuple(int(d) for d in m.tensors[t].shape) for t in m.outputs] == [(1, 1536), (1, 16, 4, 1536), (1, 500, 128), (1, 14795)]


def test_front_end_variant_rejects_what_it_cannot_do(built_lib):
    """CPU: a selector that reaches beyond the padded clip is a malformed graph, not an out-of-bounds read."""
    cfg = sm.tiny_perch_config(pad=(0, 0))                      # builder sizes the selector from n + pads: consistent
    host.HipClassifier(sm.build_model(cfg), plan_only=True).close()
    blob = bytearray(sm.build_model(sm.tiny_perch_config()))
    # shrink the clip PAD constant [[0,0],[80,80]] -> [[0,0],[0,0]] in place: the selector now overruns
    pat = np.asarray([[0, 0], [80, 80]], np.int32).tobytes()
    i = bytes(blob).find(pat)
    assert i > 0
    blob[i:i + len(pat)] = np.zeros(4, np.int32).tobytes()
    with pytest.raises(host.HipError):
        host.HipClassifier(bytes(blob), plan_only=True)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["tiny", "tiny_two_lanes", "tiny_fft1024", "tiny_perch_order"])
def test_tiny_perch_like_vs_oracle(gpu, variant):
    cfg = sm.tiny_perch_config()
    if variant == "tiny_fft1024":
        cfg = sm.tiny_perch_config(specs=(sm.SpecConfig(640, 320, 60.0, 16000.0, 1024),), n_samples=16000, pad=(160, 160), n_mels=128)
    kw = {}
    if variant == "tiny_perch_order":                            # four outputs in Perch's order, picked by explicit option
        cfg = sm.tiny_perch_config(perch_outputs=True)
        kw = dict(logits_output=3, embedding_output=0)
    blob = sm.build_model(cfg)
    n = 37 if variant == "tiny_two_lanes" else 5
    x = sm.synth_clips(n, cfg.n_samples, cfg.sample_rate)
    x[1] = 0.0                                                   # silence: every band sits on the log floor
    x[2, : cfg.n_samples // 2] = 0.0
    ref = oracle_logits_emb(blob, x)
    c = host.HipClassifier(blob, max_batch=64, **kw)
    try:
        got, emb = c.predict_batch(x.reshape(-1), n, want_embeddings=True)
        assert np.isfinite(got).all()
        assert (got.argmax(1) == ref[0].argmax(1)).all()
        assert np.abs(softmax64(got) - softmax64(ref[0])).max() <= 1e-4
        assert np.abs(got - ref[0]).max() < 1e-3 and np.abs(emb - ref[1]).max() < 1e-3
        spec_t = next(s for s in c.describe()["steps"] if s["kernel"] == "frontend")
        assert spec_t
    finally:
        c.close()


@pytest.mark.gpu
def test_perch_size_vs_oracle_and_softmax_topk(gpu):
    """Perch dimensions, 3 clips (one silent): logits within the reference's own "EQUIVALENT" band (< 1e-3,
    cmd/perch-benchmark/main.go:455-462), softmax within 1e-4, embedding within 1e-3; device softmax + top-10 vs the Go
    restatement of perchSoftmax."""
    cfg = sm.perch_config()
    blob = sm.build_model(cfg)
    x = sm.synth_clips(3, cfg.n_samples, cfg.sample_rate)
    x[2] = 0.0
    ref = oracle_logits_emb(blob, x)
    c = host.HipClassifier(blob, max_batch=8)
    try:
        got, emb = c.predict_batch(x.reshape(-1), 3, want_embeddings=True)
        assert (got.argmax(1) == ref[0].argmax(1)).all()
        assert np.abs(got - ref[0]).max() < 1e-3 and np.abs(emb - ref[1]).max() < 1e-3
        assert np.abs(softmax64(got) - softmax64(ref[0])).max() <= 1e-4
        p = host.Perch(c, [f"sp{i}" for i in range(cfg.n_classes)])
        top = p.predict_batch(x.reshape(-1), 3)
        want = np.stack([G.softmax(r) for r in ref[0]])
        for r in range(3):
            assert top[r][0][0] == f"sp{int(want[r].argmax())}"
            assert abs(top[r][0][1] - float(want[r].max())) <= 1e-4
            assert [t[1] for t in top[r]] == sorted((t[1] for t in top[r]), reverse=True)
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("orient", ["n", "t"])
def test_expand_dw_both_orientations_vs_oracle(gpu, orient, monkeypatch):
    """The fused expand+depthwise kernel with rows and columns swapped (tall, narrow images): every fused layer forced to one
    orientation, on the v2.4-style geometry sweep (wide images, 3x3 / 5x5, stride 1 / 2, asymmetric SAME padding) and on the
    Perch-style tiny model, vs the oracle."""
    from test_parity_gpu import _geo_cfg, assert_parity
    monkeypatch.setenv("BNHIP_EXPDW_ORIENT", orient)
    for cfg in (_geo_cfg(1), _geo_cfg(4), sm.tiny_perch_config()):
        blob = sm.build_model(cfg)
        x = sm.synth_clips(3, cfg.n_samples, cfg.sample_rate)
        ref = oracle_logits_emb(blob, x)[0]
        c = host.HipClassifier(blob, max_batch=8)
        try:
            got = c.predict_batch(x.reshape(-1), 3)
            ed = [s for s in c.describe()["steps"] if s["kernel"] == "expand_dw" and not s["name"].startswith("stem")]
        finally:
            c.close()
        assert ed and all((s["shape"] >= 22) == (orient == "t") for s in ed), [(s["name"], s["shape"]) for s in ed]
        assert_parity(got, ref)
        assert np.abs(got - ref).max() < 1e-3


@pytest.mark.gpu
def test_bf16_precision_drift_is_bounded(gpu):
    """"precision":"bf16" (MFMA operands rounded to bf16, fp32 accumulate and storage) on the Perch-size stand-in: top-1
    unchanged and softmax drift far inside what the reference accepts for reduced-precision Perch (~0.08 on an f16 GPU,
    internal/inference/openvino_parity_functional_test.go:156-158); the f32 engine on the same clips is the control."""
    cfg = sm.perch_config()
    blob = sm.build_model(cfg)
    x = sm.synth_clips(3, cfg.n_samples, cfg.sample_rate, first=11)
    ref = oracle_logits_emb(blob, x)[0]
    out = {}
    for prec in ("f32", "bf16"):
        c = host.HipClassifier(blob, max_batch=8, precision=prec)
        try:
            assert c.describe()["precision"] == prec
            out[prec] = c.predict_batch(x.reshape(-1), 3)
        finally:
            c.close()
    d32 = np.abs(softmax64(out["f32"]) - softmax64(ref)).max()
    d16 = np.abs(softmax64(out["bf16"]) - softmax64(ref)).max()
    print(f"softmax drift vs oracle: f32 {d32:.2e}, bf16 {d16:.2e}; max |logit| diff bf16 {np.abs(out['bf16'] - ref).max():.3e}")
    assert d32 <= 1e-4
    assert (out["bf16"].argmax(1) == ref.argmax(1)).all() and d16 <= 0.02
    assert d16 > d32            # the option really changes the arithmetic


@pytest.mark.gpu
def test_bf16_activation_storage_plan_and_drift(gpu, monkeypatch):
    """"precision":"bf16" also keeps the 6x-expanded tensors as bf16 in HBM (engine.cpp mark_bf16_storage: outputs of the
    split-bf16 expand GEMMs, of the depthwise kernels and of the fused expand + depthwise kernel, consumed by depthwise /
    projection).  The plan must mark exactly those, an f32 engine none; against the same engine with fp32 storage
    (BNHIP_BF16_ACT=0) the only new rounding is the depthwise input, so logits move little; and the oracle bounds of the
    drift test above still hold.  Run on the Perch-size stand-in (fused and unfused blocks, both depthwise forms) and on the
    v2.4-topology model."""
    for cfg, n in ((sm.perch_config(), 3), (sm.SynthConfig(), 4)):
        blob = sm.build_model(cfg)
        x = sm.synth_clips(n, cfg.n_samples, cfg.sample_rate, first=23)
        ref = oracle_logits_emb(blob, x)[0]
        f32 = host.HipClassifier(blob, max_batch=8)
        try:
            assert not any(s["out_bf16"] or s["in_bf16"] for s in f32.describe()["steps"])
        finally:
            f32.close()
        out = {}
        for act in ("1", "0"):
            monkeypatch.setenv("BNHIP_BF16_ACT", act)
            c = host.HipClassifier(blob, max_batch=8, precision="bf16")
            try:
                st = c.describe()["steps"]
                marked = [s for s in st if s["out_bf16"]]
                if act == "1":
                    kinds = {s["kernel"] for s in marked}
                    assert marked and kinds <= {"pw_gemm", "dwconv", "expand_dw", "conv_direct"}, kinds
                    # every marked value is read by a step that knows: its consumers carry in_bf16
                    outs = {s["out_v"] for s in marked}
                    assert outs == {v for v in outs if any(t["in_bf16"] for t in st)}
                    assert all(s["out_bf16"] for s in st if s["kernel"] == "expand_dw")      # the fused kernels' outputs all qualify
                    assert not st[-1]["out_bf16"]                                            # never a graph output
                else:
                    assert not marked
                out[act] = c.predict_batch(x.reshape(-1), n)
            finally:
                c.close()
        monkeypatch.delenv("BNHIP_BF16_ACT")
        a, b = out["1"], out["0"]
        assert np.isfinite(a).all() and (a.argmax(1) == ref.argmax(1)).all() and (a.argmax(1) == b.argmax(1)).all()
        d_store = np.abs(softmax64(a) - softmax64(b)).max()
        d_ref = np.abs(softmax64(a) - softmax64(ref)).max()
        print(f"{cfg.name}: softmax drift bf16-storage vs fp32-storage {d_store:.2e}, vs oracle {d_ref:.2e}, "
              f"max |logit| diff {np.abs(a - b).max():.3e}")
        assert d_ref <= 0.02 and d_store <= 0.01


@pytest.mark.gpu
def test_lds_staged_depthwise_vs_oracle(gpu, monkeypatch):
    """Plain depthwise layers forced onto the LDS-staged kernel (the fused kernel's second phase on a copied footprint:
    3x3 / 5x5, stride 1 / 2, channel counts that are not multiples of 32, fused squeeze-excite sums) vs the oracle, on the
    Perch-style tiny model (b1/b2 are expansion-free blocks) and a v2.4-style one with fusion switched off."""
    from test_parity_gpu import _geo_cfg, assert_parity
    monkeypatch.setenv("BNHIP_DW_LDS", "1")
    for cfg, nofuse in ((sm.tiny_perch_config(), False), (_geo_cfg(2), True), (_geo_cfg(5), True)):
        if nofuse:
            monkeypatch.setenv("BNHIP_NO_FUSE_EXPDW", "1")
        blob = sm.build_model(cfg)
        x = sm.synth_clips(3, cfg.n_samples, cfg.sample_rate)
        ref = oracle_logits_emb(blob, x)[0]
        c = host.HipClassifier(blob, max_batch=8)
        try:
            got = c.predict_batch(x.reshape(-1), 3)
            dw = [s for s in c.describe()["steps"] if s["kernel"] == "dwconv"]
        finally:
            c.close()
            monkeypatch.delenv("BNHIP_NO_FUSE_EXPDW", raising=False)
        assert dw and all(s["dw_lds"] == 1 for s in dw), [(s["name"], s["dw_lds"]) for s in dw]
        assert_parity(got, ref)
        assert np.abs(got - ref).max() < 1e-3


def _fe_cfg(i):
    """Seeded log-mel front-end geometries: frame / transform / hop / padding / band-count combinations the Perch-size model
    does not have (frame = transform, odd hops, one-sided padding, frames that end exactly at the clip end)."""
    rng = np.random.default_rng(500 + i)
    lfft = int(rng.choice([512, 1024, 2048]))
    L = int(rng.choice([lfft, lfft, lfft // 2, lfft * 5 // 8, lfft - 2]))
    hop = int(rng.choice([L // 2, L // 4, 94, 160, 278]))
    pad = [(0, 0), (L // 4, L // 4), (hop, 0), (0, L // 2), (7, 13)][int(rng.integers(0, 5))]
    frames = int(rng.integers(9, 40))
    from math import gcd
    sub = gcd(L, hop)                                     # tf.signal.frame reshapes the (padded) clip into sub-frames of gcd(L, hop)
    n_samples = L + hop * (frames - 1) - pad[0] - pad[1] + sub * int(rng.integers(0, max(hop // sub, 1)))   # + a tail the framing ignores
    n_mels = int(rng.choice([16, 40, 64, 96, 128]))
    fmax = 7000.0 if (lfft == 2048 and i % 2) else 16000.0       # 2048-point transforms under a narrow and under a full-band (1020-bin) mel bank
    return sm.tiny_perch_config(n_samples=n_samples, sample_rate=32000, specs=(sm.SpecConfig(L, hop, 60.0, fmax, lfft),),
                                n_mels=n_mels, pad=pad, log_floor=float(rng.choice([1e-3, 1e-2, 0.5])), log_scale=float(rng.choice([0.1, 1.0])),
                                normalize=bool(rng.integers(0, 3) == 0), compress=str(rng.choice(["log", "log", "pow"])),
                                time_major=bool(rng.integers(0, 4) != 0), seed=900 + i)


@pytest.mark.parametrize("i", range(10))
def test_front_end_geometry_sweep_plans(built_lib, i):
    cfg = _fe_cfg(i)
    c = host.HipClassifier(sm.build_model(cfg), plan_only=True)
    try:
        kinds = [s["kernel"] for s in c.describe()["steps"]]
        assert "stft" in kinds and ("clip_minmax" in kinds) == cfg.normalize
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(10))
def test_front_end_geometry_sweep_vs_oracle(gpu, i):
    cfg = _fe_cfg(i)
    blob = sm.build_model(cfg)
    x = sm.synth_clips(4, cfg.n_samples, cfg.sample_rate, first=7 * i)
    x[3] = 0.0
    ref, ref_emb = oracle_logits_emb(blob, x)
    c = host.HipClassifier(blob, max_batch=8)
    try:
        got, emb = c.predict_batch(x.reshape(-1), 4, want_embeddings=True)
    finally:
        c.close()
    assert (got.argmax(1) == ref.argmax(1)).all()
    assert np.abs(softmax64(got) - softmax64(ref)).max() <= 1e-4 and np.abs(got - ref).max() < 1e-3 and np.abs(emb - ref_emb).max() < 1e-3


@pytest.mark.parametrize("emb_first", [False, True])
def test_birdnet_v3_output_rule(built_lib, emb_first):
    """BirdNET v3.0 graphs (160000 samples, two outputs) come with the 1280-wide embedding port first or second depending on
    the export; the reference picks it by size (internal/inference/onnx/detection.go:91-106).  A 160000-sample stand-in with a
    1280-d embedding in either order must bind logits / embedding the same way, and the oracle agrees on the GPU."""
    cfg = sm.tiny_perch_config(n_samples=160000, specs=(sm.SpecConfig(640, 1280, 60.0, 16000.0, 1024),), pad=(0, 0), n_mels=32,
                               top=1280, n_classes=40, emb_first=emb_first)
    c = host.HipClassifier(sm.build_model(cfg), plan_only=True)
    try:
        d = c.describe()
        assert (c.num_species(), c.emb_dim) == (40, 1280)
        assert (d["logits_output"], d["embedding_output"]) == ((1, 0) if emb_first else (0, 1))
    finally:
        c.close()


@pytest.mark.gpu
def test_birdnet_v3_output_rule_vs_oracle(gpu):
    cfg = sm.tiny_perch_config(n_samples=160000, specs=(sm.SpecConfig(640, 1280, 60.0, 16000.0, 1024),), pad=(0, 0), n_mels=32,
                               top=1280, n_classes=40, emb_first=True)
    blob = sm.build_model(cfg)
    x = sm.synth_clips(3, cfg.n_samples, cfg.sample_rate)
    outs = Interpreter(blob).invoke(x)                       # graph order: embedding, logits
    c = host.HipClassifier(blob, max_batch=4)
    try:
        got, emb = c.predict_batch(x.reshape(-1), 3, want_embeddings=True)
    finally:
        c.close()
    assert got.shape == (3, 40) and emb.shape == (3, 1280)
    assert np.abs(got - outs[1]).max() < 1e-3 and np.abs(emb - outs[0]).max() < 1e-3


@pytest.mark.gpu
def test_bf16_gemm_kernel_equals_the_one_product_path_of_the_split_kernel(gpu, monkeypatch):
    """k_pw_b16 (pw_b16.hip: A fragments straight from global memory, W tile through LDS in fragment order) performs exactly
    the arithmetic of k_pw_bx3 with one product per operand pair; 64 clips so that the 128-row tiles it serves are chosen
    (small grids shrink to 64-row tiles, which stay on k_pw_bx3).  Bit-identical logits and embeddings, squeeze-excite scaled
    projections, fp32 and bf16-stored operands, K tails (232, 136 are not multiples of 32) included."""
    import ctypes
    cfg = sm.perch_config()
    blob = sm.build_model(cfg)
    n = 64
    x = sm.synth_clips(n, cfg.n_samples, cfg.sample_rate, first=3)
    lib = host.load_library()
    lib.bnhip_debug_pw_b16_launches.restype = ctypes.c_long
    out = {}
    for on in ("2", "0"):                                     # 2: every 128-row tile on k_pw_b16; 0: none
        monkeypatch.setenv("BNHIP_PW_B16", on)
        c = host.HipClassifier(blob, max_batch=n, precision="bf16", autotune=False, lanes=1)
        try:
            before = lib.bnhip_debug_pw_b16_launches()
            out[on] = [a.copy() for a in c.predict_batch(x.reshape(-1), n, want_embeddings=True)]
            used = lib.bnhip_debug_pw_b16_launches() - before
        finally:
            c.close()
        assert (used >= 10) if on == "2" else (used == 0), used
    assert np.array_equal(out["2"][0], out["0"][0]), np.abs(out["2"][0] - out["0"][0]).max()
    assert np.array_equal(out["2"][1], out["0"][1])


@pytest.mark.gpu
def test_weights_stationary_gemm_equals_the_tiled_kernels(gpu, monkeypatch):
    """k_pw_b16s (pw_b16.hip: skinny projections - K, N of a few dozen - with the whole weight matrix in a wave's registers, no
    LDS, epilogue straight from the accumulators, squeeze-excite scale reloaded per clip, bf16 residual stream) forced onto every
    layer it accepts: same image, same K order, one product per pair - bit-identical to the tiled kernels."""
    import ctypes
    cfg = sm.perch_config()
    blob = sm.build_model(cfg)
    n = 16
    x = sm.synth_clips(n, cfg.n_samples, cfg.sample_rate, first=5)
    lib = host.load_library()
    lib.bnhip_debug_pw_b16_launches.restype = ctypes.c_long
    out = {}
    monkeypatch.setenv("BNHIP_PW_B16", "0")                    # (so that the counter below counts k_pw_b16s launches only)
    for on in ("2", "0"):
        monkeypatch.setenv("BNHIP_PW_B16S", on)
        c = host.HipClassifier(blob, max_batch=n, precision="bf16", autotune=False, lanes=1)
        try:
            before = lib.bnhip_debug_pw_b16_launches()
            out[on] = [a.copy() for a in c.predict_batch(x.reshape(-1), n, want_embeddings=True)]
            used = lib.bnhip_debug_pw_b16_launches() - before
        finally:
            c.close()
        assert (used >= 5) if on == "2" else (used == 0), used
    assert np.array_equal(out["2"][0], out["0"][0]), np.abs(out["2"][0] - out["0"][0]).max()
    assert np.array_equal(out["2"][1], out["0"][1])


@pytest.mark.gpu
@pytest.mark.parametrize("model,prec", [("perch", "f32"), ("v24", "f32"), ("v24", "bf16")])
def test_weights_in_lds_gemm_equals_the_tiled_kernels(gpu, monkeypatch, model, prec):
    """k_pw_ws (pw_ws.hip; a tuner candidate by default, BNHIP_PW_WS=0 takes it away, 2 forces it): the 6x expands keep their
    weight columns in LDS for the life of a block, whose waves walk 16-row tile pairs with A streamed from global memory through
    a register ring - no tile writes, one barrier.  Forced onto every layer it accepts: same image, same K order, same product
    order per accumulator - bit-identical to the tiled kernels, in the six-product form of the fp32 engines (Perch dimensions:
    K = 136, five slabs with a K tail, ring depth 1; v2.4: K = 192, six slabs, ring depth 3, and K = 320, ten slabs, ring depth 5)
    and the one-product form of the bf16 ones (64-column blocks, bf16 A fragments as loaded).  The row counts here give waves
    odd and even tile counts, single-tile tails and the staggered second half of every block."""
    import ctypes
    cfg = sm.perch_config() if model == "perch" else sm.SynthConfig()
    blob = sm.build_model(cfg)
    n = 16 if model == "perch" else 24
    x = sm.synth_clips(n, cfg.n_samples, cfg.sample_rate, first=9)
    lib = host.load_library()
    lib.bnhip_debug_pw_ws_launches.restype = ctypes.c_long
    out = {}
    for on in ("2", "0"):
        monkeypatch.setenv("BNHIP_PW_WS", on)
        c = host.HipClassifier(blob, max_batch=n, precision=prec, autotune=False, lanes=1)
        try:
            before = lib.bnhip_debug_pw_ws_launches()
            out[on] = c.predict_batch(x.reshape(-1), n).copy()
            used = lib.bnhip_debug_pw_ws_launches() - before
        finally:
            c.close()
        assert (used >= 3) if on == "2" else (used == 0), used
    assert np.array_equal(out["2"], out["0"]), np.abs(out["2"] - out["0"]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_tuning_file_reproduces_the_plan_and_is_ignored_by_other_plans(gpu, tmp_path, monkeypatch, prec):
    """BNHIP_TUNE_FILE (tools/profile_round.sh: one tuning for the bench, the kernel trace and every PMC pass): the first engine
    times its candidates and writes the file, the second reads it - same tiles, same kernel flavours (incl. the streamed-operand
    and weights-stationary GEMM codes of round 4), same bits; an engine with another plan neither uses nor overwrites it."""
    cfg = sm.tiny_perch_config()
    blob = sm.build_model(cfg)
    x = sm.synth_clips(8, cfg.n_samples, cfg.sample_rate, first=2)
    path = tmp_path / "tune.txt"
    monkeypatch.setenv("BNHIP_TUNE_FILE", str(path))
    plans, outs = [], []
    for _ in range(2):
        c = host.HipClassifier(blob, max_batch=8, precision=prec)
        try:
            plans.append([(s["name"], s["nt"], s["wm"], s["nt_full"], s["wm_full"], s["shape"], s["dw_lds"], s["bx"]) for s in c.describe()["steps"]])
            outs.append(c.predict_batch(x.reshape(-1), 8).copy())
        finally:
            c.close()
        assert path.exists()
    assert plans[0] == plans[1] and np.array_equal(outs[0], outs[1])
    text = path.read_text()
    c = host.HipClassifier(blob, max_batch=4, precision=prec)       # another plan (batch size): tunes for itself
    try:
        got = c.predict_batch(x[:4].reshape(-1), 4)
    finally:
        c.close()
    assert path.read_text() == text
    assert np.abs(softmax64(got) - softmax64(outs[0][:4])).max() <= 1e-4
    # an edited file: the derived column (tile count of the per-tile sums, mel quads) is recomputed from the plan, never read -
    # garbage there changes nothing; a row that asks for a kernel form its layer is not eligible for makes the engine ignore
    # the file and tune for itself.  Either way the same logits.
    lines = text.splitlines()
    junk = [lines[0]] + [" ".join(f[:9] + ["12345"] + f[10:]) for f in (l.split(" ") for l in lines[1:])]
    path.write_text("\n".join(junk) + "\n")
    c = host.HipClassifier(blob, max_batch=8, precision=prec)
    try:
        plan = [(s["name"], s["nt"], s["wm"], s["nt_full"], s["wm_full"], s["shape"], s["dw_lds"], s["bx"]) for s in c.describe()["steps"]]
        again = c.predict_batch(x.reshape(-1), 8).copy()
    finally:
        c.close()
    assert plan == plans[0] and np.array_equal(again, outs[0])
    bad = [lines[0]] + [" ".join(f[:6] + ["9999"] + f[7:]) for f in (l.split(" ") for l in lines[1:])]      # no such tile shape
    path.write_text("\n".join(bad) + "\n")
    c = host.HipClassifier(blob, max_batch=8, precision=prec)
    try:
        self_tuned = c.predict_batch(x.reshape(-1), 8).copy()
    finally:
        c.close()
    assert np.abs(softmax64(self_tuned) - softmax64(outs[0])).max() <= 1e-4


@pytest.mark.gpu
def test_bf16_residual_stream_plan_and_drift(gpu, monkeypatch):
    """"precision":"bf16" keeps the residual stream - projection outputs, read by the next expand, the next residual add and the
    ratio-1 blocks' depthwise kernels - as bf16 where every reader can widen it (engine.cpp mark_bf16_storage; BNHIP_BF16_RESID=0
    keeps block outputs fp32).  The plan shows it (projections write bf16, fused expand + depthwise steps read bf16), the softmax
    moves by less than the option's own drift against the oracle, top-1 stays."""
    cfg = sm.perch_config()
    blob = sm.build_model(cfg)
    x = sm.synth_clips(3, cfg.n_samples, cfg.sample_rate, first=21)
    ref = oracle_logits_emb(blob, x)[0]
    out, plans = {}, {}
    for on in ("1", "0"):
        monkeypatch.setenv("BNHIP_BF16_RESID", on)
        c = host.HipClassifier(blob, max_batch=8, precision="bf16")
        try:
            plans[on] = {s["name"]: (s["in_bf16"], s["out_bf16"]) for s in c.describe()["steps"]}
            out[on] = c.predict_batch(x.reshape(-1), 3)
        finally:
            c.close()
    proj = [n for n in plans["1"] if n.endswith("/project")]
    fused = [n for n in plans["1"] if n.endswith("/expand+dw")]
    assert sum(plans["1"][n][1] for n in proj) >= 20 and sum(plans["0"][n][1] for n in proj) == 0
    assert sum(plans["1"][n][0] for n in fused) >= 10 and sum(plans["0"][n][0] for n in fused) == 0
    d_on = np.abs(softmax64(out["1"]) - softmax64(ref)).max()
    d_off = np.abs(softmax64(out["0"]) - softmax64(ref)).max()
    print(f"softmax drift vs oracle: bf16 residual stream {d_on:.2e}, fp32 residual stream {d_off:.2e}")
    assert (out["1"].argmax(1) == ref.argmax(1)).all() and d_on <= 1e-3 and d_off <= 1e-3
    assert np.abs(softmax64(out["1"]) - softmax64(out["0"])).max() <= 1e-3
