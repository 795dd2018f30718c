"""Model container + planner, CPU only: writer -> (oracle reader | C++ reader via plan_only)."""
import hashlib
import os
import re

import numpy as np
import pytest

from birdnet_go_amd import host, synth_model as sm, tflite_schema as S
from birdnet_go_amd.tflite_build import GraphBuilder
from oracle.interp import Interpreter
from oracle.tflite_reader import read_model

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_writer_is_deterministic_and_matches_golden_digest(tiny_blob, tiny_cfg):
    assert sm.build_model(tiny_cfg) == tiny_blob
    g = np.load(os.path.join(GOLD, "tiny_logits.npz"))
    assert hashlib.sha256(tiny_blob).digest() == g["sha256"].tobytes()


def test_oracle_reader_roundtrip(tiny_blob, tiny_cfg):
    m = read_model(tiny_blob)
    assert m.tensors[m.inputs[0]].shape == [1, tiny_cfg.n_samples]
    assert m.tensors[m.outputs[0]].shape == [1, tiny_cfg.n_classes]
    names = [o.name for o in m.ops]
    for need in ("RFFT2D", "GATHER", "CAST", "FULLY_CONNECTED", "POW", "REVERSE_V2", "TRANSPOSE", "CONCATENATION",
                 "CONV_2D", "DEPTHWISE_CONV_2D", "MEAN", "LOGISTIC", "MUL", "ADD"):
        assert need in names
    # weights survive bit-exactly
    w = [t for t in m.tensors if t.name == "stem/w"][0]
    rng = np.random.default_rng(tiny_cfg.seed)
    want = rng.standard_normal((tiny_cfg.stem, 3, 3, 2)).astype(np.float32) * np.float32(1.6 / np.sqrt(18))
    assert np.array_equal(w.data, want)


def test_oracle_matches_golden(tiny_blob, tiny_cfg):
    g = np.load(os.path.join(GOLD, "tiny_logits.npz"))
    x = sm.synth_clips(3, tiny_cfg.n_samples, tiny_cfg.sample_rate)
    got = Interpreter(tiny_blob).invoke(x)[0]
    # BLAS summation order may differ between hosts: tolerance, not bit-equality
    assert np.abs(got - g["logits_f32"]).max() < 2e-5
    assert np.abs(got - g["logits_f64"]).max() < 2e-5


def test_oracle_batch_equals_single(tiny_blob, tiny_cfg):
    x = sm.synth_clips(3, tiny_cfg.n_samples, tiny_cfg.sample_rate)
    it = Interpreter(tiny_blob)
    whole = it.invoke(x)[0]
    for i in range(3):
        assert np.abs(it.invoke(x[i])[0][0] - whole[i]).max() < 1e-5


def test_oracle_input_size_mismatch(tiny_blob):
    with pytest.raises(ValueError, match="input size mismatch"):
        Interpreter(tiny_blob).invoke(np.zeros(100, np.float32))


def test_frontend_equals_direct_numpy_stft(tiny_blob, tiny_cfg):
    """The graph front-end == a direct numpy restatement of MelSpecLayerSimple (real-part STFT)."""
    m = read_model(tiny_blob)
    x = sm.synth_clips(2, tiny_cfg.n_samples, tiny_cfg.sample_rate)
    keep = {}
    Interpreter(m, "f64").invoke(x, keep=keep)
    cat = [o for o in m.ops if o.name == "CONCATENATION"][0]
    spec = keep[cat.outputs[0]]
    xn = x.astype(np.float64)
    xn = xn - xn.min(1, keepdims=True)
    xn = xn / (xn.max(1, keepdims=True) + np.float64(np.float32(1e-6)))
    xn = (xn - 0.5) * 2.0
    for c, sp in enumerate(tiny_cfg.specs):
        F = sm.n_frames(tiny_cfg.n_samples, sp.frame_length, sp.frame_step)
        idx = np.arange(F)[:, None] * sp.frame_step + np.arange(sp.frame_length)[None, :]
        fr = xn[:, idx] * sm.hann_periodic(sp.frame_length).astype(np.float64)
        re = np.fft.rfft(fr, axis=-1).real
        mel = sm.mel_weight_matrix(tiny_cfg.n_mels, sp.frame_length // 2 + 1, tiny_cfg.sample_rate, sp.fmin, sp.fmax)
        v = re @ mel.astype(np.float64)
        expo = np.float64(np.float32(1.0 / (1.0 + np.exp(tiny_cfg.mag_scale))))
        y = np.power(np.power(v, 2.0), expo)
        want = np.transpose(y[:, :, ::-1], (0, 2, 1))
        assert np.allclose(spec[..., c], want, rtol=1e-4, atol=1e-5)


def test_cpp_reader_and_planner_fuse_everything(built_lib, full_blob):
    clf = host.HipClassifier(full_blob, plan_only=True)
    assert (clf.n_samples, clf.num_species(), clf.emb_dim) == (144000, 6522, 0)
    d = clf.describe()
    kinds = [s["kernel"] for s in d["steps"]]
    # front-end (FFT path): clip_minmax, normalize, 2 x stft, one banded mel + compression + store launch
    assert kinds.count("stft") == 2 and kinds.count("frontend") == 2 and kinds.count("clip_minmax") == 1
    assert "elementwise" not in kinds, "an op fell back to the unfused elementwise path"
    # 16 MBConv blocks: 11 with the fused expand+depthwise kernel (Cin <= 128) + the stem fused with b1's depthwise in the same
    # kernel family; 4 plain depthwise (the wide blocks)
    assert kinds.count("expand_dw") == 12 and kinds.count("dwconv") == 4 and kinds.count("se") == 16
    assert kinds.count("conv_direct") == 0
    # every squeeze-excite mean comes from sums emitted by the producing depthwise kernel; only the GAP uses the mean pass
    assert sum(s["fused_sum"] for s in d["steps"]) == 16 and kinds.count("mean") == 2
    pw = [s for s in d["steps"] if s["kernel"] == "pw_gemm"]
    assert sum(s["fused_scale"] for s in pw) == 16 and sum(s["fused_res"] for s in pw) == 9
    # FFT front-end: clip_minmax, normalize, 2 x stft, one banded mel + pow + store launch for both channels; on request
    # (BNHIP_FUSE_MEL) the mel projection moves into the STFT kernels' epilogue instead
    names = [s["name"] for s in d["steps"]]
    assert len(d["steps"]) == 61 and "melband0+1" in names and "mel0" not in names
    os.environ["BNHIP_FUSE_MEL"] = "1"
    try:
        n1 = [s["name"] for s in host.HipClassifier(full_blob, plan_only=True).describe()["steps"]]
    finally:
        del os.environ["BNHIP_FUSE_MEL"]
    assert len(n1) == 60 and "stft0+mel" in n1 and "stft1+mel" in n1
    # with the folded-GEMM front-end: clip_minmax + one k_frontend launch per channel
    d0 = host.HipClassifier(full_blob, plan_only=True, frontend_fft=0).describe()
    assert len(d0["steps"]) == 59 and [s["kernel"] for s in d0["steps"]].count("frontend") == 2
    assert d["specs"][0]["hop"] == 278 and d["specs"][1]["hop"] == 280 and d["specs"][0]["frames"] == 511
    assert abs(d["specs"][0]["p2"] - 1.0 / (1.0 + np.exp(1.23))) < 1e-6
    with pytest.raises(host.HipError, match="plan-only"):
        clf.predict(np.zeros(144000, np.float32))
    clf.close()


def test_cpp_reader_embedding_output(built_lib):
    blob = sm.build_model(sm.tiny_config(emit_embeddings=True))
    clf = host.HipClassifier(blob, plan_only=True)
    assert clf.emb_dim == 64 and clf.num_species() == 50
    clf.close()


def test_cpp_reader_rejects_garbage(built_lib, tiny_blob):
    with pytest.raises(host.HipError) as e:
        host.HipClassifier(b"\x00" * 64, plan_only=True)
    assert e.value.code == host.E_MODEL
    # truncated file: must fail cleanly (bounds-checked), never crash
    for cut in (9, 64, 1000, len(tiny_blob) // 2, len(tiny_blob) - 5000):
        with pytest.raises(host.HipError):
            host.HipClassifier(tiny_blob[:cut], plan_only=True)
    # corrupted offsets
    rng = np.random.default_rng(1)
    for _ in range(20):
        b = bytearray(tiny_blob)
        for p in rng.integers(8, 4096, 8):
            b[p] ^= 0xFF
        try:
            host.HipClassifier(bytes(b), plan_only=True).close()
        except host.HipError:
            pass


def test_unsupported_graph_reports_op(built_lib):
    g = GraphBuilder()
    # in-graph SOFTMAX (and the rest of the float builtin vocabulary) is served by the generic tier since round 2; what
    # stays unsupported is reported by operator name: a stand-alone complex op outside the recognised front-end
    x = g.tensor([1, 1000], name="INPUT")
    y = g.op("SOFTMAX", [x], [1, 1000], dict(beta=1.0))
    assert host.HipClassifier(g.finish([x], [y]), plan_only=True).num_species() == 1000
    g = GraphBuilder()
    x = g.tensor([1, 1000], name="INPUT")
    y = g.op("COMPLEX_ABS", [x], [1, 1000], {})
    with pytest.raises(host.HipError, match="COMPLEX_ABS") as e:
        host.HipClassifier(g.finish([x], [y]), plan_only=True)
    assert e.value.code == host.E_UNSUPPORTED


def test_magnitude_frontend_plans_onto_the_fft_path(built_lib):
    """COMPLEX_ABS graphs cannot use the folded-GEMM kernel (the magnitude is non-linear): the planner routes them through
    normalize -> stft -> mel GEMM -> finish; an fft_length the FFT kernel does not cover is reported, not guessed."""
    specs = (sm.SpecConfig(512, 94, 0.0, 3000.0), sm.SpecConfig(512, 94, 500.0, 15000.0))
    blob = sm.build_model(sm.tiny_config(complex_mode="abs", specs=specs))
    with pytest.raises(host.HipError, match="fft_length") as e:          # the default tiny geometry has a 256-point branch
        host.HipClassifier(sm.build_model(sm.tiny_config(complex_mode="abs")), plan_only=True)
    assert e.value.code == host.E_UNSUPPORTED
    d = host.HipClassifier(blob, plan_only=True).describe()
    kinds = [s["kernel"] for s in d["steps"]]
    assert kinds.count("stft") == 2 and "frontend" in kinds
    names = [s["name"] for s in d["steps"]]
    assert names[:2] == ["clip_minmax", "normalize"] and "melband0+1" in names
    # a dense mel matrix (or BNHIP_NO_MEL_BANDED) keeps the GEMM + finish pair
    os.environ["BNHIP_NO_MEL_BANDED"] = "1"
    try:
        n2 = [s["name"] for s in host.HipClassifier(blob, plan_only=True).describe()["steps"]]
    finally:
        del os.environ["BNHIP_NO_MEL_BANDED"]
    assert "mel0" in n2 and "mel1" in n2 and "melspec0+1" in n2
    # the real-part graph keeps the folded GEMM unless asked otherwise
    # the real-part graph can use either front-end: FFT by default where the frame length is covered, folded GEMM on request
    d2 = host.HipClassifier(sm.build_model(sm.tiny_config(specs=specs)), plan_only=True, frontend_fft=0).describe()
    assert "stft" not in [s["kernel"] for s in d2["steps"]]
    d3 = host.HipClassifier(sm.build_model(sm.tiny_config(specs=specs)), plan_only=True).describe()
    assert [s["kernel"] for s in d3["steps"]].count("stft") == 2
    # mixed: the default tiny geometry has one 512-point and one 256-point branch
    d4 = host.HipClassifier(sm.build_model(sm.tiny_config()), plan_only=True).describe()
    k4 = [s["kernel"] for s in d4["steps"]]
    assert k4.count("stft") == 1 and k4.count("frontend") >= 2       # (normalize + the folded-GEMM 256-point branch)
    # the oracle executes the op generically
    assert np.isfinite(Interpreter(blob).invoke(sm.synth_clips(1, 12000))[0]).all()


def test_dense_only_graphs_plan(built_lib):
    """Bat head (CustomClassifier) and FP16 range-filter (DEQUANTIZE) graphs: no STFT, plain FC stacks."""
    head = sm.build_dense_model([1024, 30])
    clf = host.HipClassifier(head, plan_only=True)
    assert (clf.n_samples, clf.num_species(), clf.emb_dim) == (1024, 30, 0)
    assert [s["kernel"] for s in clf.describe()["steps"]] == ["pw_gemm"]
    rf = sm.build_dense_model([3, 64, 128, 6522], final_sigmoid=True, fp16_weights=True, input_scale=[90.0, 180.0, 48.0])
    clf = host.HipClassifier(rf, plan_only=True)
    assert (clf.n_samples, clf.num_species()) == (3, 6522)
    assert [s["kernel"] for s in clf.describe()["steps"]] == ["pw_gemm"] * 3      # LOGISTIC folded into the last FC
    # oracle executes the same file, fp16 constants included
    x = np.array([[60.17, 24.94, 22.0], [-33.9, 151.2, 48.0]], np.float32)
    y = Interpreter(rf).invoke(x)[0]
    assert y.shape == (2, 6522) and (y > 0).all() and (y < 1).all()
    m = read_model(rf)
    w16 = [t for t in m.tensors if t.name == "fc0/w_f16"][0]
    assert w16.dtype == np.float16 and w16.data.shape == (64, 3)


def test_oracle_torch_conv_backend_equals_numpy_ops(tiny_blob, tiny_cfg):
    """bench.py's cpu_baseline runs the oracle with torch-CPU convolutions: same semantics as the numpy ops."""
    x = sm.synth_clips(2, tiny_cfg.n_samples, tiny_cfg.sample_rate)
    a = Interpreter(tiny_blob).invoke(x)[0]
    b = Interpreter(tiny_blob, conv_backend="torch").invoke(x)[0]
    assert np.abs(a - b).max() < 2e-5


def test_malformed_graphs_are_model_errors_not_crashes(built_lib):
    """operand counts, absent (-1) operands and constant dtypes are validated before the planner indexes
    anything; a well-framed file with a corrupted operator list returns BNHIP_E_MODEL (or plans, when the mutation happens
    to be harmless) - it never takes the process down."""
    from graphgen import mutated_graph
    outcomes = {"ok": 0, "model": 0, "other": 0}
    for seed in range(24):
        for mut in range(12):
            blob = mutated_graph(seed, 1000 * seed + mut)
            try:
                host.HipClassifier(blob, plan_only=True).close()
                outcomes["ok"] += 1
            except host.HipError as e:
                assert e.code in (host.E_MODEL, host.E_UNSUPPORTED, host.E_INVALID), (seed, mut, str(e))
                outcomes["model" if e.code == host.E_MODEL else "other"] += 1
    assert outcomes["model"] > 20, outcomes


def test_specific_malformations_name_the_problem(built_lib):
    def conv_graph(mut):
        g = GraphBuilder()
        x = g.tensor([1, 8, 8, 4], name="INPUT")
        w = g.const(np.zeros((8, 3, 3, 4), np.float32))
        b = g.const(np.zeros(8, np.float32))
        y = g.op("CONV_2D", [x, w, b], [1, 8, 8, 8], dict(padding=0, stride_w=1, stride_h=1, fused_activation_function=0,
                                                          dilation_w_factor=1, dilation_h_factor=1))
        m = g.op("MEAN", [y, g.const(np.asarray([1, 2], np.int32))], [1, 8], dict(keep_dims=0))
        mut(g)
        return g.finish([x], [m])

    cases = {
        "required operand": lambda g: g.ops[0]["inputs"].__setitem__(1, -1),
        "no outputs": lambda g: g.ops[0].__setitem__("outputs", []),
        "must be float32": lambda g: g.tensors[1].__setitem__("type", S.INT32),
        "must be int32": lambda g: g.tensors[4].__setitem__("type", S.FLOAT32),
        "filter dimensions": lambda g: g.tensors[3].__setitem__("shape", [1, 8, 8, 16]),
        "bias length": lambda g: (g.tensors[2].__setitem__("shape", [4]), g.buffers.__setitem__(2, np.zeros(4, np.float32))),
        "operands, got": lambda g: g.ops[1]["inputs"].pop(),
    }
    for needle, mut in cases.items():
        with pytest.raises(host.HipError, match=needle) as e:
            host.HipClassifier(conv_graph(mut), plan_only=True)
        assert e.value.code == host.E_MODEL, needle


def test_shipped_tunings_are_well_formed():
    """birdnet-go_amd/tune/<plan key>.tune (bnhip.h "tune_dir"): the header repeats what the file name says (plan hash, batch, depth,
    host depth, precision, split-bf16 mode), one row per step, rows in step order - a stale or hand-edited file is ignored by the
    engine anyway, but the package should not ship one."""
    import glob
    import os
    import re
    from birdnet_go_amd import host
    files = sorted(glob.glob(os.path.join(host.TUNE_DIR, "*.tune")))
    assert len(files) >= 5, files
    for f in files:
        m = re.match(r"([0-9a-f]{16})_b(\d+)_d(\d+)_h(\d+)_p(\d+)_x(\d+)_l(\d+)_s([0-9a-f]+)\.tune$", os.path.basename(f))
        assert m, f
        lines = open(f).read().splitlines()
        hd = lines[0].split()
        assert hd[0] == "bnhip-tuning-2" and int(hd[1]) == len(lines) - 1, f
        assert (int(hd[2]), int(hd[3]), int(hd[4]), int(hd[5]), int(hd[6])) == tuple(int(m.group(i)) for i in (2, 3, 4, 5, 6)), f
        assert int(hd[7], 16) == int(m.group(1), 16), f
        for i, row in enumerate(lines[1:]):
            q = row.split(" ", 10)
            assert len(q) == 11 and int(q[0]) == i and 0 <= int(q[2]) <= 8 and 0 <= int(q[3]) <= 12, (f, row)
