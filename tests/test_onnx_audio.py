"""ONNX classifiers with an IN-GRAPH audio front-end - every ONNX classifier the reference ships has one
(internal/classifier/model_catalog.go:412-426 BirdNET v2.4 "dfttrunc", :490-501 the BattyBirdNET backbone
`birdnet-v2.4-embeddings-fp32-dfttrunc.onnx`, :273-311 Perch v2 "with in-graph DFT"; internal/classifier/birdnet_v3_onnx.go:44-48
"its mel front-end is a Conv1d").  The real files are absent from the snapshot, so the same synthetic graphs the TFLite tests
use are written as ONNX in the four forms exporters emit (birdnet-go_amd/onnx_audio.py) and must

  * mean the same thing to the independent ONNX oracle as the TFLite file means to the TFLite oracle (CPU),
  * plan onto the SAME fused front-end kernels (fp64 FFT + banded mel) and the same launch count as the TFLite form (CPU), and
  * produce, on the GPU, what the oracle computes - and, for the fused forms, exactly what the TFLite container produces.

Precision: a recognised front-end runs the fp64 FFT of the TFLite path whatever form the file gives the transform in (ORT
would evaluate a DFT MatMul / Conv / STFT in fp32); an unrecognised one runs literally, as fp32 GEMMs / convolutions.
"""
import numpy as np
import pytest

from birdnet_go_amd import host, onnx_build as ob, synth_model as sm
from birdnet_go_amd.synth_model import SpecConfig
from oracle import gofuncs as G, onnx_interp
from oracle.interp import Interpreter

FORMS = ["matmul", "conv1d", "stft", "dft"]


def _cfgs():
    mag = dict(complex_mode="abs", specs=(SpecConfig(512, 94, 0.0, 3000.0), SpecConfig(512, 94, 500.0, 15000.0)))
    return {"v24": sm.tiny_config(emit_embeddings=True), "magnitude": sm.tiny_config(**mag), "perch": sm.tiny_perch_config()}


def _build(cfg, form, trunc=True):
    try:
        return sm.build_model(cfg, container="onnx", dft=form, trunc=trunc)
    except ValueError:                       # dft="stft" cannot express frames shorter than the transform
        return None


@pytest.mark.parametrize("form", FORMS)
def test_onnx_transcription_means_what_the_tflite_graph_means(form):
    """Two independent oracles, two containers, one set of weights."""
    for name, cfg in _cfgs().items():
        for trunc in (True, False):
            blob = _build(cfg, form, trunc)
            if blob is None:
                continue
            x = sm.synth_clips(2, cfg.n_samples, cfg.sample_rate)
            want = Interpreter(sm.build_model(cfg)).invoke(x)
            got = onnx_interp.run(blob, x)
            assert len(want) == len(got)
            for a, b in zip(got, want):
                assert np.abs(a.reshape(b.shape) - b).max() < 2e-5, (name, form, trunc)


@pytest.mark.parametrize("form", FORMS)
def test_onnx_front_ends_plan_onto_the_fused_kernels(built_lib, form):
    for name, cfg in _cfgs().items():
        blob = _build(cfg, form)
        if blob is None:
            continue
        t = host.HipClassifier(sm.build_model(cfg), plan_only=True)
        o = host.HipClassifier(blob, plan_only=True)
        try:
            assert (o.n_samples, o.num_species(), o.emb_dim) == (t.n_samples, t.num_species(), t.emb_dim)
            tk = [s["kernel"] for s in t.describe()["steps"]]
            ok = [s["kernel"] for s in o.describe()["steps"]]
            # a Conv1d that only carries cos rows (real part) cannot give its window back: it runs literally; every other form fuses
            if form == "conv1d" and cfg.complex_mode == "real":
                assert "stft" not in ok and "conv_generic" in ok
            else:
                assert ok == tk, (name, form)
        finally:
            t.close(); o.close()


def test_untruncated_dft_matmul_and_unknown_uses_run_literally(built_lib):
    """A DFT basis the graph uses in a way the recogniser does not know (here: as a graph output) is the GEMM it literally is."""
    cfg = sm.tiny_config()
    b = ob.OnnxBuilder(opset=17)
    L = 256
    x = b.input("clip", ["N", 4096])
    idx = (np.arange(31)[:, None] * 1 + np.arange(2)[None, :]).astype(np.int64)          # 31 frames of 2 x 128 samples, hop 128
    fr = b.node("Reshape", [b.node("Gather", [b.node("Reshape", [x, b.init(np.asarray([1, 32, 128], np.int64))]), b.init(idx)], axis=1),
                            b.init(np.asarray([1, 31, L], np.int64))])
    n, k = np.arange(L)[:, None], np.arange(L // 2 + 1)[None, :]
    re = b.node("MatMul", [b.node("Mul", [fr, b.init(sm.hann_periodic(L))]), b.init(np.cos(2 * np.pi * n * k / L).astype(np.float32))])
    b.output(b.node("Identity", [re]), ["N", 31, L // 2 + 1])
    blob = b.finish()
    c = host.HipClassifier(blob, plan_only=True)
    try:
        kinds = [s["kernel"] for s in c.describe()["steps"]]
        assert "stft" not in kinds and "pw_gemm" in kinds and "copy" in kinds          # Gather as a strided view, the basis as a dense layer
    finally:
        c.close()
    assert cfg is not None


@pytest.mark.gpu
@pytest.mark.parametrize("form", FORMS)
def test_onnx_audio_models_hip_vs_oracle_and_vs_tflite_container(gpu, form):
    for name, cfg in _cfgs().items():
        blob = _build(cfg, form)
        if blob is None:
            continue
        x = sm.synth_clips(5, cfg.n_samples, cfg.sample_rate)
        ref = onnx_interp.run(blob, x)
        # (autotune off: the create-time tuner picks tiles by timing, and two engines may then sum in different orders)
        o = host.HipClassifier(blob, max_batch=8, autotune=False)
        t = host.HipClassifier(sm.build_model(cfg), max_batch=8, autotune=False)
        try:
            got = o.predict_batch(x.reshape(-1), 5)
            li = 3 if len(ref) == 4 else 0                    # Perch order: embedding, spatial, spectrogram, logits
            want = ref[li].reshape(got.shape)
            assert np.isfinite(got).all() and (got.argmax(1) == want.argmax(1)).all(), (name, form)
            if cfg.perch_outputs:
                sm_ = lambda v: np.exp(v - v.max(1, keepdims=True)) / np.exp(v - v.max(1, keepdims=True)).sum(1, keepdims=True)
                assert np.abs(sm_(got.astype(np.float64)) - sm_(want.astype(np.float64))).max() <= 1e-4, (name, form)
            else:
                sig = lambda v: 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
                assert np.abs(sig(got) - sig(want)).max() <= 1e-4, (name, form)
            tf = t.predict_batch(x.reshape(-1), 5)
            if form != "conv1d":
                assert np.array_equal(got, tf), (name, form)      # same plan, same weights: the container does not matter
            else:
                # the Conv1d form folds the window into its filters; recovered as hypot(cos row, sin row) it is the original
                # to an ulp, not to the bit (and a cos-only Conv1d is not recognised at all: literal fp32 convolution)
                assert np.abs(got - tf).max() < (1e-3 if cfg.complex_mode == "real" else 1e-4), (name, form)
        finally:
            o.close(); t.close()


@pytest.mark.gpu
def test_silence_through_the_dfttrunc_onnx_form(gpu):
    """The reference benchmark's own input (144000 zeros, cmd/benchmark/benchmark.go:99-101) amplifies front-end rounding
    through the power-law compression: the recognised ONNX form must behave like the TFLite form on it."""
    cfg = sm.tiny_config()
    x = np.zeros((2, cfg.n_samples), np.float32)
    o = host.HipClassifier(sm.build_model(cfg, container="onnx", dft="matmul"), max_batch=2, autotune=False)
    t = host.HipClassifier(sm.build_model(cfg), max_batch=2, autotune=False)
    try:
        a, b = o.predict_batch(x.reshape(-1), 2), t.predict_batch(x.reshape(-1), 2)
        assert np.isfinite(a).all() and np.array_equal(a, b)
    finally:
        o.close(); t.close()


@pytest.mark.gpu
def test_bat_pipeline_from_two_onnx_files(gpu):
    """Bat.Predict (internal/classifier/bat_onnx.go:220-342) the way the reference deploys it: the shared backbone is
    `birdnet-v2.4-embeddings-fp32-dfttrunc.onnx` (bat_onnx.go:252, model_catalog.go:490-501: two outputs, logits then the
    embedding), the regional head a second ONNX file (:282)."""
    cfg = sm.tiny_config(emit_embeddings=True)
    backbone_blob = sm.build_model(cfg, container="onnx", dft="matmul", trunc=True)
    head_blob, _ = ob.build_dense_head([cfg.top, 17], style="gemm", seed=23)
    labels = [f"Batus species{i}_Bat {i}" for i in range(17)]
    backbone = host.HipClassifier(backbone_blob, max_batch=4)
    head = host.CustomClassifier(head_blob, labels, max_batch=4)
    try:
        assert backbone.emb_dim == cfg.top
        bat = host.Bat(backbone, head, threshold=0.2)
        x = sm.synth_clips(3, cfg.n_samples, cfg.sample_rate)
        for i in range(3):
            got = bat.predict(x[i])
            emb = onnx_interp.run(backbone_blob, x[i:i + 1])[1]
            scores = G.sigmoid_f32div(onnx_interp.run(head_blob, emb)[0][0])
            order = np.argsort(-scores, kind="stable")
            want = [(labels[j], float(scores[j])) for j in order if scores[j] >= 0.2][:10]
            assert [g[0] for g in got] == [w[0] for w in want]
            assert np.allclose([g[1] for g in got], [w[1] for w in want], atol=2e-5)
    finally:
        backbone.close(); head.close()
