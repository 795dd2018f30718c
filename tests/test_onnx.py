"""ONNX container (VERDICT r1 "Next" #8): dense heads as the reference ships them (`internal/classifier/bat_onnx.go:252-282`,
`internal/inference/onnx/custom_classifier.go:148-174`).  Writer: birdnet-go_amd/onnx_build.py; engine reader:
csrc/model_onnx.cpp; oracle: oracle/onnx_interp.py (independent protobuf parser + ONNX operator semantics)."""
import numpy as np
import pytest

import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, onnx_build as ob, synth_model as sm
from oracle import gofuncs as G, onnx_interp

STYLES = [("gemm", "Relu", None), ("matmul", "Relu", "Sigmoid"), ("bn", "Tanh", None), ("matmul", "LeakyRelu", "Softmax"),
          ("gemm", "Sigmoid", None)]


@pytest.mark.parametrize("style,act,final", STYLES)
def test_onnx_dense_head_plans(built_lib, style, act, final):
    blob, _ = ob.build_dense_head([1024, 64, 30], style=style, hidden_act=act, final=final)
    clf = host.HipClassifier(blob, plan_only=True)
    assert (clf.n_samples, clf.num_species(), clf.emb_dim) == (1024, 30, 0)
    kinds = [s["kernel"] for s in clf.describe()["steps"]]
    assert kinds.count("pw_gemm") == 2, kinds
    if style in ("matmul", "bn") and act in ("Relu",):
        assert "elementwise" not in kinds, kinds          # MatMul + Add (+Relu) fold into one dense step
    y = onnx_interp.run(blob, np.random.default_rng(0).standard_normal((3, 1024)).astype(np.float32))[0]
    assert y.shape == (3, 30) and np.isfinite(y).all()
    clf.close()


def test_onnx_reader_rejects_malformed_and_reports_unsupported(built_lib):
    blob, _ = ob.build_dense_head([16, 8])
    for cut in (1, 7, len(blob) // 3, len(blob) - 9):
        with pytest.raises(host.HipError) as e:
            host.HipClassifier(blob[:cut], plan_only=True)
        assert e.value.code in (host.E_MODEL, host.E_UNSUPPORTED)
    rng = np.random.default_rng(3)
    for _ in range(200):                                   # byte flips: clean error or a model, never a crash
        b = bytearray(blob)
        for p in rng.integers(0, len(b), 4):
            b[p] ^= int(rng.integers(1, 256))
        try:
            host.HipClassifier(bytes(b), plan_only=True).close()
        except host.HipError:
            pass
    b = ob.OnnxBuilder()
    x = b.input("x", ["N", 8])
    y = b.node("LSTM", [x])
    b.output(y, ["N", 8])
    with pytest.raises(host.HipError, match="LSTM") as e:
        host.HipClassifier(b.finish(), plan_only=True)
    assert e.value.code == host.E_UNSUPPORTED


def test_onnx_fp16_initializers_widen(built_lib):
    blob, _ = ob.build_dense_head([32, 16, 5], fp16_weights=True)
    clf = host.HipClassifier(blob, plan_only=True)
    assert clf.num_species() == 5
    clf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("style,act,final", STYLES)
def test_onnx_dense_head_hip_vs_oracle(gpu, style, act, final):
    blob, _ = ob.build_dense_head([1024, 96, 38], style=style, hidden_act=act, final=final, seed=5)
    x = np.random.default_rng(1).standard_normal((70, 1024)).astype(np.float32)
    ref = onnx_interp.run(blob, x)[0]
    clf = host.HipClassifier(blob, max_batch=64)
    got = clf.predict_batch(x.reshape(-1), 70)
    clf.close()
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), (style, act, final)


@pytest.mark.gpu
def test_bat_pipeline_with_onnx_head(gpu, tiny_cfg):
    """Bat.Predict (classifier/bat_onnx.go:220-342) with the head in the container the reference ships it in: TFLite
    backbone -> 64-d embedding -> ONNX regional head -> plain sigmoid (onnx/postprocess.go:8-10) -> threshold -> top-10."""
    from oracle.interp import Interpreter
    cfg = sm.tiny_config(emit_embeddings=True)
    backbone_blob = sm.build_model(cfg)
    head_blob, _ = ob.build_dense_head([cfg.top, 17], style="gemm", seed=23)
    labels = [f"Batus species{i}_Bat {i}" for i in range(17)]
    backbone = host.HipClassifier(backbone_blob, max_batch=4)
    head = host.CustomClassifier(head_blob, labels, max_batch=4)
    bat = host.Bat(backbone, head, threshold=0.2)
    x = sm.synth_clips(2, cfg.n_samples, cfg.sample_rate)
    for i in range(2):
        got = bat.predict(x[i])
        _, emb = Interpreter(backbone_blob).invoke(x[i:i + 1])
        scores = G.sigmoid_f32div(onnx_interp.run(head_blob, emb)[0][0])
        order = np.argsort(-scores, kind="stable")
        want = [(labels[j], float(scores[j])) for j in order if scores[j] >= 0.2][:10]
        assert [g[0] for g in got] == [w[0] for w in want]
        assert np.allclose([g[1] for g in got], [w[1] for w in want], atol=1e-5)
    backbone.close(); head.close()
