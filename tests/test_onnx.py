"""ONNX container (VERDICT r1 "Next" #8): dense heads as the reference ships them (`internal/classifier/bat_onnx.go:252-282`,
`internal/inference/onnx/custom_classifier.go:148-174`).  Writer: birdnet-go_amd/onnx_build.py; engine reader:
csrc/model_onnx.cpp; oracle: oracle/onnx_interp.py (independent protobuf parser + ONNX operator semantics)."""
import numpy as np
import pytest

import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, onnx_build as ob, synth_model as sm
from oracle import gofuncs as G, onnx_interp
from oracle import onnx_interp as oi

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


# ------------------------------------------------------------------------------------------------ convolutional ONNX graphs
CNN_VARIANTS = [("torch", False), ("torch", True), ("tf", False), ("tf", True)]


@pytest.mark.parametrize("style,nhwc", CNN_VARIANTS)
def test_onnx_cnn_plans_onto_the_fused_kernels(built_lib, style, nhwc):
    """NCHW Conv graphs (a8: `internal/inference/onnx/classifier.go:268-430` runs convolutional classifiers through ORT) are
    lowered channels-last: the MBConv / squeeze-excite / residual patterns land on the same fused kernels as the TFLite form,
    the second output is recognised as the embedding, and nothing but the tf-style extras needs the generic tier."""
    blob = ob.build_cnn(style=style, nhwc_input=nhwc, emit_embedding=True)
    c = host.HipClassifier(blob, plan_only=True)
    try:
        assert (c.n_samples, c.num_species(), c.emb_dim) == (40 * 56, 21, 48)
        kinds = [s["kernel"] for s in c.describe()["steps"]]
        assert kinds.count("expand_dw") == 3 and kinds.count("se") >= 3 and "conv_generic" not in kinds
        assert not any(k.startswith("generic_copy") or k == "transpose" for k in kinds), kinds     # no layout copies at all
    finally:
        c.close()


def test_onnx_cnn_reader_reports_what_it_cannot_lower(built_lib):
    def graph(mut):
        b = ob.OnnxBuilder()
        x = b.input("x", ["N", 4, 8, 8])
        w = np.zeros((6, 4, 3, 3), np.float32)
        y = mut(b, x, w)
        b.output(y, ["N", 6, 8, 8])
        return b.finish()
    cases = {
        "group": lambda b, x, w: b.node("Conv", [x, b.init(w[:, :2])], kernel_shape=[3, 3], group=2, pads=[1, 1, 1, 1]),
        "weight shape": lambda b, x, w: b.node("Conv", [x, b.init(w[:, :3])], kernel_shape=[3, 3], pads=[1, 1, 1, 1]),
        "kernel_shape": lambda b, x, w: b.node("Conv", [x, b.init(w)], kernel_shape=[5, 5], pads=[1, 1, 1, 1]),
        "padding": lambda b, x, w: b.node("Conv", [x, b.init(w)], kernel_shape=[3, 3], pads=[3, 3, 3, 3]),
        "auto_pad": lambda b, x, w: b.node("Conv", [x, b.init(w)], kernel_shape=[3, 3], auto_pad="SAME_SIDEWAYS"),
        "count_include_pad": lambda b, x, w: b.node("AveragePool", [x], kernel_shape=[2, 2], count_include_pad=1),
        "ceil_mode": lambda b, x, w: b.node("MaxPool", [x], kernel_shape=[2, 2], strides=[2, 2], ceil_mode=1),
    }
    for key, mut in cases.items():
        with pytest.raises(host.HipError, match=key) as e:
            host.HipClassifier(graph(mut), plan_only=True)
        assert e.value.code in (host.E_MODEL, host.E_UNSUPPORTED), key


@pytest.mark.gpu
@pytest.mark.parametrize("style,nhwc", CNN_VARIANTS)
def test_onnx_cnn_hip_vs_oracle(gpu, style, nhwc):
    """HIP (ONNX reader -> channels-last IR -> fused kernels) vs the oracle's NCHW numpy execution of the same file: explicit
    pads and auto_pad, depthwise groups, squeeze-excite through GlobalAveragePool, HardSigmoid, unfolded BatchNormalization,
    AveragePool, ReduceMean without keepdims, NHWC input behind a Transpose, two outputs."""
    blob = ob.build_cnn(style=style, nhwc_input=nhwc, emit_embedding=True, seed=9)
    x = np.random.default_rng(4).standard_normal((5, 40, 56, 1) if nhwc else (5, 1, 40, 56)).astype(np.float32)
    ref = oi.run(blob, x)
    c = host.HipClassifier(blob, max_batch=8)
    try:
        got, emb = c.predict_batch(x.reshape(-1), 5, want_embeddings=True)
    finally:
        c.close()
    assert np.abs(got - ref[0]).max() < 1e-4 * max(1.0, np.abs(ref[0]).max()) and np.abs(emb - ref[1]).max() < 1e-4
    assert (got.argmax(1) == ref[0].argmax(1)).all()


@pytest.mark.gpu
def test_onnx_cnn_multichannel_image_and_image_output(gpu):
    """A 3-channel NCHW input (real layout change in front of the first Conv), odd sizes with asymmetric SAME padding, MaxPool,
    channel Concat, a per-channel constant [1,C,1,1] and an image-shaped graph output (leaves in NCHW order)."""
    rng = np.random.default_rng(12)
    b = ob.OnnxBuilder()
    x = b.input("img", ["N", 3, 19, 23])
    w1 = (rng.standard_normal((8, 3, 3, 3)) * 0.3).astype(np.float32)
    t = b.node("Conv", [x, b.init(w1), b.init(rng.standard_normal(8).astype(np.float32) * 0.1)], kernel_shape=[3, 3], strides=[2, 2], auto_pad="SAME_UPPER")
    t = b.node("Relu", [t])
    p = b.node("MaxPool", [t], kernel_shape=[2, 2], strides=[2, 2])
    q = b.node("AveragePool", [t], kernel_shape=[2, 2], strides=[2, 2])
    t = b.node("Concat", [p, q], axis=1)
    t = b.node("Mul", [t, b.init(rng.uniform(0.5, 1.5, (1, 16, 1, 1)).astype(np.float32))])
    w2 = (rng.standard_normal((4, 16, 1, 1)) * 0.3).astype(np.float32)
    t = b.node("Conv", [t, b.init(w2)], kernel_shape=[1, 1])
    b.output(t, ["N", 4, 5, 6])
    blob = b.finish()
    xv = rng.standard_normal((3, 3, 19, 23)).astype(np.float32)
    ref = oi.run(blob, xv)[0]
    assert ref.shape == (3, 4, 5, 6)
    c = host.HipClassifier(blob, max_batch=4)
    try:
        assert c.num_species() == 4 * 5 * 6
        got = c.predict_batch(xv.reshape(-1), 3).reshape(3, 4, 5, 6)
    finally:
        c.close()
    assert np.abs(got - ref).max() < 1e-5


SEEDS = list(range(40))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_onnx_cnn_plans_and_oracle_runs(built_lib, seed):
    """CPU: every seeded random NCHW graph is executed by the oracle and accepted by the reader + planner (no GPU needed)."""
    from onnxgen import random_cnn, random_image
    blob, in_shape, n_cls = random_cnn(seed)
    y = oi.run(blob, random_image(seed, in_shape, 2))[0]
    assert y.shape == (2, n_cls) and np.isfinite(y).all()
    c = host.HipClassifier(blob, plan_only=True)
    try:
        assert c.num_species() == n_cls and c.n_samples == int(np.prod(in_shape))
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", SEEDS)
def test_random_onnx_cnn_hip_vs_oracle(gpu, seed):
    from onnxgen import random_cnn, random_image
    blob, in_shape, n_cls = random_cnn(seed)
    x = random_image(seed, in_shape, 3)
    ref = oi.run(blob, x)[0]
    c = host.HipClassifier(blob, max_batch=4)
    try:
        got = c.predict_batch(x.reshape(-1), 3)
    finally:
        c.close()
    assert np.abs(got - ref).max() <= 2e-5 * max(1.0, float(np.abs(ref).max())), (seed, np.abs(got - ref).max())


@pytest.mark.parametrize("op", ["QuantizeLinear", "DequantizeLinear", "QLinearConv", "QLinearMatMul", "MatMulInteger", "ConvInteger", "DynamicQuantizeLinear"])
def test_onnx_quantized_operators_are_reported_unsupported_by_name(built_lib, op):
    """The reference's INT8 ONNX variants (`int8-arm`, `int8-arm-dfttrunc`: internal/classifier/model_catalog.go:315-318,430-433) are
    aarch64-only builds (`Requirements.Arch`) and not a deployment an MI355X host sees; a file of that family handed to the engine
    anyway is refused with BNHIP_E_UNSUPPORTED and the operator's name, which the Go shim turns into "fall back to the existing
    backend" (model_openvino.go:227-230) - never a crash, never a silent float reinterpretation of quantised tensors."""
    b = ob.OnnxBuilder()
    x = b.input("x", ["N", 8])
    y = b.node(op, [x, b.init(np.asarray([0.1], np.float32))])
    b.output(y, ["N", 8])
    with pytest.raises(host.HipError, match=op) as e:
        host.HipClassifier(b.finish(), plan_only=True)
    assert e.value.code == host.E_UNSUPPORTED
