"""General convolutions (kh x kw over >= 4 input channels: fused-MBConv blocks, ResNet-style stacks - anything that is not the
1x1 / depthwise vocabulary of the v2.4 topology) run as an implicit GEMM on the f32 MFMA (k_pw_gemm<IM>): kernel sizes, strides,
dilations, SAME / VALID padding, channel counts that are not tile multiples, fused bias + activation, vs the oracle."""
import numpy as np
import pytest

from birdnet_go_amd import host, tflite_schema as S
from birdnet_go_amd.tflite_build import GraphBuilder
from oracle.interp import Interpreter

CASES = [
    # H, W, Cin, [(Cout, k, stride, dilation, padding, fused act)]
    (19, 23, 4, [(16, 3, 1, 1, "SAME", 1), (24, 3, 2, 1, "SAME", 0), (10, 5, 1, 1, "VALID", 3)]),
    (16, 40, 8, [(8, 3, 1, 2, "SAME", 0), (20, 1, 2, 1, "SAME", 1), (32, 3, 1, 1, "VALID", 0)]),
    (9, 64, 12, [(36, 5, 2, 1, "SAME", 1), (7, 3, 1, 1, "SAME", 0)]),
    (33, 17, 32, [(48, 3, 1, 1, "SAME", 1), (48, 3, 2, 1, "VALID", 0), (64, 2, 1, 1, "SAME", 1)]),
]


def build(case, seed):
    H, W, C, layers = case
    rng = np.random.default_rng(seed)
    g = GraphBuilder(description="conv stack")
    x = g.tensor([1, H, W, C], name="image")
    t = x
    for li, (co, k, s, d, pad, act) in enumerate(layers):
        w = (rng.standard_normal((co, k, k, C)) / np.sqrt(k * k * C)).astype(np.float32)
        b = (rng.standard_normal(co) * 0.1).astype(np.float32)
        e = d * (k - 1) + 1
        Ho, Wo = (-(-H // s), -(-W // s)) if pad == "SAME" else ((H - e) // s + 1, (W - e) // s + 1)
        t = g.op("CONV_2D", [t, g.const(w, f"c{li}/w"), g.const(b, f"c{li}/b")], [1, Ho, Wo, co],
                 dict(padding=S.PAD_SAME if pad == "SAME" else S.PAD_VALID, stride_w=s, stride_h=s, fused_activation_function=act,
                      dilation_w_factor=d, dilation_h_factor=d), name=f"c{li}")
        H, W, C = Ho, Wo, co
    m = g.op("MEAN", [t, g.const(np.asarray([1, 2], np.int32))], [1, C], dict(keep_dims=0))
    wh = (rng.standard_normal((9, C)) / np.sqrt(C)).astype(np.float32)
    y = g.op("FULLY_CONNECTED", [m, g.const(wh, "head/w"), g.const(np.zeros(9, np.float32), "head/b")], [1, 9],
             dict(fused_activation_function=S.ACT_NONE))
    return g.finish([x], [y, t])


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_convolutions_plan_as_implicit_gemm(built_lib, ci):
    c = host.HipClassifier(build(CASES[ci], 3 + ci), plan_only=True)
    try:
        kinds = [s["kernel"] for s in c.describe()["steps"]]
        cin, want = CASES[ci][2], 0
        for (co, k, s, d, pad, act) in CASES[ci][3]:
            want += 1 if k * k * cin >= 32 else 0        # (a 1x1 stride-2 convolution over 8 channels stays on the direct kernel)
            cin = co
        assert kinds.count("conv_igemm") == want and want >= 2, kinds
    finally:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("ci", range(len(CASES)))
def test_implicit_gemm_convolutions_vs_oracle(gpu, ci):
    blob = build(CASES[ci], 3 + ci)
    H, W, C, _ = CASES[ci]
    x = np.random.default_rng(40 + ci).standard_normal((6, H, W, C)).astype(np.float32)
    ref = Interpreter(blob).invoke(x)
    for nb in (6, 1):
        c = host.HipClassifier(blob, max_batch=8)
        try:
            got, feat = c.predict_batch(x[:nb].reshape(-1), nb, want_embeddings=True)
        finally:
            c.close()
        assert np.abs(got - ref[0][:nb]).max() < 2e-5 * max(1.0, float(np.abs(ref[0]).max()))
        assert np.abs(feat.reshape(ref[1][:nb].shape) - ref[1][:nb]).max() < 2e-5 * max(1.0, float(np.abs(ref[1]).max()))
