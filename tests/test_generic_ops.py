"""Planner coverage beyond the self-authored BirdNET topology (VERDICT r1 "Next" #2): graph passes + generic kernel tier.

CPU part: every random graph plans (`plan_only`) and the oracle executes it; the graph passes fold what they claim to.
GPU part: HIP (through the C ABI) vs the numpy oracle on >= 50 seeded random graphs and on the targeted patterns of a
TF -> TFLite EfficientNet export (PAD + VALID stride-2 convs, unfolded batch norm, pools, concat in the body, ...)."""
import numpy as np
import pytest

import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, tflite_schema as S
from birdnet_go_amd.tflite_build import GraphBuilder
from oracle.interp import Interpreter

from graphgen import f32, i32, random_graph, random_input

N_RANDOM = 64


def _close(got, ref, what, rtol=3e-4):
    scale = max(float(np.abs(ref).max()), 1.0)
    err = float(np.abs(got - ref).max())
    assert np.isfinite(got).all(), f"{what}: non-finite output"
    assert err <= rtol * scale, f"{what}: max |diff| {err:.3e} vs scale {scale:.3e}"


@pytest.mark.parametrize("seed", range(N_RANDOM))
def test_random_graph_plans_and_oracle_runs(built_lib, seed):
    blob, shape, ops = random_graph(seed)
    clf = host.HipClassifier(blob, plan_only=True)
    assert clf.n_samples == int(np.prod(shape)), ops
    d = clf.describe()
    assert d["steps"], ops
    y = Interpreter(blob).invoke(random_input(seed, shape, 2))[0]
    assert y.shape == (2, clf.num_species()) and np.isfinite(y).all(), ops
    clf.close()


def _bn_graph(depthwise=False, sub=False):
    g = GraphBuilder()
    rng = np.random.default_rng(5)
    x = g.tensor([1, 9, 11, 4], name="INPUT")
    if depthwise:
        y = g.op("DEPTHWISE_CONV_2D", [x, g.const(rng.standard_normal((1, 3, 3, 4)).astype(np.float32)), -1], [1, 9, 11, 4],
                 dict(padding=S.PAD_SAME, stride_w=1, stride_h=1, depth_multiplier=1, fused_activation_function=0,
                      dilation_w_factor=1, dilation_h_factor=1))
        co = 4
    else:
        y = g.op("CONV_2D", [x, g.const(rng.standard_normal((8, 3, 3, 4)).astype(np.float32)), g.const(f32(rng.standard_normal(8)))],
                 [1, 9, 11, 8], dict(padding=S.PAD_SAME, stride_w=1, stride_h=1, fused_activation_function=0,
                                     dilation_w_factor=1, dilation_h_factor=1))
        co = 8
    sh = [1, 9, 11, co]
    y = g.op("MUL", [y, g.const(f32(rng.uniform(0.5, 2.0, co)))], sh, {})
    if sub:
        y = g.op("SUB", [g.const(f32(rng.standard_normal(co))), y], sh, {})        # c - y
    y = g.op("ADD", [y, g.const(f32(rng.standard_normal(co)))], sh, dict(fused_activation_function=S.ACT_RELU6))
    m = g.op("MEAN", [y, g.const(i32([1, 2]))], [1, co], dict(keep_dims=0))
    return g.finish([x], [m]), (9, 11, 4)


@pytest.mark.parametrize("variant", ["conv", "dw", "sub"])
def test_unfolded_batchnorm_is_folded_into_the_convolution(built_lib, variant):
    blob, _ = _bn_graph(depthwise=variant == "dw", sub=variant == "sub")
    kinds = [s["kernel"] for s in host.HipClassifier(blob, plan_only=True).describe()["steps"]]
    assert "elementwise" not in kinds, kinds          # MUL / SUB / ADD(+RELU6) all live in the conv's weights, bias and activation


def _pad_conv_graph(H, W, k, depthwise):
    g = GraphBuilder()
    rng = np.random.default_rng(H * 100 + W + k)
    C = 8
    x = g.tensor([1, H, W, C], name="INPUT")
    pt, pb = (k - 1) // 2 - (1 if H % 2 == 0 else 0), (k - 1) // 2       # keras correct_pad
    pl, pr = (k - 1) // 2 - (1 if W % 2 == 0 else 0), (k - 1) // 2
    Hp, Wp = H + pt + pb, W + pl + pr
    p = g.op("PAD", [x, g.const(i32([[0, 0], [pt, pb], [pl, pr], [0, 0]]))], [1, Hp, Wp, C], {})
    Ho, Wo = (Hp - k) // 2 + 1, (Wp - k) // 2 + 1
    if depthwise:
        y = g.op("DEPTHWISE_CONV_2D", [p, g.const((rng.standard_normal((1, k, k, C)) / k).astype(np.float32)), g.const(f32(rng.standard_normal(C) * 0.1))],
                 [1, Ho, Wo, C], dict(padding=S.PAD_VALID, stride_w=2, stride_h=2, depth_multiplier=1, fused_activation_function=0,
                                      dilation_w_factor=1, dilation_h_factor=1))
        co = C
    else:
        co = 16
        y = g.op("CONV_2D", [p, g.const((rng.standard_normal((co, k, k, C)) / k / 3).astype(np.float32)), g.const(f32(rng.standard_normal(co) * 0.1))],
                 [1, Ho, Wo, co], dict(padding=S.PAD_VALID, stride_w=2, stride_h=2, fused_activation_function=S.ACT_RELU,
                                       dilation_w_factor=1, dilation_h_factor=1))
    m = g.op("MEAN", [y, g.const(i32([1, 2]))], [1, co], dict(keep_dims=0))
    return g.finish([x], [m]), (H, W, C)


@pytest.mark.parametrize("H,W,k,dw", [(12, 16, 3, False), (11, 15, 3, True), (12, 13, 5, True), (9, 9, 5, False)])
def test_pad_in_front_of_a_valid_conv_is_folded(built_lib, H, W, k, dw):
    blob, _ = _pad_conv_graph(H, W, k, dw)
    names = [s["kernel"] for s in host.HipClassifier(blob, plan_only=True).describe()["steps"]]
    assert "copy" not in names, names                  # no materialised PAD


def test_depth_multiplier_and_exotic_ops_report_by_name(built_lib):
    g = GraphBuilder()
    x = g.tensor([1, 64], name="INPUT")
    sel = g.const(i32([0, 5, 7]))
    y = g.op("GATHER", [x, sel], [1, 3], dict(axis=1, batch_dims=0))
    with pytest.raises(host.HipError, match="GATHER") as e:
        host.HipClassifier(g.finish([x], [y]), plan_only=True)
    assert e.value.code == host.E_UNSUPPORTED


# ------------------------------------------------------------------------------------------------ GPU: HIP vs oracle
@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(N_RANDOM))
def test_random_graph_hip_vs_oracle(gpu, seed):
    blob, shape, ops = random_graph(seed)
    x = random_input(seed, shape, 5)
    ref = Interpreter(blob).invoke(x)[0]
    clf = host.HipClassifier(blob, max_batch=4)                     # 5 clips through a max_batch of 4: two chunks
    try:
        got = clf.predict_batch(x.reshape(-1), 5)
    finally:
        clf.close()
    _close(got, ref, f"seed {seed}: {ops}")


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["conv", "dw", "sub"])
def test_folded_batchnorm_hip_vs_oracle(gpu, variant):
    blob, shape = _bn_graph(depthwise=variant == "dw", sub=variant == "sub")
    x = random_input(3, shape, 3)
    clf = host.HipClassifier(blob, max_batch=4)
    _close(clf.predict_batch(x.reshape(-1), 3), Interpreter(blob).invoke(x)[0], variant, rtol=1e-4)
    clf.close()


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,k,dw", [(12, 16, 3, False), (11, 15, 3, True), (12, 13, 5, True), (9, 9, 5, False), (48, 64, 3, True)])
def test_folded_pad_hip_vs_oracle(gpu, H, W, k, dw):
    blob, shape = _pad_conv_graph(H, W, k, dw)
    x = random_input(H + W, shape, 3)
    clf = host.HipClassifier(blob, max_batch=4)
    _close(clf.predict_batch(x.reshape(-1), 3), Interpreter(blob).invoke(x)[0], f"pad {H}x{W} k{k} dw{dw}", rtol=1e-4)
    clf.close()
