"""Pins the oracle's restatement of the reference's in-tree Go arithmetic against the known answers
the reference's own tests hold.  CPU only."""
import math

import numpy as np
import pytest

from oracle import gofuncs as G


# ---- internal/classifier/analyze_test.go:471-540 (TestApplySigmoidToPredictions)
def test_sigmoid_zero():
    r = G.sigmoid_sensitivity([0, 0, 0], 1.0)
    assert np.allclose(r, 0.5, atol=1e-4)


def test_sigmoid_symmetry():
    r = G.sigmoid_sensitivity([-2, -1, 0, 1, 2], 1.0)
    assert abs(r[0] + r[4] - 1.0) < 1e-4 and abs(r[1] + r[3] - 1.0) < 1e-4 and abs(r[2] - 0.5) < 1e-4


def test_sigmoid_sensitivity_effect():
    r = G.sigmoid_sensitivity([1.0], 2.0)
    assert r[0] > 1.0 / (1.0 + math.exp(-1.0))
    assert abs(r[0] - 1.0 / (1.0 + math.exp(-2.0))) < 1e-4


def test_sigmoid_empty():
    assert G.sigmoid_sensitivity(np.zeros(0, np.float32), 1.0).size == 0


# ---- internal/classifier/analyze_test.go:649-700 (TestGetTopKResults)
def test_topk_normal_case():
    conf = [0.9, 0.7, 0.8, 0.6, 0.5]
    c, i = G.topk(conf, 3)
    assert list(i) == [0, 2, 1] and np.all(np.diff(c) <= 0)


def test_topk_k_equals_len():
    c, i = G.topk([0.5, 0.9, 0.7], 3)
    assert np.allclose(c, [0.9, 0.7, 0.5], atol=1e-4)


def test_topk_k_greater_than_len():
    c, i = G.topk([0.8, 0.9], 5)
    assert len(c) == 2 and np.allclose(c, [0.9, 0.8])


def test_topk_empty_and_nonpositive_k():
    assert len(G.topk(np.zeros(0, np.float32), 3)[0]) == 0
    assert len(G.topk([0.1, 0.2], 0)[0]) == 0


def test_topk_matches_full_sort_random():
    rng = np.random.default_rng(0)
    for n, k in [(6522, 10), (100, 10), (11, 10), (10, 10)]:
        x = rng.random(n).astype(np.float32)
        c, i = G.topk(x, k)
        want = np.sort(x)[::-1][:k]
        assert np.array_equal(c, want) and np.array_equal(x[i], c)


# ---- internal/audiocore/convert/pcm_test.go:256-338 (TestConvertToFloat32)
def test_pcm16_known_answers():
    assert abs(G.pcm_to_f32(bytes([0, 0]), 16)[0]) < 1e-6
    assert abs(G.pcm_to_f32(bytes([0xFF, 0x7F]), 16)[0] - np.float32(32767) / np.float32(32768)) < 1e-5
    assert abs(G.pcm_to_f32(bytes([0x00, 0x80]), 16)[0] + 1.0) < 1e-6
    r = G.pcm_to_f32(bytes([0, 0, 0, 0x40, 0, 0xC0]), 16)
    assert np.allclose(r, [0.0, 0.5, -0.5], atol=1e-5)


def test_pcm24_32_known_answers():
    assert abs(G.pcm_to_f32(bytes([0, 0, 0]), 24)[0]) < 1e-6
    assert G.pcm_to_f32(bytes([0xFF, 0xFF, 0xFF]), 24)[0] < 0
    assert abs(G.pcm_to_f32(bytes([0, 0, 0, 0]), 32)[0]) < 1e-6
    assert G.pcm_to_f32(bytes([0, 0, 0, 0x80]), 32)[0] == -1.0
    with pytest.raises(ValueError):
        G.pcm_to_f32(bytes([0]), 8)


# ---- internal/inference/onnx/postprocess_test.go
def test_onnx_sigmoid_and_softmax():
    assert abs(G.sigmoid_f32div([0])[0] - 0.5) < 1e-6
    s = G.softmax([1, 2, 3])
    assert abs(float(np.sum(s, dtype=np.float32)) - 1.0) < 1e-6
    ref = np.exp(np.array([1, 2, 3], np.float64) - 3)
    assert np.allclose(s, ref / ref.sum(), atol=1e-7)


# ---- internal/audiocore/ultrasonic/filter_test.go
SR, N = 256000, 144000


def test_us_flat_tone_low_cv():
    t = np.arange(N) / SR
    cv, ok = G.us_frame_cv(0.01 * np.sin(2 * np.pi * 40000.0 * t), SR)
    assert ok and cv < 0.15


def test_us_burst_high_cv():
    s = np.zeros(N)
    i = np.arange(N // 3, 2 * N // 3)
    s[i] = 0.5 * np.sin(2 * np.pi * 45000.0 * i / SR)
    cv, ok = G.us_frame_cv(s, SR)
    assert ok and cv > 0.15


def test_us_guards():
    assert not G.us_frame_cv(np.zeros(100), SR)[1]                      # fewer samples than FFT
    assert not G.us_frame_cv(np.zeros(20000), 48000, split_hz=30000)[1]  # split >= Nyquist
    assert not G.us_frame_cv(np.zeros(20000), SR, fft_size=6000)[1]      # not a power of two
    assert not G.us_frame_cv(np.zeros(8192), SR)[1]                      # one frame
    assert G.us_frame_cv(np.zeros(8192 + 4096), SR)[1]                   # two frames = minimum


def test_is_unlikely_threshold():
    assert G.is_unlikely(0.05) and G.is_unlikely(0.14)
    assert not G.is_unlikely(0.15) and not G.is_unlikely(0.50)


def test_hanning_and_fft_and_cv():
    w = G.hanning(8)
    assert abs(w[0]) < 1e-10 and abs(w[7]) < 1e-10 and int(np.argmax(w)) in (3, 4)
    n = 256
    d = G.fft_c128(np.sin(2 * np.pi * 10 * np.arange(n) / n))
    mag = np.abs(d[:n // 2])
    assert int(np.argmax(mag)) == 10 and mag.max() > 50.0
    assert np.allclose(d, np.fft.fft(np.sin(2 * np.pi * 10 * np.arange(n) / n)), atol=1e-9)
    assert abs(G.coefficient_of_variation([5, 5, 5, 5])) < 1e-10
    assert abs(G.coefficient_of_variation([1, 2, 3]) - math.sqrt(2.0 / 3.0) / 2.0) < 1e-10
    assert G.coefficient_of_variation([1]) == 0.0 and G.coefficient_of_variation([]) == 0.0


def test_us_scale_invariance():
    s = np.zeros(N)
    i = np.arange(N // 4, N // 2)
    s[i] = 0.5 * np.sin(2 * np.pi * 45000.0 * i / SR)
    cv1, ok1 = G.us_frame_cv(s, SR)
    cv2, ok2 = G.us_frame_cv(s * 10.0, SR)
    assert ok1 and ok2 and abs(cv1 - cv2) < 0.01


# ---- resampler edges: internal/audiocore/resample/resample.go:161-169
def test_resample_edge_truncation():
    out = G.resample_edge_out([0.0, 1.0, -1.0, 2.0, -2.0, 0.99999, -0.00002])
    assert list(out) == [0, 32767, -32767, 32767, -32767, 32766, 0]
