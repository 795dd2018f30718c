"""GPU polyphase resampler (section 8 rows a2 / f2).  Length rules and guards run on CPU; sample parity vs the scipy oracle
needs the GPU.  The reference pins only output length (+-5 %): resample_test.go:26-55,151-168."""
import numpy as np
import pytest

from birdnet_go_amd import host
from oracle import resample as R

RATES = [(48000, 32000), (44100, 48000), (256000, 48000), (16000, 48000), (48000, 48000), (22050, 48000)]


def test_length_rule_matches_reference_tolerance(built_lib):
    for fr, to in RATES:
        rs = host.Resampler(fr, to)
        for n in (1, 100, 4800, 144000):
            got = rs.estimate_output(n)
            assert got == R.expected_length(n, fr, to)
            assert abs(got - n * to / fr) <= max(1, 0.05 * n * to / fr)       # the reference's own +-5 % bound
    with pytest.raises(host.HipError):
        host.Resampler(0, 48000)


def test_equal_rates_pass_through_without_gpu(built_lib):
    x = np.random.default_rng(0).standard_normal(1000).astype(np.float32)
    assert np.array_equal(host.Resampler(48000, 48000).resample_f32(x), x)
    with pytest.raises(host.HipError, match="not a multiple of 2"):
        host.Resampler(48000, 32000).resample_to(b"\x00\x01\x02")


@pytest.mark.gpu
@pytest.mark.parametrize("fr,to", [r for r in RATES if r[0] != r[1]])
def test_f32_matches_scipy_design(fr, to):
    rng = np.random.default_rng(fr + to)
    n = 30000
    t = np.arange(n) / fr
    x = (0.4 * np.sin(2 * np.pi * 1000.0 * t) + 0.3 * np.sin(2 * np.pi * 5500.0 * t) + 0.05 * rng.standard_normal(n)).astype(np.float32)
    xb = np.stack([x, x[::-1].copy(), np.zeros(n, np.float32)])
    got = host.Resampler(fr, to).resample_f32(xb)
    ref = R.resample_f64(xb, fr, to)
    assert got.shape == ref.shape
    assert np.abs(got - ref).max() < 5e-6
    assert np.abs(got[2]).max() == 0.0


@pytest.mark.gpu
def test_pcm16_edges_follow_the_reference_wrapper():
    rng = np.random.default_rng(9)
    n = 48000
    x = (0.9 * np.sin(2 * np.pi * 440.0 * np.arange(n) / 48000) * 32767).astype(np.int16)
    x[:100] = 32767                                   # drives the filter overshoot into the clamp
    x[100:200] = -32768
    got = host.Resampler(48000, 32000).resample_to(x)
    ref = R.resample_pcm16(x, 48000, 32000)
    assert got.shape == ref.shape == (32000,)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and (d > 0).mean() < 0.01       # truncation can flip one LSB where fp32/fp64 sums straddle an integer
    assert got.max() == 32767 and got.min() == -32767   # clamp(+-1) * 32767: -32768 is unreachable, as in the reference
    raw = host.Resampler(48000, 32000).resample_to(x.tobytes())
    assert raw == got.tobytes()


@pytest.mark.gpu
def test_tone_survives_and_alias_is_rejected():
    fr, to, n = 48000, 32000, 48000
    t = np.arange(n) / fr
    keep = host.Resampler(fr, to).resample_f32(np.sin(2 * np.pi * 3000.0 * t).astype(np.float32))
    kill = host.Resampler(fr, to).resample_f32(np.sin(2 * np.pi * 20000.0 * t).astype(np.float32))   # above the new Nyquist
    assert 0.95 < np.abs(keep[2000:-2000]).max() < 1.05
    assert np.abs(kill[2000:-2000]).max() < 0.02


# ------------------------------------------------------------------------------------------------ the filter spec, as numbers
# the reference's resampler values are unpinned, so the project states its own spec.  Prototype low-pass =
# Kaiser window (beta 5.0), 20 x max(L, M) + 1 taps, cutoff at the lower of the two Nyquist frequencies (what
# scipy.signal.resample_poly designs by default), applied zero-phase (no group delay in the output: sample i is time i / rate_out):
#   passband   0 .. 0.80 x Nyquist_low : ripple <= 0.03 dB (peak to peak)
#   cutoff     Nyquist_low             : -6.0 dB
#   stopband   >= 1.25 x Nyquist_low   : attenuation >= 55 dB (the first side lobe; 50 dB asserted on the GPU with finite tones)
SPEC_PAIRS = [(48000, 32000), (256000, 48000), (44100, 48000), (32000, 48000)]


@pytest.mark.parametrize("fr,to", SPEC_PAIRS)
def test_filter_design_meets_the_stated_spec(fr, to):
    from math import gcd
    from scipy import signal
    g = gcd(fr, to)
    L, M = to // g, fr // g
    half = 10 * max(L, M)
    h = signal.firwin(2 * half + 1, 1.0 / max(L, M), window=("kaiser", 5.0)) * L
    n = 1 << 22
    H = np.abs(np.fft.rfft(h, n)) / L
    f = np.arange(len(H)) * (fr * L) / n
    nyq = min(fr, to) / 2
    pb = H[f <= 0.80 * nyq]
    assert 20 * np.log10(pb.max() / pb.min()) <= 0.03
    assert abs(20 * np.log10(H[np.argmin(np.abs(f - nyq))]) + 6.02) < 0.05
    assert -20 * np.log10(H[f >= 1.25 * nyq].max()) >= 55.0


@pytest.mark.gpu
@pytest.mark.parametrize("fr,to", SPEC_PAIRS)
def test_gpu_resampler_meets_the_stated_spec(fr, to):
    """The same three figures measured THROUGH the GPU resampler with tones (Hann-weighted RMS over the middle of the clip),
    plus the zero group delay: an impulse stays where it was."""
    n = fr // 2
    t = np.arange(n) / fr
    nyq = min(fr, to) / 2
    rs = host.Resampler(fr, to)

    def gain_db(freq):
        y = rs.resample_f32(np.sin(2 * np.pi * freq * t).astype(np.float32)).astype(np.float64)
        mid = y[len(y) // 4: 3 * len(y) // 4]
        w = np.hanning(len(mid))
        return 10 * np.log10(2 * np.sum(w * mid * mid) / np.sum(w) + 1e-30)
    for frac in (0.1, 0.5, 0.8):
        assert abs(gain_db(frac * nyq)) <= 0.05, (fr, to, frac)
    # (the -6 dB point sits ON the output's Nyquist frequency, where a tone and its image fold onto each other: it is checked
    # on the design above, not with a tone)
    if 1.25 * nyq < fr / 2:                              # (only a down-sampler can be fed a tone above the new Nyquist)
        assert gain_db(1.3 * nyq) <= -50.0
    imp = np.zeros(n, np.float32)
    imp[n // 2] = 1.0
    y = rs.resample_f32(imp)
    assert abs(int(np.argmax(np.abs(y))) - round((n // 2) * to / fr)) <= 1
