"""WAV ingest + framing (section 8f row 1), CPU only; conversion checked against the oracle's restatement
of convert/pcm.go and, when the reference checkout is present (this container), against tawnyowl.wav."""
import os
import struct

import numpy as np
import pytest

from birdnet_go_amd import wav
from oracle import gofuncs as G


def make_wav(samples_int, bits, rate=48000, extensible=False, channels=1):
    bps = bits // 8
    if bits == 16:
        data = np.asarray(samples_int, "<i2").tobytes()
    elif bits == 32:
        data = np.asarray(samples_int, "<i4").tobytes()
    else:
        data = b"".join(int(v & 0xFFFFFF).to_bytes(3, "little") for v in samples_int)
    if extensible:
        fmt = struct.pack("<HHIIHH", 0xFFFE, channels, rate, rate * bps * channels, bps * channels, bits)
        fmt += struct.pack("<HHI", 22, bits, 4) + struct.pack("<H", 1) + bytes.fromhex("000000001000800000aa00389b71")
    else:
        fmt = struct.pack("<HHIIHH", 1, channels, rate, rate * bps * channels, bps * channels, bits)
    body = b"WAVE" + b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 4) + b"abcd" + \
        b"data" + struct.pack("<I", len(data)) + data
    return b"RIFF" + struct.pack("<I", len(body)) + body


@pytest.mark.parametrize("bits", [16, 24, 32])
@pytest.mark.parametrize("ext", [False, True])
def test_read_matches_go_restatement(bits, ext):
    rng = np.random.default_rng(bits)
    lim = 1 << (bits - 1)
    v = rng.integers(-lim, lim, 1000, dtype=np.int64)
    v[:3] = [-lim, lim - 1, 0]
    raw = make_wav(v, bits, extensible=ext)
    s, rate, b = wav.read_wav(raw)
    assert (rate, b) == (48000, bits)
    data = raw[raw.index(b"data") + 8:]
    assert np.array_equal(s, G.pcm_to_f32(data, bits))


def test_rejects_bad_files():
    with pytest.raises(wav.WavError):
        wav.read_wav(b"nope")
    with pytest.raises(wav.WavError, match="bit depth"):
        wav.read_wav(make_wav([0, 1], 16).replace(struct.pack("<HH", 2, 16), struct.pack("<HH", 1, 8), 1))


def test_framing_overlap_and_tail():
    x = np.arange(48000 * 7, dtype=np.float32)
    c, t = wav.frame_clips(x, 48000, 3.0, 0.0)
    assert c.shape == (3, 144000) and list(t) == [0.0, 3.0, 6.0]
    assert c[2, 47999] == x[-1] and c[2, 48000] == 0.0            # zero-padded tail
    c, t = wav.frame_clips(x, 48000, 3.0, 1.5)
    assert list(t) == [0.0, 1.5, 3.0, 4.5] and np.array_equal(c[1], x[72000:72000 + 144000])
    c, t = wav.frame_clips(x[:48000 * 6 + 100], 48000, 3.0, 0.0)     # 100-sample tail < min_tail: dropped
    assert c.shape[0] == 2
    c, t = wav.frame_clips(x[:1000], 48000, 3.0, 0.0)                # short file still yields one padded clip
    assert c.shape[0] == 1
    with pytest.raises(ValueError):
        wav.frame_clips(x, 48000, 3.0, 3.0)


@pytest.mark.skipif(not os.path.exists("/root/reference/tawnyowl.wav"), reason="reference checkout absent")
def test_tawnyowl_is_five_clips():
    s, rate, bits = wav.read_wav("/root/reference/tawnyowl.wav")
    assert (rate, bits, s.size) == (48000, 32, 720000)
    c, t = wav.frame_clips(s, rate)
    assert c.shape == (5, 144000) and np.abs(s).max() <= 1.0
