"""Streaming resampler state (VERDICT r1 "Next" #9; reference: internal/audiocore/resample/resample.go:44-224, fed ~100 ms
frames by analysis/buffer_consumer.go:118,192).  Contract: any chunking of a stream yields, concatenated (+ flush), exactly
the samples of one one-shot call over the whole stream - bit for bit - with the reference wrapper's PCM16 edges, its
"empty input writes nothing", "destination too small fails before the state advances" and NewResampler(equal rates) = nil."""
import numpy as np
import pytest

from birdnet_go_amd import host


def test_new_resampler_equal_rates_is_none(built_lib):
    assert host.StreamResampler.new(48000, 48000) is None          # NewResampler returns nil, nil (resample.go:58-60)
    lib = host.load_library()
    import ctypes as C
    lib.bnhip_resampler_create.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    h = C.c_void_p(123)
    assert lib.bnhip_resampler_create(0, 48000, 48000, C.byref(h)) == 0 and not h.value
    assert lib.bnhip_resampler_create(0, 0, 48000, C.byref(h)) == host.E_INVALID
    lib.bnhip_resampler_destroy(None)                               # Close on nil is safe


def _signal(n, rate, seed):
    rng = np.random.default_rng(seed)
    t = np.arange(n) / rate
    return 0.45 * np.sin(2 * np.pi * 997.0 * t) + 0.3 * np.sin(2 * np.pi * 6151.0 * t) + 0.1 * rng.standard_normal(n)


@pytest.mark.gpu
@pytest.mark.parametrize("fr,to", [(48000, 32000), (44100, 48000), (256000, 48000), (16000, 48000), (22050, 48000)])
def test_chunked_pcm16_equals_one_shot_bit_for_bit(gpu, fr, to):
    n = int(fr * 1.3) + 17
    pcm = np.clip(_signal(n, fr, fr + to) * 32767 * 1.2, -32768, 32767).astype("<i2")      # includes clipped peaks
    one = host.Resampler(fr, to).resample_to(pcm)
    rng = np.random.default_rng(7)
    for frame in (fr // 10, 1, 997, None):                          # ~100 ms frames, single samples at the start, odd, random
        rs = host.StreamResampler(fr, to)
        out, pos, calls = [], 0, 0
        while pos < n:
            m = frame if frame else int(rng.integers(1, fr // 5))
            if frame == 1 and calls >= 300:
                m = n - pos                                          # 300 one-sample calls, then the rest
            chunk = pcm[pos:pos + m]
            got = rs.resample_into(chunk.tobytes())
            assert len(got) <= rs.estimate_output_bytes(chunk.size * 2)
            out.append(np.frombuffer(got, "<i2"))
            pos += chunk.size
            calls += 1
        out.append(np.frombuffer(rs.flush(), "<i2"))
        got = np.concatenate(out)
        assert got.size == one.size, (frame, got.size, one.size)
        assert np.array_equal(got, one), (frame, int(np.abs(got.astype(int) - one.astype(int)).max()))
        # after flush the handle starts a new stream: same input, same output again
        again = np.concatenate([np.frombuffer(rs.resample_into(pcm.tobytes()), "<i2"), np.frombuffer(rs.flush(), "<i2")])
        assert np.array_equal(again, one)
        rs.close()


@pytest.mark.gpu
def test_chunked_f32_equals_one_shot_and_edges(gpu):
    fr, to = 48000, 32000
    x = _signal(50000, fr, 3).astype(np.float32)
    one = host.Resampler(fr, to).resample_f32(x)
    rs = host.StreamResampler(fr, to)
    parts = [rs.process_f32(x[i:i + 4800]) for i in range(0, x.size, 4800)]
    parts.append(rs.flush(pcm16=False))
    assert np.array_equal(np.concatenate(parts), one)
    # empty input writes nothing and does not disturb the stream (resample.go:100-102)
    assert rs.resample_into(b"") == b""
    # odd byte count is a validation error (resample.go:104-111)
    with pytest.raises(host.HipError, match="not a multiple of 2"):
        rs.resample_to(b"\x01\x02\x03", bytearray(64))
    # a destination smaller than the estimate fails BEFORE the state advances (resample.go:137-144): the same frame then
    # succeeds with a proper buffer and the stream continues as if the failed call never happened
    pcm = (x[:9600] * 20000).astype("<i2")
    ref = host.StreamResampler(fr, to)
    a1 = ref.resample_into(pcm[:4800].tobytes())
    a2 = ref.resample_into(pcm[4800:].tobytes())
    b1 = rs.resample_into(pcm[:4800].tobytes())
    with pytest.raises(host.HipError, match="too small"):
        rs.resample_to(pcm[4800:].tobytes(), bytearray(16))
    b2 = rs.resample_into(pcm[4800:].tobytes())
    assert a1 == b1 and a2 == b2
    rs.close(); ref.close()
    with pytest.raises(host.HipError, match="closed"):
        rs.resample_into(pcm.tobytes())
