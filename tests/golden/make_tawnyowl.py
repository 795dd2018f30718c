"""Generates tests/golden/tawnyowl_pcm32.npz from the reference's own audio fixture /root/reference/tawnyowl.wav
(RIFF WAVE_FORMAT_EXTENSIBLE, mono, 48 kHz, 32-bit PCM, 720 000 samples = five 3 s clips).
/root/reference does not exist on the GPU box, so the samples travel as a committed fixture: the int32 PCM words,
delta-coded (first differences, wrapping int32 arithmetic) and zlib-compressed by numpy.  Run in the build container:
    python tests/golden/make_tawnyowl.py
"""
import hashlib
import os
import struct

import numpy as np

SRC = "/root/reference/tawnyowl.wav"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tawnyowl_pcm32.npz")

raw = open(SRC, "rb").read()
i = raw.index(b"data")
n = struct.unpack_from("<I", raw, i + 4)[0]
pcm = np.frombuffer(raw[i + 8:i + 8 + n], "<i4")
assert pcm.size == 720000
delta = np.diff(pcm.astype(np.int64), prepend=0).astype(np.int64)
delta32 = ((delta + 2**31) % 2**32 - 2**31).astype(np.int32)           # wrapping difference
np.savez_compressed(DST, delta=delta32, sample_rate=np.int32(48000), bits=np.int32(32),
                    sha256=np.frombuffer(hashlib.sha256(pcm.tobytes()).digest(), np.uint8))
back = np.cumsum(np.load(DST)["delta"].astype(np.int64)).astype(np.int64)
back = ((back + 2**31) % 2**32 - 2**31).astype(np.int32)
assert np.array_equal(back, pcm)
print(DST, os.path.getsize(DST), "bytes")
