"""Device-resident ultrasonic frame-CV entry (bnhip_us_frame_cv_device; reference: internal/audiocore/ultrasonic/filter.go:20-145):
int16 PCM converted in the kernel (int16 / 32768 as float64, convert/pcm.go:108-113) and float64 input, vs the line-by-line
C restatement of the Go code, on the reference's own test signals and the config-4 chirps."""
import ctypes as C

import numpy as np
import pytest

from birdnet_go_amd import host
from oracle import gofuncs as G

from test_parity_gpu import _DevBuf


@pytest.mark.gpu
def test_us_frame_cv_device_pcm16_and_f64_match_go_restatement(gpu):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    lib = host.load_library()
    lib.bnhip_us_frame_cv_device.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_void_p]
    n, rate = 144000, 256000
    t = np.arange(n) / rate
    tone = 0.01 * np.sin(2 * np.pi * 40000.0 * t)                                  # filter_test.go: steady tone -> CV < 0.15
    burst = np.where((np.arange(n) >= n // 3) & (np.arange(n) < 2 * n // 3), 0.5 * np.sin(2 * np.pi * 45000.0 * t), 0.0)
    pcm = np.stack([np.round(tone * 32767).astype(np.int16), np.round(burst * 32767).astype(np.int16)] + list(bench.bat_chirps(3)))
    B = pcm.shape[0]
    frames = 1 + (n - 8192) // 4096
    want = np.array([G.us_frame_cv(p.astype(np.float64) / 32768.0, rate)[0] for p in pcm])
    for dtype, data in ((1, pcm), (0, pcm.astype(np.float64) / 32768.0)):
        d_in, d_scr, d_cv = _DevBuf(data.nbytes), _DevBuf(B * frames * 8), _DevBuf(B * 8)
        try:
            d_in.upload(data)
            rc = lib.bnhip_us_frame_cv_device(0, d_in.ptr, dtype, B, n, rate, 8192, 4096, 20000, d_scr.ptr, d_cv.ptr, None)
            assert rc == frames, host.load_library().bnhip_last_error()
            got = d_cv.download((B,), np.float64)
        finally:
            d_in.free(); d_scr.free(); d_cv.free()
        assert np.abs(got - want).max() <= 1e-9 * np.abs(want).max(), (dtype, got, want)
    assert want[0] < 0.15 < want[1]                                                 # the reference tests' verdicts
    # guards answer with an error here (the host entry returns (0, false) for them)
    assert lib.bnhip_us_frame_cv_device(0, 1, 1, 1, 100, rate, 8192, 4096, 20000, 1, 1, None) == host.E_INVALID
