"""ORACLE (test infrastructure only).  The resampler's filter arithmetic is NOT restatable from the reference: it lives
in github.com/tphakala/go-audio-resampler v1.7.0 (go.mod:38), which is absent from /root/reference, and the reference's
tests pin only the output length (+-5 %, resample_test.go:26-55,151-168).  PARITY UNPINNED for sample values.  The
project's own filter spec is scipy.signal.resample_poly's default design, so scipy itself is the checker; the PCM16
edges around it restate internal/audiocore/resample/resample.go:120-124,161-169 (via oracle/c/oracle.c)."""
from math import gcd

import numpy as np
from scipy import signal

from . import gofuncs


def resample_f64(x, rate_in, rate_out):
    g = gcd(int(rate_in), int(rate_out))
    return signal.resample_poly(np.asarray(x, np.float64), rate_out // g, rate_in // g, axis=-1)


def resample_pcm16(pcm, rate_in, rate_out):
    """int16 -> int16 through the reference wrapper's edges around the float resampler."""
    x = np.asarray(pcm, np.int16).astype(np.float32) / np.float32(32768.0)
    y = resample_f64(x, rate_in, rate_out).astype(np.float32)
    return gofuncs.resample_edge_out(y)


def expected_length(n_in, rate_in, rate_out):
    g = gcd(int(rate_in), int(rate_out))
    L, M = rate_out // g, rate_in // g
    return -(-n_in * L // M)
