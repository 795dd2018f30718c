"""C-ABI surface, CPU only: the library loads, exports every symbol include/bnhip.h declares, and
fails loudly (never falls back) when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from birdnet_go_amd import host

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "bnhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(bnhip_[a-z0-9_]+)\s*\(", src)))


def test_header_and_library_agree(built_lib):
    lib = C.CDLL(built_lib)
    fns = header_functions()
    assert len(fns) >= 18
    for f in fns:
        assert hasattr(lib, f), f"libbnhip.so does not export {f}"
    assert sorted(host.SYMBOLS) == fns, "host.py SYMBOLS out of sync with include/bnhip.h"


def test_no_torch_or_cxx_types_in_header():
    src = open(os.path.join(ROOT, "include", "bnhip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)     # declarations only
    assert "std::" not in src and "torch" not in src and "at::" not in src and "hip" not in src.replace("bnhip", "").replace("hip_stream", "")


def test_version_and_error_strings(built_lib):
    lib = host.load_library()
    assert b"gfx950" in lib.bnhip_version()
    assert isinstance(lib.bnhip_last_error(), bytes)


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="GPU present")
def test_no_gpu_is_a_loud_sentinel_error(built_lib, tiny_blob):
    with pytest.raises(host.ErrHIPUnavailable) as e:
        host.HipClassifier(tiny_blob)
    assert e.value.code == host.E_NO_DEVICE
    with pytest.raises(host.ErrHIPUnavailable):
        host.init()
    with pytest.raises(host.HipError):
        host.us_frame_cv(np.zeros((1, 144000)), 256000)


def test_invalid_arguments(built_lib):
    lib = host.load_library()
    out = C.c_void_p()
    assert lib.bnhip_model_create(None, 0, None, C.byref(out)) == host.E_INVALID
    assert lib.bnhip_model_info(None, None, None, None) == host.E_INVALID
    assert lib.bnhip_predict(None, None, 1, None, None) == host.E_INVALID
    lib.bnhip_model_destroy(None)   # idempotent / NULL-safe like Close()


def test_us_guards_need_no_gpu(built_lib):
    # the reference's guard clauses (filter.go:21-37) answer before any device work
    cv, ok = host.us_frame_cv(np.zeros((2, 100)), 256000)
    assert not ok.any() and (cv == 0).all()
    cv, ok = host.us_frame_cv(np.zeros((1, 20000)), 48000, split_hz=30000)
    assert not ok.any()
    cv, ok = host.us_frame_cv(np.zeros((1, 20000)), 256000, fft_size=6000)
    assert not ok.any()
    cv, ok = host.us_frame_cv(np.zeros((1, 8192)), 256000)
    assert not ok.any()
