import os
import sys

import pytest

os.environ.setdefault("BNHIP_HOST_DIAG", "1")      # the host pipeline's per-call diagnostic switches (hostpipe.cpp diag_env) exist in test processes only

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import birdnet_go_amd  # noqa: E402,F401  (alias -> ./birdnet-go_amd)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built_lib():
    """libbnhip.so built in-tree (cross-compiles for gfx950 without a GPU)."""
    from birdnet_go_amd import build
    return build.build()


@pytest.fixture(scope="session")
def tiny_cfg():
    from birdnet_go_amd import synth_model as sm
    return sm.tiny_config()


@pytest.fixture(scope="session")
def tiny_blob(tiny_cfg):
    from birdnet_go_amd import synth_model as sm
    return sm.build_model(tiny_cfg)


@pytest.fixture(scope="session")
def full_blob():
    from birdnet_go_amd import synth_model as sm
    return sm.build_model(sm.SynthConfig())


@pytest.fixture(scope="session")
def gpu(built_lib):
    """A usable gfx950 device behind the C ABI (GPU tests fail loudly here when the HIP library or device is missing)."""
    from birdnet_go_amd import host
    assert host.init() >= 1
    return True
