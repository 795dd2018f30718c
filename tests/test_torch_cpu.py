"""The timed CPU baseline (oracle/torch_cpu.py) computes what the parity oracle computes (CPU only)."""
import numpy as np

from birdnet_go_amd import synth_model as sm
from oracle.interp import Interpreter
from oracle.torch_cpu import TorchCPU


def test_torch_cpu_matches_the_oracle_on_both_front_end_variants():
    for cfg in (sm.tiny_config(), sm.tiny_config(complex_mode="abs", compress="log", normalize=False, time_major=True)):
        blob = sm.build_model(cfg)
        x = sm.synth_clips(3, cfg.n_samples, cfg.sample_rate)
        want = Interpreter(blob).invoke(x)
        got = TorchCPU(blob).invoke(x)
        assert len(want) == len(got)
        for a, b in zip(want, got):
            assert a.shape == b.shape and np.abs(a - b).max() < 1e-4


def test_torch_cpu_full_topology_one_clip(full_blob):
    x = sm.synth_clips(1, 144000, 48000)
    want = Interpreter(full_blob).invoke(x)[0]
    got = TorchCPU(full_blob).invoke(x)[0]
    sig = lambda v: 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
    assert (want.argmax(1) == got.argmax(1)).all() and np.abs(sig(want) - sig(got)).max() < 1e-4
