"""Blocking calls of 16 ... 192 clips (pageable fp32 / int16): serial path vs the pipelined paths (BNHIP_HOST_PIPE_MIN, BNHIP_HOST_RAMP: read once)."""
import os

os.environ.setdefault("BNHIP_HOST_DIAG", "1")      # per-call switches of the host pipeline are read only in a process that sets this
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, synth_model as sm

clf = host.HipClassifier(sm.build_model(), max_batch=256)
x = sm.synth_clips(256)
pcm = (np.clip(x, -1, 1) * 32767).astype(np.int16)


def t(fn, reps=30):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


row = []
for n in (16, 32, 48, 64, 96, 128, 160, 192):
    a = t(lambda: clf.predict_batch(x[:n].reshape(-1), n))
    b = t(lambda: clf.predict_pcm16(pcm[:n].reshape(-1), n))
    row.append(f"{n}: {a:.3f}/{b:.3f}")
print(f"pipe_min {os.environ.get('BNHIP_HOST_PIPE_MIN', '128')} ramp {os.environ.get('BNHIP_HOST_RAMP', 'default')} nosplit {os.environ.get('BNHIP_HOST_NOSPLIT', '0')}  f32/pcm16 ms  " + "  ".join(row), flush=True)
clf.close()
