"""Digests of the logits of calls of 1 / 8 / 64 / 256 clips on an engine built WITHOUT the create-time tuner (so the kernels and tiles are the
same in every process): run once per library (BNHIP_LIB=<path>) to show that a kernel change kept every bit.  Also times one clip."""
import hashlib
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, synth_model as sm

x = sm.synth_clips(256)
for kw in ({}, {"precision": "bf16"}):
    clf = host.HipClassifier(sm.build_model(), max_batch=256, autotune=False, **kw)
    d = [hashlib.sha256(clf.predict_batch(x[:n].reshape(-1), n).tobytes()).hexdigest()[:10] for n in (1, 8, 64, 256)]
    clf.close()
    print(os.environ.get("BNHIP_LIB", "lib"), kw, d, flush=True)
clf = host.HipClassifier(sm.build_model(), max_batch=256)
for n in (1, 8):
    f = lambda: clf.predict_batch(x[:n].reshape(-1), n)
    f(); f()
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    ts.sort()
    print(f"n={n}: median {ts[len(ts) // 2] * 1e3:.4f} ms  min {ts[0] * 1e3:.4f} ms", flush=True)
clf.close()
