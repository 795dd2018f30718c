"""Two-phase call of n clips: the default chunks (64-clip units) vs four equal chunks (BNHIP_HOST_CHUNKS, read per call)."""
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


for n in (128, 144, 160, 192, 224, 256):
    res = []
    q = (n + 3) // 4
    for sched in (None, ",".join(str(min(q, n - i * q)) for i in range(4)), ",".join(str(v) for v in ((n + 2) // 3, (n + 2) // 3, n - 2 * ((n + 2) // 3)))):
        if sched:
            os.environ["BNHIP_HOST_CHUNKS"] = sched
        else:
            os.environ.pop("BNHIP_HOST_CHUNKS", None)
        a = t(lambda: clf.predict_batch(x[:n].reshape(-1), n))
        b = t(lambda: clf.predict_pcm16(pcm[:n].reshape(-1), n))
        res.append(f"[{sched or 'default'}] {a:.3f}/{b:.3f}")
    print(f"n={n} f32/pcm16 ms: " + "   ".join(res), flush=True)
clf.close()
