"""Latency of small blocking calls (1, 8, 64 clips) and the 256-clip pinned call; env knobs under test are read by the library."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
blob = sm.build_model()
x = sm.synth_clips(256, 144000, 48000)
clf = host.HipClassifier(blob, max_batch=256)
def med(fn, reps=40):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3
with host.PinnedArray((256, 144000), np.float32) as pi, host.PinnedArray((256, 6522), np.float32) as po:
    pi.array[:] = x
    r = {n: med(lambda: clf.predict_batch(pi.array[:n].reshape(-1), n, out=po.array[:n]), 40 if n < 64 else 15) for n in (1, 8, 32, 64, 256)}
print(os.environ.get("TAG", ""), " ".join(f"n={n}: {v:.3f} ms" for n, v in r.items()), "launches", len(clf.describe()["steps"]), flush=True)
clf.close()
