"""What would a plan tuned for the host pipeline's chunk size buy?  A 256-clip blocking call through an engine whose max_batch
(= the batch its tiles were tuned at) is 256 / 128 / 64, cut into the same 64-clip chunks (BNHIP_HOST_RAMP=64).
python tools/debug/chunk_plan.py <max_batch>"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, synth_model as sm

mb = int(sys.argv[1])
blob = sm.build_model()
clf = host.HipClassifier(blob, max_batch=mb)
x = sm.synth_clips(256)
pcm = (np.clip(x, -1, 1) * 32767).astype(np.int16)
ncls = clf.num_species()


def t(fn, reps=20):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


with host.PinnedArray((256, x.shape[1]), np.float32) as pi, host.PinnedArray((256, x.shape[1]), np.int16) as pp, \
        host.PinnedArray((256, ncls), np.float32) as po:
    pi.array[:] = x; pp.array[:] = pcm
    a = t(lambda: clf.predict_batch(pi.array.reshape(-1), 256, out=po.array))
    b = t(lambda: clf.predict_pcm16(pp.array.reshape(-1), 256, out=po.array))
    print(f"max_batch {mb} ramp {os.environ.get('BNHIP_HOST_RAMP')}: f32_256_pinned {a:.3f} ms  pcm16_256_pinned {b:.3f} ms", flush=True)
clf.close()
