"""Blocking 256-clip host call (pinned caller memory) under different chunk units: BNHIP_HOST_RAMP=<unit>."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
blob = sm.build_model()
x = sm.synth_clips(256, 144000, 48000)
pcm = (np.clip(x, -1, 1) * 32767).astype(np.int16)
clf = host.HipClassifier(blob, max_batch=256)
def med(fn, reps=15):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3
with host.PinnedArray((256, 144000), np.float32) as pi, host.PinnedArray((256, 144000), np.int16) as pp, host.PinnedArray((256, 6522), np.float32) as po:
    pi.array[:] = x; pp.array[:] = pcm
    a = med(lambda: clf.predict_batch(pi.array.reshape(-1), 256, out=po.array))
    b = med(lambda: clf.predict_pcm16(pp.array.reshape(-1), 256, out=po.array))
    print(f"unit {os.environ.get('BNHIP_HOST_RAMP', 'default')}: f32_256 {a:.3f} ms = {256 / a:.1f} k clips/s, pcm16_256 {b:.3f} ms = {256 / b:.1f} k clips/s", flush=True)
clf.close()
