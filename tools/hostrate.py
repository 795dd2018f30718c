import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
blob = sm.build_model()
clf = host.HipClassifier(blob, max_batch=256)
x = sm.synth_clips(256)
def t(fn, reps):
    fn(); fn()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps
for n, reps in ((1, 300), (8, 200), (256, 20)):
    dt = t(lambda: clf.predict_batch(x[:n].reshape(-1), n), reps)
    print(f"fp32 n={n}: {dt*1e3:.3f} ms  {n/dt:.0f} clips/s")
pcm = (np.clip(x, -1, 1) * 32767).astype(np.int16)
dt = t(lambda: clf.predict_pcm16(pcm.reshape(-1), 256), 20)
print(f"pcm16 n=256: {dt*1e3:.3f} ms  {256/dt:.0f} clips/s")
bigp = np.tile(pcm, (8, 1))
dt = t(lambda: clf.predict_pcm16(bigp.reshape(-1), 2048), 5)
print(f"pcm16 n=2048 (8 chunks): {dt*1e3:.3f} ms  {2048/dt:.0f} clips/s")
big = np.tile(x, (8, 1))
dt = t(lambda: clf.predict_batch(big.reshape(-1), 2048), 5)
print(f"fp32 n=2048 (8 chunks): {dt*1e3:.3f} ms  {2048/dt:.0f} clips/s")
dt = t(lambda: clf.predict_topk(x[:1].reshape(-1), 1, 10, 0, 1.0), 300)
print(f"predict_topk n=1: {dt*1e3:.3f} ms")
clf.close()
