import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
blob = sm.build_model()
x = sm.synth_clips(256)
pcm = np.tile((np.clip(x, -1, 1) * 32767).astype(np.int16), (4, 1))       # 1024 clips
out = np.zeros((1024, 6522), np.float32)
def rate(clf, n):
    for _ in range(2): clf.predict_pcm16(pcm[:n].reshape(-1), n, out=out[:n])
    ts = []
    for _ in range(6):
        t0 = time.perf_counter(); clf.predict_pcm16(pcm[:n].reshape(-1), n, out=out[:n]); ts.append(time.perf_counter() - t0)
    return n / sorted(ts)[len(ts) // 2]
one = host.HipClassifier(blob, max_batch=256)
r1 = {n: rate(one, n) for n in (512, 1024)}
ref = out[:512].copy()
one.close()
two = host.HipClassifier(blob, max_batch=256, devices=[0, 0], replicate="peer")
r2 = {n: rate(two, n) for n in (512, 1024)}
print("single engine", r1, "two engines on one GPU", r2, "ratio", {n: r2[n] / r1[n] for n in r1})
print("max diff", np.abs(out[:512] - ref).max())
two.close()
