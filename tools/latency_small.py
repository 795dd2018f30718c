"""Per-launch table of a small call (default one clip) through the device entry: what bounds single-clip latency.
    python tools/latency_small.py [n_clips]
"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import birdnet_go_amd  # noqa
from birdnet_go_amd import host, synth_model as sm

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg = sm.SynthConfig()
blob = sm.build_model(cfg)
clf = host.HipClassifier(blob, device=0, max_batch=256)
x = sm.synth_clips(n, cfg.n_samples, cfg.sample_rate)
for _ in range(5):
    clf.predict_batch(x.reshape(-1), n)
ts = []
for _ in range(50):
    t0 = time.perf_counter(); clf.predict_batch(x.reshape(-1), n); ts.append(time.perf_counter() - t0)
print(f"n={n}: host call median {1e3 * np.median(ts):.3f} ms  min {1e3 * min(ts):.3f} ms")
clf.profile_enable(True)
for _ in range(5):
    clf.predict_batch(x.reshape(-1), n)
classes, steps = clf.profile_read(per_step=True)
clf.profile_enable(False)
steps = [r for r in steps if r["launches"] > 0]
tot = sum(r["ms"] / r["launches"] for r in steps)
print(f"launches per call: {len(steps)}   sum of per-launch times: {tot * 1e3:.1f} us")
for r in steps:
    print(f"{r['step']:3d} {r['kernel']:12s} {r['name'][:28]:28s} {1e3 * r['ms'] / r['launches']:8.1f} us")
