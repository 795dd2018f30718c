"""Per-chunk timeline (BNHIP_HOST_TRACE=1) of one blocking 256-clip host call: pageable vs pinned caller memory, f32 vs int16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["BNHIP_HOST_TRACE"] = "1"
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
N = int(os.environ.get("N", "256"))
blob = sm.build_model()
x = np.tile(sm.synth_clips(256, 144000, 48000), ((N + 255) // 256, 1))[:N]
clf = host.HipClassifier(blob, max_batch=256)
out = np.zeros((N, 6522), np.float32)
with host.PinnedArray((N, 144000), np.float32) as pi, host.PinnedArray((N, 6522), np.float32) as po:
    pi.array[:] = x
    for name, a, o in (("pageable", x, out), ("pinned", pi.array, po.array)):
        for rep in range(3):
            sys.stderr.flush()
            print(f"==== {name} rep {rep}", file=sys.stderr, flush=True)
            t0 = time.perf_counter(); clf.predict_batch(a.reshape(-1), N, out=o); dt = time.perf_counter() - t0
            print(f"==== {name} rep {rep}: {dt * 1e3:.3f} ms", file=sys.stderr, flush=True)
clf.close()
