"""Explicit chunk schedules for the 256-clip blocking call (BNHIP_HOST_CHUNKS, read per call): one engine, every schedule
timed in turn, twice round (same box, interleaved).  python tools/debug/host_chunks.py"""
import os

os.environ.setdefault("BNHIP_HOST_DIAG", "1")      # per-call switches of the host pipeline are read only in a process that sets this
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, synth_model as sm

SCHEDS = ["", "64,64,64,64", "32,64,96,64", "32,96,128", "48,80,128", "32,64,64,64,32", "32,96,96,32", "16,48,96,96", "64,128,64", "128,128"]


def med(fn, reps=15):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


if __name__ == "__main__":
    clf = host.HipClassifier(sm.build_model(), max_batch=256)
    x = sm.synth_clips(256)
    pcm = (np.clip(x, -1, 1) * 32767).astype(np.int16)
    ncls = clf.num_species()
    out = np.zeros((256, ncls), np.float32)
    with host.PinnedArray((256, x.shape[1]), np.float32) as pi, host.PinnedArray((256, x.shape[1]), np.int16) as pp, \
            host.PinnedArray((256, ncls), np.float32) as po:
        pi.array[:] = x; pp.array[:] = pcm
        base = clf.predict_batch(x.reshape(-1), 256).copy()
        for rnd in range(2):
            for sc in SCHEDS:
                if sc:
                    os.environ["BNHIP_HOST_CHUNKS"] = sc
                else:
                    os.environ.pop("BNHIP_HOST_CHUNKS", None)
                same = np.array_equal(clf.predict_batch(x.reshape(-1), 256, out=out), base)
                a = med(lambda: clf.predict_batch(x.reshape(-1), 256, out=out))
                b = med(lambda: clf.predict_pcm16(pcm.reshape(-1), 256, out=out))
                c = med(lambda: clf.predict_batch(pi.array.reshape(-1), 256, out=po.array))
                d = med(lambda: clf.predict_pcm16(pp.array.reshape(-1), 256, out=po.array))
                print(f"{sc or 'default':16s} f32 {a:6.3f}  pcm16 {b:6.3f}  f32 pinned {c:6.3f}  pcm16 pinned {d:6.3f} ms  same={same}", flush=True)
    clf.close()
