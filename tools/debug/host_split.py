"""Two-phase blocking call (hostpipe.cpp host_run_split) at different cuts of the plan: BNHIP_HOST_SPLIT=<step> (-1: whole-plan chunks;
read at create time), under different schedules (BNHIP_HOST_PLAN / BNHIP_HOST_CHUNKS, read per call).  Prints the 256-clip rates from
pinned and pageable caller memory and a digest of the logits (every cut and schedule must give the same bits)."""
import hashlib
import os

os.environ.setdefault("BNHIP_HOST_DIAG", "1")      # per-call switches of the host pipeline are read only in a process that sets this
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, synth_model as sm

blob = sm.build_model()
clf = host.HipClassifier(blob, max_batch=256)
x = sm.synth_clips(256)
pcm = (np.clip(x, -1, 1) * 32767).astype(np.int16)
ncls = clf.num_species()
lib = host.load_library()


def t(fn, reps=16):
    fn(); fn()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort()
    return ts[len(ts) // 2] * 1e3


PLANS = [p for p in os.environ.get("PLANS", "default").split(";")]
CHUNKS = [p for p in os.environ.get("CHUNKS", "default").split(";")]
out = np.zeros((256, ncls), np.float32)
with host.PinnedArray((256, x.shape[1]), np.float32) as pi, host.PinnedArray((256, x.shape[1]), np.int16) as pp, \
        host.PinnedArray((256, ncls), np.float32) as po:
    pi.array[:] = x; pp.array[:] = pcm
    for ch in CHUNKS:
        for plan in PLANS:
            for k, v in (("BNHIP_HOST_PLAN", plan), ("BNHIP_HOST_CHUNKS", ch)):
                if v == "default":
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            before = lib.bnhip_debug_split_calls()
            dig = hashlib.sha256(clf.predict_batch(x.reshape(-1), 256, out=out).tobytes()).hexdigest()[:12]
            dig_p = hashlib.sha256(clf.predict_pcm16(pp.array.reshape(-1), 256, out=po.array).tobytes()).hexdigest()[:12]
            dig200 = hashlib.sha256(clf.predict_batch(x[:200].reshape(-1), 200).tobytes()).hexdigest()[:12]
            c = t(lambda: clf.predict_batch(x.reshape(-1), 256, out=out))
            d = t(lambda: clf.predict_pcm16(pcm.reshape(-1), 256, out=out))
            a = t(lambda: clf.predict_batch(pi.array.reshape(-1), 256, out=po.array))
            b = t(lambda: clf.predict_pcm16(pp.array.reshape(-1), 256, out=po.array))
            print(f"split {os.environ.get('BNHIP_HOST_SPLIT', 'default')} plan [{plan}] chunks [{ch}]: f32_256_pinned {a:.3f} ms ({256 / a:.1f} k)  pcm16_256_pinned {b:.3f} ms ({256 / b:.1f} k)  "
                  f"f32_256 {c:.3f}  pcm16_256 {d:.3f}  digests f32 {dig} pcm {dig_p} 200 clips {dig200}  split calls +{lib.bnhip_debug_split_calls() - before}", flush=True)
clf.close()
