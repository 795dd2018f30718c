import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
blob = sm.build_model()
x256 = sm.synth_clips(256, 144000, 48000)
pcm256 = (np.clip(x256, -1, 1) * 32767).astype(np.int16)
pcm = np.concatenate([np.roll(pcm256, 31 * c, axis=0) for c in range(8)], axis=0)
B = 256
for eng in range(4):
    clf = host.HipClassifier(blob, max_batch=256, debug_no_reuse=True)
    steps = clf.describe()["steps"]
    def fetch_all():
        vals = {}
        for s in steps:
            for key in ("out_v", "out2_v"):
                v = s[key]
                if v >= 0 and v not in vals:
                    try:
                        vals[v] = (s["i"], s["kernel"], s["name"], clf.debug_fetch(-v - 2, B, 1 << 19).copy())
                    except Exception as ex:
                        vals[v] = (s["i"], s["kernel"], s["name"], None)
        return vals
    os.environ["BNHIP_HOST_SERIAL"] = "1"
    ref = clf.predict_pcm16(pcm.reshape(-1), 2048)
    ref2 = clf.predict_pcm16(pcm.reshape(-1), 2048)
    refvals = fetch_all()
    print(f"engine {eng}: serial deterministic {np.array_equal(ref, ref2)}; serial chunks == rolled chunk 0:",
          all(np.array_equal(ref[c * 256:(c + 1) * 256], np.roll(ref[:256], 31 * c, axis=0)) for c in range(1, 8)), flush=True)
    del os.environ["BNHIP_HOST_SERIAL"]
    found = False
    for t in range(6):
        got = clf.predict_pcm16(pcm.reshape(-1), 2048)
        per = [int((np.abs(got[c * 256:(c + 1) * 256] - ref[c * 256:(c + 1) * 256]).max(1) > 0).sum()) for c in range(8)]
        print(f"  overlapped run {t}: differing rows per chunk {per} max {np.abs(got - ref).max():.2e}", flush=True)
        if per[6] and not found:
            found = True
            gv = fetch_all()
            for v in sorted(gv, key=lambda k: gv[k][0]):
                i, kern, name, a = gv[v]
                b = refvals[v][3]
                if a is None or b is None:
                    continue
                if not np.array_equal(a, b):
                    d = np.abs(a - b)
                    r = np.nonzero(d.max(1) > 0)[0]
                    print(f"   differing value: step {i} {kern} {name} value {v}: rows {r[:8]} max {d.max():.3e} n_bad {int((d > 0).sum())} of {d.size}", flush=True)
    clf.close()
