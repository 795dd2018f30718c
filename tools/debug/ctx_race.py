"""Localise a cross-context race: run X on context 0 while Y runs on context 1, compare every plan value of context 0
against the same run made alone (debug_no_reuse keeps every value)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
from test_parity_gpu import _DevBuf
B = int(os.environ.get("B", "256"))
blob = sm.build_model()
xh = sm.synth_clips(B, 144000, 48000)
yh = np.roll(xh, 31, axis=0).copy()
clf = host.HipClassifier(blob, max_batch=B, depth=2, lanes=1, debug_no_reuse=True)
steps = clf.describe()["steps"]
x, y, o = _DevBuf(xh.nbytes), _DevBuf(yh.nbytes), _DevBuf(2 * B * 6522 * 4)
x.upload(xh); y.upload(yh)
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
clf.predict_device(x.at(0), B, o.at(0)); clf.synchronize()
ref = fetch_all()
ref_logits = o.download((2, B, 6522))[0].copy()
clf.predict_device(y.at(0), B, o.at(B * 6522 * 4)); clf.synchronize()
bad = 0
for trial in range(int(os.environ.get("TRIALS", "12"))):
    for _ in range(2):
        clf.predict_device(x.at(0), B, o.at(0))
        clf.predict_device(y.at(0), B, o.at(B * 6522 * 4))
    clf.synchronize()
    lg = o.download((2, B, 6522))[0]
    if np.array_equal(lg, ref_logits):
        continue
    bad += 1
    rows = np.nonzero(np.abs(lg - ref_logits).max(1) > 0)[0]
    print(f"trial {trial}: logits differ in rows {rows[:10]} max {np.abs(lg - ref_logits).max():.3e}")
    got = fetch_all()
    for v in sorted(got, key=lambda k: got[k][0]):
        i, kern, name, a = got[v]
        b = ref[v][3]
        if a is None or b is None:
            continue
        if not np.array_equal(a, b):
            d = np.abs(a - b)
            r = np.nonzero(d.max(1) > 0)[0]
            print(f"   first differing value: step {i} {kern} {name} value {v}: rows {r[:10]} max {d.max():.3e} n_bad {int((d > 0).sum())}")
            break
    if bad >= 3:
        break
print("bad trials:", bad)
