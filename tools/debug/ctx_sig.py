"""Failure signature of the cross-context difference: X on context 0 while Y runs on context 1; on a mismatch print WHERE the first
differing plan value differs (clip, and for the spectrogram image mel / frame / channel), the wrong and the right numbers, and
whether the inputs of that step were identical.  Run with BNHIP_LIB=<SLP build> to see the failing build."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
from test_parity_gpu import _DevBuf
B = int(os.environ.get("B", "256"))
blob = sm.build_model()
xh = sm.synth_clips(B, 144000, 48000)
yh = np.roll(xh, 31, axis=0).copy() if os.environ.get("SAME", "0") == "0" else xh.copy()
clf = host.HipClassifier(blob, max_batch=B, depth=2, lanes=1, debug_no_reuse=True)
steps = clf.describe()["steps"]
x, y, o = _DevBuf(xh.nbytes), _DevBuf(yh.nbytes), _DevBuf(2 * B * 6522 * 4)
x.upload(xh); y.upload(yh)
CAP = 1 << 19
def fetch_all():
    vals = {}
    for s in steps:
        for key in ("out_v", "out2_v"):
            v = s[key]
            if v >= 0 and v not in vals:
                try:
                    vals[v] = (s["i"], s["kernel"], s["name"], clf.debug_fetch(-v - 2, B, CAP).copy())
                except Exception:
                    vals[v] = (s["i"], s["kernel"], s["name"], None)
    return vals
clf.predict_device(x.at(0), B, o.at(0)); clf.synchronize()
ref = fetch_all()
ref_logits = o.download((2, B, 6522))[0].copy()
clf.predict_device(y.at(0), B, o.at(B * 6522 * 4)); clf.synchronize()
bad = 0
trials = int(os.environ.get("TRIALS", "16"))
for trial in range(trials):
    for _ in range(2):
        clf.predict_device(x.at(0), B, o.at(0))
        clf.predict_device(y.at(0), B, o.at(B * 6522 * 4))
    clf.synchronize()
    lg = o.download((2, B, 6522))[0]
    if np.array_equal(lg, ref_logits):
        continue
    bad += 1
    if bad > 3:
        continue
    rows = np.nonzero(np.abs(lg - ref_logits).max(1) > 0)[0]
    print(f"trial {trial}: logits differ in rows {rows[:10]} max {np.abs(lg - ref_logits).max():.3e}")
    got = fetch_all()
    n_diff = 0
    for v in sorted(got, key=lambda k: got[k][0]):
        i, kern, name, a = got[v]
        b = ref[v][3]
        if a is None or b is None or np.array_equal(a, b):
            continue
        n_diff += 1
        if n_diff > 2:
            break
        d = a != b
        r = np.nonzero(d.any(1))[0]
        print(f"   differing value: step {i} {kern} {name} value {v}: clips {r[:10]} n_bad {int(d.sum())} max {np.abs(a - b).max():.3e}")
        for c in r[:3]:
            idx = np.nonzero(d[c])[0]
            if name.startswith("melband") and a.shape[1] >= 96 * 511 * 2:
                dec = [(int(k) // (511 * 2), (int(k) // 2) % 511, int(k) & 1) for k in idx[:16]]
                print(f"      clip {c}: {len(idx)} words; (mel, frame, channel) = {dec}")
            else:
                print(f"      clip {c}: {len(idx)} words at {idx[:16]}")
            for k in idx[:6]:
                print(f"         [{k}] got {a[c, k]!r} ({a[c, k].view(np.uint32):#010x}) want {b[c, k]!r} ({b[c, k].view(np.uint32):#010x})")
print(f"bad trials: {bad} of {trials}")
