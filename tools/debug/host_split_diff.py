"""Two-phase call vs whole-plan chunks on one engine (BNHIP_HOST_NOSPLIT, read per call): where do the logits differ?"""
import os

os.environ.setdefault("BNHIP_HOST_DIAG", "1")      # per-call switches of the host pipeline are read only in a process that sets this
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, synth_model as sm

blob = sm.build_model()
clf = host.HipClassifier(blob, max_batch=256)
x = sm.synth_clips(256)
a = clf.predict_batch(x.reshape(-1), 256).reshape(256, -1).copy()
os.environ["BNHIP_HOST_NOSPLIT"] = "1"
b = clf.predict_batch(x.reshape(-1), 256).reshape(256, -1).copy()
b2 = clf.predict_batch(x.reshape(-1), 256).reshape(256, -1).copy()
os.environ["BNHIP_HOST_CHUNKS"] = "128,128"
c = clf.predict_batch(x.reshape(-1), 256).reshape(256, -1).copy()
os.environ.pop("BNHIP_HOST_CHUNKS")
os.environ.pop("BNHIP_HOST_NOSPLIT")
a2 = clf.predict_batch(x.reshape(-1), 256).reshape(256, -1).copy()
print("split repeat identical", np.array_equal(a, a2), "nosplit repeat identical", np.array_equal(b, b2))
for name, u, v in (("split vs 4x64 chunks", a, b), ("split vs 2x128 chunks", a, c), ("4x64 vs 2x128 chunks", b, c)):
    d = np.abs(u - v)
    rows = np.nonzero(d.max(1) > 0)[0]
    print(name, "max abs", d.max(), "rows differing", len(rows), rows[:10], "cols differing in first such row", int((d[rows[0]] > 0).sum()) if len(rows) else 0)
clf.close()
