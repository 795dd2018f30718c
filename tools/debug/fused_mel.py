import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
specs = (sm.SpecConfig(512, 94, 0.0, 3000.0), sm.SpecConfig(512, 94, 500.0, 15000.0))
cfg = sm.tiny_config(specs=specs)
blob = sm.build_model(cfg)
x = sm.synth_clips(2, 12000, 48000)
res = {}
for tag in ("fused", "plain"):
    if tag == "plain": os.environ["BNHIP_NO_FUSE_MEL"] = "1"
    c = host.HipClassifier(blob, max_batch=2, autotune=False, debug_no_reuse=True)
    os.environ.pop("BNHIP_NO_FUSE_MEL", None)
    st = c.describe()["steps"]
    print(tag, [(s["name"], s["out_v"]) for s in st[:6]])
    c.predict_batch(x.reshape(-1), 2)
    v = [s for s in st if s["name"].startswith("stft0") or s["name"].startswith("melband")][-1]["out_v"]
    res[tag] = c.debug_fetch(-v - 2, 2, 1 << 16)
    c.close()
a, b = res["fused"], res["plain"]
print(a.shape, b.shape)
d = np.abs(a - b)
bad = np.argwhere(d > 0)
print("differing", len(bad), "max", d.max())
F = 123
for (r, e) in bad[:30]:
    m, rem = divmod(int(e), F * 2); f, c = divmod(rem, 2)
    print(r, "mel", m, "frame", f, "ch", c, a[r, e], b[r, e])
import collections
print(collections.Counter((int(e) // (F * 2)) for r, e in bad).most_common(40))
