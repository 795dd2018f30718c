import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
blob = sm.build_model()
x256 = sm.synth_clips(256, 144000, 48000)
pcm256 = (np.clip(x256, -1, 1) * 32767).astype(np.int16)
pcm = np.concatenate([np.roll(pcm256, 31 * c, axis=0) for c in range(8)], axis=0)
def count(got):
    nbad = 0
    for c in range(1, 8):
        d = np.abs(got[c * 256:(c + 1) * 256] - np.roll(got[:256], 31 * c, axis=0))
        nbad += int((d.max(1) > 0).sum())
    return nbad
recs = []
for t in range(int(os.environ.get("N", "24"))):
    clf = host.HipClassifier(blob, max_batch=256)
    nb = sum(count(clf.predict_pcm16(pcm.reshape(-1), 2048)) for _ in range(4))
    sig = {s["name"]: (s["nt_full"], s["wm_full"], s["bx"], s["shape"], s["dw_lds"]) for s in clf.describe()["steps"]}
    recs.append((nb, sig))
    print(t, nb, flush=True)
    clf.close()
bad = [r for r in recs if r[0] > 0]; good = [r for r in recs if r[0] == 0]
print("bad", len(bad), "good", len(good))
if bad and good:
    names = list(bad[0][1].keys())
    for n in names:
        bv = {r[1][n] for r in bad}; gv = {r[1][n] for r in good}
        only_bad = bv - gv
        common = set.intersection(*[{r[1][n]} for r in bad])
        if only_bad or (common and not (common & gv)):
            print(n, "bad-only choices", only_bad, "| common to all bad", common, "| good choices", gv)
json.dump([(r[0], {k: list(v) for k, v in r[1].items()}) for r in recs], open("gpurun_out/race4.json", "w"))
