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
done = 0
for eng in range(12):
    clf = host.HipClassifier(blob, max_batch=256, debug_no_reuse=True)
    steps = clf.describe()["steps"]
    def fetch(names):
        out = {}
        for s in steps:
            if s["name"] in names or s["kernel"] in names:
                out[s["i"]] = (s["name"], clf.debug_fetch(-s["out_v"] - 2, B, 1 << 19).copy())
        return out
    want = ("frontend", "stft", "normalize", "minmax")
    os.environ["BNHIP_HOST_SERIAL"] = "1"
    ref = clf.predict_pcm16(pcm.reshape(-1), 2048)
    rv = fetch(want)
    del os.environ["BNHIP_HOST_SERIAL"]
    for t in range(6):
        got = clf.predict_pcm16(pcm.reshape(-1), 2048)
        if (np.abs(got[6 * 256:7 * 256] - ref[6 * 256:7 * 256]).max(1) > 0).any():
            gv = fetch(want)
            for i in sorted(gv):
                a, b = gv[i][1], rv[i][1]
                bad = np.argwhere(a != b)
                print(f"engine {eng} run {t}: step {i} {gv[i][0]} shape {a.shape}: {len(bad)} differing elements")
                if len(bad) and gv[i][0].startswith("melband"):
                    # out[b][m][f][c], 96 x 511 x 2
                    for (r, e) in bad[:24]:
                        m, rem = divmod(int(e), 511 * 2); f, c = divmod(rem, 2)
                        hits = np.argwhere(b == a[r, e])
                        desc = []
                        for (r2, e2) in hits[:4]:
                            m2, rem2 = divmod(int(e2), 511 * 2); f2, c2 = divmod(rem2, 2)
                            desc.append(f"clip {r2} (d={int(r2) - int(r)}) mel {m2} frame {f2} ch {c2}")
                        print(f"     clip {r} mel {m} frame {f} ch {c}: got {a[r, e]:.6g} ref {b[r, e]:.6g}; got-value found in ref at: {desc}")
            done += 1
            break
    clf.close()
    if done >= 2:
        break
