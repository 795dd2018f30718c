import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
blob = sm.build_model()
x256 = sm.synth_clips(256, 144000, 48000)
pcm256 = (np.clip(x256, -1, 1) * 32767).astype(np.int16)
pcm = np.concatenate([np.roll(pcm256, 31 * c, axis=0) for c in range(8)], axis=0)
for hd in (2, 1):
    clf = host.HipClassifier(blob, max_batch=256, host_depth=hd)
    got = clf.predict_pcm16(pcm.reshape(-1), 2048)
    base = clf.predict_pcm16(pcm256.reshape(-1), 256)     # 2 chunks of 128
    print("host_depth", hd, "chunk0 vs 256-call maxdiff", np.abs(base - got[:256]).max())
    for c in range(1, 8):
        d = np.abs(got[c * 256:(c + 1) * 256] - np.roll(got[:256], 31 * c, axis=0))
        rows = np.nonzero(d.max(1) > 0)[0]
        print(f"  chunk {c}: differing rows {rows.size} max {d.max():.3e} rows {rows[:12]}")
    again = clf.predict_pcm16(pcm.reshape(-1), 2048)
    print("  repeat identical:", np.array_equal(again, got), np.abs(again - got).max())
    clf.close()
