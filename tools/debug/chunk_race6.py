import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
blob = sm.build_model()
x256 = sm.synth_clips(256, 144000, 48000)
pcm256 = (np.clip(x256, -1, 1) * 32767).astype(np.int16)
pcm = np.concatenate([np.roll(pcm256, 31 * c, axis=0) for c in range(8)], axis=0)
clf = host.HipClassifier(blob, max_batch=256)
os.environ["BNHIP_HOST_SERIAL"] = "1"
ref = clf.predict_pcm16(pcm.reshape(-1), 2048)
del os.environ["BNHIP_HOST_SERIAL"]
def rows(got): return int((np.abs(got - ref).max(1) > 0).sum())
print("overlapped:", [rows(clf.predict_pcm16(pcm.reshape(-1), 2048)) for _ in range(8)], flush=True)
clf.profile_enable(True)
print("overlapped + event pair around every launch:", [rows(clf.predict_pcm16(pcm.reshape(-1), 2048)) for _ in range(8)], flush=True)
clf.profile_read()
clf.profile_enable(False)
print("overlapped again:", [rows(clf.predict_pcm16(pcm.reshape(-1), 2048)) for _ in range(8)], flush=True)
clf.close()
