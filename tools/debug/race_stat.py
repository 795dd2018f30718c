"""Count rows whose logits differ between the serial and the overlapped execution of the same 2048-clip host call."""
import os, sys
os.environ.setdefault("BNHIP_HOST_DIAG", "1")      # per-call switches of the host pipeline are read only in a process that sets this
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
blob = sm.build_model()
x256 = sm.synth_clips(256, 144000, 48000)
pcm256 = (np.clip(x256, -1, 1) * 32767).astype(np.int16)
pcm = np.concatenate([np.roll(pcm256, 31 * c, axis=0) for c in range(8)], axis=0)
tot = runs = 0
for eng in range(int(os.environ.get("ENGINES", "5"))):
    clf = host.HipClassifier(blob, max_batch=256)
    os.environ["BNHIP_HOST_SERIAL"] = "1"
    ref = clf.predict_pcm16(pcm.reshape(-1), 2048)
    del os.environ["BNHIP_HOST_SERIAL"]
    for _ in range(8):
        got = clf.predict_pcm16(pcm.reshape(-1), 2048)
        tot += int((np.abs(got - ref).max(1) > 0).sum()); runs += 1
    clf.close()
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: {tot} differing rows over {runs} overlapped 2048-clip calls", flush=True)
