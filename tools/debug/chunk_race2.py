import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
from test_parity_gpu import _DevBuf
blob = sm.build_model()
x256 = sm.synth_clips(256, 144000, 48000)
pcm256 = (np.clip(x256, -1, 1) * 32767).astype(np.int16)
xs = np.concatenate([np.roll(x256, 31 * c, axis=0) for c in range(8)], axis=0)
pcm = np.concatenate([np.roll(pcm256, 31 * c, axis=0) for c in range(8)], axis=0)

def report(tag, got):
    nbad = 0
    for c in range(1, 8):
        d = np.abs(got[c * 256:(c + 1) * 256] - np.roll(got[:256], 31 * c, axis=0))
        nbad += int((d.max(1) > 0).sum())
    print(f"{tag}: differing rows {nbad}", flush=True)

for kw, tag in (({}, "host pcm16"), ({"debug_no_reuse": True}, "host pcm16 no_reuse"), ({"autotune": False}, "host pcm16 no autotune"),
                ({"bf16x3": 0}, "host pcm16 bf16x3=0"), ({"lanes": 1}, "host pcm16 lanes=1")):
    clf = host.HipClassifier(blob, max_batch=256, **kw)
    for r in range(3):
        report(tag + f" run {r}", clf.predict_pcm16(pcm.reshape(-1), 2048))
    if not kw:
        for r in range(2):
            report(f"host f32 run {r}", clf.predict_batch(xs.reshape(-1), 2048))
    clf.close()
# device-pointer path, different data per call, engine built like the host path's (lanes 2)
for kw, tag in (({"depth": 2}, "device depth2 lanes2-built"), ({"depth": 2, "lanes": 1}, "device depth2 lanes1")):
    clf = host.HipClassifier(blob, max_batch=256, **kw)
    xd = [_DevBuf(x256.nbytes) for _ in range(8)]
    for c in range(8):
        xd[c].upload(xs[c * 256:(c + 1) * 256])
    o = _DevBuf(2048 * 6522 * 4)
    for r in range(3):
        for c in range(8):
            clf.predict_device(xd[c].at(0), 256, o.at(c * 256 * 6522 * 4))
        clf.synchronize()
        report(tag + f" run {r}", o.download((2048, 6522)))
    clf.close()
