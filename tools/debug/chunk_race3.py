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
pcm = np.concatenate([np.roll(pcm256, 31 * c, axis=0) for c in range(8)], axis=0)
xq = (pcm.astype(np.float32) / np.float32(32768.0))
pcm_same = np.tile(pcm256, (8, 1))

def report(tag, got, roll=31):
    nbad = 0
    for c in range(1, 8):
        d = np.abs(got[c * 256:(c + 1) * 256] - np.roll(got[:256], roll * c, axis=0))
        nbad += int((d.max(1) > 0).sum())
    print(f"{tag}: differing rows {nbad}", flush=True)

clf = host.HipClassifier(blob, max_batch=256)
for r in range(3):
    report(f"host pcm16 run {r}", clf.predict_pcm16(pcm.reshape(-1), 2048))
for r in range(3):
    report(f"host f32 quantised data run {r}", clf.predict_batch(xq.reshape(-1), 2048))
for r in range(3):
    report(f"host pcm16 identical chunks run {r}", clf.predict_pcm16(pcm_same.reshape(-1), 2048), roll=0)
pcm32 = (pcm.astype(np.int32) << 16)
for r in range(2):
    report(f"host pcm32 run {r}", clf.predict_pcm(pcm32.tobytes(), 32, 2048))
clf.close()
clf = host.HipClassifier(blob, max_batch=256, depth=2)
xd = [_DevBuf(x256.nbytes) for _ in range(8)]
for c in range(8):
    xd[c].upload(xq[c * 256:(c + 1) * 256])
o = _DevBuf(2048 * 6522 * 4)
for r in range(3):
    for c in range(8):
        clf.predict_device(xd[c].at(0), 256, o.at(c * 256 * 6522 * 4))
    clf.synchronize()
    report(f"device depth2 quantised data run {r}", o.download((2048, 6522)))
clf.close()
