import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
from test_parity_gpu import _DevBuf
blob = sm.build_model()
B = 256
xs = [sm.synth_clips(B, 144000, 48000, first=k * B) for k in range(4)]
for kw in ({"depth": 2, "lanes": 1}, {"depth": 2}, {"depth": 2, "lanes": 1, "host_depth": 1}):
    clf = host.HipClassifier(blob, max_batch=B, **kw)
    xd = [_DevBuf(x.nbytes) for x in xs]
    for d, x in zip(xd, xs): d.upload(x)
    o = [_DevBuf(B * 6522 * 4) for _ in range(4)]
    for i in range(6): clf.predict_device(xd[i % 4].at(0), B, o[i % 4].at(0))
    clf.synchronize()
    t0 = time.perf_counter()
    N = 40
    for i in range(N): clf.predict_device(xd[i % 4].at(0), B, o[i % 4].at(0))
    clf.synchronize()
    dt = time.perf_counter() - t0
    print(kw, f"{dt / N * 1e3:.3f} ms/step {B * N / dt:.0f} clips/s", flush=True)
    clf.close()
    for d in xd + o: d.free()
