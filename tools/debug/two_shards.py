"""{"devices":[0,0],"replicate":"peer"} at 2 x 256 clips through the blocking host-pointer entry (bnhip_predict): one handle, two
engines + worker threads on the one GPU a driver box has - the sharding / replication plumbing an 8-GPU handle runs, on hardware.
Prints one JSON line: ms per call, clips/s, the single-engine rate beside it, and that both produce the same logits."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, synth_model as sm

blob = sm.build_model()
x = sm.synth_clips(512, 144000, 48000, first=7)
res = {}
outs = {}
for name, kw in (("one_engine", {}), ("devices_0_0_peer", {"devices": [0, 0], "replicate": "peer"})):
    clf = host.HipClassifier(blob, max_batch=256, **kw)
    try:
        d = clf.describe()
        outs[name] = clf.predict_batch(x.reshape(-1), 512).copy()
        for _ in range(2):
            clf.predict_batch(x.reshape(-1), 512)
        t0 = time.perf_counter()
        reps = 6
        for _ in range(reps):
            clf.predict_batch(x.reshape(-1), 512)
        dt = (time.perf_counter() - t0) / reps
        res[name] = {"ms_per_512_clip_call": dt * 1e3, "clips_per_s": 512 / dt, "devices": d.get("devices"), "weight_replication": d.get("weight_replication"),
                     "tune_source": d.get("tune_source"), "tune_sources": d.get("tune_sources"), "plans_identical": d.get("plans_identical")}
    finally:
        clf.close()
# (round 6: every engine of one plan adopts one tuning - the recorded one, or the first engine's - and a clip's bits do not depend on
# the call geometry, so the 2 x 256-clip shards and the one-engine 512-clip pipelined call agree bit for bit: the difference is 0.0;
# through round 5 each engine timed its own tile candidates and the two differed by 2.5e-5)
res["max_abs_logit_diff"] = float(np.abs(outs["one_engine"] - outs["devices_0_0_peer"]).max())
res["ratio"] = res["devices_0_0_peer"]["clips_per_s"] / res["one_engine"]["clips_per_s"]
print(json.dumps(res))
