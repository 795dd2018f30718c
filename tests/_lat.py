import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
blob = sm.build_model()
x = sm.synth_clips(8)
clf = host.HipClassifier(blob, max_batch=256)
for n in (1,):
    for it in range(3): clf.predict_batch(x[:n].reshape(-1), n)
    clf.profile_enable(True)
    for it in range(20): clf.predict_batch(x[:n].reshape(-1), n)
    classes, steps = clf.profile_read(per_step=True)
    clf.profile_enable(False)
    tot = 0
    for r in steps:
        print(r)
    for r in classes: print(r)
