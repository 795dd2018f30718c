"""Runs one depth-1 batch of the full model with the split-bf16 path forced on (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import birdnet_go_amd  # noqa
from birdnet_go_amd import host, synth_model as sm
blob = sm.build_model(sm.SynthConfig())
B = 256
clf = host.HipClassifier(blob, max_batch=B, bf16x3=int(os.environ.get("BX3", "1")), lanes=1)
x = sm.synth_clips(8, 144000, 48000)
x = np.tile(x, (B // 8, 1))
for _ in range(2):
    clf.predict_batch(x.reshape(-1), B)
