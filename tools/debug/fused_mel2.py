import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
from oracle.interp import Interpreter
specs = (sm.SpecConfig(512, 94, 0.0, 3000.0), sm.SpecConfig(512, 94, 500.0, 15000.0))
cfg = sm.tiny_config(specs=specs)
blob = sm.build_model(cfg)
x = sm.synth_clips(4, 12000, 48000)
ref = Interpreter(blob).invoke(x)[0]
for zero in (False, True):
    xx = x.copy()
    if zero: xx[2] = 0.0
    ref = Interpreter(blob).invoke(xx)[0]
    for tag in ("fused", "plain"):
        for nr in (False, True):
            if tag == "plain": os.environ["BNHIP_NO_FUSE_MEL"] = "1"
            c = host.HipClassifier(blob, max_batch=8, autotune=False, debug_no_reuse=nr)
            os.environ.pop("BNHIP_NO_FUSE_MEL", None)
            got = c.predict_batch(xx.reshape(-1), 4)
            print("zero clip", zero, tag, "no_reuse", nr, "max diff per clip vs oracle", np.abs(got - ref).max(1))
            c.close()
