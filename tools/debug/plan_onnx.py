import sys
import birdnet_go_amd
from birdnet_go_amd import host, synth_model as sm
cfg = {"tiny": sm.tiny_config, "abs": lambda: sm.tiny_config(complex_mode="abs"), "perch": sm.tiny_perch_config,
       "full": sm.SynthConfig, "fullperch": sm.perch_config}[sys.argv[2]]()
blob = sm.build_model(cfg, container="onnx", dft=sys.argv[1])
clf = host.HipClassifier(blob, plan_only=True)
d = clf.describe()
print(len(d["steps"]), sorted(set(s["kernel"] for s in d["steps"])))
