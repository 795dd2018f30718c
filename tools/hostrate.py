"""PCIe-inclusive rates of the blocking host-pointer entries (what a cgo caller gets): pageable numpy memory in,
logits back in caller memory on return.  `python tools/hostrate.py [--json out.json]` (the same leg bench.py attaches
as `host_pointer`)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, synth_model as sm
from bench import host_pointer_rates

if __name__ == "__main__":
    blob = sm.build_model()
    clf = host.HipClassifier(blob, max_batch=256)
    x = sm.synth_clips(256)
    res = host_pointer_rates(clf, x, reps_small=200, reps_mid=20, reps_big=8)
    for k, v in res.items():
        if isinstance(v, dict):
            rate = v.get("clips_per_s", v.get("windows_per_s", 0.0))
            extra = "  ".join(f"{kk} {vv:.3f}" for kk, vv in v.items() if kk.endswith("_ms"))
            print(f"{k:18s} {v['ms']:9.3f} ms  {rate:9.0f} /s  {extra}")
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as fh:
            json.dump(res, fh, indent=1)
    clf.close()
