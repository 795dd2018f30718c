"""Developer aid (not collected by pytest): layer-by-layer comparison of the HIP engine against the
numpy oracle on a real GPU.  Usage: python tests/gpu_debug.py [tiny|full] [n_clips]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import birdnet_go_amd  # noqa: E402
from birdnet_go_amd import host, synth_model as sm  # noqa: E402
from oracle.interp import Interpreter  # noqa: E402
from oracle.tflite_reader import read_model  # noqa: E402


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    cfg = sm.tiny_config() if which == "tiny" else sm.SynthConfig()
    blob = sm.build_model(cfg)
    m = read_model(blob)
    x = sm.synth_clips(n, cfg.n_samples, cfg.sample_rate)
    keep = {}
    t = time.time()
    ref = Interpreter(m).invoke(x, keep=keep)[0]
    print(f"oracle {time.time() - t:.2f}s")
    clf = host.HipClassifier(blob, max_batch=max(4, n), debug_no_reuse=True)
    t = time.time()
    got = clf.predict_batch(x.reshape(-1), n)
    print(f"hip {time.time() - t:.3f}s")
    d = np.abs(got - ref)
    print("logits max|diff|", d.max(), "ref range", ref.min(), ref.max(), "finite", np.isfinite(got).all())
    sig = lambda v: 1 / (1 + np.exp(-v.astype(np.float64)))
    print("prob max|diff|", np.abs(sig(got) - sig(ref)).max(), "top1 equal", (got.argmax(1) == ref.argmax(1)).all())
    bad = 0
    for op in m.ops:
        ti = op.outputs[0]
        try:
            g = clf.debug_fetch(ti, n, int(np.prod(m.tensors[ti].shape)))
        except host.HipError:
            continue
        r = np.asarray(keep[ti], np.float32).reshape(n, -1)
        if g.shape != r.shape:
            print("shape mismatch", m.tensors[ti].name, g.shape, r.shape)
            continue
        e = np.abs(g - r).max()
        scale = np.abs(r).max() + 1e-30
        flag = "" if e / scale < 1e-3 else "   <<<<<<"
        if flag:
            bad += 1
        print(f"{op.name:18s} {m.tensors[ti].name:26s} {str(m.tensors[ti].shape):20s} max|d| {e:.3e} rel {e / scale:.2e}{flag}")
        if bad > 3:
            break


if __name__ == "__main__":
    main()
