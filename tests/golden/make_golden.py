"""Regenerates the committed golden fixtures (run from the repo root: python tests/golden/make_golden.py).

There is no runnable reference (no Go, no TFLite runtime, no real weights: SURVEY.md section 0), so
these vectors are produced by the oracle restatement itself on the deterministic synthetic models;
they pin oracle, model writer and HIP engine against drift between rounds.  PARITY UNPINNED vs
real TFLite (stated in DESIGN.md)."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import birdnet_go_amd  # noqa: E402,F401
from birdnet_go_amd import synth_model as sm  # noqa: E402
from oracle.interp import Interpreter  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    cfg = sm.tiny_config()
    blob = sm.build_model(cfg)
    x = sm.synth_clips(3, cfg.n_samples, cfg.sample_rate)
    l32 = Interpreter(blob, "f32").invoke(x)[0]
    l64 = Interpreter(blob, "f64").invoke(x)[0]
    np.savez_compressed(os.path.join(HERE, "tiny_logits.npz"), logits_f32=l32, logits_f64=l64.astype(np.float64),
                        sha256=np.frombuffer(hashlib.sha256(blob).digest(), np.uint8))
    # full-size model: logits of 2 clips, top-32 classes only (keeps the fixture small)
    full = sm.build_model(sm.SynthConfig())
    xf = sm.synth_clips(2, 144000, 48000)
    lf = Interpreter(full, "f32").invoke(xf)[0]
    idx = np.argsort(-lf, axis=1)[:, :32]
    np.savez_compressed(os.path.join(HERE, "full_top32.npz"), idx=idx.astype(np.int32),
                        logits=np.take_along_axis(lf, idx, 1),
                        sha256=np.frombuffer(hashlib.sha256(full).digest(), np.uint8))
    print("wrote fixtures; tiny max|f32-f64| =", np.abs(l32 - l64).max())


if __name__ == "__main__":
    main()
