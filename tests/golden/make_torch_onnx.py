"""Parity fixtures from an independent writer and executor (run from the repo root in the build container:
    python tests/golden/make_torch_onnx.py

For every model in `torch_models.MODELS` (BirdNET-v2.4-shaped with the DFT written as `torch.stft` and as a strided Conv1d, a
Perch-shaped NCHW model with four outputs, a bat head - tiny and full size):
  * `torch.onnx.export` writes the ONNX file (the small ones are committed under tests/golden/torch_onnx/, the large ones are
    re-exported by the tests; their sha256 is recorded so a different torch build is noticed);
  * torch-CPU computes the outputs for the fixture clips (SURVEY section 8d config-2 tones, silence, the reference's
    tawnyowl.wav windows; Perch: the reference benchmark's U[-1, 1] noise) in float32 AND float64;
  * everything lands in tests/golden/torch_onnx/fixtures.npz.
The tests then hold oracle/onnx_interp.py (CPU) and the HIP engine (GPU) to these numbers: top-1 identical,
|sigmoid diff| <= 1e-4, logits < 1e-3 (cmd/perch-benchmark/main.go:455-462 "EQUIVALENT")."""
import hashlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import torch_models as tm  # noqa: E402

OUT = os.path.join(HERE, "torch_onnx")


def main():
    os.makedirs(OUT, exist_ok=True)
    fx = {"torch_version": np.frombuffer(torch.__version__.encode(), np.uint8)}
    for name, (family, _, ins, outs, li) in tm.MODELS.items():
        model, blob, n = tm.build(name)
        x = tm.fixture_inputs(family, n, HERE)
        o32 = tm.forward_np(model, x, torch.float32)
        o64 = tm.forward_np(model, x.astype(np.float64), torch.float64)
        fx[name + "/sha256"] = np.frombuffer(hashlib.sha256(blob).digest(), np.uint8)
        fx[name + "/onnx_bytes"] = np.int64(len(blob))
        full = name.endswith(("_full", "full_stft", "full_conv"))
        for k, (a, b) in enumerate(zip(o32, o64)):
            if full and k != li:
                continue                                   # full size: the logits only (the other outputs are covered at tiny size)
            fx[f"{name}/f32_{k}"] = a.astype(np.float32)
            fx[f"{name}/f64_{k}"] = b.astype(np.float64) if not full else b.astype(np.float32)    # (full size: fp64 result rounded once, to keep the file small)
        if name in tm.COMMITTED_ONNX:
            with open(os.path.join(OUT, name + ".onnx"), "wb") as f:
                f.write(blob)
        d = np.abs(o32[li] - o64[li]).max()
        print(f"{name}: onnx {len(blob)} bytes, {x.shape[0]} clips, logits std {o32[li].std():.3f}, max |f32 - f64| {d:.2e}, top-1 {o32[li].argmax(1).tolist()}")
    np.savez_compressed(os.path.join(OUT, "fixtures.npz"), **fx)
    print("wrote", os.path.join(OUT, "fixtures.npz"), os.path.getsize(os.path.join(OUT, "fixtures.npz")), "bytes")


if __name__ == "__main__":
    main()
