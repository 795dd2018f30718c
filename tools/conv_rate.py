"""Throughput of the implicit-GEMM convolution on a ResNet-style layer stack (not part of the v2.4 path; a sanity number for
graphs with real convolutions): python tools/conv_rate.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host
import test_conv_igemm as T

case = (56, 56, 64, [(64, 3, 1, 1, "SAME", 1), (64, 3, 1, 1, "SAME", 1), (128, 3, 2, 1, "SAME", 1), (128, 3, 1, 1, "SAME", 1)])
blob = T.build(case, 1)
B = 128
clf = host.HipClassifier(blob, max_batch=B)
x = np.random.default_rng(0).standard_normal((B, 56, 56, 64)).astype(np.float32)
from test_parity_gpu import _DevBuf
dx, dl = _DevBuf(x.nbytes), _DevBuf(B * 9 * 4)
dx.upload(x)
clf.profile_enable(True)
for _ in range(3):
    clf.predict_device(dx.at(0), B, dl.at(0))
clf.synchronize()
clf.profile_read(per_step=True)
clf.predict_device(dx.at(0), B, dl.at(0)); clf.synchronize()
agg, per = clf.profile_read(per_step=True)
for r in per:
    ms = r["ms"] / r["launches"]
    print(f"{r['kernel']:12s} {r['name']:16s} {ms * 1e3:8.1f} us {r['flops'] / r['launches'] / (ms * 1e-3) / 1e12:6.1f} TF")
clf.close()
