"""Kernel-level parity of the split-bf16 GEMM family, without a model: tools/ubench/pw_lab --fuzz drives the library's own dispatcher
(launch_pw_bx3) on random shapes - K tails, ragged and odd N, single rows, squeeze-excite scale / residual / swish on and off - through
the tiled kernels (k_pw_bx3 64- / 128-row, k_pw_bx3p, k_pw_b16 both tile heights), k_pw_ws (forced onto every layer it accepts) and
k_pw_lat (every call it accepts), and compares every output bit with the tiled k_pw_bx3, which itself is held to an fp64 dot product
on sampled outputs.  The model-level tests only ever see the v2.4 / Perch layer shapes."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_gemm_family_is_bit_identical_on_random_shapes(gpu, built_lib, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "pw_lab")
    libdir = os.path.dirname(built_lib)
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "birdnet-go_amd", "csrc"), "-o", exe,
                    os.path.join(ROOT, "tools", "ubench", "pw_lab.cpp"), "-L", libdir, "-lbnhip", "-Wl,-rpath," + libdir],
                   check=True, capture_output=True, timeout=600)
    import re
    total = ws = lat = 0
    for seed in (1, 2):
        r = subprocess.run([exe, "--fuzz", str(seed), "30"], capture_output=True, text=True, timeout=600)
        last = r.stdout.strip().splitlines()[-1]
        m = re.match(r"fuzz: 30 shapes, (\d+) comparisons \((\d+) on k_pw_ws, (\d+) on k_pw_lat\), 0 mismatches$", last)
        assert r.returncode == 0 and m, r.stdout[-3000:]
        total += int(m.group(1)); ws += int(m.group(2)); lat += int(m.group(3))
    assert total >= 200 and ws >= 20 and lat >= 10, (total, ws, lat)      # (the round-5 kernels really were among the candidates)
