"""Build driver: compiles the HIP/C++ engine for gfx950 into lib/libbnhip.so (in-tree, so the
built library travels with the repo snapshot to the GPU box)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libbnhip.so")
SOURCES = ["kernels.hip", "generic.hip", "resample.hip", "stft.hip", "engine.cpp", "graph_passes.cpp", "tflite_model.cpp", "model_onnx.cpp", "hostpipe.cpp", "api.cpp"]
HEADERS = ["kernels.h", "engine.h", "tflite_model.h", "model_onnx.h", "hostpipe.h", "fft_r8.h", os.path.join("..", "..", "include", "bnhip.h")]
# -fno-slp-vectorize: the SLP vectorizer turns independent scalar fmaf chains into dependent v_pk_fma_f32 chains.  On gfx950 /
# ROCm 7.2 such a chain (k_mel_banded's band sums) returned wrong LOW halves in a few lanes whenever another stream's kernels
# shared the CU (2-4 % of clips wrong with two contexts in flight, never when run alone: tools/debug/race_stat.py, DESIGN.md
# section 10); packed fp32 buys no issue slots on this part anyway ("v_pk_fma_f32 pairs").
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]
NO_VGPR_FORM = set()     # sources to compile without the VGPR-form MFMA rewrite (none at present)


def _digest():
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    stamp = os.path.join(LIBDIR, ".build_digest")
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # per-object stamps: a translation unit is recompiled only when it, a header or the flags changed
    hh = hashlib.sha256()
    for f in HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            hh.update(fh.read())
    hh.update(" ".join(FLAGS).encode())
    objs, cmds = [], []
    for src in SOURCES:
        obj = os.path.join(LIBDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        h = hh.copy()
        with open(os.path.join(CSRC, src), "rb") as fh:
            h.update(fh.read())
        ostamp, odig = obj + ".digest", h.hexdigest()
        if not force and os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read() == odig:
            continue
        flags = [f for f in FLAGS if src not in NO_VGPR_FORM or f not in ("-mllvm", "-amdgpu-mfma-vgpr-form")]
        cmds.append(([hipcc, "-x", "hip"] + flags + ["-c", os.path.join(CSRC, src), "-o", obj], ostamp, odig))
    # the translation units are independent: compile them side by side
    procs = []
    for cmd, ostamp, odig in cmds:
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        if os.path.exists(ostamp):
            os.remove(ostamp)
        procs.append((cmd, ostamp, odig, subprocess.Popen(cmd)))
    failed = None
    for cmd, ostamp, odig, pr in procs:
        if pr.wait() != 0:
            failed = failed or (pr.returncode, cmd)
        else:
            with open(ostamp, "w") as fh:
                fh.write(odig)
    if failed:
        raise subprocess.CalledProcessError(*failed)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    with open(stamp, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
