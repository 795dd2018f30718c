"""Build driver: compiles the HIP/C++ engine for gfx950 into lib/libbnhip.so (in-tree, so the
built library travels with the repo snapshot to the GPU box)."""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.environ.get("BNHIP_LIBDIR") or os.path.join(HERE, "lib")      # (BNHIP_LIBDIR + BNHIP_EXTRA_FLAGS: an A/B build beside the shipped one)
LIB = os.path.join(LIBDIR, "libbnhip.so")
SOURCES = ["kernels.hip", "expdw.hip", "pw_b16.hip", "pw_ws.hip", "generic.hip", "resample.hip", "stft.hip", "engine.cpp", "graph_passes.cpp", "tflite_model.cpp", "model_onnx.cpp", "hostpipe.cpp", "numa.cpp", "windows.cpp", "api.cpp"]
HEADERS = ["kernels.h", "pw_common.h", "pw_split.h", "engine.h", "tflite_model.h", "model_onnx.h", "hostpipe.h", "numa.h", "windows.h", "fft_r8.h", os.path.join("..", "..", "include", "bnhip.h")]
# -fno-slp-vectorize: gfx950 hazard, reproduced standalone (tools/ubench/pkf32_vs_bf16mfma.hip, profiles/r04_pk_hazard.txt): a
# packed-fp32 VALU instruction whose op_sel bit for src1 is set (v_pk_fma_f32 / v_pk_mul_f32 ... op_sel:[0,1,..]: the LOW result
# computed from src1's HIGH half) returns a wrong low half in lanes 48-63 while another wave on the same CU executes
# v_mfma_f32_16x16x32_bf16 - i.e. whenever the other pipeline context runs a split-bf16 GEMM.  The SLP vectoriser forms exactly
# these when it packs two scalar fmaf chains that share a broadcast operand (176 of them in stft.hip: 2-4 % of clips wrong with
# two contexts in flight).  tests/test_isa_audit.py disassembles the built library and fails on any such instruction.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-unused-value", "-ffp-contract=off", "-fno-slp-vectorize", "-mllvm", "-amdgpu-mfma-vgpr-form"]
FLAGS += os.environ.get("BNHIP_EXTRA_FLAGS", "").split()
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
