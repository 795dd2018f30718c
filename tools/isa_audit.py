"""Disassemble the gfx950 code objects inside libbnhip.so and look for instruction forms this repo bans.

Banned (tools/ubench/pkf32_vs_bf16mfma.hip, DESIGN.md "packed-fp32 hazard"): a packed-fp32 VALU instruction (v_pk_fma_f32,
v_pk_mul_f32, v_pk_add_f32) whose op_sel bit for SRC1 is set, i.e. whose LOW result is computed from src1's HIGH half.  On
gfx950 such an instruction returns a wrong low half in lanes 48-63 while another wave on the same CU executes
v_mfma_f32_16x16x32_bf16 - which is what the split-bf16 GEMMs of the other pipeline context do all the time.  The compiler's
SLP vectoriser forms exactly these (hence -fno-slp-vectorize); hand-written vector code could too, so the built library is
checked instead of trusting the flag.

    python tools/isa_audit.py [path/to/lib.so]      -> prints offenders, exit code 1 if any
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
PK = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\b(.*)")
OPSEL = re.compile(r"op_sel:\[([01,]+)\]")


def code_objects(path):
    """-> [(triple, bytes)] for every device entry of every clang offload bundle in the file."""
    b = open(path, "rb").read()
    out = []
    for m in re.finditer(re.escape(MAGIC), b):
        base = m.start()
        n = struct.unpack_from("<Q", b, base + 24)[0]
        at = base + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", b, at)
            triple = b[at + 24:at + 24 + tl].decode()
            at += 24 + tl
            if "amdgcn" in triple and size:
                out.append((triple, b[base + off:base + off + size]))
    return out


def audit(path):
    """-> (n_packed_f32_instructions, [(kernel, instruction text)]) over all gfx950 code objects of `path`."""
    offenders, n_pk = [], 0
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        kernel = "?"
        for line in txt.splitlines():
            if line.endswith(">:"):
                kernel = line.split("<")[-1][:-2]
                continue
            m = PK.search(line)
            if not m:
                continue
            n_pk += 1
            sel = OPSEL.search(m.group(2))
            if sel:
                bits = sel.group(1).split(",")
                if len(bits) > 1 and bits[1] == "1":
                    offenders.append((kernel, line.strip()))
    return n_pk, offenders


if __name__ == "__main__":
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "birdnet-go_amd", "lib", "libbnhip.so")
    n, bad = audit(lib)
    print(f"{lib}: {n} packed-fp32 instructions, {len(bad)} with op_sel on src1")
    for k, ins in bad[:40]:
        print(f"  {k}: {ins}")
    sys.exit(1 if bad else 0)
