"""Disassemble the gfx950 code objects inside libbnhip.so and look for instruction forms this repo bans.

Banned (tools/ubench/pkf32_vs_bf16mfma.hip, DESIGN.md "packed-fp32 hazard"): a packed-fp32 VALU instruction (v_pk_fma_f32,
v_pk_mul_f32, v_pk_add_f32) whose op_sel bit for SRC1 is set, i.e. whose LOW result is computed from src1's HIGH half.  On
gfx950 such an instruction returns a wrong low half in lanes 48-63 while another wave on the same CU executes
v_mfma_f32_16x16x32_bf16 - which is what the split-bf16 GEMMs of the other pipeline context do all the time.  The compiler's
SLP vectoriser forms exactly these (hence -fno-slp-vectorize); hand-written vector code could too, so the built library is
checked instead of trusting the flag.

    python tools/isa_audit.py [path/to/lib.so]      -> prints offenders, exit code 1 if any
    python tools/isa_audit.py --classes <kernel substring> [path/to/lib.so]
        -> static instruction-class histogram of every kernel whose (demangled) name contains the substring: VALU split into
           transcendental / f32 mul / add / fma / integer / moves / selects / compares / other, MFMA, LDS, VMEM, SALU, branches,
           waits (round 6).  Static = every path of the kernel counted once (four variants of phase 1, the cold
           activation switch): the executed mix per wave is tools/pmc_classes.sh; this listing says which code the classes are.
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


def insn_class(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op in ("v_exp_f32_e32", "v_rcp_f32_e32", "v_log_f32_e32", "v_sqrt_f32_e32", "v_rsq_f32_e32", "v_rcp_iflag_f32_e32"):
        return "valu:transcendental"
    if op.startswith(("v_pk_mul_f32", "v_mul_f32")):
        return "valu:mul_f32"
    if op.startswith(("v_pk_add_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32")):
        return "valu:add_f32"
    if op.startswith(("v_fma", "v_pk_fma")):
        return "valu:fma"
    if op.startswith("v_cndmask"):
        return "valu:select"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "valu:move"
    if op.startswith("v_cmp"):
        return "valu:compare"
    if op.startswith(("v_readfirstlane", "v_readlane", "v_writelane")):
        return "valu:lane"
    if re.match(r"v_(add|sub|subrev|mul_lo|mul_hi|mul_u32|mul_i32|mad|lshl|lshr|ashr|and|or|xor|not|min_[iu]|max_[iu]|add3|bfe|bfi|lshlrev|lshrrev|ashrrev|med3_[iu])", op):
        return "valu:integer"
    if op.startswith("v_"):
        return "valu:other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "scratch_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_"):
        return "salu"
    return "other"


def classes(path, pattern):
    """-> {demangled kernel name: Counter(class -> static count)} for kernels whose name contains `pattern`."""
    import collections
    out = {}
    for triple, blob in code_objects(path):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob); f.flush()
            txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in txt.splitlines():
            if line.endswith(">:"):
                sym = line.split("<")[-1][:-2]
                name = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip().replace("bnhip::", "").split("(")[0]
                cur = collections.Counter() if pattern in name else None
                if cur is not None:
                    out[name] = cur
                continue
            if cur is None:
                continue
            m = re.match(r"^\t(\S+)", line)
            if m:
                cur[insn_class(m.group(1))] += 1
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
    if len(sys.argv) > 2 and sys.argv[1] == "--classes":
        lib = sys.argv[3] if len(sys.argv) > 3 else os.path.join(here, "birdnet-go_amd", "lib", "libbnhip.so")
        for name, c in sorted(classes(lib, sys.argv[2]).items()):
            valu = sum(v for k, v in c.items() if k.startswith("valu:"))
            print(f"{name}\n    VALU {valu}: " + ", ".join(f"{k[5:]} {v}" for k, v in sorted(c.items()) if k.startswith("valu:")) +
                  "\n    " + ", ".join(f"{k} {v}" for k, v in sorted(c.items()) if not k.startswith("valu:")))
        sys.exit(0)
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "birdnet-go_amd", "lib", "libbnhip.so")
    n, bad = audit(lib)
    print(f"{lib}: {n} packed-fp32 instructions, {len(bad)} with op_sel on src1")
    for k, ins in bad[:40]:
        print(f"  {k}: {ins}")
    sys.exit(1 if bad else 0)
