"""VGPR / LDS / scratch of the kernels in libbnhip.so (code-object notes): python tools/kernel_regs.py [substring]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_audit as ia
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "birdnet-go_amd", "lib", "libbnhip.so")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
for triple, blob in ia.code_objects(lib):
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(blob); f.flush()
        txt = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    for m in re.finditer(r"\.group_segment_fixed_size:\s*(\d+).*?\.name:\s*(\S+).*?\.private_segment_fixed_size:\s*(\d+).*?\.vgpr_count:\s*(\d+).*?\.vgpr_spill_count:\s*(\d+)", txt, re.S):
        lds, name, priv, vg, sp = m.groups()
        d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().replace("bnhip::", "").split("(")[0]
        if pat in d:
            print(f"{d[:90]:90s} vgpr {vg:>4s} lds {lds:>6s} scratch {priv:>4s} spilled {sp}")
