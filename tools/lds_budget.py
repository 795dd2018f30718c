"""Per pointwise layer of a plan: MFMA-pipe cycles against LDS-pipe cycles of the split-bf16 GEMM tile the engine would launch,
per slab and CU - the accounting behind DESIGN.md section 11 (i).  Needs no GPU (plan-only engine).

    python tools/lds_budget.py [--workload v24|perch] [--precision f32|bf16] [--batch 256] [--tune-file profiles/r04_tune.txt]

The tiles are the tuned ones when a tuning file of that plan is given (BNHIP_TUNE_FILE's format: the file the bench and every
profile pass of the round ran with), else the planner's defaults.

Model (MI355X_MICROARCH.md, LDS section): ds_read_b128 = 4 LDS cycles per wave-instruction (one 16 x 32 bf16 fragment),
ds_write_b128 ~ 13 cycles per wave-instruction (1 KB), v_mfma_f32_16x16x32_bf16 = 16 cycles on the wave's SIMD.  A block of
4 waves computes (64 wm) rows x (16 nt) columns; a wave holds wm row tiles and reads nt x planes fragments per slab, each used wm
times; the block writes its nt x planes KB of weight tile once per slab.  Co-resident blocks per CU: 4 (the kernels' occupancy at
<= 128 VGPRs).  products = 6 (fp32 engines: three planes) or 1 (bf16 engines: one plane)."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import birdnet_go_amd  # noqa: F401
from birdnet_go_amd import host, synth_model as sm

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="v24")
    ap.add_argument("--precision", default="f32")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--tune-file", default=None)
    a = ap.parse_args()
    tuned = {}
    if a.tune_file:
        for line in open(a.tune_file).read().splitlines()[1:]:
            f = line.split(" ", 10)
            tuned[f[10]] = (int(f[4]), int(f[5]))             # name -> (nt_full, wm_full)
    cfg = sm.perch_config() if a.workload == "perch" else sm.SynthConfig()
    clf = host.HipClassifier(sm.build_model(cfg), plan_only=True, max_batch=a.batch, precision=a.precision)
    six = a.precision == "f32"
    planes, products = (3, 6) if six else (1, 1)
    print(f"{'layer':18s} {'M':>8s} {'K':>5s} {'N':>5s} tile      slabs  mfma/SIMD  lds rd  lds wr  lds/mfma   note")
    for s in clf.describe()["steps"]:
        if s.get("kernel") not in ("pw_gemm", "pw") or not s.get("bx"):
            continue
        K, N, M = s["C"], s["Co"], a.batch * s["H"] * s["W"]
        ntf, wmf = tuned.get(s["name"], (s["nt_full"], s["wm_full"]))
        nt = ntf or 4
        wm = 1 if wmf in (5, 7, 10) else 2
        blocks_per_cu = 4
        waves = 4 * blocks_per_cu
        mfma = blocks_per_cu * wm * nt * products * 16            # per SIMD: one wave of each co-resident block
        rd = waves * nt * planes * 4                              # LDS-array cycles, all 16 waves of the CU
        wr = blocks_per_cu * nt * planes * 13
        slabs = (K + 31) // 32
        note = "weight columns fit LDS (k_pw_ws)" if 65 <= K <= 192 and N >= 64 and not s.get("fused_scale") else ""
        print(f"{s['name']:18s} {M:8d} {K:5d} {N:5d} {64 * wm:3d}x{16 * nt:<4d} {slabs:5d}  {mfma:9d}  {rd:6d}  {wr:6d}  {(rd + wr) / mfma:8.2f}   {note}")
    clf.close()
