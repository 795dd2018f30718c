#!/usr/bin/env python
"""Per-kernel PMC averages from rocprofv3 rocpd databases (one db per --pmc pass).
Usage: python tools/pmc_summary.py [--traffic-json OUT] [--mfma-json OUT] [--binding FILE] [--window SKIP:COUNT] <db> [<db> ...]
  --window: only COUNT consecutive engine dispatches ending SKIP before the last (e.g. the timed steps: skips the create-time
    autotune launches, which would otherwise pollute the per-kernel averages with other layers' shapes)
  -> CSV on stdout: kernel, dispatches, avg_us, then every counter as an average PER DISPATCH OF THE PASS THAT COLLECTED IT.
     All passes must have run the same kernel instantiations (BNHIP_TUNE_FILE pins the create-time tuner across processes):
     a kernel missing from a pass is an error, not a silent zero.
  --traffic-json: {kernel: {dispatches, hbm_bytes_per_launch, fetch_kib_raw, write_kib}},
    hbm bytes = FETCH_SIZE[KiB] * 1024 * 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE[KiB] * 1024.
  --binding: a JSON object ({lib_digest, tune_sha256, plan_signature}: what the passes ran on) copied into both JSON outputs as
    "_binding"; bench.py quotes a counter file only when its lib_digest is the loaded library's.
  --mfma-json: MFMA-pipe utilisation per kernel and per kernel class:
    SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the chip) / (kernel time * 2.4 GHz * 1024 SIMDs), kernel time taken from the
    pass that collected the counter; plus VALU : MFMA instruction ratio."""
import json
import re
import sqlite3
import sys
from collections import defaultdict

CLK_GHZ, N_SIMD = 2.4, 1024


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("void ", "").replace("bnhip::", "")[:70]


def kclass(k):
    for pre, c in (("k_expand_dw", "expand_dw"), ("k_pw_", "pw_gemm"), ("k_stft", "stft"), ("k_mel", "frontend"), ("k_frontend", "frontend"), ("k_normalize", "frontend"),
                   ("k_dwconv", "dwconv"), ("k_se", "se"), ("k_mean", "mean"), ("k_clip_minmax", "clip_minmax"), ("k_stem", "conv_direct"), ("k_conv", "conv_direct")):
        if k.startswith(pre):
            return c
    return "other"


def load(path, window):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    pm = [t for t in tabs if "pmc_event" in t][0]
    ip = [t for t in tabs if "info_pmc" in t][0]
    rows = list(cur.execute(
        f"select d.event_id, s.display_name, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    if window:          # keep COUNT consecutive bnhip dispatches ending SKIP dispatches before the last one (the timed steps)
        skip, count = window
        mine = [r for r in rows if "bnhip" in r[1]]
        rows = mine[len(mine) - skip - count:len(mine) - skip]
    disp = {eid: (short(name), en - st) for eid, name, st, en in rows}
    vals = defaultdict(lambda: defaultdict(float))
    for eid, cname, v in cur.execute(f"select e.event_id, p.name, e.value from {pm} e join {ip} p on e.pmc_id = p.id"):
        if eid in disp:
            vals[eid][cname] += v
    return disp, vals


def main():
    argv = sys.argv[1:]
    tj = mj = None
    window = None
    binding = None
    while argv and argv[0] in ("--traffic-json", "--mfma-json", "--window", "--binding"):
        if argv[0] == "--traffic-json":
            tj = argv[1]
        elif argv[0] == "--mfma-json":
            mj = argv[1]
        elif argv[0] == "--binding":
            binding = json.load(open(argv[1]))
        else:
            window = tuple(int(v) for v in argv[1].split(":"))
        argv = argv[2:]
    per = {}                 # counter -> {kernel: (sum, dispatches, ns)} from the pass that collected it
    cnt0, dur0 = defaultdict(int), defaultdict(float)
    kernel_sets = []
    for pi, path in enumerate(argv):
        disp, vals = load(path, window)
        ks = defaultdict(lambda: [0, 0.0])
        for eid, (name, d) in disp.items():
            ks[name][0] += 1; ks[name][1] += d
            if pi == 0:
                cnt0[name] += 1; dur0[name] += d
        kernel_sets.append(set(ks))
        sums = defaultdict(lambda: defaultdict(float))
        for eid, cv in vals.items():
            for c, v in cv.items():
                sums[c][disp[eid][0]] += v
        for c, byk in sums.items():
            per[c] = {k: (byk.get(k, 0.0), ks[k][0], ks[k][1]) for k in ks}
    common = set.intersection(*kernel_sets) if kernel_sets else set()
    odd = set.union(*kernel_sets) - common if kernel_sets else set()
    if odd:
        print(f"# ERROR: the passes did not run the same kernels ({len(odd)} differ, e.g. {sorted(odd)[:3]}): pin the tuner with BNHIP_TUNE_FILE", file=sys.stderr)
    counters = list(per)
    print("kernel,dispatches,avg_us," + ",".join(f"{c}_per_dispatch" for c in counters))
    order = sorted(common, key=lambda k: -dur0[k])
    for k in order:
        cells = []
        for c in counters:
            s, n, _ = per[c][k]
            cells.append(f"{s / max(n, 1):.4g}")
        print(f"\"{k}\",{cnt0[k]},{dur0[k] / cnt0[k] / 1e3:.1f}," + ",".join(cells))
    if tj:
        out = {}
        for k in order:
            if not k.startswith("k_") or "FETCH_SIZE" not in per or "WRITE_SIZE" not in per:
                continue
            fs, fn, _ = per["FETCH_SIZE"][k]; ws, wn, _ = per["WRITE_SIZE"][k]
            f, w = fs / max(fn, 1), ws / max(wn, 1)
            out[k] = {"dispatches": cnt0[k], "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024, "fetch_kib_raw": f, "write_kib": w}
        if binding:
            out["_binding"] = binding
        json.dump(out, open(tj, "w"), indent=1)
    if mj and "SQ_VALU_MFMA_BUSY_CYCLES" in per:
        ker, cls = {}, defaultdict(lambda: [0.0, 0.0, 0.0, 0.0])
        for k in order:
            busy, n, ns = per["SQ_VALU_MFMA_BUSY_CYCLES"][k]
            valu = per.get("SQ_INSTS_VALU", {}).get(k, (0, 0, 0))[0]
            mfma = per.get("SQ_INSTS_MFMA", {}).get(k, (0, 0, 0))[0]
            denom = ns * CLK_GHZ * N_SIMD
            ker[k] = {"mfma_util": busy / denom if denom else 0.0, "avg_us": ns / max(n, 1) / 1e3, "dispatches": n,
                      "valu_per_mfma": (valu / mfma) if mfma else None}
            c = cls[kclass(k)]
            c[0] += busy; c[1] += denom; c[2] += valu; c[3] += mfma
        json.dump({"denominator": f"kernel time (the PMC pass's own dispatch durations) x {CLK_GHZ} GHz x {N_SIMD} SIMDs",
                   "numerator": "SQ_VALU_MFMA_BUSY_CYCLES summed over the class's dispatches",
                   "classes": {c: {"mfma_util": v[0] / v[1] if v[1] else 0.0, "valu_per_mfma": (v[2] / v[3]) if v[3] else None} for c, v in cls.items()},
                   "clock_note": "the denominator uses the 2.4 GHz maximum clock; MFMA-heavy kernels of this workload run at 1.93-2.00 GHz (shader-clock vs wall-clock "
                                 "stamps, tools/ubench/ws_trace.hip), so the pipe is busy for ~1.2x the stated share of the cycles that existed",
                   "kernels": ker, **({"_binding": binding} if binding else {})}, open(mj, "w"), indent=1)
    sys.exit(2 if odd else 0)


if __name__ == "__main__":
    main()
