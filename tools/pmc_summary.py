#!/usr/bin/env python
"""Per-kernel PMC sums from rocprofv3 rocpd databases (one db per --pmc pass).
Usage: python tools/pmc_summary.py [--traffic-json OUT] [--window SKIP:COUNT] <db> [<db> ...]
  --window: only COUNT consecutive engine dispatches ending SKIP before the last (e.g. the timed steps: skips the create-time
    autotune launches, which would otherwise pollute the per-kernel averages with other layers' shapes)
  -> CSV on stdout (kernel, dispatches, avg_us, counter sums per dispatch)
  --traffic-json: also write {kernel: {dispatches, hbm_bytes_per_launch, fetch_kib_raw, write_kib}} where
    hbm bytes = FETCH_SIZE[KiB] * 1024 * 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE[KiB] * 1024."""
import json
import re
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    return name.replace("void ", "").replace("bnhip::", "")[:72]


WINDOW = None


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    pm = [t for t in tabs if "pmc_event" in t][0]
    ip = [t for t in tabs if "info_pmc" in t][0]
    # per dispatch: name, grid, duration
    disp = {}
    rows = list(cur.execute(
        f"select d.event_id, s.display_name, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.workgroup_size_x, d.start, d.end "
        f"from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
    if WINDOW:          # keep COUNT consecutive bnhip dispatches ending SKIP dispatches before the last one (the timed steps)
        skip, count = WINDOW
        mine = [r for r in rows if "bnhip" in r[1]]
        rows = mine[len(mine) - skip - count:len(mine) - skip]
    for eid, name, gx, gy, gz, wx, st, en in rows:
        disp[eid] = (short(name), gx * gy * gz // max(wx, 1), en - st)
    vals = defaultdict(lambda: defaultdict(float))
    for eid, cname, v in cur.execute(f"select e.event_id, p.name, e.value from {pm} e join {ip} p on e.pmc_id = p.id"):
        vals[eid][cname] += v
    return disp, vals


def main():
    agg = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    dur = defaultdict(float)
    counters = []
    argv = sys.argv[1:]
    tj = None
    global WINDOW
    while argv and argv[0] in ("--traffic-json", "--window"):
        if argv[0] == "--traffic-json":
            tj = argv[1]
        else:
            WINDOW = tuple(int(v) for v in argv[1].split(":"))
        argv = argv[2:]
    sys.argv = [sys.argv[0]] + argv
    for path in sys.argv[1:]:
        disp, vals = load(path)
        seen = set()
        for eid, (name, blocks, d) in disp.items():
            if eid not in vals:
                continue
            key = name
            for c, v in vals[eid].items():
                agg[key][c] += v
                if c not in counters:
                    counters.append(c)
            if (path, key) not in seen:
                seen.add((path, key))
        # duration/count from the first db only
        if path == sys.argv[1]:
            for eid, (name, blocks, d) in disp.items():
                cnt[name] += 1
                dur[name] += d
    print("kernel,dispatches,avg_us," + ",".join(f"{c}_per_dispatch" for c in counters))
    for k in sorted(agg, key=lambda k: -dur[k]):
        if not cnt[k]:
            continue
        print(f"\"{k}\",{cnt[k]},{dur[k] / cnt[k] / 1e3:.1f}," + ",".join(f"{agg[k].get(c, 0) / cnt[k]:.4g}" for c in counters))
    if tj:
        out = {}
        for k in sorted(agg, key=lambda k: -dur[k]):
            if not cnt[k] or not k.startswith("k_"):
                continue
            f = agg[k].get("FETCH_SIZE", 0) / cnt[k]
            w = agg[k].get("WRITE_SIZE", 0) / cnt[k]
            out[k] = {"dispatches": cnt[k], "hbm_bytes_per_launch": f * 1024 * 2 + w * 1024, "fetch_kib_raw": f, "write_kib": w}
        json.dump(out, open(tj, "w"), indent=1)


if __name__ == "__main__":
    main()
