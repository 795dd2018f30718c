#!/usr/bin/env python
"""Per-(kernel, grid) counter table from rocprofv3 PMC databases, normalised per wave when SQ_WAVES was collected.
Usage: [LASTN=n] python tools/pmc_layers.py <kernel-substring> <db> [<db> ...]
LASTN keeps only the last n matching dispatches (e.g. one step's worth: skips the create-time autotune launches)."""
import collections
import os
import re
import sqlite3
import sys


def load(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    pm = [t for t in tabs if "pmc_event" in t][0]
    ip = [t for t in tabs if "info_pmc" in t][0]
    disp = {}
    for eid, name, gx, wx, st, en in cur.execute(
            f"select d.event_id, s.display_name, d.grid_size_x, d.workgroup_size_x, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id"):
        disp[eid] = (name, gx // max(wx, 1), en - st)
    vals = collections.defaultdict(dict)
    for eid, cname, v in cur.execute(f"select e.event_id, p.name, e.value from {pm} e join {ip} p on e.pmc_id = p.id"):
        vals[eid][cname] = vals[eid].get(cname, 0) + v
    return disp, vals


def main():
    pat = sys.argv[1]
    for path in sys.argv[2:]:
        disp, vals = load(path)
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.Counter()
        dur = collections.Counter()
        items = [(eid, v) for eid, v in sorted(disp.items()) if pat in v[0]]
        lastn = int(os.environ.get("LASTN", "0"))
        if lastn:
            items = items[-lastn:]
        for eid, (name, blocks, d) in items:
            key = (re.sub(r"\(.*", "", name).replace("void bnhip::", ""), blocks)
            cnt[key] += 1
            dur[key] += d
            for c, v in vals[eid].items():
                agg[key][c] += v
        for key in sorted(agg):
            n = cnt[key]
            a = agg[key]
            w = a.get("SQ_WAVES", 0) / n
            if w:
                print(key, f"{dur[key] / n / 1e3:.0f}us waves={w:.0f}", " ".join(f"{c[3:]}={a[c] / n / w:.0f}" for c in a if c != "SQ_WAVES"))
            else:
                print(key, f"{dur[key] / n / 1e3:.0f}us", " ".join(f"{c[3:]}={a[c] / n:.3g}" for c in a))


if __name__ == "__main__":
    main()
