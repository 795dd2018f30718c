#!/usr/bin/env python
"""Ordered per-launch table of the LAST bench step from a rocprofv3 --kernel-trace sqlite database:
index, kernel, workgroups, duration_us.  Usage: python tools/step_trace.py <db> <launches_per_step>"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    n = int(sys.argv[2])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    rows = list(cur.execute(f"select s.display_name, d.grid_size_x, d.workgroup_size_x, d.grid_size_y, d.start, d.end from {kd} d join {ks} s "
                            f"on d.kernel_id = s.id order by d.start"))
    rows = [r for r in rows if "bnhip" in r[0]][-n:]
    print("i,kernel,workgroups,us")
    for i, (name, gx, wx, gy, st, en) in enumerate(rows):
        name = re.sub(r"\(.*", "", name).replace("void bnhip::", "")
        print(f"{i},\"{name}\",{gx // max(wx, 1) * max(gy, 1)},{(en - st) / 1e3:.1f}")


if __name__ == "__main__":
    main()
