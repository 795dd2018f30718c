#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace [--pmc ...]) as per-kernel stats.
Usage: python tools/prof_summary.py <results.db> [--csv out.csv] [--window SKIP_TAIL:COUNT]
--window restricts the kernel table to COUNT consecutive bnhip dispatches ending SKIP_TAIL dispatches before the last one
(e.g. 63:1260 = the 20 timed steps of `bench.py --steps 20`: the create-time autotune launches come before them, the
8-clip consistency check (61 launches) after them), so the per-kernel averages are those of the timed region."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*\)$", "", name)
    name = name.replace("void ", "").replace("bnhip::", "")
    return name[:70]


def main():
    path = sys.argv[1]
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if "kernel_dispatch" in t][0]
    ks = [t for t in tabs if "kernel_symbol" in t][0]
    where = ""
    if "--window" in sys.argv:
        skip, count = (int(v) for v in sys.argv[sys.argv.index("--window") + 1].split(":"))
        ids = [r[0] for r in cur.execute(f"select d.id from {kd} d join {ks} s on d.kernel_id = s.id "
                                         f"where s.display_name like '%bnhip%' order by d.start")]
        ids = ids[len(ids) - skip - count:len(ids) - skip]
        cur.execute("create temp table win (id integer primary key)")
        cur.executemany("insert into win values (?)", [(i,) for i in ids])
        where = "where d.id in (select id from win) "
    rows = cur.execute(
        f"select s.display_name, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
        f"max(s.arch_vgpr_count), max(s.accum_vgpr_count), max(s.sgpr_count), max(d.group_segment_size) "
        f"from {kd} d join {ks} s on d.kernel_id = s.id {where}group by s.display_name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_ms,avg_us,min_us,max_us,pct,vgpr,agpr,sgpr,lds_bytes"]
    for n, c, t, mn, mx, vg, ag, sg, lds in rows:
        lines.append(f"\"{short(n)}\",{c},{t / 1e6:.3f},{t / c / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100 * t / total:.1f},{vg},{ag},{sg},{lds}")
    # PMC counters if present
    pm = [t for t in tabs if "pmc_event" in t]
    ip = [t for t in tabs if "info_pmc" in t]
    if pm and ip:
        try:
            prow = cur.execute(
                f"select s.display_name, p.name, sum(e.value), count(*) from {pm[0]} e join {ip[0]} p on e.pmc_id = p.id "
                f"join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id group by s.display_name, p.name").fetchall()
            if prow:
                lines.append("")
                lines.append("kernel,counter,sum,dispatches")
                for n, cn, v, c in prow:
                    lines.append(f"\"{short(n)}\",{cn},{v},{c}")
        except sqlite3.Error as e:  # schema differences between rocprofv3 builds
            lines.append(f"# pmc query failed: {e}")
    out = "\n".join(lines)
    if "--csv" in sys.argv:
        open(sys.argv[sys.argv.index("--csv") + 1], "w").write(out + "\n")
    print(out)


if __name__ == "__main__":
    main()
