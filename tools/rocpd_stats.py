#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite result (kernel-trace) into a per-kernel table (like --stats CSV)."""
import sqlite3, sys, re
db = sys.argv[1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
namecol = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else "name")
q = "select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by s.%s order by 3 desc" % (namecol, kd, ks, namecol)
rows = list(c.execute(q))
tot = sum(r[2] for r in rows)
print("%-90s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
for n, cnt, t, mn, mx in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    n = re.sub(r"\(.*", "", n)[:90]
    print("%-90s %7d %12.1f %10.1f %10.1f %10.1f %6.2f" % (n, cnt, t / 1e3, t / 1e3 / cnt, mn / 1e3, mx / 1e3, 100.0 * t / tot))
print("TOTAL kernel time us: %.1f" % (tot / 1e3))
# optional: argv[3] = substring -> list individual dispatches (chronological) of matching kernels with grid size
if len(sys.argv) > 3:
    pat = sys.argv[3]
    gcols = [x for x in cols if "grid" in x or "workgroup" in x]
    q = "select s.%s, d.start, d.end-d.start, %s from %s d join %s s on d.kernel_id=s.id order by d.start" % (
        namecol, ",".join("d." + x for x in gcols), kd, ks)
    print("individual dispatches matching", pat, gcols)
    for row in c.execute(q):
        if pat in row[0]:
            print("%-40s %10.1f us  %s" % (re.sub(r"\(.*", "", row[0])[:40], row[2] / 1e3, row[3:]))
