#!/usr/bin/env python
"""Chronological listing of the LAST n kernel dispatches of a rocprofv3 rocpd database, with grid sizes and idle gaps.
Usage: rocpd_timeline.py results.db [n_last]"""
import sqlite3, sys, re
c = sqlite3.connect(sys.argv[1])
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 700
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
namecol = "kernel_name" if "kernel_name" in scol else ("display_name" if "display_name" in scol else "name")
gcols = [x for x in cols if x in ("grid_size_x", "grid_size_y", "workgroup_size_x", "grid_size", "workgroup_size")] or [x for x in cols if "grid" in x]
q = "select s.%s, d.start, d.end, %s from %s d join %s s on d.kernel_id=s.id order by d.start" % (
    namecol, ",".join("d." + x for x in gcols), kd, ks)
rows = list(c.execute(q))[-n_last:]
prev_end = None
busy = gap = 0.0
print("# cols: t_us dur_us gap_us kernel", gcols)
t0 = rows[0][1]
for r in rows:
    g = 0.0 if prev_end is None else (r[1] - prev_end) / 1e3
    d = (r[2] - r[1]) / 1e3
    busy += d
    gap += max(g, 0.0)
    nm = re.sub(r"\(.*", "", r[0])
    nm = re.sub(r"^void ", "", nm)[:70]
    print("%9.1f %8.1f %6.1f  %-70s %s" % ((r[1] - t0) / 1e3, d, g, nm, r[3:]))
    prev_end = max(prev_end or 0, r[2])
print("# busy %.1f us, gaps %.1f us, span %.1f us" % (busy, gap, (rows[-1][2] - t0) / 1e3))
