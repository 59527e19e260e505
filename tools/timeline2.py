#!/usr/bin/env python
"""Kernel timeline of the last complete search in a rocprofv3 --kernel-trace database: start, end, duration (ms), queue, grid, kernel.
usage: timeline2.py <results.db> [min_us] [which]   (which: -1 last search, -2 the one before, ...)"""
import re
import sqlite3
import sys

db = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
which = int(sys.argv[3]) if len(sys.argv) > 3 else -2
c = sqlite3.connect(db)
tab = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
suf = tab.replace('rocpd_kernel_dispatch', '')
ks = {r[0]: r[1] for r in c.execute("select id, kernel_name from rocpd_info_kernel_symbol%s" % suf)}
rows = list(c.execute("select kernel_id,start,end,queue_id,grid_size_x from %s order by start" % tab))


def nm(n):
    m = re.search(r'ckm\d+([a-z_0-9]+kernel)(ILi(\d+)E(Lb(\d)E)?)?', n)
    if not m:
        return n[:30]
    return m.group(1) + ("<%s%s>" % (m.group(3), ("," + m.group(5)) if m.group(5) else "") if m.group(3) else "")


# a search starts at the first ssv kernel after a gap without ssv kernels
starts = []
last_ssv_end = -1e18
for i, r in enumerate(rows):
    if 'ssv_kernel' in ks[r[0]]:
        if r[1] - last_ssv_end > 20e6:
            starts.append(i)
        last_ssv_end = max(last_ssv_end, r[2])
i0 = starts[which]
i1 = starts[which + 1] if which + 1 < 0 and which + 1 + len(starts) < len(starts) else len(rows)
t0 = rows[i0][1]
agg = {}
for r in rows[i0:i1]:
    n = nm(ks[r[0]])
    fam = n.split('<')[0]
    a = agg.setdefault(fam, [0, 0.0, 1e18, 0.0])
    a[0] += 1; a[1] += (r[2] - r[1]) / 1e6; a[2] = min(a[2], (r[1] - t0) / 1e6); a[3] = max(a[3], (r[2] - t0) / 1e6)
    if (r[2] - r[1]) >= min_us * 1e3:
        print("%8.3f %8.3f %7.3f q%-3d g%-8d %s" % ((r[1] - t0) / 1e6, (r[2] - t0) / 1e6, (r[2] - r[1]) / 1e6, r[3], r[4], n))
print("\n# family: launches, sum of durations ms, first start, last end")
for f, a in sorted(agg.items(), key=lambda kv: kv[1][2]):
    print("%-24s %5d %9.3f %9.3f %9.3f" % (f, a[0], a[1], a[2], a[3]))
