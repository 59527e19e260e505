#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace run (rocpd sqlite output) per kernel: calls, total, average.
usage: rocprof_summary.py <results.db> [steps]   (steps: divide totals to get per-step figures)"""
import collections
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    c = sqlite3.connect(db)
    tab = [r[0] for r in c.execute("select name from sqlite_master where type='table' and name like 'rocpd_kernel_dispatch%'")][0]
    suf = tab.replace('rocpd_kernel_dispatch', '')
    ks = {r[0]: r[1] for r in c.execute("select id, kernel_name from rocpd_info_kernel_symbol%s" % suf)}
    per = collections.defaultdict(lambda: [0, 0])
    fam = collections.defaultdict(lambda: [0, 0])
    for kid, s, e in c.execute("select kernel_id,start,end from rocpd_kernel_dispatch%s" % suf):
        n = ks[kid]
        m = re.search(r'ckm\d+([a-z_]+kernel(?:_h)?)(ILi(\d+)E)?', n)
        name = (m.group(1) + ("<%s>" % m.group(3) if m.group(3) else "")) if m else n[:48]
        per[name][0] += 1; per[name][1] += e - s
        f = m.group(1) if m else n[:48]
        fam[f][0] += 1; fam[f][1] += e - s
    tot = sum(v[1] for v in fam.values())
    print("# per kernel family (all template instances), %g timed+warmup steps" % steps)
    print("%-34s %8s %12s %12s %12s %6s" % ("kernel", "calls", "total_ms", "ms_per_step", "avg_us", "pct"))
    for n, v in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        print("%-34s %8d %12.3f %12.3f %12.1f %6.1f" % (n, v[0], v[1] / 1e6, v[1] / 1e6 / steps, v[1] / v[0] / 1e3, 100.0 * v[1] / tot))
    print("\n# per template instance")
    print("%-34s %8s %12s %12s" % ("kernel", "calls", "total_ms", "avg_us"))
    for n, v in sorted(per.items(), key=lambda kv: -kv[1][1]):
        print("%-34s %8d %12.3f %12.1f" % (n, v[0], v[1] / 1e6, v[1] / v[0] / 1e3))


if __name__ == "__main__":
    main()
