#!/usr/bin/env python
"""Print a merged timeline (kernels, memory copies, HIP API calls) of the last `window_ms` of a rocprofv3 rocpd database.
usage: timeline.py <results.db> [window_ms] [min_api_us]"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    window = float(sys.argv[2]) if len(sys.argv) > 2 else 130.0
    min_api = float(sys.argv[3]) if len(sys.argv) > 3 else 100.0
    c = sqlite3.connect(db)
    suf = [r[0] for r in c.execute("select name from sqlite_master where name like 'rocpd_region%'")][0].replace('rocpd_region', '')
    strings = dict(c.execute("select id, string from rocpd_string%s" % suf))
    ks = dict(c.execute("select id, kernel_name from rocpd_info_kernel_symbol%s" % suf))
    ev = []
    for kid, s, e in c.execute("select kernel_id, start, end from rocpd_kernel_dispatch%s" % suf):
        m = re.search(r'ckm\d+([a-z0-9_]+kernel)(ILi(\d+)E)?', ks[kid])
        name = (m.group(1) + ("<%s>" % m.group(3) if m.group(3) else "")) if m else ks[kid][:40]
        ev.append((s, e, 'K', name))
    for s, e, nid, size in c.execute("select start, end, name_id, size from rocpd_memory_copy%s" % suf):
        ev.append((s, e, 'C', "%s %d B" % (strings.get(nid, '?'), size)))
    for s, e, nid in c.execute("select start, end, name_id from rocpd_region%s" % suf):
        if (e - s) / 1e3 >= min_api:
            ev.append((s, e, 'A', strings.get(nid, '?')))
    ev.sort()
    t_end = max(e for _, e, _, _ in ev)
    t0 = t_end - window * 1e6
    for s, e, kind, name in ev:
        if s >= t0:
            print("%9.3f %9.3f %s %s" % ((s - t0) / 1e6, (e - s) / 1e6, kind, name))


if __name__ == "__main__":
    main()
