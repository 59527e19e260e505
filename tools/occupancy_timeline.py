#!/usr/bin/env python
"""Where the device's time goes in a rocprofv3 kernel trace (csv: *_kernel_trace.csv): the union of the SSV launches' intervals (the
throughput-bound phase: the VALUs issue for every CU), the time only chain kernels run (latency-bound: one wavefront per work item, the
device mostly idle), and the time nothing runs (host gaps).  Prints the totals for the window and a coarse timeline.
usage: occupancy_timeline.py <kernel_trace.csv> [bucket_ms] [skip_first_ms]"""
import csv
import re
import sys


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def measure(iv):
    return sum(e - s for s, e in iv)


def subtract(a, b):
    """a minus b, both unions"""
    out, j = [], 0
    for s, e in a:
        cur = s
        while j < len(b) and b[j][1] <= cur:
            j += 1
        k = j
        while k < len(b) and b[k][0] < e:
            if b[k][0] > cur:
                out.append([cur, b[k][0]])
            cur = max(cur, b[k][1])
            k += 1
        if cur < e:
            out.append([cur, e])
    return out


def main():
    path = sys.argv[1]
    bucket = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
    skip = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    rows = []
    for r in csv.DictReader(open(path)):
        m = re.search(r'ckm::([a-z0-9_]+kernel(?:_h)?)', r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else "other"))
    t0 = min(r[0] for r in rows) + skip * 1e6
    rows = [r for r in rows if r[1] > t0]
    t1 = max(r[1] for r in rows)
    fam = {}
    for s, e, k in rows:
        fam.setdefault("ssv" if k.startswith("ssv_kernel") else ("copy" if k == "other" else "chain"), []).append((max(s, t0), e))
    ssv, chain, copy = union(fam.get("ssv", [])), union(fam.get("chain", [])), union(fam.get("copy", []))
    anyk = union([tuple(x) for x in ssv + chain + copy])
    wall = (t1 - t0) / 1e6
    only_chain = subtract(union([tuple(x) for x in chain + copy]), ssv)
    print("# %s" % path)
    print("window %.1f ms: SSV launch(es) running %.1f ms (%.1f %%), only chain/copy kernels %.1f ms (%.1f %%), nothing on the device %.1f ms (%.1f %%)"
          % (wall, measure(ssv) / 1e6, 100 * measure(ssv) / 1e6 / wall, measure(only_chain) / 1e6, 100 * measure(only_chain) / 1e6 / wall,
             wall - measure(anyk) / 1e6, 100 * (wall - measure(anyk) / 1e6) / wall))
    per = {}
    for s, e, k in rows:
        per.setdefault(k, [0, 0.0]); per[k][0] += 1; per[k][1] += (e - max(s, t0)) / 1e6
    print("kernel time summed (overlapping): " + ", ".join("%s %.0f" % (k, v[1]) for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])[:10]))
    # the chain kernels that run while NO SSV launch is on the device: what the tails consist of
    tail = {}
    for s, e, k in rows:
        if k.startswith("ssv_kernel"):
            continue
        for a, b in subtract([[max(s, t0), e]], ssv):
            tail[k] = tail.get(k, 0.0) + (b - a) / 1e6
    print("kernel time outside the SSV phases (summed): " + ", ".join("%s %.0f" % kv for kv in sorted(tail.items(), key=lambda kv: -kv[1])[:10]))
    nb = int(wall / bucket) + 1
    line = []
    for i in range(nb):
        a, b = t0 + i * bucket * 1e6, t0 + (i + 1) * bucket * 1e6
        f = measure([[max(s, a), min(e, b)] for s, e in ssv if e > a and s < b]) / (bucket * 1e6)
        g = measure([[max(s, a), min(e, b)] for s, e in anyk if e > a and s < b]) / (bucket * 1e6)
        line.append("S" if f > 0.9 else ("s" if f > 0.5 else ("c" if g > 0.5 else ".")))
    print("timeline, %g ms per character (S: SSV > 90 %% of the bucket, s: > 50 %%, c: chain kernels only, .: idle):" % bucket)
    for i in range(0, nb, 100):
        print("  " + "".join(line[i:i + 100]))


if __name__ == "__main__":
    main()
