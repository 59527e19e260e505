#!/usr/bin/env python
"""Where the device's time goes in a rocprofv3 kernel trace (csv: *_kernel_trace.csv): the union of the SSV launches' intervals (the
throughput-bound phase: the VALUs issue for every CU), the time only chain kernels run (latency-bound: one wavefront per work item, the
device mostly idle), and the time nothing runs (host gaps).  Prints the totals for the window and a coarse timeline.
usage: occupancy_timeline.py <kernel_trace.csv> [bucket_ms] [skip_first_ms | last-step]
"last-step": the window is the last step of a bench.py --config cfg3 run -- from the end of the previous step's count_sets_kernel (the
QA table of the warm pass) to the end of the last one."""
import bisect
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
    last_step = len(sys.argv) > 3 and sys.argv[3] == "last-step"
    skip = float(sys.argv[3]) if len(sys.argv) > 3 and not last_step else 0.0
    rows = []
    for r in csv.DictReader(open(path)):
        m = re.search(r'ckm::([a-z0-9_]+kernel(?:_h)?)', r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else "other"))
    t0 = min(r[0] for r in rows) + skip * 1e6
    t1 = max(r[1] for r in rows)
    if last_step:
        ends = sorted(e for _s, e, k in rows if k == "count_sets_kernel")
        clusters = [[ends[0], ends[0]]]
        for e in ends[1:]:
            if e - clusters[-1][1] > 5e8:
                clusters.append([e, e])
            clusters[-1][1] = e
        if len(clusters) >= 2:
            t0, t1 = clusters[-2][1], clusters[-1][1]
    rows = [(s, min(e, t1), k) for s, e, k in rows if e > t0 and s < t1]
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
    ssv_starts = [a for a, _b in ssv]
    for s, e, k in rows:
        if k.startswith("ssv_kernel"):
            continue
        lo = max(0, bisect.bisect_left(ssv_starts, s) - 1)
        for a, b in subtract([[max(s, t0), e]], ssv[lo:bisect.bisect_right(ssv_starts, e) + 1]):
            tail[k] = tail.get(k, 0.0) + (b - a) / 1e6
    print("kernel time outside the SSV phases (summed): " + ", ".join("%s %.0f" % kv for kv in sorted(tail.items(), key=lambda kv: -kv[1])[:10]))
    idle = subtract([[t0, t1]], [list(x) for x in anyk])
    nossv = subtract([[t0, t1]], [list(x) for x in ssv])
    print("longest stretches with nothing on the device (ms at ms): " + ", ".join("%.0f @ %.0f" % ((b - a) / 1e6, (a - t0) / 1e6) for a, b in sorted(idle, key=lambda ab: ab[0] - ab[1])[:12]))
    print("stretches without an SSV launch: %d, %.1f ms in all; longest (ms at ms): " % (len(nossv), measure(nossv) / 1e6) + ", ".join("%.0f @ %.0f" % ((b - a) / 1e6, (a - t0) / 1e6) for a, b in sorted(nossv, key=lambda ab: ab[0] - ab[1])[:12]))
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
