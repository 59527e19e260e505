"""What the device does during a pass of the gene finder (rocprofv3 --kernel-trace of tools/gene_pass.py): the window is the LAST run of
kernels that no gap of 0.3 s interrupts.  Prints the share of the window with a kernel running, the mean number of kernels running at
once, and per kernel family its launches, summed duration and EXPOSED time -- the time during which only launches of that family ran
(what the pass would lose if the family took no time and nothing else moved).
usage: gene_pass_timeline.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys


def fam(n):
    m = re.search(r"ckm::(?:gene::)?([a-z0-9_]+kernel)(<[^>]*>)?", n)
    if "g_map" in n:
        tags = re.findall(r"\{lambda\(([^)]*)\)#(\d+)\}", n)
        kind = "waves" if "g_map_waves" in n else "map"
        outer = [t for t in tags if "Nodes" in t[0]]
        inner = [t for t in tags if "Nodes" not in t[0]]
        if outer:
            return "%s helper#%s(%s)" % (kind, outer[0][1], "int" if "int" in outer[0][0] else "")
        return "%s #%s" % (kind, inner[0][1] if inner else "?")
    if m:
        return m.group(1) + (m.group(2) or "")
    return n[:40]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), fam(r["Kernel_Name"])))
rows.sort()
# the last gap-free run
cut, end = 0, rows[0][1]
for k, (s, e, _f) in enumerate(rows):
    if s - end > 300e6:
        cut = k
    end = max(end, e)
rows = rows[cut:]
t0, t1 = rows[0][0], max(e for _s, e, _f in rows)
ev = []
for s, e, f in rows:
    ev.append((s, 1, f)); ev.append((e, -1, f))
ev.sort()
live = collections.Counter()
busy = conc = 0.0
exposed = collections.Counter()
hist = collections.Counter()
prev = t0
for t, d, f in ev:
    dt = t - prev
    if dt > 0:
        n = sum(live.values())
        if n:
            busy += dt; conc += dt * n
            fams = [k for k, v in live.items() if v]
            if len(fams) == 1:
                exposed[fams[0]] += dt
        hist[min(n, 16)] += dt
    live[f] += d
    prev = t
W = t1 - t0
print("window %.3f s, %d launches; a kernel running %.1f %% of it; %.1f kernels at once on average while any runs" % (W / 1e9, len(rows), 100 * busy / W, conc / max(busy, 1)))
print("time by number of kernels running: " + "  ".join("%s%d: %.0f%%" % (">=" if k == 16 else "", k, 100 * v / W) for k, v in sorted(hist.items())))
tot = collections.defaultdict(lambda: [0, 0.0])
for s, e, f in rows:
    tot[f][0] += 1; tot[f][1] += e - s
print("\n%-40s %8s %12s %10s %12s" % ("kernel", "launches", "summed ms", "avg us", "exposed ms"))
for f, (n, d) in sorted(tot.items(), key=lambda kv: -kv[1][1])[:24]:
    print("%-40s %8d %12.1f %10.1f %12.1f" % (f, n, d / 1e6, d / n / 1e3, exposed[f] / 1e6))
