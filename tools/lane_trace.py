#!/usr/bin/env python
"""Merges the CKM_TRACE=2 points of the library (monotonic ms + worker address) with find()'s per-batch lines into one time line.
usage: lane_trace.py <stderr file> [last_ms] [all]"""
import re
import sys


def main():
    last = float(sys.argv[2]) if len(sys.argv) > 2 else 4000.0
    every = len(sys.argv) > 3
    ev = []
    for l in open(sys.argv[1]).read().splitlines():
        m = re.match(r"ckm-trace (0x\S+)\s+([\d.]+) (.*)", l)
        if m:
            if every or re.search(r"plan ready|chain queued|chain drained|cascade done", m.group(3)):
                ev.append((float(m.group(2)), m.group(1)[-5:], m.group(3)))
            continue
        m = re.match(r"find-trace lane (\d+) batch (\d+) ingest ([\d.]+) search ([\d.]+) .. ([\d.]+)", l)
        if m:
            for g, lab in ((3, "ingest start"), (4, "search start"), (5, "search end")):
                ev.append((float(m.group(g)), "L" + m.group(1), "b%s %s" % (m.group(2), lab)))
            continue
        m = re.match(r"find-trace lane (\d+) batch (\d+) written ([\d.]+)", l)
        if m:
            ev.append((float(m.group(3)), "L" + m.group(1), "b%s written" % m.group(2)))
    ev.sort()
    t0 = ev[-1][0] - last
    for t, w, lab in ev:
        if t >= t0:
            print("%9.1f %-6s %s" % (t - t0, w, lab))


if __name__ == "__main__":
    main()
