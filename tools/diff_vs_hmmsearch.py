#!/usr/bin/env python
"""Row diff of two domtblout files -- a real `hmmsearch` table against the one this repository writes for the same HMM file and
genes.faa (BASELINE.md leg A; the reference call: checkm/hmmer.py:61-74 with the options of checkm/markerGeneFinder.py:140-142).

The scan-half oracle is a restatement of HMMER that nothing in this image can pin ("parity unpinned"): the
day a box has HMMER on PATH this tool (and tests/test_vs_hmmsearch.py, which skips itself until then) reports, per class:
  identical        every column of the row equal as text
  last_digit       same (target, query, domain) and coordinates; a %6.1f / %9.2g / %4.2f column differs by one unit of its last digit
  coords           same (target, query), different hmm/ali/env coordinates or domain count
  only_hmmsearch   rows HMMER reports and we do not;  only_ours: the opposite
Usage:  diff_vs_hmmsearch.py <hmm file> <genes.faa> [--keep]      (runs hmmsearch and the MI355X scan, prints a JSON summary)
        diff_vs_hmmsearch.py --tables <hmmsearch.tbl> <ours.tbl>
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

NUM = {6: "g", 7: "f1", 8: "f1", 11: "g", 12: "g", 13: "f1", 14: "f1", 21: "f2"}


def read_rows(path):
    rows = []
    with open(path) as f:
        for line in f:
            if not line.strip():
                break
            if line[0] == '#':
                continue
            t = line.split()
            rows.append(t[:22])
    return rows


def _close(kind, a, b):
    if a == b:
        return True
    try:
        x, y = float(a), float(b)
    except ValueError:
        return False
    if kind == "f1":
        return abs(x - y) <= 0.1000001
    if kind == "f2":
        return abs(x - y) <= 0.0100001
    if x == 0 or y == 0:
        return abs(x - y) < 1e-300
    return abs(x - y) <= 0.11 * 10 ** (int(__import__("math").floor(__import__("math").log10(max(abs(x), abs(y))))) - 0)   # one unit of the 2nd significant digit


def diff_tables(theirs, ours):
    A, B = read_rows(theirs), read_rows(ours)
    ka = {}
    for r in A:
        ka.setdefault((r[0], r[3]), []).append(r)
    kb = {}
    for r in B:
        kb.setdefault((r[0], r[3]), []).append(r)
    out = {"rows_hmmsearch": len(A), "rows_ours": len(B), "identical": 0, "last_digit": 0, "coords": 0, "only_hmmsearch": 0, "only_ours": 0, "examples": []}
    for k, ra in ka.items():
        rb = kb.get(k)
        if rb is None:
            out["only_hmmsearch"] += len(ra)
            if len(out["examples"]) < 5:
                out["examples"].append({"only_hmmsearch": ra[0]})
            continue
        if len(ra) != len(rb):
            out["coords"] += max(len(ra), len(rb))
            if len(out["examples"]) < 5:
                out["examples"].append({"domain_count": [len(ra), len(rb)], "pair": list(k)})
            continue
        for x, y in zip(ra, rb):
            if x == y:
                out["identical"] += 1
            elif x[15:21] != y[15:21] or x[9:11] != y[9:11] or x[2] != y[2] or x[5] != y[5]:
                out["coords"] += 1
                if len(out["examples"]) < 5:
                    out["examples"].append({"theirs": x, "ours": y})
            elif all(_close(NUM[c], x[c], y[c]) for c in NUM):
                out["last_digit"] += 1
            else:
                out["coords"] += 1
                if len(out["examples"]) < 5:
                    out["examples"].append({"theirs": x, "ours": y})
    for k, rb in kb.items():
        if k not in ka:
            out["only_ours"] += len(rb)
            if len(out["examples"]) < 5:
                out["examples"].append({"only_ours": rb[0]})
    return out


def merge(diffs):
    tot = {k: 0 for k in ("rows_hmmsearch", "rows_ours", "identical", "last_digit", "coords", "only_hmmsearch", "only_ours")}
    ex = []
    for d in diffs:
        for k in tot:
            tot[k] += d[k]
        ex += d["examples"]
    tot["examples"] = ex[:5]
    return tot


def run(hmm, faa, keep=False):
    exe = shutil.which("hmmsearch")
    if exe is None:
        raise SystemExit("hmmsearch is not on PATH")
    tmp = tempfile.mkdtemp(prefix="ckm_vs_hmmer_")
    theirs, ours = os.path.join(tmp, "hmmsearch.tbl"), os.path.join(tmp, "ours.tbl")
    subprocess.check_call([exe, "--domtblout", theirs, "--noali", "--notextw", "-E", "0.1", "--domE", "0.1", "--cpu", "1", hmm, faa], stdout=subprocess.DEVNULL)
    from checkm_amd.markerGeneFinder import scan_files
    scan_files(hmm, [faa], [ours])
    d = diff_tables(theirs, ours)
    d["hmmsearch"] = subprocess.run([exe, "-h"], stdout=subprocess.PIPE).stdout.decode(errors="replace").split("\n")[1].strip("# ")
    if not keep:
        shutil.rmtree(tmp)
    else:
        d["dir"] = tmp
    return d


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--tables":
        print(json.dumps(diff_tables(sys.argv[2], sys.argv[3]), indent=1))
    elif len(sys.argv) >= 3:
        print(json.dumps(run(sys.argv[1], sys.argv[2], "--keep" in sys.argv), indent=1))
    else:
        raise SystemExit(__doc__)
