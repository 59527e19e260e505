#!/usr/bin/env python
"""Row diff of two domtblout files -- a real `hmmsearch` table against the one this repository writes for the same HMM file and
genes.faa (BASELINE.md leg A; the reference call: checkm/hmmer.py:61-74 with the options of checkm/markerGeneFinder.py:140-142).

The scan-half oracle is a restatement of HMMER that nothing in this image can pin ("parity unpinned"): the
day a box has HMMER on PATH this tool (and tests/test_vs_hmmsearch.py, which skips itself until then) reports, per class:
  identical        every column of the row equal as text
  last_digit       same (target, query, domain) and coordinates; a %6.1f / %9.2g / %4.2f column differs by one unit of its last digit
  coords           same (target, query), different hmm/ali/env coordinates or domain count
  only_hmmsearch   rows HMMER reports and we do not;  only_ours: the opposite
and, when the HMM file is known, DECISION_RELEVANT: the rows among those that would change what CheckM does with them -- CheckM reads
only the printed digits and compares them with the model's GA / TC / NC cutoffs, or with E <= 1e-10 and an aligned fraction >= 0.7 where a
model has none, after the pseudogene cut at 0.3 (checkm/resultsParser.py:340-377 vetHit): a paired row whose two versions vet differently,
a row only one side reports that passes.  The known deviations of the restatement (D1 bias-filter rescaling, D2
optimal-accuracy gating, D4 libm expf, D5 summation order) predict "last_digit" rows; this count says how many of them matter.
Usage:  diff_vs_hmmsearch.py <hmm file> <genes.faa> [--keep]      (runs hmmsearch and the MI355X scan, prints a JSON summary)
        diff_vs_hmmsearch.py --tables <hmmsearch.tbl> <ours.tbl> [--hmm <hmm file>]
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


PSEUDOGENE_LENGTH, E_VALUE, LENGTH = 0.3, 1e-10, 0.7          # checkm/defaultValues.py:36-38


def models_of(hmm_file):
    """{query accession as CheckM keys it: (acc, ga, tc, nc)} -- the STICKY header view CheckM vets with (hmmerModelParser.py:54-83)."""
    from checkm_amd.hmmerModelParser import models_dict, read_headers
    return {k: (m.acc, m.ga, m.tc, m.nc) for k, m in models_dict(read_headers(hmm_file)).items()}


def vet(row, models, ignore_thresholds=False):
    """vetHit on one domtblout row's PRINTED columns (checkm/resultsParser.py:340-377; the query accession is the name when the
    accession column is '-', checkm/hmmer.py:263-266)."""
    acc = row[4] if row[4] != '-' else row[3]
    if acc not in models:
        return None
    macc, ga, tc, nc = models[acc]
    qlen, full_e, full_sc, dom_sc, ali_from, ali_to = float(row[5]), float(row[6]), float(row[7]), float(row[13]), int(row[17]), int(row[18])
    frac = float(ali_to - ali_from) / qlen
    if frac < PSEUDOGENE_LENGTH:
        return False
    if nc is not None and not ignore_thresholds and 'TIGR' in macc:
        return nc[0] <= full_sc and nc[1] <= dom_sc
    if ga is not None and not ignore_thresholds:
        return ga[0] <= full_sc and ga[1] <= dom_sc
    if tc is not None and not ignore_thresholds:
        return tc[0] <= full_sc and tc[1] <= dom_sc
    if nc is not None and not ignore_thresholds:
        return nc[0] <= full_sc and nc[1] <= dom_sc
    if full_e > E_VALUE:
        return False
    return frac >= LENGTH


def diff_tables(theirs, ours, hmm_file=None):
    A, B = read_rows(theirs), read_rows(ours)
    models = models_of(hmm_file) if hmm_file else None
    ka = {}
    for r in A:
        ka.setdefault((r[0], r[3]), []).append(r)
    kb = {}
    for r in B:
        kb.setdefault((r[0], r[3]), []).append(r)
    out = {"rows_hmmsearch": len(A), "rows_ours": len(B), "identical": 0, "last_digit": 0, "coords": 0, "only_hmmsearch": 0, "only_ours": 0, "examples": []}
    if models is not None:
        out.update({"decision_relevant": 0, "decision_relevant_last_digit": 0, "decision_examples": []})

    def relevant(x, y, cls):
        """x / y: the row as hmmsearch / we print it (None: that side has no such row)"""
        if models is None:
            return
        vx = vet(x, models) if x is not None else False
        vy = vet(y, models) if y is not None else False
        if bool(vx) != bool(vy):
            out["decision_relevant"] += 1
            if cls == "last_digit":
                out["decision_relevant_last_digit"] += 1
            if len(out["decision_examples"]) < 5:
                out["decision_examples"].append({"class": cls, "theirs": x, "ours": y, "vet_theirs": vx, "vet_ours": vy})
    for k, ra in ka.items():
        rb = kb.get(k)
        if rb is None:
            out["only_hmmsearch"] += len(ra)
            for x in ra:
                relevant(x, None, "only_hmmsearch")
            if len(out["examples"]) < 5:
                out["examples"].append({"only_hmmsearch": ra[0]})
            continue
        if len(ra) != len(rb):
            out["coords"] += max(len(ra), len(rb))
            # (CheckM keeps the best-scoring passing domain of a marker on an ORF: what matters is whether ANY domain passes on each side)
            if models is not None and any(vet(x, models) for x in ra) != any(vet(y, models) for y in rb):
                relevant(ra[0], None, "coords") if any(vet(x, models) for x in ra) else relevant(None, rb[0], "coords")
            if len(out["examples"]) < 5:
                out["examples"].append({"domain_count": [len(ra), len(rb)], "pair": list(k)})
            continue
        for x, y in zip(ra, rb):
            if x == y:
                out["identical"] += 1
            elif x[15:21] != y[15:21] or x[9:11] != y[9:11] or x[2] != y[2] or x[5] != y[5]:
                out["coords"] += 1
                relevant(x, y, "coords")
                if len(out["examples"]) < 5:
                    out["examples"].append({"theirs": x, "ours": y})
            elif all(_close(NUM[c], x[c], y[c]) for c in NUM):
                out["last_digit"] += 1
                relevant(x, y, "last_digit")
            else:
                out["coords"] += 1
                relevant(x, y, "coords")
                if len(out["examples"]) < 5:
                    out["examples"].append({"theirs": x, "ours": y})
    for k, rb in kb.items():
        if k not in ka:
            out["only_ours"] += len(rb)
            for y in rb:
                relevant(None, y, "only_ours")
            if len(out["examples"]) < 5:
                out["examples"].append({"only_ours": rb[0]})
    return out


from checkm_amd.parity import KNOWN_DEVIATIONS  # noqa: E402  (the restatement's declared deviations from HMMER)


def merge(diffs):
    keys = ["rows_hmmsearch", "rows_ours", "identical", "last_digit", "coords", "only_hmmsearch", "only_ours"]
    if diffs and all("decision_relevant" in d for d in diffs):
        keys += ["decision_relevant", "decision_relevant_last_digit"]
    tot = {k: 0 for k in keys}
    ex, dex = [], []
    for d in diffs:
        for k in tot:
            tot[k] += d[k]
        ex += d["examples"]
        dex += d.get("decision_examples", [])
    tot["examples"] = ex[:5]
    if "decision_relevant" in tot:
        tot["decision_examples"] = dex[:5]
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
    d = diff_tables(theirs, ours, hmm)
    d["known_deviations"] = KNOWN_DEVIATIONS
    d["hmmsearch"] = subprocess.run([exe, "-h"], stdout=subprocess.PIPE).stdout.decode(errors="replace").split("\n")[1].strip("# ")
    if not keep:
        shutil.rmtree(tmp)
    else:
        d["dir"] = tmp
    return d


if __name__ == "__main__":
    if len(sys.argv) >= 4 and sys.argv[1] == "--tables":
        hmm = sys.argv[sys.argv.index("--hmm") + 1] if "--hmm" in sys.argv else None
        print(json.dumps(diff_tables(sys.argv[2], sys.argv[3], hmm), indent=1))
    elif len(sys.argv) >= 3:
        print(json.dumps(run(sys.argv[1], sys.argv[2], "--keep" in sys.argv), indent=1))
    else:
        raise SystemExit(__doc__)
