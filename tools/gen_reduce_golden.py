#!/usr/bin/env python
"""Generates tests/golden/reduce_cases.json by running the REFERENCE's own classes (imported
read-only from /root/reference) on random synthetic domtblout tables, Pfam clan files and marker
sets.  Pins R1-R7 of SURVEY.md section 8a, quirks included.  Run here only (the GPU box has no
/root/reference); the JSON it writes is committed.

  PYTHONDONTWRITEBYTECODE=1 python tools/gen_reduce_golden.py
"""
import json
import os
import random
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = tempfile.mkdtemp(prefix="ckm_refdata_")
os.makedirs(os.path.join(DATA, "pfam"))
os.environ["CHECKM_DATA_PATH"] = DATA
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from checkm.hmmerModelParser import HmmModel  # noqa: E402
from checkm.markerSets import MarkerSet  # noqa: E402
from checkm.resultsParser import ResultsManager, ResultsParser  # noqa: E402
from checkm.defaultValues import DefaultValues  # noqa: E402

PFAM_FILE = DefaultValues.PFAM_CLAN_FILE


def rand_models(rng):
    models = []
    n = rng.randint(4, 14)
    used = set()
    for i in range(n):
        kind = rng.random()
        if kind < 0.5:
            acc = "PF%05d.%d" % (rng.randint(1, 99999), rng.randint(1, 30))
            name = "Pfam_%d" % i
        elif kind < 0.85:
            acc = "TIGR%05d" % rng.randint(1, 4000)
            name = acc
        else:
            acc = None
            name = "custom_%d" % i
        if (acc or name) in used:
            continue
        used.add(acc or name)
        m = {"name": name, "acc": acc or name, "leng": rng.randint(40, 600), "ga": None, "tc": None, "nc": None}
        r = rng.random()
        if r < 0.75:
            base = rng.choice([10.0, 20.0, 22.5, 25.0, 27.3, 40.0])
            if rng.random() < 0.7:
                m["ga"] = [base, base - rng.choice([0, 0.5, 3.0])]
            if rng.random() < 0.6:
                m["tc"] = [base + 2.0, base + 1.0]
            if rng.random() < 0.6:
                m["nc"] = [base - 1.0, base - 2.5]
        models.append(m)
    return models


def rand_pfam_dat(rng, models):
    pf = [m for m in models if m["acc"].startswith("PF")]
    lines = []
    clans = ["CL%04d" % rng.randint(1, 5) for _ in range(3)]
    ids = {}
    for m in pf:
        ids[m["acc"]] = "ID_" + m["acc"].replace(".", "_")
    for m in pf:
        lines.append("# STOCKHOLM 1.0")
        lines.append("#=GF ID   %s" % ids[m["acc"]])
        lines.append("#=GF AC   %s" % m["acc"])
        if rng.random() < 0.6:
            lines.append("#=GF CL   %s" % rng.choice(clans))
        if len(pf) > 1 and rng.random() < 0.2:
            other = rng.choice([x for x in pf if x is not m])
            lines.append("#=GF NE   %s;" % ids[other["acc"]] if False else "#=GF NE   %s" % ids[other["acc"]])
        lines.append("//")
    return "\n".join(lines) + "\n"


def fmt_row(t):
    return ("%-20s %-10s %5d %-20s %-10s %5d %9.2g %6.1f %5.1f %3d %3d %9.2g %9.2g %6.1f %5.1f %5d %5d %5d %5d %5d %5d %4.2f %s"
            % tuple(t))


def rand_domtblout(rng, models):
    contigs = ["NODE_%d_length_%d_cov_%.2f" % (i, rng.randint(1000, 90000), rng.uniform(1, 50)) for i in range(1, 5)] + ["k141_7", "scaffold"]
    rows = []
    weird = ["gene-A", "orf_x", "c1_1a", "plain"]
    for m in models:
        nh = rng.choice([0, 1, 1, 2, 3, 4, 7])
        base_orf = rng.randint(1, 30)
        for h in range(nh):
            r = rng.random()
            if r < 0.08:
                tname = rng.choice(weird)
            else:
                c = rng.choice(contigs)
                num = base_orf + rng.choice([0, 1, 1, 2, 5, 9]) if rng.random() < 0.7 else rng.randint(1, 60)
                tname = "%s_%d" % (c, num)
            tlen = rng.randint(60, 900)
            ndom = rng.choice([1, 1, 1, 2, 3])
            full_score = round(rng.uniform(5, 300), 1)
            full_e = float("%9.2g" % (10 ** rng.uniform(-120, -1)))
            for d in range(ndom):
                dom_score = round(min(full_score, rng.uniform(3, full_score + 1)), 1)
                hf = rng.randint(1, max(1, m["leng"] // 2))
                ht = rng.randint(hf, m["leng"])
                af = rng.randint(1, max(1, tlen // 2))
                at = min(tlen, af + rng.randint(0, max(1, int(m["leng"] * rng.uniform(0.1, 1.3)))))
                ef, et = max(1, af - rng.randint(0, 5)), min(tlen, at + rng.randint(0, 5))
                ie = float("%9.2g" % (full_e * 10 ** rng.uniform(0, 3)))
                rows.append([tname, "-", tlen, m["name"], m["acc"] if m["acc"] != m["name"] else "-", m["leng"], full_e, full_score,
                             round(rng.uniform(0, 3), 1), d + 1, ndom, ie / 10, ie, dom_score, round(rng.uniform(0, 2), 1),
                             hf, ht, af, at, ef, et, round(rng.uniform(0.5, 1.0), 2),
                             "# %d # %d # 1 # ID=1_1;partial=00" % (rng.randint(1, 5000), rng.randint(5000, 9000))])
    rng.shuffle(rows)
    hdr = "# target name accession tlen query name accession qlen E-value score bias # of c-Evalue i-Evalue score bias from to from to from to acc description\n#--- --- ---\n"
    return hdr + "\n".join(fmt_row(r) for r in rows) + ("\n" if rows else "") + "#\n# [ok]\n"


def rand_marker_sets(rng, models):
    accs = [m["acc"] for m in models] + ["PF99999.1", "TIGR09999"]
    rng.shuffle(accs)
    sets, i = [], 0
    while i < len(accs):
        k = rng.randint(1, 4)
        sets.append(sorted(accs[i:i + k]))
        i += k
    if rng.random() < 0.3 and len(sets) > 1:
        sets[-1] = sorted(set(sets[-1]) | {sets[0][0]})     # a marker listed in two sets
    return sets


FLAG_SETS = [dict(), dict(ignore_thresholds=True), dict(skip_adj=True), dict(skip_pseudogene=True),
             dict(ignore_thresholds=True, evalue=1e-5, length=0.3), dict(skip_adj=True, skip_pseudogene=True)]


def run_reference(models, pfam_text, table_text, marker_sets, fl):
    with open(PFAM_FILE, "w") as f:
        f.write(pfam_text)
    tab = os.path.join(DATA, "table.txt")
    with open(tab, "w") as f:
        f.write(table_text)
    mdict = {}
    for m in models:
        keys = {"name": m["name"], "acc": m["acc"], "leng": m["leng"]}
        hm = HmmModel(keys)
        hm.ga = tuple(m["ga"]) if m["ga"] else None
        hm.tc = tuple(m["tc"]) if m["tc"] else None
        hm.nc = tuple(m["nc"]) if m["nc"] else None
        mdict[m["acc"]] = hm
    rm = ResultsManager("bin", mdict, fl.get("ignore_thresholds", False), fl.get("evalue", DefaultValues.E_VAL),
                        fl.get("length", DefaultValues.LENGTH), fl.get("skip_pseudogene", False))
    rp = ResultsParser({"bin": mdict})
    rp.parseHmmerResults(tab, rm, fl.get("skip_adj", False))
    ms = MarkerSet(0, "k__Bacteria", 10, [set(s) for s in marker_sets])
    view = [[k, [[h.target_name, h.target_length, h.hmm_from, h.hmm_to, h.ali_from, h.ali_to, h.env_from, h.env_to, h.dom_score, h.full_e_value]
                 for h in v]] for k, v in rm.markerHits.items()]
    return {"markerHits": view,
            "geneCounts": rm.geneCounts(ms, rm.markerHits, False),
            "geneCountsIndividual": rm.geneCounts(ms, rm.markerHits, True)}


def main():
    rng = random.Random(20250614)
    cases = []
    for c in range(60):
        models = rand_models(rng)
        pfam = rand_pfam_dat(rng, models)
        table = rand_domtblout(rng, models)
        sets = rand_marker_sets(rng, models)
        runs = []
        for fl in FLAG_SETS:
            runs.append({"flags": fl, "expected": run_reference(models, pfam, table, sets, fl)})
        cases.append({"models": models, "pfam_dat": pfam, "domtblout": table, "marker_sets": sets, "runs": runs})
    # hand-made edge cases: scores EQUAL to the cutoffs, at values float32 cannot hold exactly (25.3, 27.3, 0.7 ...)
    for thr, sc in ((25.3, 25.3), (27.3, 27.3), (27.3, 27.2), (110.7, 110.7), (20.0, 20.0)):
        models = [{"name": "edgeGA", "acc": "PF00001.1", "leng": 100, "ga": [thr, thr], "tc": None, "nc": None},
                  {"name": "TIGR00001", "acc": "TIGR00001", "leng": 100, "ga": None, "tc": [thr + 5, thr + 5], "nc": [thr, thr]}]
        rows = []
        for m in models:
            rows.append(["ctg_1", "-", 200, m["name"], m["acc"], 100, 1e-30, sc, 0.0, 1, 1, 1e-32, 1e-30, sc, 0.0, 1, 100, 5, 95, 1, 100, 0.99, "-"])
            rows.append(["ctg_7", "-", 200, m["name"], m["acc"], 100, 1e-30, sc + 0.1, 0.0, 1, 1, 1e-32, 1e-30, sc - 0.1, 0.0, 1, 100, 5, 95, 1, 100, 0.99, "-"])
        table = "# edge\n" + "\n".join(fmt_row(r) for r in rows) + "\n#\n"
        sets = [["PF00001.1"], ["TIGR00001"]]
        runs = [{"flags": fl, "expected": run_reference(models, "", table, sets, fl)} for fl in FLAG_SETS]
        cases.append({"models": models, "pfam_dat": "", "domtblout": table, "marker_sets": sets, "runs": runs})
    out = os.path.join(ROOT, "tests", "golden", "reduce_cases.json")
    with open(out, "w") as f:
        json.dump({"generator": "tools/gen_reduce_golden.py", "reference": "Ecogenomics/CheckM v1.2.4 classes imported from /root/reference",
                   "cases": cases}, f, indent=0)
    n = sum(len(r["expected"]["markerHits"]) for c in cases for r in c["runs"])
    print("wrote %s: %d cases x %d flag sets, %d marker lists" % (out, len(cases), len(FLAG_SETS), n))


if __name__ == "__main__":
    main()
