#!/usr/bin/env python
"""Goldens for the QA output formats 1-9 in tab mode (resultsParser.py:219-319, 680-966), produced by the REFERENCE's own
ResultsParser / ResultsManager (imported read-only from /root/reference) on the tables of tests/golden/reduce_cases.json.
Writes tests/golden/summary_cases.json (committed).  Run here only."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = tempfile.mkdtemp(prefix="ckm_refdata_")
os.makedirs(os.path.join(DATA, "pfam"))
os.environ["CHECKM_DATA_PATH"] = DATA
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")

from checkm.hmmerModelParser import HmmModel  # noqa: E402
from checkm.markerSets import MarkerSet, BinMarkerSets  # noqa: E402
from checkm.resultsParser import ResultsManager, ResultsParser  # noqa: E402
from checkm.defaultValues import DefaultValues  # noqa: E402


class FakeAAI(object):
    def __init__(self, d):
        self.aaiMeanBinHetero = d


def main():
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "reduce_cases.json")))["cases"]
    out = []
    for ci in (0, 3, 7, 12, 20, 41):
        case = cases[ci]
        open(DefaultValues.PFAM_CLAN_FILE, "w").write(case["pfam_dat"])
        work = tempfile.mkdtemp(prefix="ckm_sum_")
        binIds = ["binB", "binA"]
        tabs = {"binB": case["domtblout"], "binA": cases[(ci + 1) % len(cases)]["domtblout"] if False else case["domtblout"]}
        models = {}
        for m in case["models"]:
            hm = HmmModel({"name": m["name"], "acc": m["acc"], "leng": m["leng"]})
            hm.ga = tuple(m["ga"]) if m["ga"] else None
            hm.tc = tuple(m["tc"]) if m["tc"] else None
            hm.nc = tuple(m["nc"]) if m["nc"] else None
            models[m["acc"]] = hm
        rp = ResultsParser({b: models for b in binIds})
        bms = {}
        stats = {}
        genes = {}
        for k, b in enumerate(binIds):
            stats[b] = {"Genome size": 2000000 + 17 * k, "# ambiguous bases": 3 * k, "# scaffolds": 40 + k, "# contigs": 44 + k, "N50 (scaffolds)": 81234 + k,
                        "N50 (contigs)": 70001, "Mean scaffold length": 50000.6 + k, "Mean contig length": 45454.5, "Longest scaffold": 300123,
                        "Longest contig": 250321, "GC": 0.51234 + 0.01 * k, "GC std": 0.02345, "Coding density": 0.9012, "Translation table": 11,
                        "# predicted genes": 1987 + k}
            rm = ResultsManager(b, models, False, DefaultValues.E_VAL, DefaultValues.LENGTH, False, stats[b])
            path = os.path.join(work, b + ".txt")
            open(path, "w").write(tabs[b])
            rp.parseHmmerResults(path, rm, k == 1)          # binA without the adjacency correction: different hit lists
            rp.results[b] = rm
            s = BinMarkerSets(b, BinMarkerSets.TAXONOMIC_MARKER_SET)
            s.addMarkerSet(MarkerSet(7 + k, "k__Bacteria;p__Test", 100 + k, [set(x) for x in case["marker_sets"]]))
            s.addMarkerSet(MarkerSet(0, "root", 5000, [set(x) for x in case["marker_sets"][:1]]))
            bms[b] = s
            # a genes.faa with prodigal-style headers for every target the tables mention (format 9)
            names = []
            for line in tabs[b].split("\n"):
                if line and not line.startswith("#"):
                    n = line.split()[0]
                    if n not in names:
                        names.append(n)
            os.makedirs(os.path.join(work, "bins", b))
            txt = ""
            for j, n in enumerate(names):
                txt += ">%s # %d # %d # %d # ID=1_%d;partial=00\nMKV%sLLA*\n" % (n, 100 + 10 * j, 400 + 10 * j, 1 if j % 2 else -1, j, "AC" * (j % 5))
            open(os.path.join(work, "bins", b, "genes.faa"), "w").write(txt)
            genes[b] = txt
        aai = FakeAAI({"binA": 12.5})
        outputs = {}
        for fmt in range(1, 10):
            for indiv in (False, True):
                of = os.path.join(work, "out_%d_%d.txt" % (fmt, indiv))
                try:
                    rp.printSummary(fmt, aai, bms, indiv, None, True, of, work)
                    outputs["%d_%d" % (fmt, int(indiv))] = open(of).read()
                except Exception as e:                      # e.g. format 9 on an ORF name without '_<n>'
                    sys.stdout = sys.__stdout__
                    outputs["%d_%d" % (fmt, int(indiv))] = {"raises": type(e).__name__}
        os.makedirs(os.path.join(work, "storage"))
        rp.cacheResults(work, bms, False)
        caches = {n: open(os.path.join(work, "storage", n)).read() for n in (DefaultValues.BIN_STATS_EXT_OUT, DefaultValues.MARKER_GENE_STATS)}
        summaries = {b: {str(f): repr(rp.results[b].getSummary(bms[b], False, outputFormat=f)) for f in (1, 2, 5, 6, 7, 8)} for b in binIds}
        out.append({"caches": caches, "summaries": summaries, "reduce_case": ci, "bin_stats": stats, "genes_faa": genes, "marker_sets": case["marker_sets"], "outputs": outputs})
    json.dump({"generator": "tools/gen_summary_golden.py", "cases": out}, open(os.path.join(ROOT, "tests", "golden", "summary_cases.json"), "w"), indent=1)
    print("wrote summary_cases.json", len(out), "cases;", sum(len(v) for c in out for v in c["outputs"].values() if isinstance(v, str)), "bytes of table text")


if __name__ == "__main__":
    main()
