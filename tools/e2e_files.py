#!/usr/bin/env python
"""End-to-end timing of the drop-in classes on FILES (cfg2 shape: 43 profiles x N bins x 2000 called genes per bin):
MarkerGeneFinder.find (reads the genes.faa files, scans, writes one domtblout per bin) -> ResultsParser.analyseResults
(QA from the resident hits) -> the same from the domtblout text (what a later `checkm qa` does).  Prints one JSON line.
    python tools/e2e_files.py [--bins 100] [--orfs 2000]
Needs a GPU.  The reference's equivalent is one hmmsearch process per bin plus the Python parsers."""
import argparse
import json
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bins", type=int, default=100)
    ap.add_argument("--orfs", type=int, default=2000)
    args = ap.parse_args()
    from synthdata import synth
    from checkm_amd.defaultValues import DefaultValues
    from checkm_amd.markerGeneFinder import MarkerGeneFinder, release_scan
    from checkm_amd.markerSets import MarkerSetParser
    from checkm_amd.resultsParser import ResultsParser
    work = tempfile.mkdtemp(prefix="ckm_e2e_")
    profs = synth.cpr43_profiles()
    hmm = os.path.join(work, "cpr43_synth.hmm")
    synth.write_hmm(hmm, profs)
    root = os.path.join(work, "data", "pfam")
    os.makedirs(root)
    with open(os.path.join(root, "Pfam-A.hmm.dat"), "w") as f:
        for p in profs:
            if p.acc.startswith("PF"):
                f.write("# STOCKHOLM 1.0\n#=GF ID   %s\n#=GF AC   %s\n//\n" % (p.name, p.acc))
    DefaultValues.set_data_root(os.path.join(work, "data"))
    binfiles = []
    for b in range(args.bins):
        recs = synth.make_bin(profs, 1000 + b, n_orfs=args.orfs)
        path = os.path.join(work, "bin_%04d.faa" % b)
        synth.write_fasta(path, recs)
        binfiles.append(path)
    out = os.path.join(work, "out")
    os.makedirs(os.path.join(out, "storage"))
    t = {}
    for rep in ("first", "second"):           # the first call pays context creation and the profile upload
        t0 = time.perf_counter()
        models = MarkerGeneFinder(8).find(binfiles, out, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, hmm, False, False, True)
        t["find_%s_s" % rep] = time.perf_counter() - t0
        if rep == "first":
            release_scan(out)
    with open(os.path.join(out, "storage", DefaultValues.BIN_STATS_OUT), "w") as f:
        for b in models:
            f.write("%s\t%s\n" % (b, repr({"GC": 0.5, "Genome size": 2000000})))
    msp = MarkerSetParser()
    sets = msp.getMarkerSets(out, list(models), hmm)
    t0 = time.perf_counter()
    rp = ResultsParser(models)
    rp.analyseResults(out, DefaultValues.BIN_STATS_OUT, DefaultValues.HMMER_TABLE_OUT)
    t["qa_resident_s"] = time.perf_counter() - t0
    rows_a = {b: rp.results[b].geneCountsForSelectedMarkerSet(sets[b], False) for b in models}
    release_scan(out)
    t0 = time.perf_counter()
    rp2 = ResultsParser(models)
    rp2.analyseResults(out, DefaultValues.BIN_STATS_OUT, DefaultValues.HMMER_TABLE_OUT)
    t["qa_from_text_s"] = time.perf_counter() - t0
    rows_b = {b: rp2.results[b].geneCountsForSelectedMarkerSet(sets[b], False) for b in models}
    assert rows_a == rows_b, "QA from resident hits and from the domtblout text differ"
    nrows = sum(1 for b in models for line in open(os.path.join(out, "bins", b, DefaultValues.HMMER_TABLE_OUT)) if line[:1] != "#")
    wall = t["find_second_s"] + t["qa_resident_s"]
    print(json.dumps({"workload": "43 profiles x %d bins x %d called genes, from genes.faa files to the QA table" % (args.bins, args.orfs),
                      "bins": args.bins, "domtblout_rows": nrows, **{k: round(v, 3) for k, v in t.items()},
                      "bins_per_hour_files": round(args.bins / wall * 3600.0), "mean_completeness": round(sum(r[6] for r in rows_a.values()) / len(rows_a), 2)}))


if __name__ == "__main__":
    main()
