"""Driver of tests/test_gpu_workers.py: ONE process runs the unmodified call sequence of `checkm analyze` + `checkm qa`
(MarkerGeneFinder.find -> ResultsParser.analyseResults -> printSummary -> cacheResults); with CKM_GPUS naming several devices find()
itself spawns one worker per device (checkm_amd/workers.py).  argv: <workdir> <marker file> <tag> <formats, comma separated>."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    work, marker, tag, fmts = sys.argv[1], sys.argv[2], sys.argv[3], [int(x) for x in sys.argv[4].split(",")]
    from checkm_amd.defaultValues import DefaultValues
    from checkm_amd.markerGeneFinder import MarkerGeneFinder, SCAN_CACHE, release_scan
    from checkm_amd.markerSets import MarkerSetParser
    from checkm_amd.resultsParser import ResultsParser
    DefaultValues.set_data_root(os.path.join(work, "data"))
    files = sorted(os.path.join(work, f) for f in os.listdir(work) if f.endswith(".faa"))
    out = os.path.join(work, "out_" + tag)
    models = MarkerGeneFinder(2).find(files, out, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, marker, False, False, True)
    ent = SCAN_CACHE[(os.path.abspath(out), DefaultValues.HMMER_TABLE_OUT)]
    with open(os.path.join(work, "mode_%s.txt" % tag), "w") as f:
        f.write("workers %d\n" % (len(ent["pool"].devs) if ent.get("pool") is not None else 0))
        if ent.get("pool") is not None:
            f.write("owners %s\n" % " ".join("%s:%d" % kv for kv in sorted(ent["owners"].items())))
    binIds = sorted(models)
    os.makedirs(os.path.join(out, "storage"), exist_ok=True)
    with open(os.path.join(out, "storage", DefaultValues.BIN_STATS_OUT), "w") as f:
        for k, b in enumerate(binIds):
            f.write("%s\t%s\n" % (b, repr({"GC": 0.5, "Genome size": 1000 + k, "# ambiguous bases": 0, "# scaffolds": 3, "# contigs": 3, "N50 (scaffolds)": 10,
                                             "N50 (contigs)": 10, "Mean scaffold length": 9.0, "Mean contig length": 9.0, "Longest scaffold": 20,
                                             "Longest contig": 20, "GC std": 0.01, "Coding density": 0.9, "Translation table": 11, "# predicted genes": 60})))
    sets = MarkerSetParser().getMarkerSets(out, binIds, marker)
    rp = ResultsParser(models)
    rp.analyseResults(out, DefaultValues.BIN_STATS_OUT, DefaultValues.HMMER_TABLE_OUT)
    for fmt in fmts:
        rp.printSummary(fmt, None, sets, False, None, True, os.path.join(work, "table_%s_fmt%d.tsv" % (tag, fmt)), None)
    rp.cacheResults(out, sets, False)
    release_scan()


if __name__ == "__main__":
    main()
