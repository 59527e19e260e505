"""Worker of tests/test_gpu_dist.py: one rank of a `torch.distributed.run` launch of the PRODUCT path
(MarkerGeneFinder.find -> ResultsParser.analyseResults -> printSummary) over a shared output directory.
argv: <workdir> <marker file> <format>.  Rank 0 writes the table to <workdir>/table_world<N>.tsv."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    work, marker, fmt = sys.argv[1], sys.argv[2], int(sys.argv[3])
    from checkm_amd import dist as cdist
    from checkm_amd.defaultValues import DefaultValues
    from checkm_amd.markerGeneFinder import MarkerGeneFinder
    from checkm_amd.markerSets import MarkerSetParser
    from checkm_amd.resultsParser import ResultsParser
    rank, _local, world = cdist.env_rank()
    DefaultValues.set_data_root(os.path.join(work, "data"))
    files = sorted(os.path.join(work, f) for f in os.listdir(work) if f.endswith(".faa"))
    out = os.path.join(work, "out_world%d" % world)
    models = MarkerGeneFinder(2).find(files, out, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, marker, False, False, True)
    binIds = sorted(models)
    if rank == 0:
        os.makedirs(os.path.join(out, "storage"), exist_ok=True)
        with open(os.path.join(out, "storage", DefaultValues.BIN_STATS_OUT), "w") as f:
            for k, b in enumerate(binIds):
                f.write("%s\t%s\n" % (b, repr({"GC": 0.5, "Genome size": 1000 + k})))
    cdist.barrier()
    msp = MarkerSetParser()
    sets = msp.getMarkerSets(out, binIds, marker)
    rp = ResultsParser(models)
    rp.analyseResults(out, DefaultValues.BIN_STATS_OUT, DefaultValues.HMMER_TABLE_OUT)
    owned = sorted(b for b in rp.results)
    rp.printSummary(fmt, None, sets, False, None, True, os.path.join(work, "table_world%d_fmt%d.tsv" % (world, fmt)) if rank == 0 else None, None)
    with open(os.path.join(work, "owned_world%d_rank%d.txt" % (world, rank)), "w") as f:
        f.write("\n".join(owned) + "\n")
    cdist.barrier()
    cdist.shutdown()


if __name__ == "__main__":
    main()
