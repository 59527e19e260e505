#!/usr/bin/env python
"""Host-only timing of the marker-file work of a lineage_wf run (no GPU needed): builds the synthetic lineage world, writes a Lineage
marker file for N bins and times MarkerSetParser.markerAccessionsForBins (called by MarkerGeneFinder.find for the analyze pass) and
MarkerSetParser.getMarkerSets (called by qa).  usage: python tools/prof_marker_files.py [nbins=1000] [workdir]"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from synthdata import synth_lineage as sl                    # noqa: E402
from checkm_amd.defaultValues import DefaultValues            # noqa: E402
from checkm_amd.markerSets import MarkerSetParser, _parse_marker_sets   # noqa: E402


def main():
    nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    wd = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="prof_marker_")
    data = os.path.join(wd, "lineage_data")
    t0 = time.perf_counter()
    w = sl.World(data) if not os.path.exists(os.path.join(data, "hmms", "checkm.hmm")) else sl.World(data, write=False)
    print("world: %.1f s (%s)" % (time.perf_counter() - t0, wd))
    DefaultValues.set_data_root(data)
    binIds = ["bin_%04d" % b for b in range(nb)]
    w.write_marker_files(wd, binIds)
    lin = os.path.join(wd, "lineage.ms")
    print("lineage.ms: %.1f MB for %d bins" % (os.path.getsize(lin) / 1e6, nb))
    msp = MarkerSetParser(8)
    t0 = time.perf_counter()
    msp.markerAccessionsForBins(binIds, lin)
    print("markerAccessionsForBins: %.3f s   %s" % (time.perf_counter() - t0, _parse_marker_sets.cache_info()))
    os.makedirs(os.path.join(wd, "out", "storage"), exist_ok=True)
    pr = cProfile.Profile()
    pr.enable()
    t0 = time.perf_counter()
    msp.getMarkerSets(os.path.join(wd, "out"), binIds, lin)
    t1 = time.perf_counter()
    pr.disable()
    print("getMarkerSets: %.3f s" % (t1 - t0))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(8)


if __name__ == "__main__":
    main()
