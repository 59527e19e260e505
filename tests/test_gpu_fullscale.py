"""Parity where the numbers are quoted: the lineage_wf shape with the DEFAULT batching of MarkerGeneFinder.find --
125 M pairs per search, ramped first batches, two contexts alternating under the baton -- over 200 cfg3-shaped bins, and a slice of
configs[4] with a 10,003-profile database, each followed by the sampled oracle diff bench.py --verify runs (tools/verify_sample.py: the
written table lines of >= 40 models per sampled bin, one of every SSV launch class, against oracle/p7oracle.c; cfg3: the QA row against
oracle/reduce_oracle.py) and by cascade_fallback_lanes == 0 -- no search may have slipped onto the host-driven cascade."""
import os

import numpy as np
import pytest

from checkm_amd import markerGeneFinder as mgf
from synthdata import synth, synth_lineage as sl
from checkm_amd.defaultValues import DefaultValues
from checkm_amd.markerSets import MarkerSetParser
from checkm_amd.resultsParser import ResultsParser
from tools import verify_sample as vs

pytestmark = pytest.mark.gpu


def test_cfg3_200_bins_default_batching_sampled_oracle_diff(gpu_ctx, tmp_path, capsys):
    root = str(tmp_path)
    w = sl.World(os.path.join(root, "data"))
    DefaultValues.set_data_root(os.path.join(root, "data"))
    nb = 200
    binIds = ["bin_%04d" % b for b in range(nb)]
    files = [os.path.join(root, "%s.faa" % b) for b in binIds]
    w.write_bin_files([(b, files[b]) for b in range(nb)], jobs=max(1, min(16, (os.cpu_count() or 2) - 2)))
    lin, _tax = w.write_marker_files(root, binIds)
    out = os.path.join(root, "out")
    assert mgf.PAIR_BUDGET == 125 * 1000 * 1000 and os.environ.get("CKM_FIND_PIPELINE", "2") == "2"
    finder = mgf.MarkerGeneFinder(8)
    finder.find(files, out, DefaultValues.HMMER_TABLE_PHYLO_OUT, DefaultValues.HMMER_PHYLO_OUT, w.phylo_hmm, False, False, True)
    models = finder.find(files, out, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, lin, False, False, True)
    ent = mgf.SCAN_CACHE[(os.path.abspath(out), DefaultValues.HMMER_TABLE_OUT)]
    tot = ent["totals"]
    assert len(ent["parts"]) >= 5 and int(tot["searches"]) >= 5                       # ramped batches, then full ones
    assert int(tot.get("cascade_fallback_lanes", 0)) == 0
    assert int(tot["regions_multi"]) > 500                                            # the trace ensembles ran at scale (one stream per region)
    os.makedirs(os.path.join(out, "storage"), exist_ok=True)
    with open(os.path.join(out, "storage", DefaultValues.BIN_STATS_OUT), "w") as f:
        for b in binIds:
            f.write("%s\t%s\n" % (b, repr({"GC": 0.5, "Genome size": 1})))
    sets = MarkerSetParser().getMarkerSets(out, binIds, lin)
    rp = ResultsParser(models)
    rp.analyseResults(out, DefaultValues.BIN_STATS_OUT, DefaultValues.HMMER_TABLE_OUT)
    rp.printSummary(1, None, sets, False, None, True, None, None)
    lines = capsys.readouterr().out.strip().split("\n")
    qa_rows = {ln.split("\t")[0]: ln for ln in lines[1:]}
    assert sorted(qa_rows) == binIds
    rng = np.random.default_rng(12)
    pick = sorted(rng.choice(nb, size=3, replace=False).tolist())
    res = vs.verify(out, DefaultValues.HMMER_TABLE_OUT, w.checkm_hmm, [binIds[b] for b in pick], [files[b] for b in pick], models, k_bins=3, n_models=40, seed=5,
                    marker_sets=sets, qa_rows=qa_rows, pfam_text=open(DefaultValues.PFAM_CLAN_FILE).read())
    assert res["identical"] and res["qa_rows_identical"], res["mismatches"][:2]
    assert res["models"] >= 120 and res["rows"] >= 60 and res["launch_classes"] >= 20, res
    mgf.release_scan()


def test_cfg5_slice_10003_profiles_sampled_oracle_diff(gpu_ctx, tmp_path):
    root = str(tmp_path)
    nmodels = 10000
    w = sl.World(os.path.join(root, "cfg5_data"), n_models=nmodels)
    DefaultValues.set_data_root(os.path.join(root, "cfg5_data"))
    hmm = os.path.join(root, "pfam_like.hmm")
    import shutil
    shutil.copyfile(w.checkm_hmm, hmm)
    rng = np.random.default_rng(5)
    longs = [synth.random_profile(rng, M, "long%d" % M, "PF%05d.1" % (90000 + M)) for M in (2049, 3000, 4096)]
    for p in longs:
        p.stats = (-8.5 - 0.002 * p.M, 0.71, -9.5 - 0.002 * p.M, 0.71, -3.8, 0.71)
    synth.write_hmm(hmm, longs, mode="a")
    nb = 50
    files = []
    rng = np.random.default_rng(77)
    for b in range(nb):
        f = os.path.join(root, "mag_%03d.faa" % b)
        planted = sorted(rng.choice(nmodels, size=400, replace=False).tolist())
        synth.write_fasta(f, sl.make_lineage_bin(w.profs, planted, 900000 + b, n_orfs=2500))
        files.append(f)
    out = os.path.join(root, "out5")
    models = mgf.MarkerGeneFinder(8).find(files, out, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, hmm, False, False, True)
    ent = mgf.SCAN_CACHE[(os.path.abspath(out), DefaultValues.HMMER_TABLE_OUT)]
    tot = ent["totals"]
    assert len(next(iter(models.values()))) == nmodels + 3
    assert int(tot.get("cascade_fallback_lanes", 0)) == 0
    binIds = ["mag_%03d" % b for b in range(nb)]
    res = vs.verify(out, DefaultValues.HMMER_TABLE_OUT, hmm, binIds[:2], files[:2], models, k_bins=2, n_models=48, seed=9)
    assert res["identical"], res["mismatches"][:2]
    assert res["launch_classes"] >= 30 and res["rows"] >= 30, res
    mgf.release_scan()
