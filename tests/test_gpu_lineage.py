"""The lineage_wf-shaped configuration (cfg3 of BASELINE.json) through the reference's own call sequence
(checkm/main.py:156-160, 325-333): a 2000-profile `checkm.hmm`, per-bin model subsets that come out of a *Lineage* (and a *Taxon*)
marker file via MarkerSetParser.markerAccessionsForBins + the clan expansion, MarkerGeneFinder.find -> ResultsParser.analyseResults
-> printSummary.  Whole bins are compared with (scan oracle -> domtblout text -> reduce oracle); every bin is compared between the
resident-hits path, the text path of a later `qa`, and a scan of the bin alone."""
import os

import numpy as np
import pytest

from checkm_amd import _lib
from synthdata import synth, synth_lineage as sl
from checkm_amd.defaultValues import DefaultValues
from checkm_amd import markerGeneFinder as mgf
from checkm_amd.markerSets import MarkerSetParser
from checkm_amd.resultsParser import ResultsParser
from oracle import p7
from oracle import reduce_oracle as ro
from tests import common

pytestmark = pytest.mark.gpu

NBINS = 32
CHECKED = 4            # whole bins compared row for row with the oracles


@pytest.fixture(scope="module")
def lineage(gpu_ctx, tmp_path_factory):
    root = tmp_path_factory.mktemp("lineage")
    w = sl.World(str(root / "data"))
    DefaultValues.set_data_root(str(root / "data"))
    binIds = ["bin_%03d" % b for b in range(NBINS)]
    recs, files = [], []
    for b in range(NBINS):
        # the checked bins are kept small (the scalar oracle does ~1e6 residue*HMM/s per thread); the rest are larger
        lo, hi = (250, 400) if b < CHECKED else (400, 1200)
        r = w.bin_records(b, orf_lo=lo, orf_hi=hi)
        f = root / ("%s.faa" % binIds[b])
        synth.write_fasta(str(f), r)
        recs.append(r); files.append(str(f))
    lin, tax = w.write_marker_files(str(root), binIds)
    yield dict(w=w, root=root, binIds=binIds, recs=recs, files=files, lin=lin, tax=tax)
    mgf.release_scan()


def _bin_stats(out, binIds):
    os.makedirs(os.path.join(out, "storage"), exist_ok=True)
    with open(os.path.join(out, "storage", DefaultValues.BIN_STATS_OUT), "w") as f:
        for k, b in enumerate(binIds):
            f.write("%s\t%s\n" % (b, repr({"GC": 0.5, "Genome size": 1000 + k})))


def _omodels(models):
    return {a: {"acc": a, "ga": list(m.ga) if m.ga else None, "tc": list(m.tc) if m.tc else None, "nc": list(m.nc) if m.nc else None, "leng": m.leng}
            for a, m in models.items()}


def _summary_rows(capsys):
    lines = capsys.readouterr().out.strip().split("\n")
    return lines[0], {ln.split("\t")[0]: ln for ln in lines[1:]}


def test_lineage_marker_file_through_find_and_qa(lineage, capsys, monkeypatch):
    L = lineage
    w, binIds = L["w"], L["binIds"]
    out = str(L["root"] / "out_lineage")
    monkeypatch.setattr(mgf, "PAIR_BUDGET", 4 * 1000 * 1000)           # several ckm_search calls: batching, overlapped ingest and output
    models = mgf.MarkerGeneFinder(4).find(L["files"], out, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, L["lin"], False, False, True)
    ent = mgf.SCAN_CACHE[(os.path.abspath(out), DefaultValues.HMMER_TABLE_OUT)]
    assert len(ent["parts"]) >= 3 and sorted(models) == binIds
    # the subset of a bin = marker genes of its whole chain + their clans (markerSets.py:443-457), in database order
    msp = MarkerSetParser()
    wanted = msp.markerAccessionsForBins(binIds, L["lin"])
    nsub = []
    for b, binId in enumerate(binIds):
        genes = w.lineage.marker_genes(w.family_of(b))
        assert genes <= wanted[binId] and set(models[binId]) == wanted[binId]
        assert wanted[binId] - genes                                               # the clan expansion added families
        nsub.append(len(wanted[binId]))
    assert min(nsub) >= 300 and max(nsub) <= 1600 and len(set(nsub)) > 5
    # ---- whole bins against the scan oracle (text) ----
    hs = p7.HmmSet(w.checkm_hmm)
    index = {hs.acc(i): i for i in range(hs.n)}
    texts = {}
    for b in range(CHECKED):
        binId = binIds[b]
        sub = sorted(index[a] for a in wanted[binId])
        recs = L["recs"][b]
        rows = common.oracle_search_threaded(hs, sub, [p7.digitize(r[2]) for r in recs], [r[0] for r in recs])
        texts[binId] = hs.format_domtblout(rows, [r[0] for r in recs], [r[1] for r in recs])
        got = open(os.path.join(out, "bins", binId, DefaultValues.HMMER_TABLE_OUT)).read()
        assert got == texts[binId], binId
        assert len(rows) >= 50
    hs.close()
    # ---- QA: resident hits, one batched count, table; against the reduce oracle for the checked bins ----
    _bin_stats(out, binIds)
    sets = msp.getMarkerSets(out, binIds, L["lin"])
    dat = open(DefaultValues.PFAM_CLAN_FILE).read()
    rp = ResultsParser(models)
    rp.analyseResults(out, DefaultValues.BIN_STATS_OUT, DefaultValues.HMMER_TABLE_OUT)
    rp.printSummary(1, None, sets, False, None, True, None, None)
    head, rows1 = _summary_rows(capsys)
    assert head.split("\t")[0] == "Bin Id" and sorted(rows1) == binIds
    for b in range(CHECKED):
        binId = binIds[b]
        sel = sets[binId].selectedMarkerSet()
        assert [sorted(s) for s in sel.markerSet] == [sorted(s - DefaultValues.MARKERS_TO_EXCLUDE) for s in w.lineage.selected_sets(w.family_of(b))]
        mh, gc = ro.reduce_bin(texts[binId], _omodels(models[binId]), dat, [sorted(s) for s in sel.markerSet])
        rm = rp.results[binId]
        got = [[k, [[h.target_name, h.target_length, h.hmm_from, h.hmm_to, h.ali_from, h.ali_to, h.env_from, h.env_to, h.dom_score, h.full_e_value]
                    for h in v]] for k, v in rm.markerHits.items()]
        assert got == ro.marker_hits_view(mh), binId
        f = rows1[binId].split("\t")
        assert [int(x) for x in f[5:11]] == gc[:6] and f[11] == "%0.2f" % gc[6] and f[12] == "%0.2f" % gc[7], binId
        assert gc[6] > 2.0                                                           # planted markers are found (these small bins hold ~100 lineage plants)
        assert rm.geneCounts(sel, rm.markerHits, False) == gc                       # the batched row ...
        rm._counts.clear()
        assert rm.geneCounts(sel, rm.markerHits, False) == gc                       # ... and the per-bin launch agree
    # ---- a later `qa`: the scan is gone, every table is re-read from text; all 32 rows identical ----
    mgf.release_scan()
    rp2 = ResultsParser(models)
    rp2.analyseResults(out, DefaultValues.BIN_STATS_OUT, DefaultValues.HMMER_TABLE_OUT)
    rp2.printSummary(1, None, sets, False, None, True, None, None)
    _head, rows2 = _summary_rows(capsys)
    assert rows2 == rows1
    # ---- a bin scanned alone gives the same table (Z, domZ and the plan are per bin: what sharding over GPUs relies on) ----
    out1 = str(L["root"] / "out_single")
    for b in (1, 9, 20):
        mgf.MarkerGeneFinder(1).find([L["files"][b]], out1, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, L["lin"], False, False, True)
        a = open(os.path.join(out1, "bins", binIds[b], DefaultValues.HMMER_TABLE_OUT)).read()
        assert a == open(os.path.join(out, "bins", binIds[b], DefaultValues.HMMER_TABLE_OUT)).read(), b


def test_taxon_marker_file_and_phylo_pass(lineage, capsys):
    """taxonomy_wf's branch: ONE marker set for every bin (markerSets.py:428-441), and the tree pass of lineage_wf (phylo.hmm, every
    model for every bin, main.py:156-160)."""
    L = lineage
    w, binIds = L["w"], L["binIds"]
    out = str(L["root"] / "out_taxon")
    files = L["files"][:6]
    ids = binIds[:6]
    models = mgf.MarkerGeneFinder(2).find(files, out, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, L["tax"], False, False, True)
    msp = MarkerSetParser()
    wanted = msp.markerAccessionsForBins(ids, L["tax"])
    assert all(set(models[b]) == wanted[b] for b in ids) and len(set(id(wanted[b]) for b in ids)) == 1
    hs = p7.HmmSet(w.checkm_hmm)
    index = {hs.acc(i): i for i in range(hs.n)}
    sub = sorted(index[a] for a in wanted[ids[0]])
    recs = L["recs"][2]
    rows = common.oracle_search_threaded(hs, sub, [p7.digitize(r[2]) for r in recs], [r[0] for r in recs])
    text = hs.format_domtblout(rows, [r[0] for r in recs], [r[1] for r in recs])
    hs.close()
    assert open(os.path.join(out, "bins", ids[2], DefaultValues.HMMER_TABLE_OUT)).read() == text
    _bin_stats(out, ids)
    sets = msp.getMarkerSets(out, ids, L["tax"])
    rp = ResultsParser(models)
    rp.analyseResults(out, DefaultValues.BIN_STATS_OUT, DefaultValues.HMMER_TABLE_OUT)
    rp.printSummary(1, None, sets, False, None, True, None, None)
    _head, rows = _summary_rows(capsys)
    dat = open(DefaultValues.PFAM_CLAN_FILE).read()
    sel = sets[ids[2]].selectedMarkerSet()
    _mh, gc = ro.reduce_bin(text, _omodels(models[ids[2]]), dat, [sorted(s) for s in sel.markerSet])
    f = rows[ids[2]].split("\t")
    assert [int(x) for x in f[5:11]] == gc[:6] and f[11] == "%0.2f" % gc[6] and f[12] == "%0.2f" % gc[7]
    # tree pass: phylo.hmm as a raw HMM marker file
    mgf.MarkerGeneFinder(2).find(files[:3], out, DefaultValues.HMMER_TABLE_PHYLO_OUT, DefaultValues.HMMER_PHYLO_OUT, w.phylo_hmm, False, False, True)
    hp = p7.HmmSet(w.phylo_hmm)
    recs = L["recs"][1]
    rows = common.oracle_search_threaded(hp, range(hp.n), [p7.digitize(r[2]) for r in recs], [r[0] for r in recs])
    assert open(os.path.join(out, "bins", ids[1], DefaultValues.HMMER_TABLE_PHYLO_OUT)).read() == hp.format_domtblout(rows, [r[0] for r in recs], [r[1] for r in recs])
    assert len(set(r.model_idx for r in rows)) >= 40                                 # the 43 phylogenetic markers were planted
    hp.close()


def test_large_database_with_long_and_overlong_models(gpu_ctx, tmp_path):
    """configs[4]-shaped slice: a 2000+-profile database that also holds models of 2049, 3000 and 4096 nodes -- beyond the 2048 the SSV
    kernel's LDS image holds, searched through the exact-MSV route and the Q = 48 / 64, QH = 24 / 32 kernel classes -- and one of 4500
    nodes, beyond every class.  The database loads, the 4500-node model keeps its place and header but is flagged unsearchable and left
    out of every bin's scan with a warning, batching never runs out of memory, the rows of the three long models and of a sample of the
    others equal the oracle's in two bins, and a search that names the 4500-node model is refused with CKM_ERANGE by name."""
    rng = np.random.default_rng(5150)
    profs = []
    for i in range(2000):
        M = int(np.clip(np.rint(rng.lognormal(4.0, 0.5)), 20, 400))
        profs.append(synth.random_profile(rng, M, "m%05d" % i, "PF%05d.1" % (30000 + i)))
    longs = {}
    for k, M in enumerate((2049, 3000, 4096, 4500)):
        lp = synth.random_profile(rng, M, "long%d" % M, "PF%05d.1" % (39000 + k))
        longs[M] = lp
        profs.insert(400 * k + 300, lp)
    for p in profs:
        p.stats = (-8.5 - 0.002 * p.M, 0.71, -9.5 - 0.002 * p.M, 0.71, -3.8, 0.71)
        p.ga = (25.0, 25.0)
    hmm = str(tmp_path / "big.hmm")
    synth.write_hmm(hmm, profs)
    prof = _lib.Profiles(gpu_ctx, hmm)
    assert prof.n == 2004
    unsearchable = [i for i, h in enumerate(prof.headers) if not h["searchable"]]
    assert [prof.headers[i]["leng"] for i in unsearchable] == [4500]
    searchable = [p for p in profs if p.M <= 4096]
    long_pos = [i for i, p in enumerate(searchable) if p.M > 2048]
    files = []
    for b in range(4):
        planted = sorted(set(rng.choice(len(searchable), size=80, replace=False).tolist()) | set(long_pos))
        recs = sl.make_lineage_bin(searchable, planted, 7000 + b, n_orfs=400)
        f = str(tmp_path / ("bin%d.faa" % b))
        synth.write_fasta(f, recs)
        files.append((f, recs, planted))
    seqs = _lib.Seqs(gpu_ctx, [files[0][1][:20]])
    with pytest.raises(_lib.CkmError) as err:
        _lib.search(gpu_ctx, prof, seqs, [[0, unsearchable[0]]])
    assert err.value.code != 0 and "long4500" in str(err.value)
    seqs.close(); prof.close()
    out = str(tmp_path / "out")
    budget = mgf.PAIR_BUDGET
    mgf.PAIR_BUDGET = 500 * 1000                                          # several batches
    try:
        models = mgf.MarkerGeneFinder(4).find([f for f, _r, _p in files], out, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, hmm, False, False, True)
    finally:
        mgf.PAIR_BUDGET = budget
    assert all(len(m) == 2003 for m in models.values())                                     # the 4500-node model is in no bin's view
    hs = p7.HmmSet(hmm)
    names = [p.name for p in searchable]
    for b in (1, 3):
        f, recs, planted = files[b]
        chosen = set(names[j] for j in sorted(set(planted[:20] + list(range(0, 2000, 150)) + long_pos)))
        idx = [i for i in range(hs.n) if hs.name(i) in chosen]
        rows = common.oracle_search_threaded(hs, idx, [p7.digitize(r[2]) for r in recs], [r[0] for r in recs])
        want = hs.format_domtblout(rows, [r[0] for r in recs], [r[1] for r in recs])
        got = [l for l in open(os.path.join(out, "bins", "bin%d" % b, DefaultValues.HMMER_TABLE_OUT)) if not l.startswith("#") and l.split()[3] in chosen]
        ref = [l + "\n" for l in want.split("\n") if l and not l.startswith("#")]
        key = lambda l: (l.split()[3], l.split()[0], int(l.split()[9]))
        # (E-values depend on Z = sequences of the bin, the same on both sides; domZ per model likewise)
        assert sorted(got, key=key) == sorted(ref, key=key) and len(ref) >= 20
        assert sum(1 for l in ref if l.split()[3].startswith("long")) >= 3                 # each long model finds its planted ORF
    hs.close()
    mgf.release_scan()
