"""Host logic of the lineage_wf-shaped path that needs no device: the synthetic lineage world parses through the marker-file mirrors
(checkm/markerSets.py:428-522), the model subset of a bin is its chain's marker genes plus the clan expansion (markerSets.py:443-457)
matched by NAME or ACC as `hmmfetch -f` matches (markerSets.py:326-343), and find()'s batch plan / rank shards."""
from checkm_amd import dist as cdist, markerGeneFinder as mgf
from synthdata import synth_lineage as sl
from checkm_amd.defaultValues import DefaultValues
from checkm_amd.markerSets import BinMarkerSets, MarkerSetParser, wanted_model


def test_world_files_parse_through_the_mirrors(tmp_path):
    w = sl.World(str(tmp_path / "data"), n_models=240, seed=77)
    DefaultValues.set_data_root(str(tmp_path / "data"))
    binIds = ["b%d" % i for i in range(12)]
    lin, tax = w.write_marker_files(str(tmp_path), binIds)
    msp = MarkerSetParser()
    assert msp.markerFileType(lin) == BinMarkerSets.TREE_MARKER_SET and msp.markerFileType(tax) == BinMarkerSets.TAXONOMIC_MARKER_SET
    assert msp.hmmDatabaseFor(lin) == DefaultValues.HMM_MODELS == w.checkm_hmm
    per_bin = msp.parseLineageMarkerSetFile(lin)
    wanted = msp.markerAccessionsForBins(binIds, lin)
    from checkm_amd.pfam import PFAM
    pfam = PFAM(DefaultValues.PFAM_CLAN_FILE)
    sub = msp.markerAccessionsForBins(binIds[5:2:-1], lin)                      # a subset, in the caller's order
    assert list(sub) == binIds[5:2:-1] and all(sub[b] == wanted[b] for b in sub)
    for k, b in enumerate(binIds):
        fid = w.family_of(k)
        bms = per_bin[b]
        assert [ms.UID for ms in bms.markerSets] == w.lineage.chain(fid)
        assert bms.getMarkerGenes() == w.lineage.marker_genes(fid)
        assert [sorted(s) for s in bms.selectedMarkerSet().markerSet] == [sorted(s) for s in w.lineage.selected_sets(fid)]
        assert w.lineage.marker_genes(fid) <= wanted[b] <= set(w.accs)
        # (markerAccessionsForBins keys the expansion by the file's literals instead of building every bin's sets: same result)
        assert wanted[b] == bms.getMarkerGenes() | pfam.genesInSameClan(bms.getMarkerGenes())
    # clan expansion: a marker in a clan pulls its clan-mates in
    assert any(wanted[b] - w.lineage.marker_genes(w.family_of(k)) for k, b in enumerate(binIds))
    t = msp.markerAccessionsForBins(binIds, tax)
    assert all(t[b] is t[binIds[0]] for b in binIds)
    sets = msp.getMarkerSets(str(tmp_path), binIds, tax)
    assert sets["b3"].selectedMarkerSet().UID == "p1"
    # records: prodigal-style names, planted markers of the selected set
    recs = w.bin_records(0, orf_lo=200, orf_hi=260)
    assert 200 <= len(recs) <= 260 and all(r[2].endswith("*") and "_" in r[0] for r in recs)
    assert w.bin_records(0, orf_lo=200, orf_hi=260) == recs                      # fixed seeds


def test_wanted_model_matches_name_or_acc():
    keys = {"PF00001.1", "TIGR00042", "onlyname"}
    assert wanted_model("x", "PF00001.1", keys) and wanted_model("TIGR00042", "TIGR00042", keys)
    assert wanted_model("onlyname", "PF99999.9", keys)                            # listed by NAME although the record has an ACC
    assert wanted_model("onlyname", None, keys) and not wanted_model("y", None, keys) and not wanted_model("y", "PF2.1", keys)


def test_batches_and_shards():
    sizes = [320 * 1000] * 10 + [320 * 5000] * 3          # ~1000 and ~5000 ORFs
    nm = [500] * 13
    b = mgf.plan_batches(sizes, nm, pair_budget=2 * 1000 * 1000, res_budget=10 ** 12)
    assert [i for part in b for i in part] == list(range(13)) and all(part for part in b)
    assert all(sum(sizes[i] // 320 * nm[i] for i in part) <= 2 * 1000 * 1000 or len(part) == 1 for part in b)
    assert len(mgf.plan_batches(sizes, nm, pair_budget=10 ** 15, res_budget=10 ** 15)) == 1
    assert mgf.plan_batches([], []) == []
    # the first three batches of a pass stop at 1/8, 1/4, 1/2 of the budgets (the device starts after a short ingest; lanes start staggered)
    many = [320 * 1000] * 400
    r = mgf.plan_batches(many, [500] * 400, pair_budget=40 * 1000 * 500, res_budget=10 ** 15)
    assert [len(x) for x in r[:5]] == [5, 10, 20, 40, 40] and sum(len(x) for x in r) == 400
    assert len(mgf.plan_batches(sizes, nm, pair_budget=10 ** 15, res_budget=320 * 2000)) >= 8
    sh = cdist.shard_bins([s * n for s, n in zip(sizes, nm)], 4)
    assert sorted(i for s in sh for i in s) == list(range(13))
    loads = [sum(sizes[i] for i in s) for s in sh]
    assert max(loads) <= 1.6 * min(loads)


def test_lineage_sets_behave_like_the_dict_the_reference_returns(tmp_path):
    """getMarkerSets / parseLineageMarkerSetFile build a bin's sets when the bin is first asked for (checkm/markerSets.py:478-511 builds
    all of them): every way a caller of the reference may reach a value must build it -- dict(d), d.copy(), {**d}, pop, setdefault,
    deepcopy, pickle, two threads asking for one bin."""
    import copy
    import pickle
    import threading
    w = sl.World(str(tmp_path / "data"), n_models=240, seed=78)
    DefaultValues.set_data_root(str(tmp_path / "data"))
    binIds = ["b%d" % i for i in range(10)]
    lin, _tax = w.write_marker_files(str(tmp_path), binIds)
    msp = MarkerSetParser()

    def fresh():
        return msp.getMarkerSets(str(tmp_path), binIds, lin)
    ref = {b: fresh()[b].getMarkerGenes() for b in binIds}
    for view in (dict(fresh()), fresh().copy(), {**fresh()}, copy.deepcopy(fresh()), pickle.loads(pickle.dumps(fresh())), dict(fresh().items())):
        assert type(view) is dict and list(view) == binIds
        assert all(isinstance(v, BinMarkerSets) and v.getMarkerGenes() == ref[b] for b, v in view.items())
    d = fresh()
    assert len(d) == 10 and "b3" in d and "zz" not in d and d.get("zz") is None and list(d.keys()) == binIds
    assert d.pop("b3").getMarkerGenes() == ref["b3"] and "b3" not in d and len(d) == 9 and list(d) == [b for b in binIds if b != "b3"]
    assert d.setdefault("b4", None).getMarkerGenes() == ref["b4"]
    assert d.setdefault("new", 5) == 5 and d["new"] == 5 and list(d)[-1] == "new"
    assert all(v is not None for v in d.values())
    try:
        d["zz"]
        assert False
    except KeyError:
        pass
    d2 = fresh()
    assert d2 == d2.copy() and d2.copy() == d2                 # (Mapping equality over the built values)
    # the excluded markers apply to bins built before and after the call (TIGR00398 / TIGR00399 always, markerSets.py:277-297)
    assert all(not ({"TIGR00398", "TIGR00399"} & g) for g in ref.values())
    # two threads, one bin
    d = fresh()
    got, errs = [], []

    def ask():
        try:
            got.append(d["b7"])
        except Exception as e:          # noqa: BLE001
            errs.append(e)
    ts = [threading.Thread(target=ask) for _ in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs and len(got) == 8 and all(g is got[0] for g in got)


def test_hard_world_is_deterministic_and_leaves_the_plain_one_alone(tmp_path):
    """bench.py's hard_workload leg (synthdata/synth_lineage.py: make_lineage_bin(hard=True)): same ORF count and names as the plain bin,
    fixed seeds, the planted markers still there; the plain world's bins are what they were before the switch existed."""
    w = sl.World(str(tmp_path / "data"), n_models=240, seed=77)
    a, h = w.bin_records(3, orf_lo=400, orf_hi=500), w.bin_records(3, orf_lo=400, orf_hi=500, hard=True)
    assert a == w.bin_records(3, orf_lo=400, orf_hi=500) and h == w.bin_records(3, orf_lo=400, orf_hi=500, hard=True)
    assert [r[0] for r in a] == [r[0] for r in h] and a != h
    def poor(recs):           # proteins over a small alphabet: the low-complexity ORFs (a random 60-mer of the background uses 15 letters or more)
        return sum(1 for r in recs if len(r[2]) > 60 and len(set(r[2][:-1])) <= 8)
    assert poor(h) >= 2 and poor(a) == 0
    block = sl.scrambled(__import__("numpy").random.default_rng(1), __import__("numpy").arange(100))
    assert sorted(block.tolist()) == list(range(100)) and block.tolist() != list(range(100))
