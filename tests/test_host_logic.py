"""CPU tests of the host logic around the hot path, against goldens produced by the reference's own
classes (tools/gen_host_golden.py)."""
import json
import os

import pytest

from checkm_amd.common import binIdFromFilename
from checkm_amd.defaultValues import DefaultValues
from checkm_amd.hmmer import HMMERParser
from checkm_amd.hmmerModelParser import HmmModelParser
from checkm_amd.markerSets import BinMarkerSets, MarkerSet, MarkerSetParser

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "host_cases.json")


@pytest.fixture(scope="module")
def gold():
    with open(GOLD) as f:
        return json.load(f)


def test_sticky_header_parse(gold, tmp_path):
    p = tmp_path / "sticky.hmm"
    p.write_text(gold["sticky"]["hmm_text"])
    models = HmmModelParser(str(p)).models()
    assert set(models) == set(gold["sticky"]["models"])
    for acc, want in gold["sticky"]["models"].items():
        m = models[acc]
        got = {"name": m.name, "acc": m.acc, "leng": m.leng, "ga": list(m.ga) if m.ga else None, "tc": list(m.tc) if m.tc else None,
               "nc": list(m.nc) if m.nc else None}
        assert got == want


def _view(b):
    return {bid: {"selected_uid": bm.selectedMarkerSet().UID,
                  "sets": [[ms.UID, ms.lineageStr, ms.numGenomes, [sorted(s) for s in ms.markerSet]] for ms in bm.markerSets]} for bid, bm in b.items()}


def test_taxon_marker_file(gold, tmp_path):
    tf = tmp_path / "taxon.ms"
    tf.write_text(gold["taxon"]["file"])
    ex = tmp_path / "exclude.txt"
    ex.write_text(gold["taxon"]["exclude_file"])
    msp = MarkerSetParser()
    assert msp.markerFileType(str(tf)) == BinMarkerSets.TAXONOMIC_MARKER_SET
    assert _view(msp.getMarkerSets(str(tmp_path), ["binA", "binB"], str(tf))) == gold["taxon"]["result"]["default"]
    assert _view(msp.getMarkerSets(str(tmp_path), ["binA", "binB"], str(tf), str(ex))) == gold["taxon"]["result"]["user_exclude"]


def test_lineage_marker_file(gold, tmp_path):
    DefaultValues.set_data_root(str(tmp_path))
    (tmp_path / "selected_marker_sets.tsv").write_text(gold["lineage"]["selected_map"])
    lf = tmp_path / "lineage.ms"
    lf.write_text(gold["lineage"]["file"])
    msp = MarkerSetParser()
    assert msp.markerFileType(str(lf)) == BinMarkerSets.TREE_MARKER_SET
    assert _view(msp.getMarkerSets(str(tmp_path), ["binA", "binB"], str(lf))) == gold["lineage"]["result"]


def test_bin_id_from_filename(gold):
    for name, want in gold["binIdFromFilename"].items():
        assert binIdFromFilename(name) == want


def test_marker_set_accessors():
    """The accessor tests of checkm/test/test_markerSets.py:25-55, restated."""
    ms = MarkerSet(0, 'k__Bacteria', 1, [{'a', 'b'}, {'c'}])
    assert ms.size() == (3, 2) and ms.numMarkers() == 3 and ms.numSets() == 2
    assert ms.getMarkerGenes() == {'a', 'b', 'c'}
    b = BinMarkerSets('bin', BinMarkerSets.TAXONOMIC_MARKER_SET)
    b.addMarkerSet(ms)
    b.addMarkerSet(MarkerSet(1, 'k__Bacteria;p__X', 1, [{'d'}]))
    assert b.numMarkerSets() == 2 and b.getMarkerGenes() == {'a', 'b', 'c', 'd'}
    assert b.mostSpecificMarkerSet() is ms and b.selectedMarkerSet() is ms
    ms.removeMarkers({'c'})
    assert ms.markerSet == [{'a', 'b'}]


def test_domtblout_reader_contract(tmp_path):
    """Sample row of checkm/hmmer.py:188; '-' accession falls back to the name; a blank line ends the table."""
    row = ("NODE_925902_length_6780_cov_18.428171_754_2 -            399 PGK                  PF00162.14   384  2.2e-164  543.7   0.1   1   1  "
           "1.3e-167  2.5e-164  543.5   0.1     1   384     9   386     9   386 1.00 # 1767 # 2963 # -1 # ID=754_2;partial=00")
    p = tmp_path / "t.txt"
    p.write_text("# header\n" + row + "\n" + row.replace("PF00162.14", "-         ") + "\n\n" + row + "\n")
    with open(p) as fh:
        hp = HMMERParser(fh)
        h1, h2, h3 = hp.next(), hp.next(), hp.next()
    assert h1.target_name.endswith("754_2") and h1.target_length == 399 and h1.query_accession == "PF00162.14"
    assert h1.full_e_value == 2.2e-164 and h1.dom_score == 543.5 and (h1.ali_from, h1.ali_to) == (9, 386)
    assert h1.target_description.startswith("# 1767 # 2963")
    assert h2.query_accession == "PGK"
    assert h3 is None


def test_amino_acid_identity_matches_the_reference(gold, tmp_path):
    """AminoAcidIdentity.aai / strainHetero / run and HmmerAligner._extractSeq against values produced by the reference's own
    classes (tools/gen_host_golden.py): the leading/trailing-gap quirks, the marker id cut at the first dot, the pair report."""
    import os
    from checkm_amd.aminoAcidIdentity import AminoAcidIdentity
    from checkm_amd.hmmerAligner import HmmerAligner
    a = AminoAcidIdentity()
    for s1, s2, want in gold["aai"]:
        assert a.aai(s1, s2) == want, (s1, s2)
    for scores, thr, het, mean in gold["strain"]:
        h, m = a.strainHetero(scores, thr)
        assert {k: dict(v) for k, v in h.items()} == het and m == mean
    g = gold["aai_run"]
    od = tmp_path / "run"
    for b, fs in g["files"].items():
        (od / "bins" / b).mkdir(parents=True)
        (od / "storage" / "aai_qa" / b).mkdir(parents=True)
        for fn, txt in fs.items():
            (od / "storage" / "aai_qa" / b / fn).write_text(txt)
    (od / "bins" / "binC").mkdir()
    r = AminoAcidIdentity()
    rep = str(od / "pairs.txt")
    r.run(0.9, str(od), rep)
    assert {b: dict(d) for b, d in r.aaiRawScores.items()} == g["raw"]
    assert {b: dict(d) for b, d in r.aaiHetero.items()} == g["hetero"] and r.aaiMeanBinHetero == g["mean"]
    assert sorted(open(rep).read().strip().split("\n\n")) == g["report_sorted_blocks"]
    ha = HmmerAligner(1)
    orfs = {"c1_1": "MKV*", "c1_2": "ACD", "c9_5": "WWW*"}
    for sid, want in gold["extract_seq"]:
        assert ha._extractSeq(sid, orfs) == want


def test_gene_files_step_of_find(tmp_path, monkeypatch):
    """MarkerGeneFinder's first step (checkm/markerGeneFinder.py:108-127): -g copies proteins (plain or .gz) to bins/<id>/genes.faa;
    nucleotide bins go through a ProdigalRunner-like gene caller, `threads` at a time, and bins already called are not called again."""
    import gzip

    from checkm_amd import markerGeneFinder as mgf
    out = tmp_path / "out"
    prot = tmp_path / "a.faa"; prot.write_text(">g_1\nMKV*\n")
    with gzip.open(tmp_path / "b.faa.gz", "wt") as f:
        f.write(">h_1\nMAA*\n")
    finder = mgf.MarkerGeneFinder(3)
    ids, faa, read_from = finder._geneFiles([str(prot), str(tmp_path / "b.faa.gz")], str(out), False, True)
    # the plain file is copied in the background while the scan reads the source; the compressed one is unpacked first
    assert read_from == [str(prot), faa[1]]
    for f in finder._pending_copies:
        f.result()
    assert ids == ["a", "b"] and [open(p).read() for p in faa] == [">g_1\nMKV*\n", ">h_1\nMAA*\n"]

    calls = []

    class FakeProdigal(object):
        def __init__(self, outDir):
            self.aaGeneFile = os.path.join(outDir, "genes.faa")

        def areORFsCalled(self, bNucORFs):
            return os.path.exists(self.aaGeneFile)

        def run(self, query, bNucORFs=True):
            calls.append((os.path.basename(query), bNucORFs))
            with open(self.aaGeneFile, "w") as f:
                f.write(">%s_1\nMSS*\n" % os.path.basename(query))

    nuc = [tmp_path / ("n%d.fna" % i) for i in range(5)]
    for p in nuc:
        p.write_text(">c\nACGT\n")
    monkeypatch.setattr(mgf, "GENE_CALLER", FakeProdigal)
    ids, faa, _read = finder._geneFiles([str(p) for p in nuc], str(out), True, False)
    assert ids == ["n%d" % i for i in range(5)] and sorted(calls) == [("n%d.fna" % i, True) for i in range(5)]
    assert open(faa[3]).read() == ">n3.fna_1\nMSS*\n"
    del calls[:]
    finder._geneFiles([str(p) for p in nuc[:2]], str(out), True, False)
    assert calls == []                                   # already called: reused, as areORFsCalled decides in the reference
    # no caller at all (CheckM not importable here) and nothing on disk: a logged error and exit code 1
    monkeypatch.setattr(mgf, "GENE_CALLER", None)
    monkeypatch.setattr(mgf, "gene_caller", lambda: None)
    with pytest.raises(SystemExit) as ei:
        finder._geneFiles([str(tmp_path / "zz.fna")], str(out), True, False)
    assert ei.value.code == 1


def test_row_at_a_time_rules_match_the_reference(gold, tmp_path):
    """ResultsManager.vetHit and PFAM.filterHitsFromSameClan (the Python-level forms of what ckm_reduce does in bulk) against
    verdicts recorded from the reference's classes on 260 random hits (tools/gen_host_golden.py)."""
    from checkm_amd.hmmer import HmmerHitDOM
    from checkm_amd.hmmerModelParser import HmmModel
    from checkm_amd.pfam import PFAM
    from checkm_amd.resultsParser import ResultsManager
    r = gold["rules"]
    models = {}
    for acc, cut in r["models"].items():
        m = HmmModel({"acc": acc, "name": "n_" + acc, "leng": 100})
        m.ga, m.tc, m.nc = (tuple(cut[k]) if cut[k] is not None else None for k in ("ga", "tc", "nc"))
        models[acc] = m

    def mk(d):
        return HmmerHitDOM([str(d[k]) for k in r["hit_field_order"]])
    for case in r["vetHit"]:
        rm = ResultsManager("b", models, bIgnoreThresholds=case["bIgnoreThresholds"], evalueThreshold=1e-10, lengthThreshold=0.7,
                            bSkipPseudoGeneCorrection=case["bSkipPseudoGeneCorrection"])
        assert [rm.vetHit(mk(d)) for d in r["hits"]] == case["verdicts"]
    dat = tmp_path / "Pfam-A.hmm.dat"
    dat.write_text(r["pfam_dat"])
    for case in r["filterHitsFromSameClan"]:
        objs, mh = {}, {}
        for marker, idxs in case["input"].items():
            for idx in idxs:
                h = mk(r["hits"][idx]); objs[id(h)] = idx
                mh.setdefault(marker, []).append(h)
        res = PFAM(str(dat)).filterHitsFromSameClan(mh)
        assert {k: [objs[id(h)] for h in v] for k, v in res.items()} == case["kept"]
        assert list(res.keys()) == list(case["kept"].keys())          # non-Pfam markers first, then Pfam markers in order of survival
        assert res["absent"] == []                                   # a defaultdict(list), as the reference returns


def test_qa_cache_readers(tmp_path):
    """parseBinStatsExt / parseMarkerGeneStats read back what cacheResults wrote (resultsParser.py:161-189; the reference text of both
    caches is in tests/golden/summary_cases.json)."""
    import ast

    from checkm_amd.resultsParser import ResultsParser
    sc = json.load(open(os.path.join(os.path.dirname(GOLD), "summary_cases.json")))["cases"][0]
    (tmp_path / "storage").mkdir()
    for name, text in sc["caches"].items():
        (tmp_path / "storage" / name).write_text(text)
    rp = ResultsParser(None)
    for name, got in ((DefaultValues.BIN_STATS_EXT_OUT, rp.parseBinStatsExt(str(tmp_path))), (DefaultValues.MARKER_GENE_STATS, rp.parseMarkerGeneStats(str(tmp_path)))):
        want = {ln.split("\t")[0]: ast.literal_eval(ln.split("\t")[1]) for ln in sc["caches"][name].splitlines()}
        assert got == want and len(got) >= 2


def test_known_answers_of_the_reference_test_suite():
    """The facts the reference's own unit tests assert on this path (checkm/test/test_aminoAcidIdentity.py:27-143 and
    checkm/test/test_markerSets.py:26-60), restated as data."""
    from checkm_amd.aminoAcidIdentity import AminoAcidIdentity
    aai = AminoAcidIdentity()
    for a, b, want in (("ACGT", "ACGT", 1.0), ("ACGT", "TGCA", 0.0), ("ACGT----", "----TGCA", 0.0), ("ACGT--", "--GTAC", 1.0),
                       ("AAAACGTTTT", "---ACGG---", 3.0 / 4.0), ("ACGT", "ACGG", 3.0 / 4.0), ("A-C-G-T", "A-C-G-T", 1.0), ("A-C-G-T", "AACCGGT", 4.0 / 7.0)):
        assert aai.aai(a, b) == pytest.approx(want, abs=1e-7), (a, b)
    for scores, het, mean in (({"g1": [0.1], "g2": [0.1], "g3": [0.1]}, (0.0, 0.0, 0.0), 0.0),
                              ({"g1": [0.95], "g2": [0.95], "g3": [0.95]}, (1.0, 1.0, 1.0), 100.0),
                              ({"g1": [0.95], "g2": [0.1], "g3": [0.1]}, (1.0, 0.0, 0.0), 100.0 / 3.0),
                              ({"g1": [0.95, 0.95, 0.95], "g2": [0.1, 0.1, 0.1], "g3": [0.95, 0.1, 0.1]}, (1.0, 0.0, 1.0 / 3.0), 400.0 / 9.0)):
        h, m = aai.strainHetero({"b1": scores}, 0.9)
        assert [h["b1"][g] for g in ("g1", "g2", "g3")] == pytest.approx(list(het), abs=1e-7)
        assert m["b1"] == pytest.approx(mean, abs=1e-7)
    ms = MarkerSet(0, 'k__Bacteria', 100, [{'a', 'b'}, {'c'}])
    assert ms.size() == (3, 2) and ms.numMarkers() == 3 and ms.numSets() == 2 and ms.getMarkerGenes() == {'a', 'b', 'c'}
    bms = BinMarkerSets(0, BinMarkerSets.TAXONOMIC_MARKER_SET)
    ms1 = MarkerSet(1, 'k__Bacteria', 100, [{'a', 'b'}, {'c'}])
    ms2 = MarkerSet(2, 'k__Bacteria', 100, [{'d', 'e'}, {'f'}])
    bms.addMarkerSet(ms1); bms.addMarkerSet(ms2)
    assert bms.getMarkerGenes() == set('abcdef') and bms.mostSpecificMarkerSet() is ms1 and bms.selectedMarkerSet() is ms1
