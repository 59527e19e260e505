"""End to end through the reference's own API surface: MarkerGeneFinder.find -> ResultsParser.analyseResults
-> printSummary, on the GPU, against (scan oracle -> domtblout text -> reduce oracle)."""
import os

import numpy as np
import pytest

from synthdata import synth
from checkm_amd.defaultValues import DefaultValues
from checkm_amd.markerGeneFinder import MarkerGeneFinder, release_scan
from checkm_amd.markerSets import MarkerSetParser
from checkm_amd.resultsParser import ResultsParser
from oracle import p7
from oracle import reduce_oracle as ro
from tests import common

pytestmark = pytest.mark.gpu


def _setup(tmp_path, nbins=3):
    profs = synth.small_profiles(11, 12, 40, 300)
    hmm = common.hmm_file("s11", profs)
    root = tmp_path / "data"
    (root / "pfam").mkdir(parents=True)
    # two Pfam families in one clan so the clan filter has work to do
    pf = [p.acc for p in profs if p.acc.startswith("PF")]
    dat = ""
    for i, a in enumerate(pf):
        dat += "# STOCKHOLM 1.0\n#=GF ID   fam%d\n#=GF AC   %s\n" % (i, a)
        if i < 2:
            dat += "#=GF CL   CL0001\n"
        dat += "//\n"
    (root / "pfam" / "Pfam-A.hmm.dat").write_text(dat)
    DefaultValues.set_data_root(str(root))
    binfiles, recs_all = [], []
    for b in range(nbins):
        recs = synth.make_bin(profs, 500 + b, n_orfs=150, dup_frac=0.6)
        f = tmp_path / ("bin_%d.faa" % b)
        synth.write_fasta(str(f), recs)
        binfiles.append(str(f)); recs_all.append(recs)
    return profs, hmm, binfiles, recs_all, dat


def _oracle_tables(hmm, recs_all):
    hs = p7.HmmSet(hmm)
    texts = []
    for recs in recs_all:
        rows = hs.search(list(range(hs.n)), [p7.digitize(r[2]) for r in recs], [r[0] for r in recs])
        texts.append(hs.format_domtblout(rows, [r[0] for r in recs], [r[1] for r in recs]))
    hs.close()
    return texts


def test_find_then_qa_matches_oracles(gpu_ctx, tmp_path, capsys):
    profs, hmm, binfiles, recs_all, dat = _setup(tmp_path)
    out = tmp_path / "out"
    (out / "storage").mkdir(parents=True)
    models = MarkerGeneFinder(4).find(binfiles, str(out), DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, hmm, False, False, True)
    assert sorted(models) == ["bin_0", "bin_1", "bin_2"]
    texts = _oracle_tables(hmm, recs_all)
    for b in range(3):
        t = out / "bins" / ("bin_%d" % b) / DefaultValues.HMMER_TABLE_OUT
        assert t.read_text() == texts[b]
        assert (out / "bins" / ("bin_%d" % b) / DefaultValues.PRODIGAL_AA).exists()
    # HmmModel view (sticky parse) and pickle cache round trip
    msp = MarkerSetParser()
    msp.writeBinModels(models, str(out / "storage" / DefaultValues.CHECKM_HMM_MODEL_INFO))
    models2 = msp.loadBinModels(str(out / "storage" / DefaultValues.CHECKM_HMM_MODEL_INFO))
    assert {a: (m.name, m.leng, m.ga, m.tc, m.nc) for a, m in models2["bin_1"].items()} == {a: (m.name, m.leng, m.ga, m.tc, m.nc) for a, m in models["bin_1"].items()}
    with open(out / "storage" / DefaultValues.BIN_STATS_OUT, "w") as f:
        for b in range(3):
            f.write("bin_%d\t%s\n" % (b, repr({"GC": 0.5, "Genome size": 1000 + b})))
    sets = msp.getMarkerSets(str(out), list(models), hmm)
    want = {}
    omodels = {a: {"acc": a, "ga": list(m.ga) if m.ga else None, "tc": list(m.tc) if m.tc else None, "nc": list(m.nc) if m.nc else None, "leng": m.leng}
               for a, m in models["bin_0"].items()}
    for b in range(3):
        mh, gc = ro.reduce_bin(texts[b], omodels, dat, [sorted(s) for s in sets["bin_%d" % b].selectedMarkerSet().markerSet])
        want["bin_%d" % b] = (ro.marker_hits_view(mh), gc)

    def check(rp):
        for b, (view, gc) in want.items():
            rm = rp.results[b]
            got = [[k, [[h.target_name, h.target_length, h.hmm_from, h.hmm_to, h.ali_from, h.ali_to, h.env_from, h.env_to, h.dom_score, h.full_e_value]
                        for h in v]] for k, v in rm.markerHits.items()]
            assert got == view, b
            assert rm.geneCountsForSelectedMarkerSet(sets[b], False) == gc, b
    # (1) resident packed hits, no text round trip
    rp = ResultsParser(models)
    stats = rp.analyseResults(str(out), DefaultValues.BIN_STATS_OUT, DefaultValues.HMMER_TABLE_OUT)
    assert stats["bin_2"]["Genome size"] == 1002
    check(rp)
    rp.printSummary(1, None, sets, False, None, True, None, None)
    lines = capsys.readouterr().out.strip().split("\n")
    assert lines[0].split("\t")[0] == "Bin Id" and len(lines) == 4
    for ln in lines[1:]:
        f = ln.split("\t")
        gc = want[f[0]][1]
        assert [int(x) for x in f[5:11]] == gc[:6] and f[11] == "%0.2f" % gc[6] and f[12] == "%0.2f" % gc[7]
    rp.cacheResults(str(out), sets, False)
    assert os.path.getsize(out / "storage" / DefaultValues.BIN_STATS_EXT_OUT) > 0
    # (2) a later `qa` command: the scan is gone, the domtblout text is re-read
    release_scan()
    rp2 = ResultsParser(models2)
    rp2.analyseResults(str(out), DefaultValues.BIN_STATS_OUT, DefaultValues.HMMER_TABLE_OUT)
    check(rp2)
    # flags reach the library
    rp3 = ResultsParser(models2)
    rp3.parseBinHits(str(out), DefaultValues.HMMER_TABLE_OUT, bSkipAdjCorrection=True, bIgnoreThresholds=True)
    mh, gc = ro.reduce_bin(texts[1], omodels, dat, [sorted(s) for s in sets["bin_1"].selectedMarkerSet().markerSet], True, 1e-10, 0.7, False, True)
    assert rp3.results["bin_1"].geneCountsForSelectedMarkerSet(sets["bin_1"], False) == gc


def test_cutoffs_are_float64_end_to_end(gpu_ctx, tmp_path):
    """GA 25.10 must reach vetHit as the double 25.1 (as float32 it is 25.100000381 and would reject a 25.1 hit;
    resultsParser.py:356-367 compares Python floats parsed from text)."""
    from checkm_amd import _lib, qa as cqa
    prof0 = synth.small_profiles(5, 1, 40, 41)[0]
    prof0.acc = "PF12345.1"
    prof0.ga, prof0.tc, prof0.nc = (25.10, 25.10), None, None
    prof0.stats = (-8.0, 0.71, -9.0, 0.71, -3.5, 0.71)
    path = str(tmp_path / "cut.hmm")
    synth.write_hmm(path, [prof0])
    prof = _lib.Profiles(gpu_ctx, path)
    assert prof.headers[0]["ga"] == (25.1, 25.1)
    plan = cqa.QAPlan.for_hmm_models(prof, [[0]])
    rows = [dict(target_name="c_1", target_length=100, query_accession="PF12345.1", query_length=prof0.M, full_e_value=1e-12,
                 full_score=25.1, full_bias=0.0, dom=1, ndom=1, c_evalue=1e-12, i_evalue=1e-12, dom_score=25.1, dom_bias=0.0,
                 hmm_from=1, hmm_to=prof0.M, ali_from=1, ali_to=90, env_from=1, env_to=90, acc=0.9)]
    hc = cqa.ext_columns([rows], lambda r: 0)
    res = plan.reduce(gpu_ctx, None, None, ext=hc)
    assert list(res.hist[0]) == [0, 1, 0, 0, 0, 0] and res.completeness[0] == 100.0
    res.close(); prof.close()


def test_align_matches_the_oracle(gpu_ctx):
    """ckm_align (what hmmalign computes per sequence, reduced to the match columns CheckM keeps) against the oracle's restatement:
    whole genes with flanks, fragments, two copies in one sequence, a merged pair of adjacent ORFs, random and tiny sequences."""
    from checkm_amd import _lib
    profs = common.mixed_profiles()
    path = common.hmm_file("mixed", profs)
    rng = np.random.default_rng(41)
    recs, model = [], []
    for mi, p in enumerate(profs):
        dom = synth.sample_domain(rng, p)
        recs.append(("g%d_1" % mi, "", synth.to_text(np.concatenate([synth.random_residues(rng, 15), dom, synth.random_residues(rng, 20)])))); model.append(mi)
        a = int(rng.integers(2, p.M // 2)); b = int(rng.integers(p.M // 2 + 1, p.M))
        recs.append(("g%d_2" % mi, "", synth.to_text(synth.sample_domain(rng, p, a, b)))); model.append(mi)
        recs.append(("g%d_3" % mi, "", synth.to_text(np.concatenate([dom, synth.random_residues(rng, 6), synth.sample_domain(rng, p)])))); model.append(mi)
        recs.append(("g%d_4" % mi, "", synth.to_text(synth.random_residues(rng, int(rng.integers(1, 90)))))); model.append(mi)
    recs.append(("tiny", "", "M")); model.append(0)
    prof = _lib.Profiles(gpu_ctx, path)
    seqs = _lib.Seqs(gpu_ctx, [recs])
    hs = p7.HmmSet(path)
    got = _lib.align(gpu_ctx, prof, seqs, model, list(range(len(recs))))
    nmatched = 0
    for r, (rec, m) in enumerate(zip(recs, model)):
        rc, want = hs.align(m, p7.digitize(rec[2]))
        assert len(got[r]) == profs[m].M
        if rc != 0:        # posterior decoding left the float range (two strong copies under a one-domain model): no column is reported
            assert rec[0].endswith("_3") and (got[r] == 0).all(), rec[0]
            continue
        assert (got[r] == want).all(), (rec[0], np.nonzero(got[r] != want)[0][:5])
        nmatched += int((want > 0).sum())
    assert nmatched > 0.5 * sum(profs[m].M for m in model[::4])
    prof.close(); seqs.close(); hs.close()


def test_multi_copy_markers_are_aligned_and_scored(gpu_ctx, tmp_path):
    """HmmerAligner.makeAlignmentsOfMultipleHits -> <marker>.masked.faa -> AminoAcidIdentity.run, against the same steps taken with
    the oracles (reduce oracle for which hits are kept, scan oracle for the alignment, the reference-pinned AAI arithmetic)."""
    from checkm_amd.aminoAcidIdentity import AminoAcidIdentity
    from checkm_amd.hmmerAligner import HmmerAligner
    profs, hmm, binfiles, recs_all, dat = _setup(tmp_path)
    out = tmp_path / "out"
    (out / "storage").mkdir(parents=True)
    models = MarkerGeneFinder(4).find(binfiles, str(out), DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, hmm, False, False, True)
    sets = MarkerSetParser().getMarkerSets(str(out), list(models), hmm)
    aai_dir = str(out / "storage" / "aai_qa")
    HmmerAligner(2).makeAlignmentsOfMultipleHits(str(out), hmm, DefaultValues.HMMER_TABLE_OUT, models, sets, False, DefaultValues.E_VAL, DefaultValues.LENGTH, aai_dir)
    texts = _oracle_tables(hmm, recs_all)
    omodels = {a: {"acc": a, "ga": list(m.ga) if m.ga else None, "tc": list(m.tc) if m.tc else None, "nc": list(m.nc) if m.nc else None, "leng": m.leng}
               for a, m in models["bin_0"].items()}
    hs = p7.HmmSet(hmm)
    slot = {hs.acc(i) or hs.name(i): i for i in range(hs.n)}
    nfiles = 0
    for b in range(3):
        binId = "bin_%d" % b
        mh, _gc = ro.reduce_bin(texts[b], omodels, dat, [sorted(s) for s in sets[binId].selectedMarkerSet().markerSet])
        orfs = {r[0]: r[2] for r in recs_all[b]}
        for marker, hits in mh.items():
            if len(hits) < 2:
                continue
            hits = sorted(hits, key=lambda h: h["full_e_value"], reverse=True)
            want = []
            seen = {}
            for h in hits:
                text = "".join(orfs[s][:-1] if orfs[s].endswith("*") else orfs[s] for s in h["target_name"].split("&&"))
                seen[h["target_name"]] = text
            for name, text in seen.items():
                rc, path = hs.align(slot[marker], p7.digitize(text))
                want.append(">%s&&%s\n%s\n" % (binId, name, "".join(text.upper()[i - 1] if i > 0 else "-" for i in path)))
            f = out / "storage" / "aai_qa" / binId / (marker + ".masked.faa")
            assert f.read_text() == "".join(want), (binId, marker)
            nfiles += 1
    assert nfiles >= 3
    # the top hit of every marker in every bin, one file per marker with the hit statistics in the header (qa -o 9 path)
    top_dir = out / "top"
    rp = HmmerAligner(2).makeAlignmentTopHit(str(out), hmm, DefaultValues.HMMER_TABLE_OUT, models, False, DefaultValues.E_VAL, DefaultValues.LENGTH, True, str(top_dir))
    assert sorted(rp.results) == ["bin_0", "bin_1", "bin_2"]
    want_files = {}
    for b in range(3):
        binId = "bin_%d" % b
        mh, _gc = ro.reduce_bin(texts[b], omodels, dat, [sorted(s) for s in sets[binId].selectedMarkerSet().markerSet])
        orfs = {r[0]: r[2] for r in recs_all[b]}
        for marker, hits in mh.items():
            top = sorted(hits, key=lambda h: h["full_e_value"], reverse=True)[0]
            text = "".join(orfs[s][:-1] if orfs[s].endswith("*") else orfs[s] for s in top["target_name"].split("&&"))
            rc, path = hs.align(slot[marker], p7.digitize(text))
            want_files.setdefault(marker, []).append(">%s&&%s [e-value=%.4g,score=%.1f]\n%s\n" % (binId, top["target_name"], top["full_e_value"], top["full_score"],
                                                                                                "".join(text.upper()[i - 1] if i > 0 else "-" for i in path)))
    assert len(want_files) >= 10
    for marker in models["bin_0"]:
        f = top_dir / (marker + ".masked.faa")
        if marker in want_files:
            assert f.read_text() == "".join(want_files[marker]), marker
        else:
            assert not f.exists()
    a = AminoAcidIdentity()
    a.run(0.9, str(out), str(out / "pairs.txt"))
    assert set(a.aaiMeanBinHetero) <= {"bin_0", "bin_1", "bin_2"} and len(a.aaiMeanBinHetero) >= 1
    for binId, v in a.aaiMeanBinHetero.items():
        assert 0.0 <= v <= 100.0
    hs.close(); release_scan()


def test_keep_alignment_writes_the_domain_alignments(gpu_ctx, tmp_path):
    """bKeepAlignment (checkm/markerGeneFinder.py:138-142: hmmsearch without --noali, -o <hmmerOut>): the report holds, per reported
    domain, the alignment of the envelope's optimal-accuracy path.  Checked against the domtblout rows of the same scan: the target
    line spells the ORF from ali_from to ali_to, the model line covers hmm_from..hmm_to, gaps and inserts are where the columns say."""
    from checkm_amd import hmmer as chm
    profs = synth.small_profiles(11, 12, 40, 300)
    hmm = common.hmm_file("s11", profs)
    recs = synth.make_bin(profs, 4711, n_orfs=150, dup_frac=0.6)
    faa = str(tmp_path / "b.faa")
    synth.write_fasta(faa, recs)
    out = str(tmp_path / "out")
    MarkerGeneFinder(1).find([faa], out, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, hmm, True, False, True)
    rows = list(chm.read_domtblout(os.path.join(out, "bins", "b", DefaultValues.HMMER_TABLE_OUT)))
    text = open(os.path.join(out, "bins", "b", DefaultValues.HMMER_OUT)).read().split("\n")
    seqs = {r[0]: r[2] for r in recs}
    blocks = []                                            # (query, target, model line, target line, coordinates)
    query = None
    query_at = {}
    for i, line in enumerate(text):
        if line.startswith("Query:"):
            query = line.split()[1]
        query_at[i] = query
        if line.startswith("  == domain"):
            ml, tl = text[i + 1].split(), text[i + 3].split()
            blocks.append((query, tl[0], ml[2], tl[2], int(ml[1]), int(ml[3]), int(tl[1]), int(tl[3])))
    assert len(blocks) == len(rows) and len(rows) >= 12
    # the PP line of every block against the oracle: posterior probability of each aligned residue in its emitting state, hmmsearch's code
    hs = p7.HmmSet(hmm)
    index_of = {hs.name(m): m for m in range(hs.n)}
    pp_lines = {}
    for i, line in enumerate(text):
        if line.startswith("  == domain"):
            f = text[i + 4].split()
            assert len(f) == 2 and f[1] == "PP", text[i + 4]
            ml, tl = text[i + 1].split(), text[i + 3].split()
            pp_lines[(query_at[i], tl[0], int(ml[1]), int(ml[3]), int(tl[1]), int(tl[3]))] = f[0]

    def code(v):
        v = float(v) + 0.05
        return '*' if v >= 1.0 else chr(int(v * 10.0) + 48)
    stars = digits = dots = 0
    for h in rows:
        rc, path, pp = hs.envelope_alignment(index_of[h.query_name], p7.digitize(seqs[h.target_name]), h.env_from, h.env_to)
        assert rc == 0
        want, prev = "", 0
        for k in range(h.hmm_from, h.hmm_to + 1):
            pr = int(path[k - 1])
            if pr == 0:
                want += "."
                continue
            if prev:
                want += "".join(code(pp[j]) for j in range(prev + 1, pr))
            want += code(pp[pr])
            prev = pr
        got = pp_lines[(h.query_name, h.target_name, h.hmm_from, h.hmm_to, h.ali_from, h.ali_to)]
        assert got == want, (h.query_name, h.target_name)
        stars += want.count("*"); dots += want.count("."); digits += sum(c.isdigit() for c in want)
    assert stars > 100 and digits > 20                      # (confident cores and uncertain edges both occur)
    seen = set()
    for h in rows:
        hit = [b for b in blocks if b[0] == h.query_name and b[1] == h.target_name and (b[4], b[5], b[6], b[7]) == (h.hmm_from, h.hmm_to, h.ali_from, h.ali_to)]
        assert len(hit) >= 1, (h.query_name, h.target_name)
        _q, _t, ml, tl, hf, ht, af, at = hit[0]
        seen.add(id(hit[0]))
        assert len(ml) == len(tl)
        assert tl.replace("-", "").upper() == seqs[h.target_name][af - 1:at]                  # the aligned stretch of the ORF, inserts in lower case
        assert sum(1 for c in ml if c != ".") == ht - hf + 1                                    # every node from hmm_from to hmm_to once
        assert all((a == ".") == b.islower() for a, b in zip(ml, tl) if b != "-")              # inserts: '.' above a lower-case residue
        assert tl[0] != "-" and tl[-1] != "-" and ml[0] != "." and ml[-1] != "."              # an alignment begins and ends on a match state
    release_scan()
