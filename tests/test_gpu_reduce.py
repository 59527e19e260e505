"""GPU parity of the reduce half: libcheckm_hip's ckm_reduce / ckm_count_sets (through the Python
mirror classes) against the goldens produced by the reference's own classes.
Bar: integer-identical hit lists and histograms; completeness/contamination bit-identical float64."""
import ast
import json
import os

import pytest

from checkm_amd import _lib, qa as cqa
from synthdata import synth
from checkm_amd.defaultValues import DefaultValues
from checkm_amd.hmmerModelParser import HmmModel
from checkm_amd.markerSets import MarkerSet
from checkm_amd.resultsParser import ResultsManager, ResultsParser
from oracle import reduce_oracle as ro
from tests import common

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reduce_cases.json")


def _view(mh):
    return [[k, [[h.target_name, h.target_length, h.hmm_from, h.hmm_to, h.ali_from, h.ali_to, h.env_from, h.env_to, h.dom_score, h.full_e_value]
                 for h in v]] for k, v in mh.items()]


def test_reduce_matches_reference_goldens(gpu_ctx, tmp_path):
    with open(GOLD) as f:
        cases = json.load(f)["cases"]
    root = tmp_path / "data"
    (root / "pfam").mkdir(parents=True)
    DefaultValues.set_data_root(str(root))
    for ci, case in enumerate(cases):
        (root / "pfam" / "Pfam-A.hmm.dat").write_text(case["pfam_dat"])
        tab = tmp_path / "t.txt"
        tab.write_text(case["domtblout"])
        models = {}
        for m in case["models"]:
            hm = HmmModel({"name": m["name"], "acc": m["acc"], "leng": m["leng"]})
            hm.ga = tuple(m["ga"]) if m["ga"] else None
            hm.tc = tuple(m["tc"]) if m["tc"] else None
            hm.nc = tuple(m["nc"]) if m["nc"] else None
            models[m["acc"]] = hm
        ms = MarkerSet(0, "k__Bacteria", 10, [set(s) for s in case["marker_sets"]])
        for run in case["runs"]:
            fl = run["flags"]
            rm = ResultsManager("bin", models, fl.get("ignore_thresholds", False), fl.get("evalue", DefaultValues.E_VAL),
                                fl.get("length", DefaultValues.LENGTH), fl.get("skip_pseudogene", False))
            ResultsParser({"bin": models}).parseHmmerResults(str(tab), rm, fl.get("skip_adj", False))
            assert _view(rm.markerHits) == run["expected"]["markerHits"], (ci, fl)
            assert rm.geneCounts(ms, rm.markerHits, False) == run["expected"]["geneCounts"], (ci, fl)
            assert rm.geneCounts(ms, rm.markerHits, True) == run["expected"]["geneCountsIndividual"], (ci, fl)


def test_resident_hits_reduce_like_the_text_path(gpu_ctx, tmp_path):
    """Scan -> packed hits -> ckm_reduce must equal scan -> domtblout text -> (oracle) reduce."""
    profs = common.mixed_profiles()
    path = common.hmm_file("mixed", profs)
    bins = [synth.make_bin(profs, 3000 + b, n_orfs=200, dup_frac=0.9) for b in range(3)]
    prof = _lib.Profiles(gpu_ctx, path)
    seqs = _lib.Seqs(gpu_ctx, bins)
    hits = _lib.search(gpu_ctx, prof, seqs)
    plan = cqa.QAPlan.for_hmm_models(prof, [list(range(prof.n))] * len(bins))
    models = {}
    for hd in prof.headers:
        a = hd["acc"] or hd["name"]
        models[a] = {"acc": a, "ga": list(hd["ga"]) if hd["ga"] else None, "tc": list(hd["tc"]) if hd["tc"] else None,
                     "nc": list(hd["nc"]) if hd["nc"] else None, "leng": hd["leng"]}
    for flags in (dict(), dict(ignore_thresholds=True), dict(skip_adj=True), dict(individual_markers=True)):
        res = plan.reduce(gpu_ctx, hits, seqs, flags.get("ignore_thresholds", False), 1e-10, 0.7, False, flags.get("skip_adj", False),
                          flags.get("individual_markers", False))
        nmerged = 0
        for b in range(len(bins)):
            t = str(tmp_path / ("b%d.txt" % b))
            hits.write_domtblout(prof, seqs, b, t)
            sets = [sorted(models.keys())]
            mh, gc = ro.reduce_bin(open(t).read(), models, "", sets, flags.get("ignore_thresholds", False), 1e-10, 0.7, False,
                                   flags.get("skip_adj", False), flags.get("individual_markers", False))
            assert list(res.hist[b]) == gc[:6], (flags, b)
            assert res.completeness[b] == gc[6] and res.contamination[b] == gc[7], (flags, b, res.completeness[b], gc[6])
            want = [(k, h['target_name'].split('&&'), h['ali_from'], h['ali_to']) for k, v in mh.items() for h in v]
            got = []
            for i in range(int(res.kept_bin_off[b]), int(res.kept_bin_off[b + 1])):
                names = [seqs.names[int(hits.seq[int(res.kept_row[i])])]]
                if int(res.kept_row2[i]) != 0xFFFFFFFFFFFFFFFF:
                    names.append(seqs.names[int(hits.seq[int(res.kept_row2[i])])]); nmerged += 1
                got.append((plan.keys.names[int(res.kept_key[i])], sorted(names), int(res.kept_ali_from[i]), int(res.kept_ali_to[i])))
            assert got == want, (flags, b)
        res.close()
    hits.close(); prof.close(); seqs.close()


def test_summary_formats_match_reference_text(gpu_ctx, tmp_path):
    """ResultsParser.printSummary in tab mode, output formats 1-9, with and without --individual_markers, byte for byte against the
    text the reference's own classes printed for the same tables (tools/gen_summary_golden.py).  Format 4 lists the markers in the
    iteration order of a Python set of strings (not reproducible between processes): compared as marker -> count maps."""
    from checkm_amd.markerSets import BinMarkerSets
    with open(GOLD) as f:
        rcases = json.load(f)["cases"]
    with open(os.path.join(os.path.dirname(GOLD), "summary_cases.json")) as f:
        scases = json.load(f)["cases"]

    class FakeAAI(object):
        aaiMeanBinHetero = {"binA": 12.5}
    root = tmp_path / "data"
    (root / "pfam").mkdir(parents=True)
    DefaultValues.set_data_root(str(root))
    nok = nraise = 0
    for sc in scases:
        case = rcases[sc["reduce_case"]]
        (root / "pfam" / "Pfam-A.hmm.dat").write_text(case["pfam_dat"])
        work = tmp_path / ("w%d" % sc["reduce_case"])
        models = {}
        for m in case["models"]:
            hm = HmmModel({"name": m["name"], "acc": m["acc"], "leng": m["leng"]})
            hm.ga = tuple(m["ga"]) if m["ga"] else None
            hm.tc = tuple(m["tc"]) if m["tc"] else None
            hm.nc = tuple(m["nc"]) if m["nc"] else None
            models[m["acc"]] = hm
        binIds = ["binB", "binA"]
        rp = ResultsParser({b: models for b in binIds})
        bms = {}
        for k, b in enumerate(binIds):
            rm = ResultsManager(b, models, False, DefaultValues.E_VAL, DefaultValues.LENGTH, False, sc["bin_stats"][b])
            (work / "bins" / b).mkdir(parents=True)
            t = work / (b + ".txt")
            t.write_text(case["domtblout"])
            rp.parseHmmerResults(str(t), rm, k == 1)
            rp.results[b] = rm
            s = BinMarkerSets(b, BinMarkerSets.TAXONOMIC_MARKER_SET)
            s.addMarkerSet(MarkerSet(7 + k, "k__Bacteria;p__Test", 100 + k, [set(x) for x in sc["marker_sets"]]))
            s.addMarkerSet(MarkerSet(0, "root", 5000, [set(x) for x in sc["marker_sets"][:1]]))
            bms[b] = s
            (work / "bins" / b / "genes.faa").write_text(sc["genes_faa"][b])
        for key, want in sc["outputs"].items():
            fmt, indiv = int(key.split("_")[0]), key.endswith("_1")
            of = str(work / ("out_%s.txt" % key))
            if isinstance(want, dict):
                with pytest.raises(Exception) as e:
                    rp.printSummary(fmt, FakeAAI(), bms, indiv, None, True, of, str(work))
                assert type(e.value).__name__ == want["raises"], (sc["reduce_case"], key)
                nraise += 1
                continue
            rp.printSummary(fmt, FakeAAI(), bms, indiv, None, True, of, str(work))
            got = open(of).read()
            if fmt == 4:
                def as_maps(txt):
                    blocks = [b for b in txt.split("\n\n") if b.strip()]
                    out = []
                    for blk in blocks:
                        h, r = blk.strip("\n").split("\n")
                        hp, rv = h.split("\t"), r.split("\t")
                        out.append((hp[0], rv[0], dict(zip(hp[1:], rv[1:]))))
                    return out
                assert as_maps(got) == as_maps(want), (sc["reduce_case"], key)
            else:
                assert got == want, (sc["reduce_case"], key, got[:300], want[:300])
            nok += 1
        # the two caches `qa` leaves behind and the dictionaries they are made of (resultsParser.py:121-143, 567-676)
        (work / "storage").mkdir()
        rp.cacheResults(str(work), bms, False)
        for name, want in sc["caches"].items():
            assert (work / "storage" / name).read_text() == want, (sc["reduce_case"], name)
        for b in binIds:
            for f, want in sc["summaries"][b].items():
                # compared as dictionaries: where the reference inserts markers without hits follows set hashing
                got = rp.results[b].getSummary(bms[b], False, outputFormat=int(f))
                assert got == ast.literal_eval(want), (sc["reduce_case"], b, f)
    assert nok >= 80 and nraise >= 1
