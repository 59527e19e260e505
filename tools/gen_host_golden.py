#!/usr/bin/env python
"""Goldens for the host logic around the hot path, produced by the REFERENCE's own classes
(imported read-only from /root/reference): the sticky header parse (hmmerModelParser.py:54-83),
marker-file parsing + exclusion (markerSets.py:248-297,478-522), binIdFromFilename, and AminoAcidIdentity.aai / strainHetero /
run (aminoAcidIdentity.py:39-161) together with HmmerAligner._extractSeq (hmmerAligner.py:407-426).
Writes tests/golden/host_cases.json (committed).  Run here only."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = tempfile.mkdtemp(prefix="ckm_refdata_")
os.environ["CHECKM_DATA_PATH"] = DATA
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

from checkm.hmmerModelParser import HmmModelParser  # noqa: E402
from checkm.markerSets import MarkerSetParser  # noqa: E402
from checkm.common import binIdFromFilename  # noqa: E402
from synthdata import synth  # noqa: E402


def hmm_text():
    import numpy as np
    rng = np.random.default_rng(99)
    profs = []
    spec = [("modA", "PF00001.1", (20.0, 19.0), (21.0, 20.0), (18.0, 17.0)),
            ("modB", None, None, None, None),                # inherits ACC and all cutoffs of modA
            ("TIGR00010", "TIGR00010", None, (30.0, 30.0), (25.0, 25.0)),   # inherits GA of modA
            ("modD", "PF00002.7", (11.0, 10.5), None, None), # inherits TC/NC of the TIGR model
            ("modE", None, None, None, None)]
    for name, acc, ga, tc, nc in spec:
        p = synth.random_profile(rng, 12, name, acc)
        p.ga, p.tc, p.nc = ga, tc, nc
        p.stats = (-8.0, 0.7, -9.0, 0.7, -3.5, 0.7)
        profs.append(p)
    path = os.path.join(DATA, "sticky.hmm")
    synth.write_hmm(path, profs)
    return path, open(path).read()


def main():
    out = {}
    path, text = hmm_text()
    models = HmmModelParser(path).models()
    out["sticky"] = {"hmm_text": text,
                     "models": {k: {"name": m.name, "acc": m.acc, "leng": m.leng, "ga": m.ga, "tc": m.tc, "nc": m.nc} for k, m in models.items()}}
    # taxon marker file
    taxon = ("# [Taxon Marker File]\n"
             "Bacteria\t2\t2\tk__Bacteria\t5449\t[{'PF00380.14', 'PF00410.14'}, {'TIGR00967'}, {'TIGR00398'}, {'TIGR00399', 'PF03719.10'}]"
             "\t0\troot\t5656\t[{'PF00380.14'}, {'TIGR00398', 'TIGR00399'}]\n")
    tf = os.path.join(DATA, "taxon.ms")
    open(tf, "w").write(taxon)
    excl = os.path.join(DATA, "exclude.txt")
    open(excl, "w").write("# comment\nPF00410.14\n")
    msp = MarkerSetParser()
    res = {}
    for tag, ex in (("default", None), ("user_exclude", excl)):
        b = msp.getMarkerSets(DATA, ["binA", "binB"], tf, ex)
        res[tag] = {bid: {"selected_uid": bm.selectedMarkerSet().UID, "sets": [[ms.UID, ms.lineageStr, ms.numGenomes, [sorted(s) for s in ms.markerSet]] for ms in bm.markerSets]}
                    for bid, bm in b.items()}
    out["taxon"] = {"file": taxon, "exclude_file": open(excl).read(), "result": res}
    # lineage marker file
    open(os.path.join(DATA, "selected_marker_sets.tsv"), "w").write("10\t11\n11\t12\n12\t12\n20\t20\n")
    lineage = ("# [Lineage Marker File]\n"
               "binA\t3\t10\tk__Bacteria;p__X\t20\t[{'PF00001.1'}]\t12\tk__Bacteria\t300\t[{'PF00001.1', 'TIGR00001'}, {'PF00002.2'}]\t0\troot\t5000\t[{'PF00003.3'}]\n"
               "binB\t2\t20\tk__Archaea\t40\t[{'TIGR00002', 'TIGR00398'}]\t0\troot\t5000\t[{'PF00003.3'}]\n")
    lf = os.path.join(DATA, "lineage.ms")
    open(lf, "w").write(lineage)
    b = msp.getMarkerSets(DATA, ["binA", "binB"], lf)
    out["lineage"] = {"file": lineage, "selected_map": "10\t11\n11\t12\n12\t12\n20\t20\n",
                      "result": {bid: {"selected_uid": bm.selectedMarkerSet().UID,
                                       "sets": [[ms.UID, ms.lineageStr, ms.numGenomes, [sorted(s) for s in ms.markerSet]] for ms in bm.markerSets]} for bid, bm in b.items()}}
    # amino-acid identity of masked alignments: hand-made edge cases + random gapped pairs, through the reference's own class
    import random
    from checkm.aminoAcidIdentity import AminoAcidIdentity
    from checkm.hmmerAligner import HmmerAligner
    aai = AminoAcidIdentity()
    rnd = random.Random(7)
    pairs = [("--ACDE--", "-AACDEF-"), ("ACDE", "ACDF"), ("----", "----"), ("A-", "AC"), ("-A", "CA"), ("A", "A"), ("A", "-"), ("-CDE", "ACD-"),
             ("AC--DE", "AC--DF"), ("AC-DE", "ACWDE"), ("----A", "ACDEA"), ("A----", "A-CDE")]
    for _ in range(60):
        n = rnd.randint(1, 40)
        a = "".join(rnd.choice("ACDEFGHIK-") if rnd.random() < 0.8 else "-" for _ in range(n))
        b = "".join((c if rnd.random() < 0.7 else rnd.choice("ACDEFGHIK-")) for c in a)
        pairs.append((a, b))
    out["aai"] = [[a, b, aai.aai(a, b)] for a, b in pairs]
    scores = {"bin1": {"PF1": [0.95, 0.5, 1.0], "TIGR2": [0.2]}, "bin2": {"PF1": [0.9]}, "bin3": {"M": [0.9, 0.9000001, 0.89]}}
    out["strain"] = []
    for thr in (0.9, 0.5, 0.0):
        het, mean = aai.strainHetero(scores, thr)
        out["strain"].append([scores, thr, {k: dict(v) for k, v in het.items()}, mean])
    # run(): masked files on disk -> raw scores, heterogeneity, pair report
    od = os.path.join(DATA, "aai_run")
    files = {"binA": {"PF00318.15.masked.faa": ">binA&&c1_1\nACD-EF\n>binA&&c1_7&&c1_8\nACDWEF\n>binA&&c2_3\n-CDWE-\n", "notes.txt": "x"},
             "binB": {"TIGR00001.masked.faa": ">binB&&k_1\nAAAA\n>binB&&k_2\nAAAC\n"}}
    for b, fs in files.items():
        os.makedirs(os.path.join(od, "bins", b), exist_ok=True)
        os.makedirs(os.path.join(od, "storage", "aai_qa", b), exist_ok=True)
        for fn, txt in fs.items():
            open(os.path.join(od, "storage", "aai_qa", b, fn), "w").write(txt)
    os.makedirs(os.path.join(od, "bins", "binC"), exist_ok=True)           # a bin without multi-copy markers
    r = AminoAcidIdentity()
    rep = os.path.join(od, "pairs.txt")
    r.run(0.9, od, rep)
    out["aai_run"] = {"files": files, "raw": {b: {m: v for m, v in d.items()} for b, d in r.aaiRawScores.items()},
                      "hetero": {b: dict(d) for b, d in r.aaiHetero.items()}, "mean": r.aaiMeanBinHetero,
                      "report_sorted_blocks": sorted(open(rep).read().strip().split("\n\n"))}
    ha = HmmerAligner(1)
    orfs = {"c1_1": "MKV*", "c1_2": "ACD", "c9_5": "WWW*"}
    out["extract_seq"] = [[sid, ha._extractSeq(sid, orfs)] for sid in ("c1_1", "c1_2", "c1_1&&c1_2", "c1_2&&c9_5&&c1_1")]
    # row-at-a-time rules of the reduce half: ResultsManager.vetHit (resultsParser.py:340-377) and PFAM.filterHitsFromSameClan
    # (util/pfam.py:86-147) on random hits; hits are recorded as plain dicts, results as indices into the hit list
    from checkm.hmmer import HmmerHitDOM
    from checkm.hmmerModelParser import HmmModel
    from checkm.resultsParser import ResultsManager
    from checkm.util.pfam import PFAM
    os.makedirs(os.path.join(DATA, "pfam"), exist_ok=True)
    dat = ("# STOCKHOLM 1.0\n#=GF ID   famA\n#=GF AC   PF00001.3\n#=GF CL   CL0001\n#=GF NE   famC\n//\n"
           "# STOCKHOLM 1.0\n#=GF ID   famB\n#=GF AC   PF00002.1\n#=GF CL   CL0001\n//\n"
           "# STOCKHOLM 1.0\n#=GF ID   famC\n#=GF AC   PF00003.9\n#=GF CL   CL0001\n//\n"
           "# STOCKHOLM 1.0\n#=GF ID   famD\n#=GF AC   PF00004.2\n//\n"
           "# STOCKHOLM 1.0\n#=GF ID   famE\n#=GF AC   PF00005.2\n//\n")
    dat_path = os.path.join(DATA, "pfam", "Pfam-A.hmm.dat")
    open(dat_path, "w").write(dat)
    model_spec = {"PF00001.3": dict(ga=(25.0, 20.0)), "PF00002.1": dict(tc=(30.0, 30.0)), "PF00003.9": dict(), "PF00004.2": dict(nc=(22.0, 21.5)),
                  "PF00005.2": dict(ga=(10.0, 10.0), nc=(50.0, 50.0)), "TIGR00011": dict(ga=(10.0, 10.0), nc=(40.0, 35.0)), "TIGR00012": dict(tc=(33.3, 33.3))}
    models = {}
    for acc, cut in model_spec.items():
        m = HmmModel({"acc": acc, "name": "n_" + acc, "leng": 100})
        m.ga, m.tc, m.nc = cut.get("ga"), cut.get("tc"), cut.get("nc")
        models[acc] = m
    rnd = random.Random(21)
    accs = sorted(model_spec)
    hit_dicts = []
    for k in range(260):
        acc = rnd.choice(accs)
        a0 = rnd.randint(1, 120); a1 = a0 + rnd.randint(0, 110)
        fs = round(rnd.choice([9.9, 10.0, 20.0, 21.5, 22.0, 25.0, 30.0, 33.3, 35.0, 40.0, 50.0, rnd.uniform(0, 60)]), 1)
        ds = round(rnd.choice([fs, fs - 0.1, 10.0, 20.0, 21.5, 30.0, 33.3, 35.0, rnd.uniform(0, 60)]), 1)
        ev = rnd.choice([0.0, 1e-30, 1e-10, 1.1e-10, 9.9e-11, 1e-5])
        hit_dicts.append({"target_name": "c%d_%d" % (rnd.randint(1, 3), rnd.randint(1, 6)), "target_accession": "-", "target_length": 300, "query_name": "n_" + acc,
                          "query_accession": acc, "query_length": rnd.choice([100, 150, 37]), "full_e_value": ev, "full_score": fs, "full_bias": 0.0, "dom": 1, "ndom": 1,
                          "c_evalue": ev, "i_evalue": rnd.choice([ev, ev * 10, 1e-3]), "dom_score": ds, "dom_bias": 0.0, "hmm_from": 1, "hmm_to": 90, "ali_from": a0,
                          "ali_to": a1, "env_from": a0, "env_to": a1, "acc": 0.9, "target_description": "-"})
    order = ["target_name", "target_accession", "target_length", "query_name", "query_accession", "query_length", "full_e_value", "full_score", "full_bias", "dom", "ndom",
             "c_evalue", "i_evalue", "dom_score", "dom_bias", "hmm_from", "hmm_to", "ali_from", "ali_to", "env_from", "env_to", "acc", "target_description"]

    def mk(d):
        return HmmerHitDOM([str(d[k]) for k in order])
    vet = []
    for flags in ((False, False), (True, False), (False, True), (True, True)):
        rm = ResultsManager("b", models, bIgnoreThresholds=flags[0], evalueThreshold=1e-10, lengthThreshold=0.7, bSkipPseudoGeneCorrection=flags[1])
        vet.append({"bIgnoreThresholds": flags[0], "bSkipPseudoGeneCorrection": flags[1], "verdicts": [bool(rm.vetHit(mk(d))) for d in hit_dicts]})
    clan_cases = []
    for trial in range(12):
        chosen = [rnd.randrange(len(hit_dicts)) for _ in range(rnd.randint(3, 40))]
        mh = {}
        objs = {}
        for idx in chosen:
            h = mk(hit_dicts[idx]); objs[id(h)] = idx
            mh.setdefault(h.query_accession, []).append(h)
        res = PFAM(dat_path).filterHitsFromSameClan(mh)
        clan_cases.append({"input": {k: [objs[id(h)] for h in v] for k, v in mh.items()}, "kept": {k: [objs[id(h)] for h in v] for k, v in res.items()}})
    out["rules"] = {"pfam_dat": dat, "models": {a: {"ga": m.ga, "tc": m.tc, "nc": m.nc} for a, m in models.items()}, "hits": hit_dicts, "hit_field_order": order,
                    "vetHit": vet, "filterHitsFromSameClan": clan_cases}
    names = ["/a/b/bin.1.fna", "x.fa.gz", "genome.faa", "noext", "a.b.c.gz", "dir.d/file"]
    out["binIdFromFilename"] = {n: binIdFromFilename(n) for n in names}
    with open(os.path.join(ROOT, "tests", "golden", "host_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote host_cases.json", list(out))


if __name__ == "__main__":
    main()
