#!/usr/bin/env python
"""Goldens for the host logic around the hot path, produced by the REFERENCE's own classes
(imported read-only from /root/reference): the sticky header parse (hmmerModelParser.py:54-83),
marker-file parsing + exclusion (markerSets.py:248-297,478-522) and binIdFromFilename.
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
from checkm_amd import synth  # noqa: E402


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
    names = ["/a/b/bin.1.fna", "x.fa.gz", "genome.faa", "noext", "a.b.c.gz", "dir.d/file"]
    out["binIdFromFilename"] = {n: binIdFromFilename(n) for n in names}
    with open(os.path.join(ROOT, "tests", "golden", "host_cases.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote host_cases.json", list(out))


if __name__ == "__main__":
    main()
