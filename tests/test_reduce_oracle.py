"""Pins the reduce-half oracle (oracle/reduce_oracle.py) against goldens produced by the REFERENCE's
own classes (tools/gen_reduce_golden.py, run in the build container where /root/reference exists)."""
import json
import os

import pytest

from oracle import reduce_oracle as ro

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reduce_cases.json")


@pytest.fixture(scope="module")
def cases():
    with open(GOLD) as f:
        return json.load(f)["cases"]


def models_of(case):
    return {m["acc"]: {"acc": m["acc"], "ga": m["ga"], "tc": m["tc"], "nc": m["nc"], "leng": m["leng"]} for m in case["models"]}


def test_oracle_matches_reference_goldens(cases):
    n = 0
    for ci, case in enumerate(cases):
        for run in case["runs"]:
            fl = run["flags"]
            for indiv, key in ((False, "geneCounts"), (True, "geneCountsIndividual")):
                mh, gc = ro.reduce_bin(case["domtblout"], models_of(case), case["pfam_dat"], case["marker_sets"],
                                       fl.get("ignore_thresholds", False), fl.get("evalue", 1e-10), fl.get("length", 0.7),
                                       fl.get("skip_pseudogene", False), fl.get("skip_adj", False), indiv)
                assert ro.marker_hits_view(mh) == run["expected"]["markerHits"], (ci, fl)
                assert gc == run["expected"][key], (ci, fl, key, gc, run["expected"][key])
                n += 1
    assert n == len(cases) * 6 * 2 and len(cases) >= 65


def test_goldens_exercise_the_quirks(cases):
    """The fixture must actually contain merged ORFs, clan drops and threshold rejections."""
    merged = sum(1 for c in cases for r in c["runs"] for k, hl in r["expected"]["markerHits"] for h in hl if "&&" in h[0])
    assert merged > 10
    default = [r["expected"]["markerHits"] for c in cases for r in c["runs"] if r["flags"] == {}]
    ignore = [r["expected"]["markerHits"] for c in cases for r in c["runs"] if r["flags"] == {"ignore_thresholds": True}]
    assert default != ignore
