"""CPU sanity of the scan-half oracle (parity is UNPINNED against HMMER: none exists here).  These
check internal consistency and the properties the domain offers."""
import numpy as np
import pytest

from synthdata import synth
from oracle import p7
from tests import common


@pytest.fixture(scope="module")
def hs():
    profs = synth.small_profiles(7, 6, 20, 150)
    h = p7.HmmSet(common.hmm_file("small7", profs))
    yield h, profs
    h.close()


def test_filter_pass_rates_are_calibrated(hs):
    h, profs = hs
    rng = np.random.default_rng(1)
    n = npass = 0
    for i in range(h.n):
        for _ in range(150):
            st = h.stages(i, synth.random_residues(rng, int(rng.integers(60, 400))).astype(np.uint8))
            n += 1
            npass += st.pass_msv
    assert 0.003 < npass / n < 0.06      # F1 = 0.02 on random sequence


def test_planted_domains_are_found_with_right_coordinates(hs):
    h, profs = hs
    rng = np.random.default_rng(2)
    for i, p in enumerate(profs):
        dom = synth.sample_domain(rng, p)
        fl, fr = synth.random_residues(rng, 25), synth.random_residues(rng, 31)
        seq = np.concatenate([fl, dom, fr]).astype(np.uint8)
        rows = h.search([i], [seq], ["s_1"])
        assert len(rows) == 1
        r = rows[0]
        assert r.tlen == len(seq) and r.qlen == p.M
        assert abs(r.ali_from - 26) <= 6 and abs(r.ali_to - (25 + len(dom))) <= 6
        assert r.hmm_from <= 5 and r.hmm_to >= p.M - 4
        assert r.env_from <= r.ali_from <= r.ali_to <= r.env_to
        assert r.full_evalue < 1e-5 and 0.5 < r.acc <= 1.0


def test_evalue_scales_with_database_size(hs):
    """E = P * Z: the same hit among twice as many targets has twice the E-value, same score."""
    h, profs = hs
    rng = np.random.default_rng(3)
    seq = np.concatenate([synth.random_residues(rng, 10), synth.sample_domain(rng, profs[0]), synth.random_residues(rng, 10)]).astype(np.uint8)
    junk = [synth.random_residues(rng, 80).astype(np.uint8) for _ in range(7)]
    a = h.search([0], [seq] + junk[:3], ["t"] + ["j%d" % i for i in range(3)])[0]
    b = h.search([0], [seq] + junk, ["t"] + ["j%d" % i for i in range(7)])[0]
    assert a.full_score == b.full_score and a.ali_from == b.ali_from
    assert b.full_evalue == pytest.approx(2 * a.full_evalue, rel=1e-12)


def test_empty_and_degenerate_sequences(hs):
    h, _ = hs
    assert h.search([0, 1], [], []) == []
    x = p7.digitize("XXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXXX*")
    st = h.stages(0, x)
    assert np.isfinite(st.msv_sc) and np.isfinite(st.fwd_sc)
    assert h.search([0], [x], ["x_1"]) == []
    assert list(p7.digitize("acdX*-~bjzou")) == [0, 1, 2, 26, 27, 20, 28, 21, 22, 23, 24, 25]


def test_domtblout_text_round_trips_through_the_reduce_oracle(hs):
    from oracle import reduce_oracle as ro
    h, profs = hs
    recs = synth.make_bin(profs, 77, n_orfs=80, dup_frac=0.5)
    rows = h.search(list(range(h.n)), [p7.digitize(r[2]) for r in recs], [r[0] for r in recs])
    text = h.format_domtblout(rows, [r[0] for r in recs], [r[1] for r in recs])
    parsed = ro.parse_domtblout(text)
    assert len(parsed) == len(rows) and len(rows) >= 6
    for r, q in zip(rows, parsed):
        assert q["target_name"] == recs[r.seq_idx][0] and q["query_length"] == r.qlen
        assert (q["hmm_from"], q["hmm_to"], q["ali_from"], q["ali_to"], q["env_from"], q["env_to"]) == (r.hmm_from, r.hmm_to, r.ali_from, r.ali_to, r.env_from, r.env_to)
        assert q["full_score"] == float("%.1f" % r.full_score) and q["full_e_value"] == float("%.2g" % r.full_evalue)
        assert q["target_description"] == recs[r.seq_idx][1]


def test_oracle_output_is_pinned_against_its_own_record():
    """Regression pin (NOT a reference vector, see tools/gen_oracle_selfcheck.py): the HIP path is tested against the oracle and
    would follow an accidental change of the restatement silently; this catches it on the CPU."""
    import json
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tools"))
    import gen_oracle_selfcheck as g
    want = json.load(open(os.path.join(root, "tests", "golden", "oracle_selfcheck.json")))
    got = json.loads(json.dumps(g.build()))
    for k in want:
        assert got[k] == want[k], k
    assert got["ensemble"]["rc"] == 0 and len(got["ensemble"]["env"]) >= 2 and got["nrows"] >= 8
