"""The scan oracle against BRUTE-FORCE PATH ENUMERATION of the HMMER3 local profile, on models and targets small enough to list
every state path.  The enumeration below is written from the published definition of the search profile (Eddy 2008/2011;
HMMER User Guide, "the Plan 7 profile": N/C/J loops L/(L+3) [L/(L+2) unihit], local entry B->Mk = occ_k / sum_j occ_j (M-j+1),
local exit Mk->E = Dk->E = 1, E->C = E->J = 1/2 [E->C = 1 unihit], insert odds ratio 1, null1 p1 = L/(L+1)) and shares no code
with oracle/p7oracle.c -- it is an independent pin of the restatement's Forward scores (the scan oracle has no HMMER vectors to
be pinned against: DESIGN.md section 2).  CPU only."""
import math

import numpy as np
import pytest

from synthdata import synth
from oracle import p7


def _profile(rng, M, k):
    p = synth.random_profile(rng, M, "tiny%d" % k, "PF9%04d.1" % k)
    for j in range(0, M + 1):          # indel probabilities large enough that insert and delete paths carry weight
        mi, md, ii, dd = rng.uniform(0.05, 0.2), rng.uniform(0.05, 0.25), rng.uniform(0.2, 0.6), rng.uniform(0.2, 0.6)
        p.t[j] = [1 - mi - md, mi, md, 1 - ii, ii, 1 - dd, dd]
    p.t[0, 5:] = [1.0, 0.0]
    p.t[M, 0:3] = [1 - p.t[M, 1], p.t[M, 1], 0.0]
    p.t[M, 5:] = [1.0, 0.0]
    p.stats = (-30.0, 0.71, -30.0, 0.71, -30.0, 0.70)      # so permissive that a five-residue target clears every filter
    return p


def _read_back(path):
    """Probabilities as the FILE states them (5 decimals of -ln p): the enumeration must see what the oracle read."""
    models, lines = [], open(path).read().split("\n")
    i = 0
    while i < len(lines):
        if lines[i].startswith("LENG"):
            M = int(lines[i].split()[1])
        if lines[i].startswith("HMM "):
            i += 2
            if lines[i].split()[0] == "COMPO":
                i += 1
            pr = lambda toks: [0.0 if t == "*" else math.exp(-float(t)) for t in toks]
            t = [pr(lines[i + 1].split())]
            mat = [None]
            i += 2
            for k in range(1, M + 1):
                mat.append(pr(lines[i].split()[1:21]))
                t.append(pr(lines[i + 2].split()))
                i += 3
            models.append((M, mat, t))
        i += 1
    return models


def _enumerate(M, mat, t, x, Lcfg, multihit, free_loops=False, ungapped_uniform=False):
    """Sum and maximum over ALL state paths of the odds-ratio product for residues x (codes 0..19); returns (total, best, #paths).
    free_loops: N/C/J self-loops cost nothing (the approximation both integer filters make, repaired by a flat -3 nats);
    ungapped_uniform: the MSV model -- uniform entry 2/(M(M+1)), match-to-match for free, no inserts or deletes."""
    L = len(x)
    nj = 1.0 if multihit else 0.0
    move = (2.0 + nj) / (Lcfg + 2.0 + nj)
    loop = 1.0 if free_loops else 1.0 - move
    eC, eJ = (0.5, 0.5) if multihit else (1.0, 0.0)
    MM, MI, MD, IM, II, DM, DD = range(7)
    occ = [0.0] * (M + 1)
    occ[1] = t[0][MI] + t[0][MM]
    for k in range(2, M + 1):
        occ[k] = occ[k - 1] * (t[k - 1][MM] + t[k - 1][MI]) + (1.0 - occ[k - 1]) * t[k - 1][DM]
    Z = sum(occ[k] * (M - k + 1) for k in range(1, M + 1))
    entry = [0.0] + [occ[k] / Z for k in range(1, M + 1)]
    if ungapped_uniform:
        entry = [0.0] + [2.0 / (M * (M + 1.0))] * M
        t = [[1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0] for _ in range(M + 1)]
    e = lambda k, i: mat[k][x[i]] / synth.BGF[x[i]]
    acc = {"total": 0.0, "best": 0.0, "paths": 0}

    def N(i, w):
        if i < L:
            N(i + 1, w * loop)
        B(i, w * move)

    def B(i, w):
        if i < L:
            for k in range(1, M + 1):
                Mk(k, i + 1, w * entry[k] * e(k, i))

    def Mk(k, i, w):                       # M_k has just emitted residue i-1
        E(i, w)
        if k < M:
            if i < L:
                Mk(k + 1, i + 1, w * t[k][MM] * e(k + 1, i))
                Ik(k, i + 1, w * t[k][MI])
            Dk(k + 1, i, w * t[k][MD])

    def Ik(k, i, w):
        if w == 0.0:
            return
        if i < L:
            Mk(k + 1, i + 1, w * t[k][IM] * e(k + 1, i))
            Ik(k, i + 1, w * t[k][II])

    def Dk(k, i, w):
        if w == 0.0:
            return
        E(i, w)
        if k < M:
            if i < L:
                Mk(k + 1, i + 1, w * t[k][DM] * e(k + 1, i))
            Dk(k + 1, i, w * t[k][DD])

    def E(i, w):
        Cs(i, w * eC)
        if eJ > 0.0:
            J(i, w * eJ)

    def J(i, w):
        if i < L:
            J(i + 1, w * loop)
        B(i, w * move)

    def Cs(i, w):
        if i < L:
            Cs(i + 1, w * loop)
        else:
            acc["total"] += w * move
            acc["best"] = max(acc["best"], w * move)
            acc["paths"] += 1

    N(0, 1.0)
    return acc["total"], acc["best"], acc["paths"]


@pytest.fixture(scope="module")
def tiny(tmp_path_factory):
    rng = np.random.default_rng(2024)
    profs = [_profile(rng, M, k) for k, M in enumerate((1, 2, 3, 3, 4))]
    path = str(tmp_path_factory.mktemp("bf") / "tiny.hmm")
    synth.write_hmm(path, profs)
    hs = p7.HmmSet(path)
    hs.file_path = path
    yield hs, _read_back(path), rng
    hs.close()


def test_forward_parser_equals_the_sum_over_all_paths(tiny):
    """fwd_sc of the oracle's per-target pipeline (multihit, whole sequence) = ln of the sum over every path; null1 likewise."""
    hs, models, rng = tiny
    npaths = 0
    for m, (M, mat, t) in enumerate(models):
        for L in (1, 2, 3, 4, 5):
            for rep in range(3):
                x = [int(v) for v in rng.integers(0, 20, size=L)]
                total, best, n = _enumerate(M, mat, t, x, L, True)
                st = hs.stages(m, np.array(x, dtype=np.uint8))
                assert st.fwd_sc == pytest.approx(math.log(total), abs=2e-4), (m, x)
                assert st.null_sc == pytest.approx(L * math.log(L / (L + 1.0)) + math.log(1.0 / (L + 1.0)), abs=1e-5)
                assert math.log(best) <= st.fwd_sc + 1e-4            # Viterbi path <= Forward
                npaths += n
    assert npaths > 20000                                             # (the lists are real: tens of thousands of paths in all)


def test_envelope_forward_equals_the_unihit_sum_over_all_paths(tiny):
    """Envelope rescoring: unihit profile, N/C loops configured for the FULL target length, only residues ienv..jenv scored."""
    hs, models, rng = tiny
    for m, (M, mat, t) in enumerate(models):
        for L, (ie, je) in ((5, (1, 5)), (5, (2, 4)), (6, (3, 3)), (4, (1, 2))):
            x = [int(v) for v in rng.integers(0, 20, size=L)]
            total, _, _ = _enumerate(M, mat, t, x[ie - 1:je], L, False)
            rc, envsc = hs.envelope(m, np.array(x, dtype=np.uint8), ie, je)[:2]
            assert rc == 0 and envsc == pytest.approx(math.log(total), abs=2e-4), (m, x, ie, je)


def _decode(M, mat, t, x, Lcfg):
    """Unihit posterior decoding and optimal accuracy by enumeration.  Returns (ppM[i][k], ppI[i][k], {'N': [..], 'C': [..]}, best) where best is
    (accuracy, first match residue, last match residue, first match node, last match node) of the path that maximises the summed
    posterior of its emitting states -- HMMER's optimal-accuracy criterion; an exit is taken from a match state (a path that ends
    ...M_k D_k+1 E has the accuracy of ...M_k E and is never preferred)."""
    L = len(x)
    move = 2.0 / (Lcfg + 2.0)
    loop = 1.0 - move
    MM, MI, MD, IM, II, DM, DD = range(7)
    occ = [0.0] * (M + 1)
    occ[1] = t[0][MI] + t[0][MM]
    for k in range(2, M + 1):
        occ[k] = occ[k - 1] * (t[k - 1][MM] + t[k - 1][MI]) + (1.0 - occ[k - 1]) * t[k - 1][DM]
    Z = sum(occ[k] * (M - k + 1) for k in range(1, M + 1))
    entry = [0.0] + [occ[k] / Z for k in range(1, M + 1)]
    e = lambda k, i: mat[k][x[i]] / synth.BGF[x[i]]
    paths = []                                   # (weight, [(residue index, 'M'|'I'|'N'|'C', node)], ends_in_match)

    def N(i, w, em):
        if i < L:
            N(i + 1, w * loop, em + [(i, 'N', 0)])
        for k in range(1, M + 1):
            if i < L:
                Mk(k, i + 1, w * move * entry[k] * e(k, i), em + [(i, 'M', k)])

    def Mk(k, i, w, em):
        Cs(i, w, em, True)
        if k < M:
            if i < L:
                Mk(k + 1, i + 1, w * t[k][MM] * e(k + 1, i), em + [(i, 'M', k + 1)])
                Ik(k, i + 1, w * t[k][MI], em + [(i, 'I', k)])
            Dk(k + 1, i, w * t[k][MD], em)

    def Ik(k, i, w, em):
        if i < L:
            Mk(k + 1, i + 1, w * t[k][IM] * e(k + 1, i), em + [(i, 'M', k + 1)])
            Ik(k, i + 1, w * t[k][II], em + [(i, 'I', k)])

    def Dk(k, i, w, em):
        Cs(i, w, em, False)
        if k < M:
            if i < L:
                Mk(k + 1, i + 1, w * t[k][DM] * e(k + 1, i), em + [(i, 'M', k + 1)])
            Dk(k + 1, i, w * t[k][DD], em)

    def Cs(i, w, em, from_match):
        if i < L:
            Cs(i + 1, w * loop, em + [(i, 'C', 0)], from_match)
        else:
            paths.append((w * move, em, from_match))

    N(0, 1.0, [])
    total = sum(p[0] for p in paths)
    ppM = [[0.0] * (M + 1) for _ in range(L)]
    ppI = [[0.0] * (M + 1) for _ in range(L)]
    ppX = {'N': [0.0] * L, 'C': [0.0] * L}
    for w, em, _ in paths:
        for i, kind, k in em:
            if kind == 'M':
                ppM[i][k] += w / total
            elif kind == 'I':
                ppI[i][k] += w / total
            else:
                ppX[kind][i] += w / total
    best = None
    for w, em, from_match in paths:
        if not from_match:
            continue
        acc = sum(ppM[i][k] if kind == 'M' else ppI[i][k] if kind == 'I' else ppX[kind][i] for i, kind, k in em)
        ms = [(i, k) for i, kind, k in em if kind == 'M']
        cand = (acc, ms[0][0] + 1, ms[-1][0] + 1, ms[0][1], ms[-1][1], ms)
        if best is None or cand[0] > best[0]:
            best = cand
    return ppM, ppI, ppX, best


def test_envelope_decoding_null2_and_optimal_accuracy_by_enumeration(tiny):
    """Per envelope: the optimal-accuracy score and its alignment coordinates, and the null2 odds by expectation
    (usage of every match / insert state and of N+C over the envelope, divided by its length)."""
    hs, models, rng = tiny
    checked = 0
    for m, (M, mat, t) in enumerate(models):
        for L, (ie, je) in ((5, (1, 5)), (6, (2, 5)), (4, (1, 3)), (6, (1, 6))):
            x = [int(v) for v in rng.integers(0, 20, size=L)]
            sub = x[ie - 1:je]
            Ld = len(sub)
            ppM, ppI, ppX, best = _decode(M, mat, t, sub, L)
            rc, envsc, oasc, null2, coords, _xC, _ns = hs.envelope(m, np.array(x, dtype=np.uint8), ie, je)
            assert rc == 0
            assert oasc == pytest.approx(best[0], abs=2e-4), (m, x, ie, je)
            # a second path within float noise of the best one may differ in coordinates: only unambiguous cases are compared
            hmm_from, hmm_to, ali_from, ali_to = (int(v) for v in coords)
            assert (ali_from, ali_to, hmm_from, hmm_to) == (best[1] + ie - 1, best[2] + ie - 1, best[3], best[4]), (m, x, ie, je)
            for r in range(20):
                want = (sum(ppX['N']) + sum(ppX['C'])) / Ld
                for k in range(1, M + 1):
                    want += sum(ppM[i][k] for i in range(Ld)) / Ld * (mat[k][r] / synth.BGF[r]) + sum(ppI[i][k] for i in range(Ld)) / Ld
                assert float(null2[r]) == pytest.approx(want, rel=3e-4), (m, r)
            checked += 1
    assert checked == 20


def _regions(M, mat, t, x):
    """Domain regions of the posterior heuristics (HMMER User Guide / p7_domaindef: rt1 0.25, rt2 0.10, rt3 0.20), from enumerated
    multihit posteriors: mocc[i] = P(residue i is emitted by the model core), begin / end mass = P(a domain begins at i / ends at i).
    Returns [(i, j, is_multidomain)]."""
    L = len(x)
    move = 3.0 / (L + 3.0)
    loop = 1.0 - move
    MM, MI, MD, IM, II, DM, DD = range(7)
    occ = [0.0] * (M + 1)
    occ[1] = t[0][MI] + t[0][MM]
    for k in range(2, M + 1):
        occ[k] = occ[k - 1] * (t[k - 1][MM] + t[k - 1][MI]) + (1.0 - occ[k - 1]) * t[k - 1][DM]
    Z = sum(occ[k] * (M - k + 1) for k in range(1, M + 1))
    entry = [0.0] + [occ[k] / Z for k in range(1, M + 1)]
    e = lambda k, i: mat[k][x[i]] / synth.BGF[x[i]]
    core = [0.0] * (L + 1)
    beg = [0.0] * (L + 2)       # beg[i]: a domain's first residue is i (1-based)
    end = [0.0] * (L + 1)       # end[i]: a domain's last residue is i
    tot = [0.0]

    def flank(i, w, ev):                          # N or J: i residues consumed so far
        if i < L:
            flank(i + 1, w * loop, ev)
            for k in range(1, M + 1):
                Mk(k, i + 1, w * move * entry[k] * e(k, i), ev + [('b', i + 1), ('c', i + 1)])

    def Mk(k, i, w, ev):
        E(i, w, ev)
        if k < M:
            if i < L:
                Mk(k + 1, i + 1, w * t[k][MM] * e(k + 1, i), ev + [('c', i + 1)])
                Ik(k, i + 1, w * t[k][MI], ev + [('c', i + 1)])
            Dk(k + 1, i, w * t[k][MD], ev)

    def Ik(k, i, w, ev):
        if i < L:
            Mk(k + 1, i + 1, w * t[k][IM] * e(k + 1, i), ev + [('c', i + 1)])
            Ik(k, i + 1, w * t[k][II], ev + [('c', i + 1)])

    def Dk(k, i, w, ev):
        E(i, w, ev)
        if k < M:
            if i < L:
                Mk(k + 1, i + 1, w * t[k][DM] * e(k + 1, i), ev + [('c', i + 1)])
            Dk(k + 1, i, w * t[k][DD], ev)

    def E(i, w, ev):
        ev = ev + [('e', i)]
        flank(i, w * 0.5, ev)                     # E -> J, then J behaves as N does
        w2 = w * 0.5 * loop ** (L - i) * move     # E -> C, C emits the rest, C -> T
        tot[0] += w2
        for kind, pos in ev:
            (core if kind == 'c' else beg if kind == 'b' else end)[pos] += w2

    flank(0, 1.0, [])
    mocc = [0.0] + [core[i] / tot[0] for i in range(1, L + 1)]
    btot, etot = [0.0] * (L + 1), [0.0] * (L + 1)
    for i in range(1, L + 1):
        btot[i] = btot[i - 1] + beg[i] / tot[0]
        etot[i] = etot[i - 1] + end[i] / tot[0]
    out, i, triggered = [], -1, False
    for j in range(1, L + 1):
        if not triggered:
            if mocc[j] - (btot[j] - btot[j - 1]) < 0.10:
                i = j
            elif i == -1:
                i = j
            if mocc[j] >= 0.25:
                triggered = True
        elif mocc[j] - (etot[j] - etot[j - 1]) < 0.10:
            worst = max(min(etot[z] - etot[i - 1], btot[j] - btot[z - 1]) for z in range(i, j + 1))
            out.append((i, j, worst >= 0.20))
            i, triggered = -1, False
    return out


def test_reported_scores_and_evalues_follow_from_the_enumerated_quantities(tiny, capfd, monkeypatch):
    """The score arithmetic behind a domtblout row, rebuilt from enumerated quantities only (HMMER User Guide, "how scores and
    E-values are calculated"; p7_Pipeline): null2 bias = ln(1 + omega * prod null2(x_i)) with omega 1/256, domain bit score =
    (envelope Forward + (L - Ld) ln(L/(L+3)) - null1 - bias) / ln 2, sequence score from the whole-sequence Forward or from the sum
    of its domains, whichever is larger, lnP from the exponential tail (tau, lambda of the model), E = P * Z, c-E = P * domZ."""
    hs, models, rng = tiny
    tau, lam, omega = -30.0, 0.70, 1.0 / 256.0
    monkeypatch.setenv("P7O_TRACE_REGIONS", "1")       # the oracle then names on stderr the regions it resolves by the stochastic ensemble
    seen = ensemble = 0
    for m, (M, mat, t) in enumerate(models):
        for rep in range(36):
            L = int(rng.integers(3, 7 if M <= 2 else 6))          # (the multihit event enumeration grows fast with L and M)
            x = [int(v) for v in rng.integers(0, 20, size=L)]
            if rep % 2 == 0:                       # half of the targets carry the consensus of the model: clearly positive scores
                for k in range(1, min(M, L) + 1):
                    x[k - 1] = int(np.argmax(mat[k]))
            capfd.readouterr()
            rows = hs.search([m], [np.array(x, dtype=np.uint8)], ["t"])
            regions = _regions(M, mat, t, x)
            used_ensemble = "multi-domain region" in capfd.readouterr().err
            assert used_ensemble == any(multi for _, _, multi in regions), (m, x, regions)
            if used_ensemble:
                ensemble += 1                      # null2 by trace, envelopes by clustering: not a closed-form quantity
                continue
            if len(rows) != 1 or rows[0].ndom != 1:
                continue
            r = rows[0]
            ie, je = r.env_from, r.env_to
            assert regions == [(ie, je, False)], (m, x, regions)       # a single-domain region IS the envelope
            sub, Ld = x[ie - 1:je], je - ie + 1
            env_total, _, _ = _enumerate(M, mat, t, sub, L, False)
            ppM, ppI, ppX, best = _decode(M, mat, t, sub, L)
            null2 = []
            for res in range(20):
                v = (sum(ppX['N']) + sum(ppX['C'])) / Ld
                for k in range(1, M + 1):
                    v += sum(ppM[i][k] for i in range(Ld)) / Ld * (mat[k][res] / synth.BGF[res]) + sum(ppI[i][k] for i in range(Ld)) / Ld
                null2.append(v)
            domcorr = sum(math.log(null2[c]) for c in sub)
            bias = math.log(1.0 + omega * math.exp(domcorr))
            null1 = L * math.log(L / (L + 1.0)) + math.log(1.0 / (L + 1.0))
            envsc = math.log(env_total)
            dom_bits = (envsc + (L - Ld) * math.log(L / (L + 3.0)) - null1 - bias) / math.log(2.0)
            fwd_total, _, _ = _enumerate(M, mat, t, x, L, True)
            seq_bits = (math.log(fwd_total) - null1 - bias) / math.log(2.0)
            pre_bits = (math.log(fwd_total) - null1) / math.log(2.0)
            if envsc - domcorr > 0.0 and dom_bits > seq_bits:          # the domain sum replaces the whole-sequence score
                seq_bits, pre_bits = dom_bits, (envsc + (L - Ld) * math.log(L / (L + 3.0)) - null1) / math.log(2.0)
            tol = 3e-3                                                   # the table-driven log-sum HMMER uses is good to 1e-3 nats
            assert r.dom_score == pytest.approx(dom_bits, abs=tol), (m, x)
            assert r.dom_bias == pytest.approx(bias / math.log(2.0), abs=tol)
            assert r.full_score == pytest.approx(seq_bits, abs=tol), (m, x)
            assert r.full_bias == pytest.approx(pre_bits - seq_bits, abs=2 * tol)
            lnP_dom = 0.0 if r.dom_score < tau else -lam * (r.dom_score - tau)
            lnP_seq = 0.0 if r.full_score < tau else -lam * (r.full_score - tau)
            assert r.i_evalue == pytest.approx(math.exp(lnP_dom) * 1.0, rel=1e-5) and r.c_evalue == pytest.approx(math.exp(lnP_dom) * 1.0, rel=1e-5)
            assert r.full_evalue == pytest.approx(math.exp(lnP_seq) * 1.0, rel=1e-5)
            assert r.acc == pytest.approx(best[0] / (1.0 + abs(je - ie)), abs=1e-4)
            seen += 1
    assert seen >= 20 and ensemble >= 20, (seen, ensemble)


def test_integer_filters_approximate_the_best_path_of_their_models(tiny):
    """ViterbiFilter (16-bit, 1/500 bit) and MSVFilter (8-bit, 1/3 bit): the best path of the full profile / of the ungapped
    uniform-entry model with N/C/J loops for free, minus the flat 3 nats both filters charge for those loops.  The filters round
    every term, so agreement is to the rounding they are allowed: half a unit per term of the path."""
    hs, models, rng = tiny
    for m, (M, mat, t) in enumerate(models):
        for L in (2, 3, 4, 5, 6):
            for rep in range(3):
                x = [int(v) for v in rng.integers(0, 20, size=L)]
                st = hs.stages(m, np.array(x, dtype=np.uint8))
                _, best, _ = _enumerate(M, mat, t, x, L, True, free_loops=True)
                nterms = 2 * L + 6
                assert st.vit_sc == pytest.approx(math.log(best) - 3.0, abs=nterms * 0.5 * math.log(2.0) / 500.0 + 1e-3), (m, x)
                _, best, _ = _enumerate(M, mat, t, x, L, True, free_loops=True, ungapped_uniform=True)
                assert st.msv_sc == pytest.approx(math.log(best) - 3.0, abs=nterms * 0.5 * math.log(2.0) / 3.0 + 1e-3), (m, x)


def test_bias_filter_equals_the_sum_over_all_paths_of_the_two_state_model(tiny):
    """The composition filter: a two-state HMM (background, mean run 400; model composition, mean run M/8; start 0.999 / 0.001)
    scored as odds against the background plus the null1 length terms -- 2^L paths, all of them listed."""
    import itertools
    hs, models, rng = tiny
    compos = []
    for ln in open(hs.file_path).read().split("\n"):
        if ln.split()[:1] == ["COMPO"]:
            compos.append([math.exp(-float(v)) for v in ln.split()[1:21]])
    assert len(compos) == len(models)
    for m, (M, mat, t) in enumerate(models):
        L0, L1 = 400.0, M / 8.0
        tr = [[L0 / (L0 + 1.0), 1.0 / (L0 + 1.0)], [1.0 / (L1 + 1.0), L1 / (L1 + 1.0)]]
        pi = [0.999, 0.001]
        for L in (1, 2, 5, 8):
            x = [int(v) for v in rng.integers(0, 20, size=L)]
            odds = lambda s, c: 1.0 if s == 0 else compos[m][c] / synth.BGF[c]
            total = 0.0
            for states in itertools.product((0, 1), repeat=L):
                w = pi[states[0]] * odds(states[0], x[0])
                for i in range(1, L):
                    w *= tr[states[i - 1]][states[i]] * odds(states[i], x[i])
                total += w
            want = math.log(total) + L * math.log(L / (L + 1.0)) + math.log(1.0 / (L + 1.0))
            st = hs.stages(m, np.array(x, dtype=np.uint8))
            assert st.bias_sc == pytest.approx(want, abs=2e-5), (m, x)


def _segmentations(M, mat, t, x):
    """Every multihit path of the whole target with its weight, reduced to what the trace ensemble keeps of a path: per domain
    (first match residue, last match residue, first match node, last match node) and, for the null2-by-trace terms, the states that
    emitted the residues from the first to the last match.  Returns (total weight, {segmentation: weight}, per-residue expectation of
    the null2 odds ratio, its second moment)."""
    L = len(x)
    move = 3.0 / (L + 3.0)
    loop = 1.0 - move
    MM, MI, MD, IM, II, DM, DD = range(7)
    occ = [0.0] * (M + 1)
    occ[1] = t[0][MI] + t[0][MM]
    for k in range(2, M + 1):
        occ[k] = occ[k - 1] * (t[k - 1][MM] + t[k - 1][MI]) + (1.0 - occ[k - 1]) * t[k - 1][DM]
    Z = sum(occ[k] * (M - k + 1) for k in range(1, M + 1))
    entry = [0.0] + [occ[k] / Z for k in range(1, M + 1)]
    e = lambda k, i: mat[k][x[i]] / synth.BGF[x[i]]
    seg_w = {}
    m1 = [0.0] * (L + 1)
    m2 = [0.0] * (L + 1)
    tot = [0.0]

    def finish(w, doms):
        tot[0] += w
        key = []
        ratio = [1.0] * (L + 1)
        for em, kexit in doms:                            # em: [(residue, 'M'|'I', node)] of one domain, in order; kexit: node of the state E was entered from (M or D)
            ms = [(i, k) for i, kind, k in em if kind == 'M']
            sqfrom, sqto = ms[0][0], ms[-1][0]
            key.append((sqfrom, sqto, ms[0][1], ms[-1][1]))   # model coordinates from match states only, as p7_trace_Index takes them (trailing delete states, kexit, do not count)
            inside = [(i, kind, k) for i, kind, k in em if sqfrom <= i <= sqto]
            n = float(len(inside))
            for pos in range(sqfrom + 1, sqto + 1):       # the first residue of a domain keeps ratio 1 (HMMER's `pos <= sqfrom` loop)
                r = x[pos - 1]
                ratio[pos] = sum((mat[k][r] / synth.BGF[r]) if kind == 'M' else 1.0 for _, kind, k in inside) / n
        key = tuple(key)
        seg_w[key] = seg_w.get(key, 0.0) + w
        for pos in range(1, L + 1):
            m1[pos] += w * ratio[pos]
            m2[pos] += w * ratio[pos] * ratio[pos]

    def flank(i, w, doms):
        if i < L:
            flank(i + 1, w * loop, doms)
            for k in range(1, M + 1):
                Mk(k, i + 1, w * move * entry[k] * e(k, i), doms, [(i + 1, 'M', k)])

    def Mk(k, i, w, doms, em):
        E(i, w, doms + [(em, k)])
        if k < M:
            if i < L:
                Mk(k + 1, i + 1, w * t[k][MM] * e(k + 1, i), doms, em + [(i + 1, 'M', k + 1)])
                Ik(k, i + 1, w * t[k][MI], doms, em + [(i + 1, 'I', k)])
            Dk(k + 1, i, w * t[k][MD], doms, em)

    def Ik(k, i, w, doms, em):
        if i < L:
            Mk(k + 1, i + 1, w * t[k][IM] * e(k + 1, i), doms, em + [(i + 1, 'M', k + 1)])
            Ik(k, i + 1, w * t[k][II], doms, em + [(i + 1, 'I', k)])

    def Dk(k, i, w, doms, em):
        E(i, w, doms + [(em, k)])
        if k < M:
            if i < L:
                Mk(k + 1, i + 1, w * t[k][DM] * e(k + 1, i), doms, em + [(i + 1, 'M', k + 1)])
            Dk(k + 1, i, w * t[k][DD], doms, em)

    def E(i, w, doms):
        flank(i, w * 0.5, doms)
        finish(w * 0.5 * loop ** (L - i) * move, doms)

    flank(0, 1.0, [])
    return tot[0], seg_w, [v / tot[0] for v in m1], [v / tot[0] for v in m2]


def test_trace_ensemble_samples_paths_with_their_probabilities(tiny):
    """The 200 stochastic tracebacks of a region: the domain segmentations they produce occur with the frequencies the enumerated
    path probabilities predict (binomial 4.5 sigma: the generator is seeded, so the outcome is fixed), no impossible segmentation
    ever appears, and the summed per-residue null2 odds (null2 by trace) average to their exact expectation."""
    hs, models, rng = tiny
    N = 200
    for m in (1, 2, 3):                                # (2, 3 and 3 nodes: the enumeration with per-domain bookkeeping grows fast)
        M, mat, t = models[m]
        cons = [int(np.argmax(mat[k])) for k in range(1, M + 1)]
        for x in (cons + cons, cons + [int(rng.integers(0, 20))] + cons[:max(1, M - 1)], [int(v) for v in rng.integers(0, 20, size=5)]):
            x = x[:6]
            L = len(x)
            total, seg_w, mean, second = _segmentations(M, mat, t, x)
            rc, n2sum, segs, nseg, _env = hs.region_ensemble(m, np.array(x, dtype=np.uint8), 1, L)
            assert rc == 0
            seen = {}
            for tr in range(N):
                key = tuple(tuple(int(v) for v in segs[tr, d]) for d in range(int(nseg[tr])))
                seen[key] = seen.get(key, 0) + 1
            assert all(k in seg_w and seg_w[k] > 0.0 for k in seen), [k for k in seen if k not in seg_w]
            rare_p, rare_n = 0.0, 0
            for key, w in seg_w.items():
                p = w / total
                if N * p >= 3.0:
                    sd = math.sqrt(N * p * (1.0 - p))
                    assert abs(seen.get(key, 0) - N * p) <= 4.5 * sd + 1e-9, (m, x, key, seen.get(key, 0), N * p)
                else:
                    rare_p += p
                    rare_n += seen.get(key, 0)
            assert abs(rare_n - N * rare_p) <= 4.5 * math.sqrt(max(N * rare_p * (1.0 - rare_p), 1e-12)) + 1.0
            for pos in range(1, L + 1):
                sd = math.sqrt(max(second[pos] - mean[pos] ** 2, 0.0) / N)
                assert float(n2sum[pos - 1]) / N == pytest.approx(mean[pos], abs=4.5 * sd + 1e-5), (m, x, pos)


def test_alignment_to_the_model_is_the_optimal_accuracy_path(tiny):
    """hmmalign restated (p7o_align: unihit, whole sequence): the residue each match state emits on the optimal-accuracy path."""
    hs, models, rng = tiny
    for m, (M, mat, t) in enumerate(models):
        for L in (3, 4, 5, 6):
            x = [int(v) for v in rng.integers(0, 20, size=L)]
            for k in range(1, min(M, L - 1) + 1):           # consensus residues from position 2 on: the path has something to find
                x[k] = int(np.argmax(mat[k]))
            best = _decode(M, mat, t, x, L)[3]
            rc, path = hs.align(m, np.array(x, dtype=np.uint8))
            want = [0] * M
            for i, k in best[5]:
                want[k - 1] = i + 1
            assert rc == 0 and [int(v) for v in path] == want, (m, x)


def test_posterior_line_of_a_domain_alignment(tiny):
    """What hmmsearch prints under a domain alignment (the PP line): the posterior probability of every residue on the optimal-accuracy
    path in the state that emits it (p7o_envelope_alignment).  By enumeration: the share of the path weights in which that state
    emits that residue."""
    hs, models, rng = tiny
    checked = inserts = 0
    for m, (M, mat, t) in enumerate(models):
        for L in (3, 4, 5, 6):
            for trial in range(3):
                x = [int(v) for v in rng.integers(0, 20, size=L)]
                for k in range(1, min(M, L - 1) + 1):
                    if trial != 2 or k != 2:             # (third trial: a foreign residue in the middle, an insert or a weak match)
                        x[k] = int(np.argmax(mat[k]))
                ppM, ppI, _ppX, best = _decode(M, mat, t, x, L)
                rc, path, pp = hs.envelope_alignment(m, np.array(x, dtype=np.uint8), 1, L)
                rc2, path2 = hs.align(m, np.array(x, dtype=np.uint8))
                assert rc == 0 and rc2 == 0 and [int(v) for v in path] == [int(v) for v in path2]
                matched = {int(i): k + 1 for k, i in enumerate(path) if i}
                assert matched == {i + 1: k for i, k in best[5]}
                lo, hi = min(matched), max(matched)
                node = 0
                for i in range(1, L + 1):
                    if i in matched:
                        node = matched[i]
                        assert float(pp[i]) == pytest.approx(ppM[i - 1][node], rel=2e-4, abs=1e-6), (m, x, i)
                        checked += 1
                    elif lo < i < hi:                    # between two matched residues and not matched itself: emitted by the insert state of the node before
                        assert float(pp[i]) == pytest.approx(ppI[i - 1][node], rel=2e-4, abs=1e-6), (m, x, i)
                        inserts += 1
                    else:
                        assert float(pp[i]) == 0.0       # flanks: not on the path
    assert checked > 100


def _cluster(segs, nseg, nsamples=200):
    """Sampled segments -> envelopes, restated from the description of HMMER's ensemble clustering (single linkage; two segments link
    when they overlap by >= 0.8 of the shorter one in the sequence AND in the model and their diagonals differ by <= 4; a cluster
    counts when >= 0.25 of the traces have a segment in it; each endpoint is the outermost value reached by >= 0.02 of those traces)."""
    items = [(tr, tuple(int(v) for v in segs[tr, d])) for tr in range(nsamples) for d in range(int(nseg[tr]))]

    def linked(a, b):
        for lo, hi in ((0, 1), (2, 3)):
            nov = min(a[hi], b[hi]) - max(a[lo], b[lo]) + 1
            if nov / float(min(a[hi] - a[lo] + 1, b[hi] - b[lo] + 1)) < 0.8:
                return False
        diag = lambda s: int((s[0] - s[2] + s[1] - s[3]) / 2)          # (C integer division: toward zero)
        return abs(diag(a) - diag(b)) <= 4

    label = [-1] * len(items)
    ncl = 0
    for h in range(len(items)):
        if label[h] >= 0:
            continue
        label[h] = ncl
        stack = [h]
        while stack:
            a = stack.pop()
            for b in range(len(items)):
                if label[b] < 0 and linked(items[a][1], items[b][1]):
                    label[b] = ncl
                    stack.append(b)
        ncl += 1
    envs = []
    for c in range(ncl):
        members = [items[h] for h in range(len(items)) if label[h] == c]
        ninc = len(set(tr for tr, _ in members))
        if ninc / float(nsamples) < 0.25:
            continue
        ends = []
        for f in range(4):
            vals = [sg[f] for _, sg in members]
            order = range(min(vals), max(vals)) if f in (0, 2) else range(max(vals), min(vals), -1)
            pick = max(vals) if f in (0, 2) else min(vals)
            for v in order:
                if vals.count(v) / float(ninc) >= 0.02:
                    pick = v
                    break
            ends.append(pick)
        envs.append((ends[0], ends[1], ends[2], ends[3]))
    return sorted(envs, key=lambda s: (s[0], s[1]))


def test_ensemble_clustering_restated(tmp_path):
    """Envelopes of multi-domain regions (tandem fragments of one model): the oracle's clustering of its 200 traces against the
    restatement above, on regions with 2 to ~8 sampled domains per trace."""
    rng = np.random.default_rng(31)
    profs = [synth.random_profile(rng, M, "cl%d" % k, "PF8%04d.1" % k) for k, M in enumerate((40, 75, 120))]
    for p in profs:
        p.stats = (-30.0, 0.71, -30.0, 0.71, -30.0, 0.70)
    path = str(tmp_path / "cl.hmm")
    synth.write_hmm(path, profs)
    hs = p7.HmmSet(path)
    try:
        nenv = 0
        for m, pr in enumerate(profs):
            M = pr.M
            for copies in (2, 3, 5, 8):
                parts = [synth.random_residues(rng, 12)]
                for c in range(copies):
                    a = int(rng.integers(1, M // 2)); b = int(rng.integers(a + M // 4, M + 1))
                    parts.append(synth.sample_domain(rng, pr, a, b))
                    if c % 2:
                        parts.append(synth.random_residues(rng, 6))
                parts.append(synth.random_residues(rng, 12))
                x = np.concatenate(parts).astype(np.uint8)
                rc, _n2, segs, nseg, env = hs.region_ensemble(m, x, 1, len(x))
                assert rc == 0
                assert [tuple(int(v) for v in e) for e in env] == _cluster(segs, nseg), (m, copies)
                nenv += len(env)
        assert nenv >= 20
    finally:
        hs.close()


def test_filter_decisions_follow_the_published_thresholds(tmp_path):
    """F1 = 0.02 on the MSV score (Gumbel), again after the composition filter replaces null1, F2 = 1e-3 on the Viterbi filter
    score (Gumbel), F3 = 1e-5 on the Forward score (exponential tail): the pass flags of the oracle's cascade against the formulas,
    from its own stage scores, on targets that land on both sides of every threshold."""
    rng = np.random.default_rng(8)
    pr = synth.random_profile(rng, 60, "thr", "PF80000.1")
    mu_m, lam_m, mu_v, lam_v, tau, lam_f = -7.5, 0.71, -8.3, 0.71, -3.6, 0.70
    pr.stats = (mu_m, lam_m, mu_v, lam_v, tau, lam_f)
    path = str(tmp_path / "thr.hmm")
    synth.write_hmm(path, [pr])
    hs = p7.HmmSet(path)
    gumbel = lambda x, mu, lam: 1.0 - math.exp(-math.exp(-lam * (x - mu)))
    tail = lambda x, mu, lam: 1.0 if x < mu else math.exp(-lam * (x - mu))
    bits = lambda sc, null: (sc - null) / math.log(2.0)
    counts = [0, 0, 0, 0]
    try:
        for rep in range(400):
            frac = rep % 8
            parts = [synth.random_residues(rng, int(rng.integers(20, 200)))]
            if frac:                                  # a fragment of the model of growing length: scores from noise to clear hits
                parts.append(synth.sample_domain(rng, pr, 1, 6 + 7 * frac))
            parts.append(synth.random_residues(rng, int(rng.integers(5, 60))))
            x = np.concatenate(parts).astype(np.uint8)
            st = hs.stages(0, x)
            p_msv = gumbel(bits(st.msv_sc, st.null_sc), mu_m, lam_m)
            if abs(p_msv - 0.02) < 1e-4:
                continue                               # (on the knife edge float noise decides; never happens with these seeds)
            assert bool(st.pass_msv) == (p_msv <= 0.02)
            counts[0] += st.pass_msv
            if not st.pass_msv:
                continue
            p_bias = gumbel(bits(st.msv_sc, st.bias_sc), mu_m, lam_m)
            assert bool(st.pass_bias) == (p_bias <= 0.02)
            counts[1] += st.pass_bias
            if not st.pass_bias:
                continue
            # the Viterbi filter is only consulted when the MSV P-value has not already cleared F2 (p7_Pipeline: `if (P > F2)`)
            assert bool(st.pass_vit) == (p_bias <= 1e-3 or gumbel(bits(st.vit_sc, st.bias_sc), mu_v, lam_v) <= 1e-3)
            counts[2] += st.pass_vit
            if not st.pass_vit:
                continue
            assert bool(st.pass_fwd) == (tail(bits(st.fwd_sc, st.bias_sc), tau, lam_f) <= 1e-5)
            counts[3] += st.pass_fwd
        assert 400 > counts[0] > counts[2] > counts[3] > 20, counts
    finally:
        hs.close()
