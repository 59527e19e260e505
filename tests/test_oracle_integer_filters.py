"""A second, independently written formulation of the two INTEGER filters of the scan oracle (oracle/p7oracle.c: parity unpinned -- no
HMMER exists next to the reference, checkm/hmmer.py:61-74 only shells out to it), stated from the published definitions rather than from
the oracle's code:

  MSV filter (Eddy 2011, "Accelerated profile HMM searches", fig. 2 and methods): the score of the best set of ungapped local diagonals in
  unsigned bytes of 1/3 bit; a byte b stands for the score (b - 190) / 3 bits; emission COSTS are stored with an offset (`bias`, the
  largest emission score) so that they are non-negative; every local entry costs log2(M (M + 1) / 2) bits (uniform entry), leaving the
  core costs log2 2, moving from N / J / C costs log2((L + 3) / 3); additions saturate at 255, subtractions at 0, a row whose best cell
  would saturate ends the filter with "passes"; the N, J, C self loops are not charged and stand for 3 nats in all.
  Viterbi filter (same paper): the optimal-path score of the local multi-hit profile in signed words of 1/500 bit offset by 12000, -32768
  for "impossible", entry by occupancy (Eddy 2011 eq. for the local entry distribution: occ(k) / sum_j occ(j) (M - j + 1)), all nine core
  transitions, zero-cost insert emissions, the same 3-nat approximation.

Everything below is written in the probability / cost language of those definitions: its own HMM text reader, its own occupancy, its own
rounding, dynamic programs over dictionaries of Python integers.  What it checks: the constants path (scales, offsets, bias, the rounding
of entry / exit / move costs, the II clamp) and the recurrences of the oracle's msv_filter / vit_filter, on random profiles and on
sequences that do and do not carry the model.  What it cannot check: that these definitions are HMMER's (nothing here can)."""
import ctypes
import math
import os

import numpy as np

from synthdata import synth
from oracle import p7

_m = ctypes.CDLL("libm.so.6")
_m.expf.restype = ctypes.c_float; _m.expf.argtypes = [ctypes.c_float]
_m.logf.restype = ctypes.c_float; _m.logf.argtypes = [ctypes.c_float]
_m.roundf.restype = ctypes.c_float; _m.roundf.argtypes = [ctypes.c_float]
F = np.float32
AA = "ACDEFGHIKLMNPQRSTVWY"
# background of the null model: the amino-acid frequencies HMMER's p7_bg uses (Swiss-Prot 50.8), as published in its User Guide's tables
BG = [0.0787945, 0.0151600, 0.0535222, 0.0668298, 0.0397062, 0.0695071, 0.0229198, 0.0590092, 0.0594422, 0.0963728,
      0.0237718, 0.0414386, 0.0482904, 0.0395639, 0.0540978, 0.0683364, 0.0540687, 0.0673417, 0.0114135, 0.0304133]


def _prob(tok):
    return F(0.0) if tok == "*" else F(_m.expf(F(-1.0 * float(tok))))


def read_models(path):
    """[(M, match[k][x], trans[k][7])] k = 0..M: probabilities of an HMMER3 ASCII file (the file stores -ln p)."""
    out, lines = [], open(path).read().split("\n")
    i = 0
    while i < len(lines):
        if lines[i].startswith("HMM ") or lines[i].startswith("HMM\t"):
            i += 2
            if lines[i].split()[0] == "COMPO":
                i += 1
            ins0 = lines[i].split(); t0 = lines[i + 1].split(); i += 2
            mat, tr = [None], [[_prob(t) for t in t0[:7]]]
            del ins0
            while not lines[i].startswith("//"):
                f = lines[i].split()
                mat.append([_prob(t) for t in f[1:21]])
                tr.append([_prob(t) for t in lines[i + 2].split()[:7]])
                i += 3
            out.append((len(mat) - 1, mat, tr))
        i += 1
    return out


MM, MI, MD, IM, II, DM, DD = range(7)


def occupancy(M, tr):
    occ = [F(0.0)] * (M + 1)
    occ[1] = F(tr[0][MI] + tr[0][MM])
    for k in range(2, M + 1):
        occ[k] = F(F(occ[k - 1] * F(tr[k - 1][MM] + tr[k - 1][MI])) + F(F(F(1.0) - occ[k - 1]) * tr[k - 1][DM]))
    return occ


def lg(x):                       # natural log of a double, to float (the oracle and HMMER take log() of the ratio in double precision)
    return F(math.log(float(x))) if float(x) > 0 else F(-np.inf)


def msv_bytes(M, mat, dsq, L):
    """Byte the MSV filter ends with (xJ), or None when a row saturates."""
    S = F(3.0 / math.log(2.0))
    cost = lambda sc: -1.0 * float(_m.roundf(F(S * F(sc))))              # third-bits, as a cost
    best = max([0.0] + [float(lg(float(mat[k][x]) / BG[x])) for k in range(1, M + 1) for x in range(20)])
    bias = int(min(255.0, cost(F(-1.0 * best))))
    def emission_cost(k, x):
        c = cost(lg(float(mat[k][x]) / BG[x]))
        return 255 if c > 255 - bias else int(c) + bias
    entry = int(min(255.0, cost(_m.logf(F(2.0) / F(F(M) * F(M + 1))))))
    leave = int(min(255.0, cost(_m.logf(F(0.5)))))
    move = int(min(255.0, cost(_m.logf(F(3.0) / F(L + 3)))))
    base = 190
    em = {(k, x): emission_cost(k, x) for k in range(1, M + 1) for x in set(dsq)}
    both = (move + entry) & 0xff                                           # (the two costs are added in a byte register)
    sub = lambda a, b: max(0, a - b)
    add = lambda a, b: min(255, a + b)
    xJ, xB, prev = 0, sub(base, both), {}
    for x in dsq:
        cur, xE = {}, 0
        for k in range(1, M + 1):
            v = sub(add(max(prev.get(k - 1, 0), xB), bias), em[(k, x)])
            cur[k] = v; xE = max(xE, v)
        prev = cur
        if add(xE, bias) == 255:
            return None
        xJ = max(xJ, sub(xE, leave))
        xB = sub(max(base, xJ), both)
    return xJ


def vit_words(M, mat, tr, dsq, L):
    """Word the Viterbi filter ends with (xC), or 32767 when a cell saturates."""
    S = F(500.0 / math.log(2.0))
    NEG = -32768
    def word(sc):
        if not np.isfinite(sc):
            return NEG
        v = float(_m.roundf(F(S * F(sc))))
        return 32767 if v >= 32767.0 else NEG if v <= -32768.0 else int(v)
    sat = lambda v: 32767 if v > 32767 else NEG if v < NEG else v
    occ = occupancy(M, tr)
    Z = F(0.0)
    for k in range(1, M + 1):
        Z = F(Z + F(occ[k] * F(M - k + 1)))
    t = lambda k, a, cap=0: min(cap, word(lg(tr[k][a]))) if 1 <= k < M else NEG
    bm = {k: min(0, word(lg(float(occ[k]) / float(Z)))) for k in range(1, M + 1)}
    e = {(k, x): word(lg(float(mat[k][x]) / BG[x])) for k in range(1, M + 1) for x in set(dsq)}
    e_loop = e_move = word(F(-math.log(2.0)))
    move = word(_m.logf(F(3.0) / F(L + 3)))
    base = 12000
    xN, xB, xJ, xC = base, base + move, NEG, NEG
    Mp, Ip, Dp = {}, {}, {}
    for x in dsq:
        Mc, Ic, Dc, xE = {}, {}, {}, NEG
        for k in range(1, M + 1):
            v = sat(xB + bm[k])
            v = max(v, sat(Mp.get(k - 1, NEG) + t(k - 1, MM)), sat(Ip.get(k - 1, NEG) + t(k - 1, IM)), sat(Dp.get(k - 1, NEG) + t(k - 1, DM)))
            v = sat(v + e[(k, x)])
            Mc[k] = v; xE = max(xE, v)
            Ic[k] = max(sat(Mp.get(k, NEG) + t(k, MI)), sat(Ip.get(k, NEG) + t(k, II, -1)))
        for k in range(2, M + 1):
            dd = word(lg(tr[k - 1][DD])) if k - 1 < M else NEG
            Dc[k] = max(sat(Mc[k - 1] + t(k - 1, MD)), sat(Dc.get(k - 1, NEG) + dd))
        if xE >= 32767:
            return 32767
        xC = max(xC, xE + e_move); xJ = max(xJ, xE + e_loop); xB = max(xJ + move, xN + move)
        Mp, Ip, Dp = Mc, Ic, Dc
    return xC


def test_second_formulation_agrees_with_the_oracle(tmp_path):
    profs = synth.small_profiles(21, 14, 12, 70, with_stats=False)
    for q in profs:
        q.stats = (-8.0 - 0.01 * q.M, 0.71, -9.0 - 0.01 * q.M, 0.71, -4.0, 0.71)          # (the integer stages do not read the calibration; the reader asks for it)
    path = os.path.join(str(tmp_path), "m.hmm")
    synth.write_hmm(path, profs)
    models = read_models(path)
    hs = p7.HmmSet(path)
    assert hs.n == len(models) == len(profs)
    rng = np.random.default_rng(8)
    seqs = [r[2].rstrip("*") for r in synth.make_bin(profs, 31, n_orfs=40, dup_frac=0.3)]             # carry planted domains of the models
    seqs += ["".join(rng.choice(list(AA), p=np.asarray(BG) / sum(BG), size=int(n))) for n in (25, 60, 140, 300)]
    seqs = [s for s in seqs if set(s) <= set(AA)][:26]
    checked = high = 0
    for i, (M, mat, tr) in enumerate(models):
        assert M == hs.M(i)
        for n, s in enumerate(seqs):
            dsq = [AA.index(c) for c in s]
            st = hs.stages(i, np.asarray(dsq, dtype=np.uint8))
            if n % 3 != i % 3 and 0 <= st.msv_xJ <= 200:                          # a third of the sequences per model, and every one the model scores well on
                continue
            b = msv_bytes(M, mat, dsq, len(s))
            assert (b is None and st.msv_xJ == -1) or b == st.msv_xJ, (i, s[:20], b, st.msv_xJ)
            w = vit_words(M, mat, tr, dsq, len(s))
            assert w == st.vit_xC, (i, s[:20], w, st.vit_xC)
            checked += 1
            high += 1 if (b is None or b > 200) else 0
    assert checked >= 100 and high >= 5           # some of the pairs carry real hits (bytes well above the base of 190)
    hs.close()


def test_striped_avx2_filters_equal_the_scalar_ones(tmp_path):
    """oracle/p7simd.c (the striped AVX2 byte MSV / word Viterbi filters bench.py times as cpu_baseline kind "port-simd") against the
    scalar loops of oracle/p7oracle.c: the final xJ byte and xC word -- and so every stage decision and every row -- must be the same,
    on models of every striping remainder (M mod 32, M mod 16), on hits, non-hits, overflowing pairs and degenerate residues."""
    import pytest
    if not p7.lib().p7o_simd_available():
        pytest.skip("no AVX2 on this CPU")
    rng = np.random.default_rng(31)
    lens = [1, 2, 7, 15, 16, 17, 31, 32, 33, 47, 64, 65, 100, 129, 255, 300, 511, 700]
    profs = [synth.random_profile(rng, m, "m%d" % m, "PF%05d.1" % m) for m in lens]
    for p in profs:                      # (any calibration will do: the filters' integers do not depend on it, the pass flags must agree either way)
        p.stats = (-8.5, 0.71, -9.6, 0.71, -3.8, 0.71)
    path = str(tmp_path / "s.hmm")
    synth.write_hmm(path, profs)
    recs = synth.make_bin(profs, 77, n_orfs=60, dup_frac=0.3)
    seqs = [p7.digitize(r[2]) for r in recs]
    # a sequence with ambiguity codes and a long exact copy of a model's consensus several times over (byte overflow)
    seqs.append(p7.digitize("ACDEFGHIKLMNPQRSTVWYBJZOUX*" * 8))
    seqs.append(np.concatenate([seqs[3]] * 6))
    names = ["s%d" % i for i in range(len(seqs))]
    hs = p7.HmmSet(path)
    try:
        n = over = passed = 0
        for i in range(hs.n):
            for d in seqs:
                assert not p7.set_simd(False)
                a = hs.stages(i, d)
                assert p7.set_simd(True)
                b = hs.stages(i, d)
                assert (a.msv_xJ, a.vit_xC, a.pass_msv, a.pass_bias, a.pass_vit, a.pass_fwd) == (b.msv_xJ, b.vit_xC, b.pass_msv, b.pass_bias, b.pass_vit, b.pass_fwd), (lens[i], len(d))
                assert np.float32(a.msv_sc).view(np.uint32) == np.float32(b.msv_sc).view(np.uint32) and np.float32(a.vit_sc).view(np.uint32) == np.float32(b.vit_sc).view(np.uint32)
                n += 1; over += a.msv_xJ < 0; passed += a.pass_fwd
        assert n == len(lens) * len(seqs) and over > 0 and passed > 10
        p7.set_simd(False)
        r0 = hs.search(list(range(hs.n)), seqs, names)
        p7.set_simd(True)
        r1 = hs.search(list(range(hs.n)), seqs, names)
        assert len(r0) == len(r1) > 10
        assert hs.format_domtblout(r0, names, [""] * len(names)) == hs.format_domtblout(r1, names, [""] * len(names))
    finally:
        p7.set_simd(False)
        hs.close()
