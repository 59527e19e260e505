"""Synthetic profile HMMs and ORF sets of the shapes BASELINE.json names.

The reference ships no HMM file (custom_marker_sets/cpr_43_markers.hmm is a stripped blob,
/root/reference/.MISSING_LARGE_BLOBS:1) and no test genome, so every input of the scan half
has to be generated: HMMER3/f ASCII profiles (the format checkm/hmmerModelParser.py:54-83
skims) and prodigal-style protein FASTA (`<contig>_<n>` names, trailing '*', cf. the sample
row in checkm/hmmer.py:188).

Nothing here touches the oracle or the GPU; calibration constants (STATS LOCAL lines) come
from tools/calibrate_synth.py, which is test tooling, and are stored in synth_stats.json.
"""
import json
import os

import numpy as np

AMINO = "ACDEFGHIKLMNPQRSTVWY"
# Swiss-Prot 50.8 background used by HMMER's null model
BGF = np.array([0.0787945, 0.0151600, 0.0535222, 0.0668298, 0.0397062, 0.0695071, 0.0229198,
                0.0590092, 0.0594422, 0.0963728, 0.0237718, 0.0414386, 0.0482904, 0.0395639,
                0.0540978, 0.0683364, 0.0540687, 0.0673417, 0.0114135, 0.0304133])
BGF = BGF / BGF.sum()

_STATS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_stats.json")


class Profile(object):
    """Probability-space profile: mat/ins [M+1,20], t [M+1,7] in HMMER order MM MI MD IM II DM DD."""

    def __init__(self, name, acc, M, mat, ins, t, desc="synthetic", ga=None, tc=None, nc=None, stats=None):
        self.name, self.acc, self.M, self.desc = name, acc, M, desc
        self.mat, self.ins, self.t = mat, ins, t
        self.ga, self.tc, self.nc = ga, tc, nc
        self.stats = stats  # (mmu, mlambda, vmu, vlambda, ftau, flambda)


def random_profile(rng, M, name, acc, conc=0.25):
    """Dirichlet match emissions (conc < 1 gives peaked columns ~1.5-2 bits), background inserts."""
    mat = np.zeros((M + 1, 20))
    ins = np.zeros((M + 1, 20))
    t = np.zeros((M + 1, 7))
    mat[0, 0] = 1.0
    alpha = conc * 20 * BGF
    mat[1:] = rng.dirichlet(alpha, size=M)
    mat[1:] = 0.9 * mat[1:] + 0.1 * BGF  # no zero probabilities
    ins[:] = BGF
    for k in range(0, M + 1):
        mi = rng.uniform(0.005, 0.03)
        md = rng.uniform(0.005, 0.03)
        ii = rng.uniform(0.3, 0.6)
        dd = rng.uniform(0.2, 0.5)
        t[k] = [1 - mi - md, mi, md, 1 - ii, ii, 1 - dd, dd]
    t[0, 5:] = [1.0, 0.0]          # no D_0
    t[M, 0:3] = [1 - t[M, 1], t[M, 1], 0.0]  # M_M -> E, no D_{M+1}
    t[M, 5:] = [1.0, 0.0]
    return Profile(name, acc, M, mat, ins, t)


def _fmt(p):
    return "       *" if p <= 0.0 else "%8.5f" % (0.0 - np.log(p) + 0.0)


def _compo(prof):
    M = prof.M
    mocc = np.zeros(M + 1)
    mocc[1] = prof.t[0, 1] + prof.t[0, 0]
    for k in range(2, M + 1):
        mocc[k] = mocc[k - 1] * (prof.t[k - 1, 0] + prof.t[k - 1, 1]) + (1 - mocc[k - 1]) * prof.t[k - 1, 5]
    c = (prof.mat[1:] * mocc[1:, None]).sum(0)
    return c / c.sum()


def _row(vals, probs, n):
    """`n` numbers of a body line: -ln(p) as %8.5f, '*' for probability 0 (one format call per line, not per number)."""
    if probs.min() > 0.0:
        return _ROW_FMT[n] % tuple(vals)
    return " ".join("       *" if p <= 0.0 else "%8.5f" % v for v, p in zip(vals, probs))


_ROW_FMT = {20: " ".join(["%8.5f"] * 20), 7: " ".join(["%8.5f"] * 7)}


def hmm_text(p):
    """One profile as HMMER3/f ASCII text (layout: SURVEY.md appendix B2)."""
    out = ["HMMER3/f [3.1b2 | February 2015]\n", "NAME  %s\n" % p.name]
    if p.acc:
        out.append("ACC   %s\n" % p.acc)
    out.append("DESC  %s\n" % p.desc)
    out.append("LENG  %d\n" % p.M)
    out.append("ALPH  amino\nRF    no\nMM    no\nCONS  no\nCS    no\nMAP   no\n")
    out.append("NSEQ  1\nEFFN  1.000000\nCKSUM 0\n")
    for tag in ("ga", "tc", "nc"):
        v = getattr(p, tag)
        if v is not None:
            out.append("%s    %.2f %.2f;\n" % (tag.upper(), v[0], v[1]))
    if p.stats is not None:
        out.append("STATS LOCAL MSV      %9.4f %8.5f\n" % (p.stats[0], p.stats[1]))
        out.append("STATS LOCAL VITERBI  %9.4f %8.5f\n" % (p.stats[2], p.stats[3]))
        out.append("STATS LOCAL FORWARD  %9.4f %8.5f\n" % (p.stats[4], p.stats[5]))
    out.append("HMM     " + "".join("     %s   " % a for a in AMINO) + "\n")
    out.append("            m->m     m->i     m->d     i->m     i->i     d->m     d->d\n")
    with np.errstate(divide="ignore"):
        lm, li, lt = 0.0 - np.log(p.mat) + 0.0, 0.0 - np.log(p.ins) + 0.0, 0.0 - np.log(p.t) + 0.0
        compo = _compo(p)
        lc = 0.0 - np.log(compo) + 0.0
    out.append("  COMPO  " + _row(lc, compo, 20) + "\n")
    out.append("         " + _row(li[0], p.ins[0], 20) + "\n")
    out.append("         " + _row(lt[0], p.t[0], 7) + "\n")
    for k in range(1, p.M + 1):
        out.append("%7d  " % k + _row(lm[k], p.mat[k], 20) + "      - - - - -\n")
        out.append("         " + _row(li[k], p.ins[k], 20) + "\n")
        out.append("         " + _row(lt[k], p.t[k], 7) + "\n")
    out.append("//\n")
    return "".join(out)


def write_hmm(path, profiles, mode="w"):
    """HMMER3/f ASCII writer."""
    with open(path, mode) as f:
        for p in profiles:
            f.write(hmm_text(p))


def _cdf(prof):
    if not hasattr(prof, "_mat_cdf"):
        prof._mat_cdf = np.cumsum(prof.mat / prof.mat.sum(axis=1, keepdims=True), axis=1)
        prof._bg_cdf = np.cumsum(BGF)
    return prof._mat_cdf, prof._bg_cdf


def sample_domain(rng, prof, k_from=1, k_to=None):
    """Emit one pass through the core model nodes k_from..k_to (M/I/D walk)."""
    k_to = prof.M if k_to is None else k_to
    mat_cdf, bg_cdf = _cdf(prof)
    out = []
    k, st = k_from, "M"
    while True:
        if st == "M":
            out.append(min(19, int(np.searchsorted(mat_cdf[k], rng.random()))))
        elif st == "I":
            out.append(min(19, int(np.searchsorted(bg_cdf, rng.random()))))
        if k == k_to and st != "I":
            break
        if st == "M":
            r = rng.random()
            if r < prof.t[k, 1]:
                st = "I"
            elif r < prof.t[k, 1] + prof.t[k, 2] and k + 1 <= k_to:
                st, k = "D", k + 1
            else:
                st, k = "M", k + 1
        elif st == "I":
            if rng.random() < prof.t[k, 4]:
                st = "I"
            else:
                st, k = "M", k + 1
        else:  # D
            if k == k_to:
                break
            if rng.random() < prof.t[k, 6]:
                st, k = "D", k + 1
            else:
                st, k = "M", k + 1
    return np.array(out, dtype=np.int64)


def random_residues(rng, n):
    return rng.choice(20, size=n, p=BGF)


def to_text(codes):
    return "".join(AMINO[c] for c in codes)


def orf_lengths(rng, n):
    return np.maximum(30, np.rint(rng.lognormal(5.55, 0.55, size=n))).astype(np.int64)


def make_bin(profiles, seed, n_orfs=2000, plant=True, dup_frac=0.05, orfs_per_contig=40):
    """One synthetic bin: list of (name, description, protein-with-trailing-*).

    Every model gets one planted full-length ORF; dup_frac of the models additionally get a
    duplicate ORF or are split over two adjacent ORFs (exercises resultsParser.py:401-479).
    """
    rng = np.random.default_rng(seed)
    lens = orf_lengths(rng, n_orfs)
    flat = random_residues(rng, int(lens.sum()))
    cuts = np.cumsum(lens)[:-1]
    seqs = np.split(flat, cuts)
    if plant and profiles:
        slots = rng.permutation(n_orfs - 1)[: 2 * len(profiles)]
        for mi, p in enumerate(profiles):
            s = int(slots[2 * mi])
            dom = sample_domain(rng, p)
            fl = random_residues(rng, int(rng.integers(5, 40)))
            fr = random_residues(rng, int(rng.integers(5, 40)))
            r = rng.random()
            if r < dup_frac / 2 and p.M >= 40:
                # adjacent split: first half in ORF s, second half in ORF s+1
                cut = int(rng.integers(p.M // 3, 2 * p.M // 3))
                a = sample_domain(rng, p, 1, cut)
                b = sample_domain(rng, p, cut + 1, p.M)
                seqs[s] = np.concatenate([fl, a, random_residues(rng, 8)])
                if s + 1 < n_orfs:
                    seqs[s + 1] = np.concatenate([random_residues(rng, 8), b, fr])
            else:
                seqs[s] = np.concatenate([fl, dom, fr])
                if r > 1 - dup_frac / 2:
                    s2 = int(slots[2 * mi + 1])
                    seqs[s2] = np.concatenate([fr, sample_domain(rng, p), fl])
    out = []
    pos = 1
    lut = np.frombuffer(AMINO.encode(), dtype=np.uint8)
    for i, sq in enumerate(seqs):
        contig = i // orfs_per_contig + 1
        n = i % orfs_per_contig + 1
        name = "c%06d_%d" % (contig, n)
        end = pos + 3 * (len(sq) + 1) - 1
        desc = "# %d # %d # 1 # ID=%d_%d;partial=00;start_type=ATG;rbs_motif=None;rbs_spacer=None" % (pos, end, contig, n)
        pos = end + 50
        out.append((name, desc, lut[sq].tobytes().decode() + "*"))
    return out


def write_fasta(path, records):
    with open(path, "w") as f:
        for name, desc, seq in records:
            f.write(">%s %s\n" % (name, desc))
            for i in range(0, len(seq), 60):
                f.write(seq[i:i + 60] + "\n")


def cpr43_accessions():
    """The 43 CPR marker accessions (31 TIGRFAM + 12 Pfam) -- names only, bodies are synthetic."""
    tig = ["TIGR%05d" % n for n in (2, 9, 12, 19, 29, 43, 59, 60, 61, 62, 64, 82, 86, 92, 115, 116,
                                      158, 166, 337, 344, 362, 422, 435, 468, 755, 810, 855, 922, 952, 1009, 3631)]
    pf = ["PF00276.21", "PF00281.20", "PF00297.23", "PF00347.24", "PF00366.21", "PF00410.20",
          "PF00466.21", "PF00573.23", "PF00687.22", "PF00831.24", "PF01409.21", "PF13393.7"]
    return tig + pf


def cpr43_profiles(seed=43, with_stats=True):
    """43 profiles with M drawn from [60,900] (cfg2 of BASELINE.json)."""
    rng = np.random.default_rng(seed)
    accs = cpr43_accessions()
    Ms = rng.integers(60, 901, size=len(accs))
    profs = []
    stats = load_stats() if with_stats else {}
    for acc, M in zip(accs, Ms):
        name = acc.split(".")[0] if acc.startswith("TIGR") else "synth_" + acc.split(".")[0]
        p = random_profile(rng, int(M), name, acc)
        key = "cpr43/%s" % acc
        if key in stats:
            p.stats = tuple(stats[key]["stats"])
            if acc.startswith("TIGR"):
                p.tc = (stats[key]["cut"], stats[key]["cut"])
                p.nc = (stats[key]["cut"] - 10.0, stats[key]["cut"] - 10.0)
            else:
                p.ga = (stats[key]["cut"], stats[key]["cut"])
        profs.append(p)
    return profs


# (seed, n, mlo, mhi) families calibrated by tools/calibrate_synth.py
SMALL_SETS = [(7, 6, 20, 150), (11, 12, 40, 300), (13, 4, 300, 1100), (17, 2, 1300, 2048)]


def small_profiles(seed, n, mlo=20, mhi=150, prefix="SYN", with_stats=True):
    rng = np.random.default_rng(seed)
    stats = load_stats() if with_stats else {}
    profs = []
    for i in range(n):
        M = int(rng.integers(mlo, mhi + 1))
        acc = "PF%05d.%d" % (90000 + i, 1 + i % 5) if i % 2 == 0 else "TIGR%05d" % (90000 + i)
        p = random_profile(rng, M, "%s%03d" % (prefix, i), acc)
        key = "small/%d/%s" % (seed, acc)
        if key in stats:
            p.stats = tuple(stats[key]["stats"])
            if i % 3 != 2:       # every third model has no cutoffs: exercises the E-value branch of vetHit
                if acc.startswith("TIGR"):
                    p.tc = (stats[key]["cut"], stats[key]["cut"])
                    p.nc = (stats[key]["cut"] - 5.0, stats[key]["cut"] - 5.0)
                else:
                    p.ga = (stats[key]["cut"], stats[key]["cut"])
        profs.append(p)
    return profs


def load_stats():
    if os.path.exists(_STATS_FILE):
        with open(_STATS_FILE) as f:
            return json.load(f)
    return {}
