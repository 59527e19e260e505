"""oracle/gene_oracle.c (the restated node extraction of the gene finder CheckM runs before the scan, checkm/prodigal.py:74,86-93)
against an independent formulation written here: the C code scans every strand backwards with three per-frame registers, as the
published source does; this file cuts every frame into the stretches between its stop codons and applies the rules stretch by stretch.
Neither is pinned to a real prodigal (none exists here): the test pins the restatement against itself stated another way, plus
hand-derived nodes of a designed ORF."""
import random

from oracle import genes

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
STOPS = {11: {"TAA", "TAG", "TGA"}, 4: {"TAA", "TAG"}}
STARTS = {"ATG": genes.ATG, "GTG": genes.GTG, "TTG": genes.TTG}


def by_stretches(seq, tt, closed=False):
    slen = len(seq)
    out = []
    if slen < 3:
        return out
    for strand in (1, -1):
        s = seq if strand == 1 else "".join(COMP.get(c, "N") for c in reversed(seq))
        fix = (lambda x: x) if strand == 1 else (lambda x: slen - 1 - x)
        for f in range(3):
            pos = list(range(f, slen - 2, 3))                         # codon positions of the frame, ascending
            if not pos:
                continue
            # stretch by stretch from the right: cur_right closes the current stretch (a real stop, or the frame's last complete codon /
            # a virtual stop beyond the sequence while no stop has been met)
            cur_right, cur_real, seen_stop = (pos[-1] if not closed else slen + ((f - slen % 3) % 3)), False, False
            idx = len(pos) - 1
            saw = False
            while idx >= 0:
                i = pos[idx]
                if s[i:i + 3] in STOPS[tt]:
                    if saw:
                        out.append((fix(cur_right), genes.STOP, strand, fix(i), 0 if cur_real else 1))
                    cur_right, cur_real, seen_stop, saw = i, True, True, False
                elif cur_right < slen:
                    need = 90 if seen_stop else 60
                    cod = s[i:i + 3]
                    if cod in STARTS and cur_right - i + 3 >= need:
                        out.append((fix(i), STARTS[cod], strand, fix(cur_right), 0)); saw = True
                    elif i <= 2 and not closed and cur_right - i > 60:
                        out.append((fix(i), genes.ATG, strand, fix(cur_right), 1)); saw = True
                idx -= 1
            if saw:
                out.append((fix(cur_right), genes.STOP, strand, fix(f - 6), 0 if cur_real else 1))
    return sorted(out, key=lambda n: (n[0], n[2], n[1], n[3], n[4]))


def test_designed_orf_has_the_expected_nodes():
    orf = "ATG" + "GCT" * 40 + "TAA"
    s = "CC" + orf + "CCCC"
    n = genes.nodes(s, 11)
    assert (2, genes.ATG, 1, 125, 0) in n                      # the start at 2 closes at the TAA at 125
    assert (125, genes.STOP, 1, -4, 0) in n                    # its stop node: no earlier stop in frame 2 -> stop_val = frame - 6
    assert (0, genes.ATG, 1, 129, 1) in n and (1, genes.ATG, 1, 127, 1) in n       # frames 0 and 1 run off both ends: edge starts
    assert all(x[1] != genes.STOP or x[4] == 1 or x[0] == 125 for x in n if x[2] == 1)
    # table 4 reads TGA through: an ORF closed by TGA under table 11 stays open
    s2 = "CC" + "ATG" + "GCT" * 40 + "TGA" + "GCT" * 10 + "TAA" + "CC"
    assert (2, genes.ATG, 1, 125, 0) in genes.nodes(s2, 11)
    assert (2, genes.ATG, 1, 158, 0) in genes.nodes(s2, 4)


def test_c_scan_equals_the_stretch_formulation():
    rng = random.Random(20260925)
    cases = ["", "A", "AT", "ATG", "ATGTAA", "TTATTATTA", "N" * 50]
    for _ in range(300):
        n = rng.choice([3, 4, 5, 59, 60, 61, 62, 63, 64, 65, 89, 90, 91, 92, 93, 94, 95, 96, 200, 301, 1000, 3001])
        alpha = rng.choice(["ACGT", "ACGT", "ACGTN", "AAGCTT", "ATG", "ACGTacgtn"])
        cases.append("".join(rng.choice(alpha) for _ in range(n)))
    # stop-poor sequences: long open stretches in several frames
    for _ in range(40):
        cases.append("".join(rng.choice("ACG") + rng.choice("CG") + rng.choice("ACGT") for _ in range(rng.choice([20, 31, 70, 200]))))
    for s in cases:
        up = s.upper().replace("U", "T")
        for tt in (11, 4):
            for closed in (False, True):
                got = sorted(genes.nodes(s, tt, closed), key=lambda n: (n[0], n[2], n[1], n[3], n[4]))
                assert got == by_stretches(up, tt, closed), (s[:80], tt, closed)
