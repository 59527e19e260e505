"""oracle/gene_full.c (the restated gene finder, parity unpinned) checked without Prodigal: hand-derived cases for the pieces whose
answer can be worked out by hand, a second formulation of the dynamic program's connection window and of the coding sums, and the
properties any gene finder of this shape must have on designed genomes (planted genes recovered, proteins that translate back,
N-run masking, the table-4 signal)."""
import numpy as np

from synthdata import synth_genome as sg
from oracle import genes as og

CODON = {}
for i, a in enumerate("ACGT"):
    for j, b in enumerate("ACGT"):
        for k, c in enumerate("ACGT"):
            CODON[a + b + c] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF"[i * 16 + j * 4 + k]
COMP = str.maketrans("ACGTN", "TGCAN")


def _translate(nt, table, partial5):
    out = []
    for i in range(0, len(nt) - 2, 3):
        c = nt[i:i + 3]
        a = "X" if "N" in c else ("W" if table == 4 and c == "TGA" else CODON[c])
        out.append("M" if i == 0 and not partial5 else a)
    return "".join(out)


def test_planted_genes_are_recovered_and_translate_back():
    g = sg.make_genome(11, n_contigs=4, contig_len=(30000, 50000), sd_frac=0.7)
    seqs = [s for _c, s in g]
    t, genes, prots = og.find_genes(seqs, 11)
    assert t is not None and t.uses_sd == 1 and len(genes) > 100
    total = sum(len(s) for s in seqs)
    assert 0.80 < sum(x.end - x.begin + 1 for x in genes) / total < 0.97
    assert t.type_wt[0] > t.type_wt[1] and t.type_wt[0] > t.type_wt[2]              # ATG is the planted majority start
    assert max(range(28), key=lambda k: t.rbs_wt[k]) in (24, 27, 22, 16)              # the planted AGGAGG at 5-10 bp wins
    for x, p in zip(genes, prots):
        s = seqs[x.contig]
        nt = s[x.begin - 1:x.end]
        if x.strand == -1:
            nt = nt.translate(COMP)[::-1]
        assert (x.end - x.begin + 1) % 3 == 0
        assert p == _translate(nt, 11, x.partial_left if x.strand == 1 else x.partial_right)
        if not (x.partial_left or x.partial_right):
            assert p.endswith("*") and "*" not in p[:-1] and nt[:3] in ("ATG", "GTG", "TTG")
    # genes of a contig come in order and carry consistent strands / coordinates
    for c in range(len(seqs)):
        b = [x.begin for x in genes if x.contig == c]
        assert b == sorted(b)


def test_bins_below_20kb_are_not_trained_and_masks_hide_genes():
    small = sg.make_genome(5, n_contigs=2, contig_len=(5000, 7000))
    assert og.find_genes([s for _c, s in small], 11)[0] is None
    g = sg.make_genome(6, n_contigs=2, contig_len=(40000, 50000))
    seqs = [s for _c, s in g]
    _t, genes, _p = og.find_genes(seqs, 11)
    victim = next(x for x in genes if x.contig == 0 and x.end - x.begin > 600 and not (x.partial_left or x.partial_right))
    mid = (victim.begin + victim.end) // 2
    for run, hidden in ((49, False), (50, True), (200, True)):
        s0 = seqs[0][:mid] + "N" * run + seqs[0][mid + run:]
        _t2, g2, _p2 = og.find_genes([s0, seqs[1]], 11, mask=True)
        crossing = [x for x in g2 if x.contig == 0 and x.begin <= mid and x.end >= mid + run - 1]
        assert (not crossing) == hidden, (run, [(x.begin, x.end) for x in crossing])
        _t3, g3, _p3 = og.find_genes([s0, seqs[1]], 11, mask=False)
        assert any(x.contig == 0 and x.begin <= mid and x.end >= mid + run - 1 for x in g3) or run >= 200      # unmasked, the run is read through


def test_table_4_genomes_prefer_table_4():
    g = sg.make_genome(9, n_contigs=4, contig_len=(30000, 40000), table=4, gc=0.3)
    seqs = [s for _c, s in g]
    total = sum(len(s) for s in seqs)
    d = {}
    for tt in (11, 4):
        _t, genes, _p = og.find_genes(seqs, tt)
        d[tt] = sum(x.end - x.begin + 1 for x in genes) / total
    assert d[4] > d[11]


def test_start_site_bins_by_hand():
    """The Shine-Dalgarno bin of a designed upstream region (sequence.c's table): AGGAGG with 7 bases to the start is bin 27, GGAGG at the
    same spacing 24, AGGA 15, and an AGGAGG four bases from the start 26; a region without A/G structure is bin 0."""
    import ctypes as C
    L = og._full()
    L.pg_test_rbs.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]

    def bins(up, spacer):
        s = "C" * 30 + up + "C" * spacer + "ATG" + "GCT" * 40
        d = og.digitize(s)
        a, b = C.c_int(), C.c_int()
        L.pg_test_rbs(d.ctypes.data, len(d), 30 + len(up) + spacer, C.byref(a), C.byref(b))
        return a.value, b.value
    assert bins("AGGAGG", 7)[0] == 27
    assert bins("GGAGG", 7)[0] == 24
    assert bins("AGGA", 7)[0] == 15
    assert bins("AGGAGG", 4)[0] == 26
    assert bins("CCCCCC", 7) == (0, 0)
    assert bins("AGGCGG", 7)[1] in (19, 17, 18)           # one mismatch inside the six: the mismatch table


def test_connection_window_second_formulation():
    """dprog's candidate window, restated: node i looks back 500 nodes, and -- when the node that far back still lies at or beyond the stop
    of i's own ORF (a giant ORF) -- from the ORF's stop node 500 further back.  Compared on the node lists of real synthetic contigs."""
    import ctypes as C
    L = og._full()
    L.pg_test_window.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]
    for seed in (3, 4):
        s = sg.make_genome(seed, n_contigs=1, contig_len=(60000, 70000))[0][1]
        # a giant ORF: 4000 codons without a stop in frame 0
        rng = np.random.default_rng(seed)
        giant = "ATG" + "".join(rng.choice(["GCT", "GAA", "CTG", "AAA", "GGC"], size=4000)) + "TAA"
        s = s[:30000] + giant + s[30000:]
        d = og.digitize(s)
        cap = len(d)
        arr = [(C.c_int32 * cap)() for _ in range(5)]
        n = L.pg_test_window(d.ctypes.data, len(d), 11, arr[0], arr[1], arr[2], arr[3], arr[4], cap)
        ndx, sv, strand, typ, mn = (np.ctypeslib.as_array(a)[:n].copy() for a in arr)
        pos_of = {}
        for i in range(n):
            pos_of.setdefault(int(ndx[i]), []).append(i)
        seen_giant = 0
        for i in range(n):
            base = max(0, i - 500)
            wants_stop = (strand[i] == -1 and typ[i] != 3) or (strand[i] == 1 and typ[i] == 3)
            if wants_stop and ndx[base] >= sv[i]:
                cands = [j for j in pos_of.get(int(sv[i]), []) if j <= base]
                j = max(cands) if cands else -1
                want = 0 if j < 500 else j - 500
                seen_giant += 1
            else:
                want = 0 if base < 500 else base - 500
            assert mn[i] == want, (seed, i)
        assert seen_giant > 0
