"""Host side of the device gene caller (checkm_amd/geneFinder.py) without a GPU: the genes.faa / genes.gff / genes.fna it writes are read
back by the mirrors of the reference's own parsers (checkm/prodigal.py:185-274), and the choice between translation tables 11 and 4 follows
checkm/prodigal.py:117-133."""
import os

import numpy as np

from checkm_amd import geneFinder
from checkm_amd.prodigal import ProdigalFastaParser, ProdigalGeneFeatureParser


def _columns(rng, n, ncontigs, contig_len):
    cols = {"bin": np.zeros(n, np.uint32), "contig": np.sort(rng.integers(0, ncontigs, n)).astype(np.uint32)}
    b = rng.integers(1, contig_len - 1000, n)
    cols["begin"] = b.astype(np.int32)
    cols["end"] = (b + 3 * rng.integers(30, 300, n) - 1).astype(np.int32)
    cols["strand"] = rng.choice([1, -1], n).astype(np.int8)
    cols["start_type"] = rng.integers(0, 4, n).astype(np.uint8)
    cols["partial_left"] = rng.integers(0, 2, n).astype(np.uint8)
    cols["partial_right"] = rng.integers(0, 2, n).astype(np.uint8)
    cols["rbs_bin"] = rng.integers(-1, 28, n).astype(np.int32)
    ml = rng.integers(0, 7, n).astype(np.int32)
    ml[ml < 3] = 0
    cols["mot_len"] = ml
    cols["mot_ndx"] = rng.integers(0, 64, n).astype(np.int32)
    cols["mot_spacer"] = rng.integers(3, 15, n).astype(np.int32)
    for f in ("gc_cont", "conf", "score", "cscore", "sscore", "rscore", "uscore", "tscore"):
        cols[f] = rng.normal(0, 10, n)
    cols["proteins"] = ["".join(rng.choice(list("ACDEFGHIKLMNPQRSTVWY"), int((e - s + 1) // 3))) for s, e in zip(cols["begin"], cols["end"])]
    return cols


def _bin(seed, n=300, ncontigs=4, contig_len=6000, table=11):
    rng = np.random.default_rng(seed)
    contigs = [("ctg%d" % i, "".join(rng.choice(list("ACGT"), contig_len))) for i in range(ncontigs)]
    cols = _columns(rng, n, ncontigs, contig_len)
    return contigs, cols, geneFinder.BinGenes(contigs, table, cols, slice(0, n), 1, 1, 0.5)


def test_written_files_read_back_by_the_reference_parsers(tmp_path):
    contigs, cols, g = _bin(3)
    aa, gff, fna = (str(tmp_path / x) for x in ("genes.faa", "genes.gff", "genes.fna"))
    g.write(aa, gff, fna)
    # positions from the FASTA headers (checkm/prodigal.py:185-205): `>contig_k # begin # end # strand # ...`, k counting from 1 per contig
    pos = ProdigalFastaParser().genePositions(aa)
    assert len(pos) == g.n
    k_of = {}
    for i in range(g.n):
        c = int(cols["contig"][i])
        k_of[c] = k_of.get(c, 0) + 1
        assert pos["ctg%d_%d" % (c, k_of[c])] == [int(cols["begin"][i]), int(cols["end"][i])]
    assert ProdigalFastaParser().genePositions(fna) == pos
    # the GFF (checkm/prodigal.py:208-274): translation table, genes per contig, coding bases with overlaps counted once
    p = ProdigalGeneFeatureParser(gff)
    assert p.translationTable == 11
    assert sum(len(v) for v in p.genes.values()) == g.n
    assert sum(p.codingBases(cid) for cid, _s in contigs) == g.coding_bases()
    # the nucleotide record of a reverse-strand gene is the reverse complement of the contig's bases
    comp = str.maketrans("ACGT", "TGCA")
    with open(fna) as f:
        recs = f.read().split(">")[1:]
    for i, rec in enumerate(recs[:40]):
        head, body = rec.split("\n", 1)
        c, s, e, st = int(cols["contig"][i]), int(cols["begin"][i]), int(cols["end"][i]), int(cols["strand"][i])
        want = contigs[c][1][s - 1:e]
        assert body.replace("\n", "") == (want if st == 1 else want.translate(comp)[::-1])
    # proteins: 60 residues per line, the record's sequence unchanged
    with open(aa) as f:
        recs = f.read().split(">")[1:]
    assert [r.split("\n", 1)[1].replace("\n", "") for r in recs] == list(cols["proteins"])
    assert all(len(ln) <= 60 for r in recs for ln in r.split("\n")[1:])


def test_rows_view_and_selection_forms_agree():
    contigs, cols, g = _bin(5, n=50)
    g2 = geneFinder.BinGenes(contigs, 11, cols, list(range(50)), 1, 1, 0.5)          # an index list selects what the slice does
    assert g.rows == g2.rows and g.n == g2.n == 50
    assert g.coding_bases() == g2.coding_bases()
    empty = geneFinder.BinGenes(contigs, 11, cols, slice(0, 0), 1, 1, 0.5)
    assert empty.n == 0 and empty.coding_bases() == 0 and empty.rows == []


def test_table_choice_is_the_reference_rule():
    class G(object):
        def __init__(self, n):
            self.n = n

        def coding_bases(self):
            return self.n
    total = 1000
    # table 4 only when its density beats table 11's by MORE than 0.05 AND exceeds 0.7 (checkm/prodigal.py:131-133)
    assert geneFinder.best_table(G(600), G(800), total)[0] == 4
    assert geneFinder.best_table(G(720), G(760), total)[0] == 11          # 0.04 more is not enough
    assert geneFinder.best_table(G(700), G(750), total)[0] == 4           # (0.75 - 0.70 > 0.05 in doubles, in the reference's expression too)
    assert geneFinder.best_table(G(600), G(700), total)[0] == 11          # 0.7 is not exceeded
    assert geneFinder.best_table(G(900), G(800), total)[0] == 11
    assert geneFinder.best_table(G(0), G(0), 0) == (11, {11: 0, 4: 0})


def test_small_bin_is_refused_with_the_reason(tmp_path, monkeypatch):
    # below 20 kb the caller cannot train; the error names what CheckM would have done (-p meta) instead of returning silently
    f = tmp_path / "tiny.fna"
    f.write_text(">c1\n" + "ACGT" * 100 + "\n")

    class Fake(object):
        trained, n = False, 0

        def coding_bases(self):
            return 0
    monkeypatch.setattr(geneFinder, "call_bins", lambda bins, table, mask=True, ctx=None: [Fake() for _ in bins])
    monkeypatch.setattr(geneFinder.runtime, "get_ctx", lambda: None)
    monkeypatch.setattr(geneFinder.runtime, "get_ctx_k", lambda k: None)
    try:
        geneFinder.call_bin_files([(str(f), str(tmp_path))])
    except ValueError as e:
        assert "-p meta" in str(e) and "tiny.fna" in str(e)
    else:
        raise AssertionError("a 400-base bin was accepted")
    assert not os.path.exists(str(tmp_path / "genes.faa"))
