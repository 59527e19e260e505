"""User-supplied output of a REAL prodigal (tests/golden/prodigal_real/README.md): every (NAME.fna, NAME.gff) pair found there is compared,
line by line, with what the gene oracle (CPU) and the device gene caller (-m gpu) give for the same bin, written by the product's own
writer in prodigal's layout.  Reference call being matched: checkm/prodigal.py:80-93 (`-p single -m -f gff -g <table>`).  No pair exists
yet (no prodigal anywhere near this repository): the tests skip themselves and the gene-calling oracle stays "parity unpinned"."""
import glob
import os

import numpy as np
import pytest

from checkm_amd import geneFinder

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "prodigal_real")
FIELDS = ("bin", "contig", "begin", "end", "strand", "start_type", "partial_left", "partial_right", "rbs_bin", "mot_len", "mot_ndx", "mot_spacer",
          "gc_cont", "conf", "score", "cscore", "sscore", "rscore", "uscore", "tscore")


def pairs():
    out = []
    for fna in sorted(glob.glob(os.path.join(HERE, "*.fna"))):
        base = fna[:-4]
        for table, gff in ((11, base + ".gff"), (4, base + ".t4.gff")):
            if os.path.exists(gff):
                out.append((fna, gff, table))
    return out


def cds_lines(path):
    """(seqid, begin, end, score, strand, attributes without the ID) of every CDS line: the ID's sequence number is prodigal's own count."""
    out = []
    for line in open(path):
        if line.startswith("#") or not line.strip():
            continue
        t = line.rstrip("\n").split("\t")
        attrs = [a for a in t[8].rstrip(";").split(";") if not a.startswith("ID=")]
        out.append((t[0], int(t[3]), int(t[4]), t[5], t[6], ";".join(attrs)))
    return out


def written_by_us(contigs, table, cols, tmp_path):
    g = geneFinder.BinGenes(contigs, table, cols, slice(0, len(cols["begin"])), 1, 1, 0.5)
    g.write(str(tmp_path / "ours.faa"), str(tmp_path / "ours.gff"))
    return cds_lines(str(tmp_path / "ours.gff"))


def oracle_columns(contigs, table):
    from oracle import genes as og
    t, genes, prots = og.find_genes([s for _c, s in contigs], table)
    assert t is not None, "the bin is below 20 kb: prodigal -p single refuses it too"
    cols = {f: np.asarray([getattr(x, f) if f != "bin" else 0 for x in genes]) for f in FIELDS}
    cols["proteins"] = prots
    return cols


@pytest.mark.skipif(not pairs(), reason="no real prodigal output in tests/golden/prodigal_real (see its README): the gene oracle is parity-unpinned")
@pytest.mark.parametrize("fna,gff,table", pairs())
def test_oracle_against_prodigal_gff(fna, gff, table, tmp_path):
    contigs = geneFinder.read_contigs(fna)
    assert written_by_us(contigs, table, oracle_columns(contigs, table), tmp_path) == cds_lines(gff)


@pytest.mark.gpu
@pytest.mark.skipif(not pairs(), reason="no real prodigal output in tests/golden/prodigal_real (see its README)")
@pytest.mark.parametrize("fna,gff,table", pairs())
def test_device_against_prodigal_gff(fna, gff, table, tmp_path, gpu_ctx):
    from checkm_amd import _lib
    contigs = geneFinder.read_contigs(fna)
    cols, _per_bin, _stats = _lib.call_genes(gpu_ctx, [[s for _c, s in contigs]], table)
    assert written_by_us(contigs, table, cols, tmp_path) == cds_lines(gff)


def test_the_comparison_itself_on_a_hand_made_pair(tmp_path):
    """The machinery above on a synthetic 'prodigal' file made from the oracle's own genes with prodigal's source column and ID numbering:
    equal apart from the ID, and a changed coordinate or score is seen."""
    from synthdata import synth_genome as sg
    contigs = sg.make_genome(77, n_contigs=2, contig_len=(15000, 20000))
    cols = oracle_columns(contigs, 11)
    ours = written_by_us(contigs, 11, cols, tmp_path)
    assert len(ours) > 20
    fake = tmp_path / "real.gff"
    with open(tmp_path / "ours.gff") as f, open(fake, "w") as g:
        for line in f:
            g.write(line.replace("checkm_amd_device", "Prodigal_v2.6.3").replace("ID=1_", "ID=7_") if not line.startswith("#") else line)
    assert cds_lines(str(fake)) == ours
    text = open(fake).read().replace("\t%d\t" % ours[3][1], "\t%d\t" % (ours[3][1] + 3), 1)
    open(fake, "w").write(text)
    assert cds_lines(str(fake)) != ours
