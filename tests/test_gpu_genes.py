"""Gene calling on the device (ckm_genes_call: training, node scores, both dynamic programs, gene records, translations) against the CPU
oracle oracle/gene_full.c -- the restatement of Prodigal 2.6.3's single-genome mode as CheckM invokes it (checkm/prodigal.py:80-93;
parity unpinned: no prodigal exists next to the reference) -- gene for gene, score for score (float64 bit patterns), protein for
protein; and through MarkerGeneFinder.find from nucleotide bins."""
import os

import numpy as np
import pytest

from checkm_amd import _lib, geneFinder, synth_genome as sg
from oracle import genes as og

pytestmark = pytest.mark.gpu


def _genomes():
    out = []
    for k in range(16):
        out.append(sg.make_genome(100 + k, n_contigs=2 + k % 5, contig_len=(15000, 60000), gc=0.35 + 0.03 * (k % 10), sd_frac=0.0 if k % 4 == 0 else 0.6))
    out.append(sg.make_genome(200, n_contigs=1, contig_len=(30000, 40000)))                      # one contig: no separators; < 100 kb (CheckM's `meta` range: trained on itself here)
    out.append(sg.make_genome(201, n_contigs=3, contig_len=(7000, 9000)))                        # 20-30 kb in all
    out.append(sg.make_genome(202, n_contigs=4, contig_len=(30000, 50000), n_runs=6))            # runs of N: 10, 49, 50, 51, 120, 300 long (-m masks those >= 50)
    out.append(sg.make_genome(203, n_contigs=3, contig_len=(40000, 60000), n_runs=12, gc=0.62))
    out.append(sg.make_genome(204, n_contigs=2, contig_len=(4000, 6000)))                        # < 20 kb: cannot be trained (both sides say so)
    out.append(sg.make_genome(205, n_contigs=5, contig_len=(20000, 30000), table=4, gc=0.3))     # genes that read TGA as Trp
    return out


def _key(cols, k):
    return tuple(int(cols[f][k]) for f in ("contig", "begin", "end", "strand", "start_type", "partial_left", "partial_right", "rbs_bin", "mot_len", "mot_ndx", "mot_spacer"))


def _okey(g):
    return (g.contig, g.begin, g.end, g.strand, g.start_type, g.partial_left, g.partial_right, g.rbs_bin, g.mot_len, g.mot_ndx, g.mot_spacer)


@pytest.mark.parametrize("table", [11, 4])
def test_genes_identical_to_the_oracle(gpu_ctx, table):
    genomes = _genomes()
    cols, per_bin, stats = _lib.call_genes(gpu_ctx, [[s for _c, s in g] for g in genomes], table)
    by_bin = {}
    for k in range(len(cols["begin"])):
        by_bin.setdefault(int(cols["bin"][k]), []).append(k)
    ngenes = 0
    for b, g in enumerate(genomes):
        t, ogenes, oprots = og.find_genes([s for _c, s in g], table)
        ks = by_bin.get(b, [])
        if t is None:
            assert not per_bin["trained"][b] and not ks
            continue
        assert per_bin["trained"][b] and int(per_bin["uses_sd"][b]) == t.uses_sd, b
        assert float(per_bin["gc"][b]) == t.gc
        assert [_key(cols, k) for k in ks] == [_okey(x) for x in ogenes], b
        for f in ("gc_cont", "conf", "score", "cscore", "sscore", "rscore", "uscore", "tscore"):
            got = np.asarray([cols[f][k] for k in ks], dtype=np.float64).view(np.uint64)
            want = np.asarray([getattr(x, f) for x in ogenes], dtype=np.float64).view(np.uint64)
            assert (got == want).all(), (b, f, np.nonzero(got != want)[0][:3])
        assert [cols["proteins"][k] for k in ks] == oprots, b
        ngenes += len(ks)
        assert len(ks) >= sum(len(s) for _c, s in g) // 2500                 # the planted genes are found (about one per kb)
    assert ngenes > 2000
    assert stats["ms_dp_train"] > 0 and stats["ms_dp_find"] > 0


def test_find_from_nucleotide_bins_writes_prodigal_files(gpu_ctx, tmp_path, monkeypatch):
    """MarkerGeneFinder.find on nucleotide bins with no prodigal on PATH: genes.faa / genes.gff come from the device caller, the table
    choice follows checkm/prodigal.py:117-133, ProdigalGeneFeatureParser reads the GFF, and the scan runs on the written proteins."""
    from checkm_amd import markerGeneFinder as mgf, synth
    from checkm_amd.defaultValues import DefaultValues
    from checkm_amd.prodigal import ProdigalGeneFeatureParser
    from tests import common
    monkeypatch.setenv("CKM_GENE_CALLER", "device")
    files = []
    for k, seed in enumerate((301, 302, 303)):
        g = sg.make_genome(seed, n_contigs=3, contig_len=(25000, 40000), table=4 if k == 2 else 11, gc=0.32 if k == 2 else 0.5)
        f = tmp_path / ("bin_%d.fna" % k)
        sg.write_fasta(str(f), g)
        files.append(str(f))
    profs = common.mixed_profiles()
    hmm = common.hmm_file("mixed", profs)
    out = str(tmp_path / "out")
    models = mgf.MarkerGeneFinder(2).find(files, out, DefaultValues.HMMER_TABLE_OUT, DefaultValues.HMMER_OUT, hmm, False, False, False)
    assert sorted(models) == ["bin_0", "bin_1", "bin_2"]
    for k in range(3):
        d = os.path.join(out, "bins", "bin_%d" % k)
        faa, gff = os.path.join(d, DefaultValues.PRODIGAL_AA), os.path.join(d, DefaultValues.PRODIGAL_GFF)
        assert os.path.getsize(faa) > 0 and os.path.exists(os.path.join(d, DefaultValues.HMMER_TABLE_OUT))
        p = ProdigalGeneFeatureParser(gff)
        contigs = geneFinder.read_contigs(files[k])
        total = sum(len(s) for _c, s in contigs)
        coding = sum(p.codingBases(c) for c, _s in contigs)
        assert coding / total > 0.7
        # the same choice and the same records as a direct call
        a, b4 = geneFinder.call_bins([contigs], 11)[0], geneFinder.call_bins([contigs], 4)[0]
        best, dens = geneFinder.best_table(a, b4, total)
        assert p.translationTable == best
        assert coding == (a if best == 11 else b4).coding_bases()
        names = [ln[1:].split()[0] for ln in open(faa) if ln.startswith(">")]
        assert names[0].rsplit("_", 1)[1] == "1" and len(names) == len((a if best == 11 else b4).rows)
    mgf.release_scan(out)
