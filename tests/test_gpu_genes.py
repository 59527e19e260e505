"""Gene calling on the device (ckm_genes_call: training, node scores, both dynamic programs, gene records, translations) against the CPU
oracle oracle/gene_full.c -- the restatement of Prodigal 2.6.3's single-genome mode as CheckM invokes it (checkm/prodigal.py:80-93;
parity unpinned: no prodigal exists next to the reference) -- gene for gene, score for score (float64 bit patterns), protein for
protein; and through MarkerGeneFinder.find from nucleotide bins."""
import os

import numpy as np
import pytest

from checkm_amd import _lib, geneFinder
from synthdata import synth_genome as sg
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
    from tests import common
    genomes = _genomes() + [[("e%d" % k, s) for k, s in enumerate(g)] for g in common.edge_genomes()]      # + empty / tiny / all-N contigs, IUPAC, mask edges, the 20 kb limit
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
        if b < 22:
            assert len(ks) >= sum(len(s) for _c, s in g) // 2500             # the planted genes are found (about one per kb)
    assert ngenes > 2000
    assert stats["ms_dp_train"] > 0 and stats["ms_dp_find"] > 0


@pytest.mark.parametrize("table", [11, 4])
def test_dynamic_program_off_its_rings(gpu_ctx, table):
    """tests/common.py: dp_stress_genomes -- a 36 kb open reading frame (windows that start behind the LDS ring: global-memory fall-backs,
    16-bit distances that do not fit), two thousand starts of one frame (one class of nodes beyond its class ring: the generic candidate
    loop; overlapping starts further than a packed offset holds), node-dense repeats, two hundred short contigs (every block remainder)."""
    from tests import common
    genomes = common.dp_stress_genomes()
    cols, per_bin, stats = _lib.call_genes(gpu_ctx, genomes, table)
    by_bin = {}
    for k in range(len(cols["begin"])):
        by_bin.setdefault(int(cols["bin"][k]), []).append(k)
    for b, g in enumerate(genomes):
        t, ogenes, oprots = og.find_genes(g, table)
        ks = by_bin.get(b, [])
        assert t is not None and per_bin["trained"][b], b
        assert [_key(cols, k) for k in ks] == [_okey(x) for x in ogenes], b
        for f in ("gc_cont", "conf", "score", "cscore", "sscore", "rscore", "uscore", "tscore"):
            got = np.asarray([cols[f][k] for k in ks], dtype=np.float64).view(np.uint64)
            want = np.asarray([getattr(x, f) for x in ogenes], dtype=np.float64).view(np.uint64)
            assert (got == want).all(), (b, f, np.nonzero(got != want)[0][:3])
        assert [cols["proteins"][k] for k in ks] == oprots, b
    assert len(cols["begin"]) > 100


def test_find_from_nucleotide_bins_writes_prodigal_files(gpu_ctx, tmp_path, monkeypatch):
    """MarkerGeneFinder.find on nucleotide bins with no prodigal on PATH: genes.faa / genes.gff come from the device caller, the table
    choice follows checkm/prodigal.py:117-133, ProdigalGeneFeatureParser reads the GFF, and the scan runs on the written proteins."""
    from checkm_amd import markerGeneFinder as mgf
    from synthdata import synth
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


def test_library_writer_equals_the_python_writer(gpu_ctx, tmp_path):
    """ckm_genes_write_bin (what call_bin_files uses) writes the bytes BinGenes.write writes -- genes.faa, genes.gff, genes.fna, reverse
    complements of IUPAC codes and lower case included -- and ckm_genes_coding_union is the interval union the table choice needs."""
    genomes = [sg.make_genome(410, n_contigs=3, contig_len=(25000, 40000)), sg.make_genome(411, n_contigs=2, contig_len=(30000, 40000), sd_frac=0.0, gc=0.4)]
    s0 = list(genomes[0][1][1])
    for k, ch in zip(range(500, 20000, 377), "RYKMSWBDHVNrykmacgt" * 10):          # ambiguity codes and lower case inside genes
        s0[k] = ch
    genomes[0][1] = (genomes[0][1][0], "".join(s0))
    batch = _lib.GeneBatch(genomes)
    for table in (11, 4):
        call = _lib.GeneCall(gpu_ctx, batch, table)
        cols = call.columns()
        at = np.searchsorted(cols["bin"], np.arange(len(genomes) + 1))
        union = call.coding_union()
        assert call.genes_per_bin().tolist() == [int(at[b + 1] - at[b]) for b in range(len(genomes))]
        for b, g in enumerate(genomes):
            bg = geneFinder.BinGenes(g, table, cols, slice(int(at[b]), int(at[b + 1])), call.per_bin["trained"][b], call.per_bin["uses_sd"][b], call.per_bin["gc"][b])
            assert int(union[b]) == bg.coding_bases()
            py = [str(tmp_path / ("py_%d_%d.%s" % (table, b, e))) for e in ("faa", "gff", "fna")]
            cc = [str(tmp_path / ("c_%d_%d.%s" % (table, b, e))) for e in ("faa", "gff", "fna")]
            bg.write(*py)
            call.write_bin(b, *cc)
            for x, y in zip(py, cc):
                with open(x, "rb") as fx, open(y, "rb") as fy:
                    assert fx.read() == fy.read(), (table, b, x)
            assert os.path.getsize(cc[0]) > 10000
        call.close()


def test_batched_files_path_warns_refuses_and_matches_single_calls(gpu_ctx, tmp_path, caplog):
    """call_bin_files: sub-batches and several calls in flight give the files single calls give; a 20-100 kb bin is called with a
    warning that names CheckM's `-p meta` (checkm/prodigal.py:80-83); a bin below 20 kb is refused before anything is written."""
    import logging
    from checkm_amd.defaultValues import DefaultValues
    jobs, genomes = [], []
    for k in range(7):
        g = sg.make_genome(500 + k, n_contigs=2 + k % 3, contig_len=(20000, 35000), gc=0.4 + 0.03 * k, table=4 if k == 5 else 11)
        f = tmp_path / ("b%d.fna" % k)
        sg.write_fasta(str(f), g)
        d = tmp_path / ("o%d" % k)
        d.mkdir()
        jobs.append((str(f), str(d))); genomes.append(g)
    logger = logging.getLogger("test_gene_files")
    with caplog.at_level(logging.WARNING, logger="test_gene_files"):
        res = geneFinder.call_bin_files(jobs, bNucORFs=True, max_bases=150000, logger=logger)
    assert geneFinder.call_bin_files.last_phases["calls"] >= 6                        # several sub-batches x two tables
    assert any("-p meta" in r.getMessage() for r in caplog.records)                    # the 40-100 kb bins say what CheckM would have done
    for (f, d), g in zip(jobs, genomes):
        total = sum(len(s) for _c, s in g)
        a, b4 = geneFinder.call_bins([g], 11)[0], geneFinder.call_bins([g], 4)[0]
        best, dens = geneFinder.best_table(a, b4, total)
        assert res[f] == (best, dens)
        ref = [str(tmp_path / ("ref.%s" % e)) for e in ("faa", "gff", "fna")]
        (a if best == 11 else b4).write(*ref)
        for x, name in zip(ref, (DefaultValues.PRODIGAL_AA, DefaultValues.PRODIGAL_GFF, DefaultValues.PRODIGAL_NT)):
            with open(x, "rb") as fx, open(os.path.join(d, name), "rb") as fy:
                assert fx.read() == fy.read(), (f, name)
    # refusal comes before any device call or file
    tiny = tmp_path / "tiny.fna"
    tiny.write_text(">c1\n" + "ACGT" * 2000 + "\n")
    od = tmp_path / "otiny"
    od.mkdir()
    od2 = tmp_path / "o_after"
    od2.mkdir()
    with pytest.raises(ValueError, match="-p meta"):
        geneFinder.call_bin_files([(jobs[0][0], str(od2)), (str(tiny), str(od))])
    assert not os.listdir(str(od)) and not os.listdir(str(od2))


def test_find_ends_the_run_when_a_bin_cannot_be_called(gpu_ctx, tmp_path, monkeypatch, caplog):
    """A bin below the 20 kb the gene finder trains on, among good ones, through MarkerGeneFinder.find: the helper thread's refusal reaches the
    main thread as logger.error + sys.exit(1) -- the reference's ending for a failed prodigal (checkm/prodigal.py:100-115) -- and no bin's
    gene files exist afterwards (the refusal comes before any call); the message names what CheckM would have done (-p meta)."""
    import logging
    from checkm_amd import markerGeneFinder as mgf
    from tests import common
    monkeypatch.setenv("CKM_GENE_CALLER", "device")
    files = []
    for k, seed in enumerate((601, 602)):
        f = tmp_path / ("ok_%d.fna" % k)
        sg.write_fasta(str(f), sg.make_genome(seed, n_contigs=2, contig_len=(25000, 30000)))
        files.append(str(f))
    tiny = tmp_path / "tiny.fna"
    tiny.write_text(">c1\n" + "ACGT" * 1500 + "\n")
    files.append(str(tiny))
    hmm = common.hmm_file("mixed", common.mixed_profiles())
    out = str(tmp_path / "out")
    with caplog.at_level(logging.ERROR, logger="timestamp"):
        with pytest.raises(SystemExit) as e:
            mgf.MarkerGeneFinder(4).find(files, out, "hmmer.analyze.txt", "hmmer.analyze.ali.txt", hmm, False, False, False)
    assert e.value.code == 1
    assert any("-p meta" in r.getMessage() and "tiny.fna" in r.getMessage() for r in caplog.records)
    for b in ("ok_0", "ok_1", "tiny"):
        assert not os.path.exists(os.path.join(out, "bins", b, "genes.faa"))
    mgf.release_scan()
