"""The gene-calling pipeline of libcheckm_hip.so on the CPU: checkm_amd/csrc/gene_pipe.h + gene_dev.h compiled against a host executor
(tests/emu: one loop iteration per device thread, the cooperating kernels restated as scalar loops) and diffed with the gene oracle
(oracle/gene_full.c, the restatement of Prodigal 2.6.3's single-genome mode as CheckM invokes it, checkm/prodigal.py:80-93; parity
unpinned) gene for gene, score for score (float64 bit patterns), protein for protein.  What this pins without a GPU: the rank-based node
order, the chain arrays, the GC-frame range counts, the frame sweeps as per-reading-frame walks, the count / logarithm split of the
training loops, the -m masks as range counts, the record builder.  The kernels themselves are checked by tests/test_gpu_genes.py."""
import numpy as np
import pytest

from synthdata import synth_genome as sg
from oracle import genes as og
from tests import emu


def _genomes():
    return [sg.make_genome(100, n_contigs=2, contig_len=(15000, 60000), gc=0.35, sd_frac=0.0),               # no Shine-Dalgarno sites: upstream motifs
            sg.make_genome(101, n_contigs=3, contig_len=(15000, 30000), gc=0.38, sd_frac=0.6),
            sg.make_genome(200, n_contigs=1, contig_len=(30000, 40000)),                                      # one contig: no separators
            sg.make_genome(202, n_contigs=3, contig_len=(20000, 30000), n_runs=6),                            # runs of N around the 50-base mask edge
            sg.make_genome(204, n_contigs=2, contig_len=(4000, 6000)),                                        # < 20 kb: untrained
            sg.make_genome(205, n_contigs=3, contig_len=(20000, 30000), table=4, gc=0.3)]                     # genes that read TGA as Trp


def _key(cols, k):
    return tuple(int(cols[f][k]) for f in ("contig", "begin", "end", "strand", "start_type", "partial_left", "partial_right", "rbs_bin", "mot_len", "mot_ndx", "mot_spacer"))


def _okey(g):
    return (g.contig, g.begin, g.end, g.strand, g.start_type, g.partial_left, g.partial_right, g.rbs_bin, g.mot_len, g.mot_ndx, g.mot_spacer)


def _plain(g):
    return [c[1] if isinstance(c, tuple) else c for c in g]


@pytest.mark.parametrize("table", [11, 4])
def test_emulated_pipeline_equals_the_oracle(table):
    from tests import common
    genomes = [_plain(g) for g in _genomes()] + common.edge_genomes()
    cols, per_bin = emu.call_genes(genomes, table)
    by_bin = {}
    for k in range(len(cols["begin"])):
        by_bin.setdefault(int(cols["bin"][k]), []).append(k)
    ngenes, kinds = 0, set()
    for b, g in enumerate(genomes):
        t, ogenes, oprots = og.find_genes(g, table)
        ks = by_bin.get(b, [])
        if t is None:
            assert not per_bin["trained"][b] and not ks
            continue
        assert per_bin["trained"][b] and int(per_bin["uses_sd"][b]) == t.uses_sd, b
        kinds.add(t.uses_sd)
        assert float(per_bin["gc"][b]) == t.gc
        assert [_key(cols, k) for k in ks] == [_okey(x) for x in ogenes], b
        for f in ("gc_cont", "conf", "score", "cscore", "sscore", "rscore", "uscore", "tscore"):
            got = np.asarray([cols[f][k] for k in ks], dtype=np.float64).view(np.uint64)
            want = np.asarray([getattr(x, f) for x in ogenes], dtype=np.float64).view(np.uint64)
            assert (got == want).all(), (b, f, np.nonzero(got != want)[0][:3])
        assert [cols["proteins"][k] for k in ks] == oprots, b
        ngenes += len(ks)
    assert ngenes > 200 and kinds == {0, 1}, (ngenes, kinds)          # both start-site models were trained


def _same_as_oracle(genomes, table):
    cols, per_bin = emu.call_genes(genomes, table)
    by_bin = {}
    for k in range(len(cols["begin"])):
        by_bin.setdefault(int(cols["bin"][k]), []).append(k)
    ngenes = 0
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
        ngenes += len(ks)
    return ngenes


def test_emulated_pipeline_on_the_dynamic_programs_hard_cases():
    """A giant open reading frame, thousands of starts in one frame, node-dense repeats, hundreds of short contigs (tests/common.py:
    dp_stress_genomes) -- the inputs on which the device's dynamic program leaves its rings (tests/test_gpu_genes.py runs the same bins)."""
    from tests import common
    assert _same_as_oracle(common.dp_stress_genomes(), 11) > 100


def test_product_does_not_load_the_emulation():
    import os
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "checkm_amd")
    for d, _sub, files in os.walk(root):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp")) or (f.endswith(".h") and f != "gene_dev.h" and f != "gene_exec.h"):
                with open(os.path.join(d, f), errors="ignore") as fh:
                    text = fh.read()
                assert "gene_emu" not in text or f == "ckm_genes.hip" and "tests/test_gene_emu.py" in text, f
