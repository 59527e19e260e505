"""The gene-calling front end on the device (ckm_orf_scan: codon flags + start / stop nodes of all six frames, checkm_amd/csrc/
kernels_orf.hip) against the oracle (oracle/gene_oracle.c), node for node: contigs of every length class around the kernel's 64-base
windows and 64-codon steps, ambiguous bases, lower case, many contigs in one call, both translation tables CheckM asks prodigal for
(checkm/prodigal.py:86-93), closed and open ends, and a 2 Mb contig.  Integer work: the bar is equality."""
import random

import numpy as np
import pytest

from checkm_amd import _lib
from oracle import genes

pytestmark = pytest.mark.gpu


def oracle_nodes(contigs, tt, closed):
    out = []
    for c, s in enumerate(contigs):
        for ndx, typ, strand, sv, edge in genes.nodes(s, tt, closed, sort=False):
            out.append((c, ndx, 1 if strand == -1 else 0, typ, sv, edge))
    return sorted(out, key=lambda n: (n[0], n[1], n[2], n[3], n[4], n[5]))        # (position, forward strand first: node.c compare_nodes)


def device_nodes(ctx, contigs, tt, closed):
    cols, stats = _lib.orf_nodes(ctx, contigs, tt, closed)
    got = list(zip(cols["contig"].tolist(), cols["ndx"].tolist(), cols["strand_rev"].tolist(), cols["type"].tolist(), cols["stop_val"].tolist(), cols["edge"].tolist()))
    return got, stats


def test_nodes_equal_the_oracle(gpu_ctx):
    rng = random.Random(4)
    contigs = ["", "A", "AT", "ATG", "ATGTAA", "N" * 70, "ATG" + "GCT" * 40 + "TAA"]
    for n in list(range(57, 70)) + list(range(186, 200)) + [383, 384, 385, 575, 576, 577, 1000, 4097, 12289]:
        contigs.append("".join(rng.choice("ACGT") for _ in range(n)))
    for n in (300, 2000, 9000):                     # stop-poor: open stretches that span many 64-codon steps
        contigs.append("".join(rng.choice("ACG") + rng.choice("CG") + rng.choice("ACGT") for _ in range(n)))
    contigs.append("".join(rng.choice("ACGTNacgtn") for _ in range(5000)))
    contigs.append("".join(rng.choice("ATG") for _ in range(3000)))
    for tt in (11, 4):
        for closed in (False, True):
            got, stats = device_nodes(gpu_ctx, contigs, tt, closed)
            want = oracle_nodes(contigs, tt, closed)
            assert len(got) == len(want) and len(want) > 1000, (len(got), len(want))
            assert got == want, (tt, closed, next((a, b) for a, b in zip(got, want) if a != b))
            assert stats["bases"] == sum(len(c) for c in contigs)


def test_a_genome_sized_contig_and_many_small_ones(gpu_ctx):
    rng = np.random.default_rng(11)
    big = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=2_000_003)].tobytes().decode()
    small = [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, size=int(n))].tobytes().decode() for n in rng.integers(1, 4000, size=400)]
    contigs = [big] + small
    got, stats = device_nodes(gpu_ctx, contigs, 11, False)
    want = oracle_nodes(contigs, 11, False)
    assert got == want and len(got) > 100000
    assert stats["ms_flags"] > 0 and stats["padded_bytes"] >= stats["bases"]
