"""First slice of gene calling on the MI355X: the deterministic front end of the gene finder CheckM runs in front of
the marker-gene scan.

The reference calls `prodigal -p single|meta -q -m -f gff -g <table> -a genes.faa -i <bin>` twice per bin, tables 11 and 4
(checkm/prodigal.py:74,86-93,131-133), and keeps table 4 when it raises the coding density enough.  Prodigal is a third-party C
program that is in neither /root/reference nor this image; what it does falls into (1) a byte scan that finds the start / stop NODES of
all six frames, (2) a per-genome training pass, (3) a dynamic program over the nodes.  This module offers (1) on the device -- both
translation tables from one upload of the bin -- through libcheckm_hip's ckm_orf_scan; (2) and (3) are not built (DESIGN.md section
10), so `checkm_amd.prodigal.ProdigalRunner` still runs the external binary to produce genes.faa.  There is no CPU implementation
here: without a gfx950 device the library raises."""
from checkm_amd import _lib, runtime

ATG, GTG, TTG, STOP = 0, 1, 2, 3


def read_contigs(fastaFile):
    """[(id, sequence)] of a nucleotide FASTA file (plain or gzip), ids cut at the first whitespace as the reference's readFasta does
    (checkm/util/seqUtils.py:180-211)."""
    import gzip
    opener = gzip.open if fastaFile.endswith('.gz') else open
    out, name, seq = [], None, []
    with opener(fastaFile, 'rt') as f:
        for line in f:
            if line.startswith('>'):
                if name is not None:
                    out.append((name, ''.join(seq)))
                name, seq = line[1:].split(None, 1)[0] if line[1:].strip() else '', []
            elif name is not None:
                seq.append(line.strip())
    if name is not None:
        out.append((name, ''.join(seq)))
    return out


class OrfNodes(object):
    """Start / stop nodes of a bin's contigs for one translation table: numpy columns contig, ndx, stop_val, type, strand_rev, edge."""

    def __init__(self, contigs, transTable=11, closedEnds=False):
        cols, self.stats = _lib.orf_nodes(runtime.get_ctx(), contigs, transTable, closedEnds)
        for k, v in cols.items():
            setattr(self, k, v)
        self.n = len(self.ndx)

    def orfs(self, minLength=90):
        """(contig, start, stop, strand) of every start node whose ORF is closed by a real stop and spans at least minLength bases:
        the candidates the gene finder's dynamic program chooses among."""
        sel = (self.type != STOP) & (self.edge == 0) & (abs(self.stop_val - self.ndx) + 3 >= minLength)
        return list(zip(self.contig[sel].tolist(), self.ndx[sel].tolist(), self.stop_val[sel].tolist(), (1 - 2 * self.strand_rev[sel].astype(int)).tolist()))
