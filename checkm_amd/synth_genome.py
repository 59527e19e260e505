"""Synthetic nucleotide bins for the gene-calling leg: contigs made of genes with a codon bias, optional Shine-Dalgarno
sites upstream of their starts, both strands, separated by random intergenic sequence -- enough structure for the gene finder's training
(GC-frame bias, hexamer statistics, start-site model) to have something to learn.  Inputs only: nothing here is the product's logic."""
import numpy as np

_STOPS11 = ("TAA", "TAG", "TGA")
_COMP = str.maketrans("ACGTN", "TGCAN")


def _codon_table(rng, gc):
    """Codon usage skewed by position: third positions carry most of the GC skew, as in real genomes."""
    cods = [a + b + c for a in "ACGT" for b in "ACGT" for c in "ACGT" if a + b + c not in _STOPS11]
    w = []
    for c in cods:
        p = 1.0
        p *= (gc if c[2] in "GC" else 1.0 - gc) * 1.6 + 0.2
        p *= (0.55 if c[0] in "GC" else 0.45)
        w.append(p * rng.gamma(2.0))
    w = np.asarray(w)
    return cods, w / w.sum()


def make_genome(seed, n_contigs=6, contig_len=(30000, 80000), gc=0.5, sd_frac=0.6, n_runs=0, table=11):
    """[(contig id, sequence)]: genes of 150-2400 bp on both strands, ~88 % coding."""
    rng = np.random.default_rng(seed)
    cods, cw = _codon_table(rng, gc)
    stops = ("TAA", "TAG", "TGA") if table == 11 else ("TAA", "TAG")
    base_p = [(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]
    out = []
    for c in range(n_contigs):
        target = int(rng.integers(contig_len[0], contig_len[1]))
        parts, n = [], 0
        while n < target:
            ig = "".join(rng.choice(list("ACGT"), p=base_p, size=int(rng.integers(20, 160))))
            ncod = int(min(800, max(50, rng.lognormal(5.5, 0.6))))
            body = "".join(rng.choice(cods, p=cw, size=ncod))
            start = rng.choice(["ATG", "ATG", "ATG", "ATG", "GTG", "TTG"])
            gene = start + body + str(rng.choice(stops))
            if rng.random() < sd_frac:
                sp = "".join(rng.choice(list("ACGT"), p=base_p, size=int(rng.integers(5, 10))))
                ig = ig + "AGGAGG" + sp
            unit = ig + gene
            if rng.random() < 0.5:
                unit = unit.translate(_COMP)[::-1]
            parts.append(unit); n += len(unit)
        s = "".join(parts) + "".join(rng.choice(list("ACGT"), p=base_p, size=int(rng.integers(10, 90))))
        if n_runs:
            s = list(s)
            for _ in range(n_runs):
                a = int(rng.integers(0, max(1, len(s) - 400)))
                L = int(rng.choice([10, 49, 50, 51, 120, 300]))
                s[a:a + L] = "N" * L
            s = "".join(s)
        out.append(("c%06d" % (c + 1), s))
    return out


def write_fasta(path, contigs, width=70):
    with open(path, "w") as f:
        for cid, s in contigs:
            f.write(">%s\n" % cid)
            for i in range(0, len(s), width):
                f.write(s[i:i + width] + "\n")
