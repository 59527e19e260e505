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


# ---- nucleotide bins that carry a given set of proteins (bench.py: the from_fasta leg) ------------------------------------------------------
_AA_CODONS = {
    'A': ('GCT', 'GCC', 'GCA', 'GCG'), 'R': ('CGT', 'CGC', 'CGA', 'CGG', 'AGA', 'AGG'), 'N': ('AAT', 'AAC'), 'D': ('GAT', 'GAC'), 'C': ('TGT', 'TGC'),
    'Q': ('CAA', 'CAG'), 'E': ('GAA', 'GAG'), 'G': ('GGT', 'GGC', 'GGA', 'GGG'), 'H': ('CAT', 'CAC'), 'I': ('ATT', 'ATC', 'ATA'),
    'L': ('TTA', 'TTG', 'CTT', 'CTC', 'CTA', 'CTG'), 'K': ('AAA', 'AAG'), 'M': ('ATG',), 'F': ('TTT', 'TTC'), 'P': ('CCT', 'CCC', 'CCA', 'CCG'),
    'S': ('TCT', 'TCC', 'TCA', 'TCG', 'AGT', 'AGC'), 'T': ('ACT', 'ACC', 'ACA', 'ACG'), 'W': ('TGG',), 'Y': ('TAT', 'TAC'), 'V': ('GTT', 'GTC', 'GTA', 'GTG')}


def _codon_lut(gc):
    """[256][6] codon numbers (index into a 64 x 3 byte table) per amino-acid letter, synonymous codons repeated by a GC-skewed weight
    so that a uniform draw of a column follows a codon usage: enough for the gene finder's hexamer statistics to separate coding from
    intergenic sequence."""
    codons = [a + b + c for a in "ACGT" for b in "ACGT" for c in "ACGT"]
    num = {c: i for i, c in enumerate(codons)}
    lut = np.zeros((256, 6), dtype=np.uint8)
    lut[:] = num['GCC']
    for aa, cs in _AA_CODONS.items():
        w = np.asarray([(gc if c[2] in "GC" else 1.0 - gc) * 1.6 + 0.2 for c in cs])
        k = np.maximum(1, np.round(6 * w / w.sum()).astype(int))
        cols = [num[c] for c, n in zip(cs, k) for _ in range(n)][:6]
        while len(cols) < 6:
            cols.append(cols[len(cols) % len(cs)])
        lut[ord(aa)] = cols
    tab = np.frombuffer("".join(codons).encode(), dtype=np.uint8).reshape(64, 3)
    return lut, tab


def genome_from_proteins(proteins, seed, n_contigs=20, gc=0.5, sd_frac=0.6):
    """[(contig id, nucleotide sequence as bytes)] carrying `proteins` (amino-acid strings; a trailing '*' is dropped) as genes: ATG + one
    codon per residue + a stop, both strands, optional Shine-Dalgarno site, 20-160 intergenic bases."""
    rng = np.random.default_rng(seed)
    lut, tab = _codon_lut(gc)
    prots = [p[:-1] if p.endswith('*') else p for p in proteins]
    lens = np.asarray([len(p) for p in prots], dtype=np.int64)
    aa = np.frombuffer("".join(prots).encode(), dtype=np.uint8)
    col = rng.integers(0, 6, size=len(aa))
    nuc = tab[lut[aa, col]].reshape(-1).tobytes()                      # all coding bodies, back to back
    off = np.concatenate(([0], np.cumsum(lens))) * 3
    base_p = [(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2]
    ig_len = rng.integers(20, 160, size=len(prots) + n_contigs)
    ig_all = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.choice(4, p=base_p, size=int(ig_len.sum()) + 16 * len(prots))].tobytes()
    stops, flip, sd = rng.integers(0, 3, size=len(prots)), rng.random(len(prots)) < 0.5, rng.random(len(prots)) < sd_frac
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    per = (len(prots) + n_contigs - 1) // max(1, n_contigs)
    out, igp = [], 0
    for c in range(n_contigs):
        parts = []
        for g in range(c * per, min(len(prots), (c + 1) * per)):
            n = int(ig_len[g]); ig = ig_all[igp:igp + n]; igp += n
            if sd[g]:
                ig += b"AGGAGG" + ig_all[igp:igp + 7]; igp += 7
            unit = ig + b"ATG" + nuc[off[g]:off[g + 1]] + (b"TAA", b"TAG", b"TGA")[stops[g]]
            parts.append(unit.translate(comp)[::-1] if flip[g] else unit)
        if parts:
            n = int(ig_len[len(prots) + c]); parts.append(ig_all[igp:igp + n]); igp += n
            out.append((("c%06d" % (c + 1)).encode(), b"".join(parts)))
    return out


def write_fasta_bytes(path, contigs, width=70):
    with open(path, "wb") as f:
        for cid, s in contigs:
            f.write(b">" + cid + b"\n")
            f.write(b"\n".join(s[i:i + width] for i in range(0, len(s), width)) + b"\n")
