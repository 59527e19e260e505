"""Shared helpers for the tests: synthetic inputs on disk, oracle and product side by side."""
import os
import tempfile

import numpy as np

from synthdata import synth

_CACHE = {}


def hmm_file(tag, profs):
    """Write (once per process) a synthetic HMM file and return its path."""
    if tag not in _CACHE:
        d = tempfile.mkdtemp(prefix="ckm_test_")
        path = os.path.join(d, tag + ".hmm")
        synth.write_hmm(path, profs)
        _CACHE[tag] = path
    return _CACHE[tag]


def mixed_profiles():
    """12 short/medium + 4 long profiles: exercises several SSV and canonical Q classes."""
    return synth.small_profiles(11, 12, 40, 300) + synth.small_profiles(13, 4, 300, 1100)


def float_bits(a):
    return np.asarray(a, dtype=np.float32).view(np.uint32)


def oracle_search_threaded(hs, model_idx, dsq, names, threads=None):
    """hs.search over `model_idx` with the models spread over host threads (the C call releases the GIL; Z is the number of
    targets and domZ is per model, so the rows of a model do not depend on which other models are searched with it).
    Rows come back in model_idx order, as one call would give them."""
    import os
    from concurrent.futures import ThreadPoolExecutor
    threads = threads or min(32, os.cpu_count() or 1)
    model_idx = list(model_idx)
    if threads <= 1 or len(model_idx) <= 1:
        return hs.search(model_idx, dsq, names)
    with ThreadPoolExecutor(max_workers=threads) as ex:
        parts = list(ex.map(lambda m: hs.search([m], dsq, names), model_idx))
    return [r for part in parts for r in part]


def row_key(o):
    """Every value of a domtblout row the oracle reports, floats as bit patterns."""
    return (o.seq_idx, o.model_idx, o.tlen, o.qlen, o.full_evalue, int(float_bits(o.full_score)), int(float_bits(o.full_bias)), o.dom_idx, o.ndom,
            o.c_evalue, o.i_evalue, int(float_bits(o.dom_score)), int(float_bits(o.dom_bias)), o.hmm_from, o.hmm_to, o.ali_from, o.ali_to,
            o.env_from, o.env_to, int(float_bits(o.acc)))


def hit_key(hits, i, seq_base=0):
    return (int(hits.seq[i]) - seq_base, int(hits.model[i]), int(hits.tlen[i]), int(hits.qlen[i]), float(hits.full_evalue[i]), int(float_bits(hits.full_score[i])),
            int(float_bits(hits.full_bias[i])), int(hits.dom_idx[i]), int(hits.ndom[i]), float(hits.c_evalue[i]), float(hits.i_evalue[i]),
            int(float_bits(hits.dom_score[i])), int(float_bits(hits.dom_bias[i])), int(hits.hmm_from[i]), int(hits.hmm_to[i]), int(hits.ali_from[i]),
            int(hits.ali_to[i]), int(hits.env_from[i]), int(hits.env_to[i]), int(float_bits(hits.acc[i])))


def real_format_hmm_text(profs):
    """The synthetic profiles as the HMM files CheckM actually ships look (checkm_data_2015_01_16: hmms/checkm.hmm is HMMER3/b text written by
    HMMER 3.0, Pfam / TIGRFAM records; checkm/hmmerModelParser.py:54-83 reads their headers): version-b and version-f headers in one file,
    DATE / NSEQ / EFFN / CKSUM / BM / SM / COM lines, a DESC with spaces, GA / TC / NC with the trailing semicolon, RF / CS / MAP switched
    on with their annotation columns on every match line (three columns in 3/b, five in 3/f), records without a COMPO line, and records
    without ACC and cutoffs of their own (the sticky carry-over of CheckM's parser)."""
    import re
    out = []
    for i, p in enumerate(profs):
        txt = synth.hmm_text(p)
        head, body = txt.split("HMM     ", 1)
        ver_b = i % 2 == 0
        lines = [ln for ln in head.split("\n") if ln]
        keep = []
        for ln in lines:
            tag = ln.split()[0]
            if tag.startswith("HMMER3"):
                keep.append("HMMER3/b [3.0 | March 2010]" if ver_b else "HMMER3/f [3.1b1 | May 2013]")
            elif tag in ("RF", "MM", "CONS", "CS", "MAP", "NSEQ", "EFFN", "CKSUM"):
                continue
            elif tag == "ACC" and i % 4 == 3:
                continue                                      # no accession of its own: CheckM's parser carries the previous record's over
            elif tag in ("GA", "TC", "NC") and i % 4 == 3:
                continue
            elif tag == "DESC":
                keep.append("DESC  Ribosomal protein L%d, N-terminal domain (synthetic %s)" % (i + 1, p.name))
            elif tag == "LENG":
                keep.append(ln)
            else:
                keep.append(ln)
        keep = [k for k in keep if k]
        at = next(k for k, ln in enumerate(keep) if ln.startswith("ALPH")) + 1
        extra = ["RF    no", "CS    yes", "MAP   yes", "DATE  Fri Jan 16 12:00:%02d 2015" % (i % 60), "NSEQ  %d" % (20 + 7 * i), "EFFN  %.6f" % (1.5 + 0.37 * i),
                 "CKSUM %d" % (1234567 + 97 * i)]
        if not ver_b:
            extra.insert(1, "MM    no"); extra.insert(2, "CONS  yes")
        keep[at:at] = extra
        st = next(k for k, ln in enumerate(keep) if ln.startswith("STATS"))
        keep[st:st] = ["BM    hmmbuild --hand -o /dev/null HMM SEED", "SM    hmmsearch -Z 9421015 -E 1000 --cpu 4 HMM pfamseq"] if ver_b else \
                      ["COM   [1] hmmbuild -n %s HMM.ann SEED.ann" % p.name, "BM    hmmbuild HMM.ann SEED.ann", "SM    hmmsearch -Z 45638612 -E 1000 --cpu 4 HMM pfamseq"]
        blines = body.split("\n")
        new_body = []
        for ln in blines:
            if ln.lstrip().startswith("COMPO") and i % 3 == 1:
                continue                                      # a record without a composition line (hmmbuild writes one, hand-edited files may not)
            m = re.match(r"^(\s*\d+\s+.*?)\s+- - - - -\s*$", ln)
            if m:
                k = int(ln.split()[0])
                ann = "%6d %s %s" % (k * 2 + 1, "-", "HEC"[k % 3]) if ver_b else "%6d %s %s %s %s" % (k * 2 + 1, "acdefg"[k % 6], "-", "-", "HEC"[k % 3])
                new_body.append(m.group(1) + " " + ann)
            else:
                new_body.append(ln)
        out.append("\n".join(keep) + "\nHMM     " + "\n".join(new_body))
    return "".join(out)


def real_format_hmm_file(tag, profs):
    if tag not in _CACHE:
        d = tempfile.mkdtemp(prefix="ckm_test_")
        path = os.path.join(d, tag + ".hmm")
        with open(path, "w") as f:
            f.write(real_format_hmm_text(profs))
        _CACHE[tag] = path
    return _CACHE[tag]


def edge_genomes():
    """Bins (lists of contig strings) around the gene finder's edges: empty and one-to-three-base contigs, a contig of unknown bases only,
    lower case and IUPAC codes, hundreds of contigs too short for any node, one base short of / exactly at the 20 kb training limit, runs of
    49 and 50 unknown bases inside contigs (prodigal's -m masks from 50), a single reading frame shorter than 20 kb, a repeat genome."""
    from synthdata import synth_genome as sg
    rng = np.random.default_rng(4)
    base = [s for _c, s in sg.make_genome(700, n_contigs=3, contig_len=(20000, 30000))]
    return [base + ["", "A", "AC", "ACG"],
            ["N" * 500] + base,
            [base[0].lower(), base[1][:5000] + "RYKMSWBDHVN" * 20 + base[1][5000:], base[2]],
            ["".join(rng.choice(list("ACGT"), 80)) for _ in range(300)] + base[:1],
            [base[0][:19999]],
            [base[0][:20000]],
            [base[0] + "N" * 49 + base[1], base[2][:100] + "N" * 50 + base[2][100:]],
            ["ATG" + "GCA" * 400 + "TAA"],
            [("ATG" + "GCT" * 300 + "TAA" + "CCGGTTAACC") * 30]]


def dp_stress_genomes():
    """Bins (lists of contig strings) that push the gene finder's dynamic program off its common path: an open reading frame of 36 kb (the
    window of the nodes behind it starts further back than any ring of recent nodes holds -- dprog.c looks 500 nodes back and 500 more behind
    the node it finds there, further when a giant frame sits there); two thousand in-frame ATG codons in ONE frame, forward and reverse, and three thousand with nothing between them (one
    class of nodes fills the whole window; a stop's overlapping start is thousands of nodes away); ATG / TAA alternating through all three
    frames (windows of nothing but nodes, stops every few bases); and two hundred short contigs of 200 - 3000 bases (final-sweep sequences
    of 0 .. 200 nodes: every remainder of the 64-node blocks the device takes them in)."""
    from synthdata import synth_genome as sg
    rng = np.random.default_rng(9)
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}

    def rc(s):
        return "".join(comp[c] for c in reversed(s))
    base = [s for _c, s in sg.make_genome(710, n_contigs=3, contig_len=(25000, 35000))]
    sense = [a + b + c for a in "ACGT" for b in "ACGT" for c in "ACGT" if a + b + c not in ("TAA", "TAG", "TGA")]
    giant = "ATG" + "".join(rng.choice(sense, 12000)) + "TAA"
    many = "ATG" + "".join(("ATG" if k % 2 == 0 else str(rng.choice(sense))) for k in range(4000)) + "TAA"
    pure = "ATG" * 3000 + "TAA"                                                    # (nothing but forward starts of one frame for 3000 nodes)
    dense = "".join(rng.choice(["ATGA", "TAAC", "ATGC", "TAGG", "GTGA", "TGAC"], 3000))
    shorts = ["".join(rng.choice(list("ACGT"), int(n))) for n in rng.integers(200, 3000, 200)]
    return [[base[0][:12000] + giant + base[0][12000:], base[1]],
            [base[0], base[1][:8000] + many + base[1][8000:], base[2][:5000] + rc(many) + base[2][5000:]],
            [base[2], dense + base[0][:20000] + rc(dense), base[1][:3000] + pure + base[1][3000:6000] + rc(pure) + base[1][6000:9000]],
            shorts + base[:1]]
