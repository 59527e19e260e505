"""Synthetic lineage_wf-shaped world (cfg3/cfg4 of BASELINE.json): everything `checkm lineage_wf` reads on the
marker-gene path, generated with fixed seeds.

  <root>/hmms/phylo.hmm                 43 profiles (the tree pass, checkm/main.py:156-160)
  <root>/hmms/checkm.hmm                N_MODELS (default 2000) profiles, Pfam- and TIGRFAM-named (the analyze pass, main.py:325-333)
  <root>/pfam/Pfam-A.hmm.dat            clan / nesting annotation of the PF models (checkm/util/pfam.py:34-56)
  <root>/selected_marker_sets.tsv       internalID -> selectedID (checkm/markerSets.py:513-522)
  lineage.ms                            '# [Lineage Marker File]': one line per bin, most specific set first (markerSets.py:478-511)
  taxon.ms                              '# [Taxon Marker File]': one line, applies to every bin (markerSets.py:428-441)

The lineage tree is three levels deep (root -> NPHYLA -> NFAMILIES); a bin sits at a family and carries the chain
[family, phylum, root], so the models its scan needs are the union of the chain's marker genes plus the clan expansion
(markerSets.py:443-457): 300-1500 of the 2000 ("per-bin subset of 43 (phylo pass) + U[300,1500] (lineage pass)").
Bins hold U[1500,6000] ORFs (U[orf_lo, orf_hi] here), background residues with one planted ORF for a fraction of the bin's markers.

Nothing here touches the oracle or the GPU; STATS LOCAL lines come from tools/calibrate_synth.py (synth_stats_cfg3.json).
"""
import json
import os

import numpy as np

from synthdata import synth

_STATS_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "synth_stats_cfg3.json")

N_MODELS = 2000
NPHYLA, NFAMILIES = 8, 40


def model_lengths(n=N_MODELS, seed=2000):
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.lognormal(5.25, 0.55, size=n)), 30, 1200).astype(np.int64)      # median ~190, as Pfam-A / TIGRFAM


def accessions(n=N_MODELS):
    """Two thirds Pfam (versioned), one third TIGRFAM, interleaved."""
    return [("TIGR%05d" % (20000 + i)) if i % 3 == 2 else ("PF%05d.%d" % (20000 + i, 1 + i % 7)) for i in range(n)]


def load_stats():
    if os.path.exists(_STATS_FILE):
        with open(_STATS_FILE) as f:
            return json.load(f)
    return {}


def lineage_profiles(n=N_MODELS, seed=2000, with_stats=True):
    """The `checkm.hmm` of the synthetic world.  Every 5th model has no cutoffs (E-value branch of vetHit); TIGRFAMs carry TC+NC,
    Pfams GA, as the real databases do."""
    rng = np.random.default_rng(seed + 1)
    Ms, accs = model_lengths(n, seed), accessions(n)
    stats = load_stats() if with_stats else {}
    profs = []
    for i in range(n):
        acc = accs[i]
        name = acc if acc.startswith("TIGR") else "fam_%05d" % (20000 + i)
        p = synth.random_profile(rng, int(Ms[i]), name, acc)
        st = stats.get(acc)
        if st is not None:
            p.stats = tuple(st["stats"])
        elif with_stats:
            # uncalibrated fallback, close to the fitted values (only used when the stats file is absent)
            p.stats = (-8.5 - 0.002 * int(Ms[i]), 0.71, -9.5 - 0.002 * int(Ms[i]), 0.71, -3.8, 0.71)
        if i % 5 != 4:
            cut = 22.0 + (i % 9)
            if acc.startswith("TIGR"):
                p.tc = (cut, cut); p.nc = (cut - 8.0, cut - 8.0)
            else:
                p.ga = (cut, cut)
        profs.append(p)
    return profs


def pfam_dat(accs, seed=2000):
    """Stockholm-ish Pfam-A.hmm.dat text: ~30 % of the PF families sit in clans of 2-6 members; a few are nested in each other."""
    rng = np.random.default_rng(seed + 2)
    pf = [a for a in accs if a.startswith("PF")]
    clan_of = {}
    order = rng.permutation(len(pf))
    k, c = 0, 0
    while k < int(0.3 * len(pf)):
        size = int(rng.integers(2, 7))
        for j in order[k:k + size]:
            clan_of[pf[int(j)]] = "CL%04d" % c
        k += size; c += 1
    ids = {a: "fam_%s" % a[2:7] for a in pf}
    nested = {}
    for j in order[-40:-1:2]:
        a, b = pf[int(j)], pf[int(order[int(j) % 20])]
        if a != b:
            nested.setdefault(a, []).append(ids[b])
    out = []
    for a in pf:
        out.append("# STOCKHOLM 1.0\n#=GF ID   %s\n#=GF AC   %s\n" % (ids[a], a))
        if a in clan_of:
            out.append("#=GF CL   %s\n" % clan_of[a])
        for nb in nested.get(a, ()):
            out.append("#=GF NE   %s\n" % nb)
        out.append("//\n")
    return "".join(out)


class Lineage(object):
    """The marker-set tree: node uid -> (lineage string, nGenomes, list of collocated sets of accessions)."""

    def __init__(self, accs, seed=2000):
        rng = np.random.default_rng(seed + 3)
        n = len(accs)
        self.nodes = {}
        self.parent = {}

        def sets_of(idx):
            idx = [int(x) for x in idx]
            out, k = [], 0
            while k < len(idx):
                size = int(min(len(idx) - k, rng.choice([1, 1, 1, 2, 2, 3, 4, 6])))
                out.append(set(accs[j] for j in idx[k:k + size]))
                k += size
            return out
        perm = rng.permutation(n)
        root_idx = perm[:int(0.05 * n)]                               # ~100 universal markers
        self.nodes["0"] = ("root", 5656, sets_of(root_idx))
        self.selected = {"0": "0"}
        self.families = []
        for ph in range(NPHYLA):
            uid = "p%d" % ph
            extra = rng.permutation(n)[:int(rng.integers(int(0.08 * n), int(0.25 * n)))]
            idx = np.unique(np.concatenate([root_idx[:int(0.8 * len(root_idx))], extra]))
            rng.shuffle(idx)
            self.nodes[uid] = ("k__Bacteria;p__Synth%d" % ph, int(rng.integers(50, 2000)), sets_of(idx))
            self.parent[uid] = "0"
            self.selected[uid] = uid if ph % 4 else "0"               # some phyla defer to the root set
            for fa in range(NFAMILIES // NPHYLA):
                fid = "f%d_%d" % (ph, fa)
                more = rng.permutation(n)[:int(rng.integers(int(0.05 * n), int(0.45 * n)))]
                fidx = np.unique(np.concatenate([idx[:int(0.7 * len(idx))], more]))
                rng.shuffle(fidx)
                self.nodes[fid] = ("k__Bacteria;p__Synth%d;f__Fam%d" % (ph, fa), int(rng.integers(5, 300)), sets_of(fidx))
                self.parent[fid] = uid
                self.selected[fid] = fid if fa % 3 else uid           # every third family defers to its phylum
                self.families.append(fid)

    def chain(self, fid):
        out, u = [], fid
        while True:
            out.append(u)
            if u not in self.parent:
                return out
            u = self.parent[u]

    def line(self, binId, fid):
        ch = self.chain(fid)
        f = [binId, str(len(ch))]
        for u in ch:
            lin, ng, sets = self.nodes[u]
            f += [u, lin, str(ng), repr([set(sorted(s)) for s in sets])]
        return "\t".join(f) + "\n"

    def marker_genes(self, fid):
        g = set()
        for u in self.chain(fid):
            for s in self.nodes[u][2]:
                g |= s
        return g

    def selected_sets(self, fid):
        """Collocated sets of the set `BinMarkerSets.selectedMarkerSet()` resolves for a bin at `fid` (markerSets.py:86-121)."""
        want = self.selected[fid]
        ch = self.chain(fid)
        while want not in ch:
            want = self.selected[want]
        return self.nodes[want][2]


def sample_domain_fast(rng, prof):
    """One pass through the core model, vectorised: match residues by inverse CDF, delete runs and insert runs with the
    model's own transition probabilities (geometric run lengths).  Distributionally the walk synth.sample_domain does."""
    M = prof.M
    mat_cdf, bg_cdf = synth._cdf(prof)
    res = np.minimum(19, (rng.random(M)[:, None] > mat_cdf[1:M + 1]).sum(axis=1))
    keep = np.ones(M, dtype=bool)
    starts = np.nonzero(rng.random(M) < prof.t[1:M + 1, 2])[0]
    for s in starts:
        run = int(rng.geometric(1.0 - min(0.9, float(prof.t[min(M, s + 2), 6]))))
        keep[s + 1:s + 1 + run] = False
    keep[0] = True; keep[M - 1] = True
    ins_at = np.nonzero(rng.random(M - 1) < prof.t[1:M, 1])[0]
    if len(ins_at) == 0:
        return res[keep].astype(np.int64)
    parts, last = [], 0
    for s in ins_at:
        parts.append(res[last:s + 1][keep[last:s + 1]])
        run = int(rng.geometric(1.0 - min(0.9, float(prof.t[s + 1, 4]))))
        parts.append(np.minimum(19, np.searchsorted(bg_cdf, rng.random(run))))
        last = s + 1
    parts.append(res[last:][keep[last:]])
    return np.concatenate(parts).astype(np.int64)


def diverged(rng, dom, bg, keep):
    """A paralog of a sampled domain: each residue kept with probability `keep`, else redrawn from the background; a few short
    deletions.  What a marker family's other members look like to the marker's profile: a weak hit that passes the first filters."""
    out = dom.copy()
    redo = rng.random(len(out)) >= keep
    out[redo] = rng.choice(20, size=int(redo.sum()), p=bg)
    if len(out) > 60:
        for _ in range(int(rng.integers(0, 3))):
            a = int(rng.integers(5, len(out) - 20)); out = np.concatenate([out[:a], out[a + int(rng.integers(1, 12)):]])
    return out


def scrambled(rng, dom):
    """A domain cut into blocks of 10-25 residues and put back in random order: every block still draws an ungapped diagonal (the MSV
    filter adds those up in any order), the co-linear path the Viterbi and Forward stages ask for is gone."""
    cuts, a = [], 0
    while a < len(dom):
        b = a + int(rng.integers(10, 26)); cuts.append(dom[a:b]); a = b
    order = rng.permutation(len(cuts))
    return np.concatenate([cuts[int(k)] for k in order])


def low_complexity(rng, n):
    """An ORF of biased composition: a two- to four-letter alphabet, a short repeated motif, or a homopolymer run between them."""
    kind = int(rng.integers(0, 3))
    if kind == 0:
        letters = rng.choice(20, size=int(rng.integers(2, 5)), replace=False)
        return rng.choice(letters, size=n, p=rng.dirichlet(np.ones(len(letters)) * 2.0))
    if kind == 1:
        motif = rng.choice(20, size=int(rng.integers(2, 8)))
        s = np.tile(motif, n // len(motif) + 1)[:n].copy()
        flip = rng.random(n) < 0.08
        s[flip] = rng.choice(20, size=int(flip.sum()))
        return s
    s = rng.choice(20, size=n, p=synth.BGF)
    a = int(rng.integers(0, max(1, n // 2))); s[a:a + int(rng.integers(15, max(16, n // 2)))] = int(rng.integers(0, 20))
    return s


def make_lineage_bin(profs, planted, seed, n_orfs, phylo=None, dup_frac=0.04, orfs_per_contig=40, composition=None, paralogs=None, hard=False):
    """One bin: list of (name, desc, protein + '*').  `planted`: indices into profs that get one full-length ORF (a dup_frac share of
    them a second copy: contamination).  `phylo`: profiles of the tree pass, one ORF each as well.  `composition`: residue
    frequencies of the background (default Swiss-Prot); `paralogs` = (profile, copies): a family present many times.
    `hard` (round 6, bench.py's hard_workload leg): what real proteomes do to the filter cascade and the plain world does not -- a
    background composition of the bin's own (a Dirichlet draw around Swiss-Prot's: the GC skew of a genome shows in its amino acids),
    5 % of the ORFs of low complexity, and for every planted marker 3-5 diverged paralogs (35-75 % of the residues kept) and 2-4 block-scrambled
    copies besides it (as many as half the bin's ORFs hold; the real copies come first)."""
    rng = np.random.default_rng(seed)
    lens = synth.orf_lengths(rng, n_orfs)
    bg = synth.BGF if composition is None else np.asarray(composition, dtype=np.float64) / np.sum(composition)
    if hard and composition is None:
        bg = rng.dirichlet(np.asarray(synth.BGF, dtype=np.float64) * 40.0)
    flat = rng.choice(20, size=int(lens.sum()), p=bg)
    seqs = np.split(flat, np.cumsum(lens)[:-1])
    if hard:
        for s in rng.permutation(n_orfs)[:max(1, n_orfs // 20)]:
            seqs[int(s)] = low_complexity(rng, len(seqs[int(s)]))
    todo = list(phylo or []) + [profs[i] for i in planted]            # (a bin too small for all its plants loses lineage markers, never the 43 phylogenetic ones)
    copies = []
    for p in todo:
        copies.append((p, 1.0))
        if rng.random() < dup_frac:
            copies.append((p, 1.0))
    if hard:
        for p in todo:
            for _ in range(int(rng.integers(3, 6))):
                copies.append((p, float(rng.uniform(0.35, 0.75))))              # paralogs, close to remote
        for p in todo:
            for _ in range(int(rng.integers(2, 5))):
                copies.append((p, -1.0))                                        # scrambled homologs: filter survivors that end nowhere
    if paralogs is not None:
        copies += [(paralogs[0], 1.0)] * int(paralogs[1])
    copies = copies[:max(0, (n_orfs - 1) // 2)]
    slots = rng.permutation(n_orfs)[:len(copies)]
    for s, (p, keep) in zip(slots, copies):
        fl = rng.choice(20, size=int(rng.integers(5, 40)), p=bg)
        fr = rng.choice(20, size=int(rng.integers(5, 40)), p=bg)
        dom = sample_domain_fast(rng, p)
        seqs[int(s)] = np.concatenate([fl, dom if keep >= 1.0 else (scrambled(rng, dom) if keep < 0 else diverged(rng, dom, bg, keep)), fr])
    lut = np.frombuffer(synth.AMINO.encode(), dtype=np.uint8)
    out, pos = [], 1
    for i, sq in enumerate(seqs):
        contig, n = i // orfs_per_contig + 1, i % orfs_per_contig + 1
        end = pos + 3 * (len(sq) + 1) - 1
        out.append(("c%06d_%d" % (contig, n), "# %d # %d # 1 # ID=%d_%d;partial=00;start_type=ATG;rbs_motif=None;rbs_spacer=None" % (pos, end, contig, n),
                    lut[sq].tobytes().decode() + "*"))
        pos = end + 50
    return out


class World(object):
    """Builds the data root and the marker files under `root`; bins are generated on request."""

    def __init__(self, root, n_models=N_MODELS, seed=2000, write=True, jobs=None):
        self.root, self.seed, self.n_models = root, seed, n_models
        self.profs = lineage_profiles(n_models, seed)
        self.accs = [p.acc for p in self.profs]
        self.index = {a: i for i, a in enumerate(self.accs)}
        self.phylo = synth.cpr43_profiles()
        self.lineage = Lineage(self.accs, seed)
        self.dat = pfam_dat(self.accs, seed)
        self.checkm_hmm = os.path.join(root, "hmms", "checkm.hmm")
        self.phylo_hmm = os.path.join(root, "hmms", "phylo.hmm")
        if write:
            os.makedirs(os.path.join(root, "hmms"), exist_ok=True)
            os.makedirs(os.path.join(root, "pfam"), exist_ok=True)
            if not os.path.exists(self.checkm_hmm):
                with open(self.checkm_hmm + ".tmp", "w") as f:
                    for text in _pool_map(root, seed, n_models, "hmm", [list(range(k, min(k + 25, n_models))) for k in range(0, n_models, 25)], jobs):
                        f.write(text)
                os.replace(self.checkm_hmm + ".tmp", self.checkm_hmm)
            if not os.path.exists(self.phylo_hmm):
                synth.write_hmm(self.phylo_hmm, self.phylo)
            with open(os.path.join(root, "pfam", "Pfam-A.hmm.dat"), "w") as f:
                f.write(self.dat)
            with open(os.path.join(root, "selected_marker_sets.tsv"), "w") as f:
                for u, v in self.lineage.selected.items():
                    f.write("%s\t%s\n" % (u, v))

    def family_of(self, b):
        return self.lineage.families[(b * 7 + 3) % len(self.lineage.families)]

    def bin_records(self, b, orf_lo=1500, orf_hi=6000, completeness=(0.5, 1.0), **kw):
        rng = np.random.default_rng(self.seed * 1000 + b)
        n_orfs = int(rng.integers(orf_lo, orf_hi + 1))
        fid = self.family_of(b)
        sel = sorted(set().union(*self.lineage.selected_sets(fid)))
        frac = rng.uniform(*completeness)
        planted = [self.index[a] for a in sel if rng.random() < frac]
        return make_lineage_bin(self.profs, planted, self.seed * 1000 + b, n_orfs, phylo=self.phylo, **kw)

    def write_bin_files(self, jobs_list, jobs=None, hard=False):
        """jobs_list: [(bin index, path)] -- the genes.faa files of those bins, written by a pool of generator processes.
        hard: the harder world of make_lineage_bin (composition skew, low-complexity ORFs, paralog families)."""
        todo = [j for j in jobs_list if not os.path.exists(j[1])]
        if len(todo) <= 2:
            for b, path in todo:
                synth.write_fasta(path, self.bin_records(b, hard=hard))
            return
        chunks = [todo[k::max(1, len(todo) // 4)] for k in range(max(1, len(todo) // 4))]
        for _ in _pool_map(self.root, self.seed, self.n_models, "hardbins" if hard else "bins", chunks, jobs):
            pass

    def write_mag_files(self, jobs_list, jobs=None):
        """jobs_list: [(bin index, path)] -- cfg5's bins (5000 ORFs, 600 planted models of the whole database), by the generator pool."""
        todo = [j for j in jobs_list if not os.path.exists(j[1])]
        if not todo:
            return
        chunks = [todo[k::max(1, len(todo) // 2)] for k in range(max(1, len(todo) // 2))]
        for _ in _pool_map(self.root, self.seed, self.n_models, "mags", chunks, jobs):
            pass

    def write_nucleotide_bins(self, jobs_list, jobs=None):
        """jobs_list: [(bin index, path)] -- nucleotide FASTA files whose genes are the bins' proteins (synth_genome.genome_from_proteins)."""
        todo = [j for j in jobs_list if not os.path.exists(j[1])]
        if not todo:
            return
        chunks = [todo[k::max(1, len(todo) // 4)] for k in range(max(1, len(todo) // 4))]
        for _ in _pool_map(self.root, self.seed, self.n_models, "fna", chunks, jobs):
            pass

    def write_marker_files(self, outdir, binIds, families=None):
        """lineage.ms (one line per bin) and taxon.ms (the p0 phylum set for every bin) in `outdir`."""
        lin = os.path.join(outdir, "lineage.ms")
        with open(lin, "w") as f:
            f.write("# [Lineage Marker File]\n")
            for k, b in enumerate(binIds):
                f.write(self.lineage.line(b, families[k] if families else self.family_of(k)))
        tax = os.path.join(outdir, "taxon.ms")
        with open(tax, "w") as f:
            f.write("# [Taxon Marker File]\n")
            u = "p1"
            lin_s, ng, sets = self.lineage.nodes[u]
            f.write("\t".join(["Synth1", "1", u, lin_s, str(ng), repr([set(sorted(s)) for s in sets])]) + "\n")
        return lin, tax


# ---- generator processes: the synthetic world is a few hundred MB of text; writing it is single-threaded Python otherwise ----
_WORKER_WORLD = None


def _worker_init(root, seed, n_models):
    global _WORKER_WORLD
    _WORKER_WORLD = World(root, n_models, seed, write=False)


def _worker_task(job):
    kind, arg = job
    w = _WORKER_WORLD
    if kind == "hmm":
        return "".join(synth.hmm_text(w.profs[i]) for i in arg)
    if kind == "mags":           # cfg5: bins of 5000 ORFs with 600 of the database's models planted (bench.py: bench_cfg5)
        import numpy as np
        for b, path in arg:
            planted = sorted(np.random.default_rng(77000 + b).choice(w.n_models, size=600, replace=False).tolist())
            synth.write_fasta(path + ".tmp", make_lineage_bin(w.profs, planted, 900000 + b, n_orfs=5000))
            os.replace(path + ".tmp", path)
        return len(arg)
    if kind == "fna":            # the bin's proteins carried by a synthetic genome (bench.py: the from_fasta leg)
        from synthdata import synth_genome as sg
        for b, path in arg:
            prots = [r[2] for r in w.bin_records(b)]
            g = sg.genome_from_proteins(prots, 7000 + b, n_contigs=20, gc=0.35 + 0.3 * (b % 11) / 10.0, sd_frac=0.0 if b % 6 == 5 else 0.6)
            sg.write_fasta_bytes(path + ".tmp", g)
            os.replace(path + ".tmp", path)
        return len(arg)
    for b, path in arg:
        synth.write_fasta(path + ".tmp", w.bin_records(b, hard=(kind == "hardbins")))
        os.replace(path + ".tmp", path)
    return len(arg)


def _pool_map(root, seed, n_models, kind, args, jobs=None):
    """Results of the tasks in order.  'spawn' processes: the caller may already hold a HIP context, which must not be forked."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    jobs = jobs or int(os.environ.get("CKM_SYNTH_JOBS", "0")) or min(32, max(1, (os.cpu_count() or 2) - 2))
    jobs = max(1, min(jobs, len(args)))
    import sys
    main_file = getattr(sys.modules.get("__main__"), "__file__", None)
    spawnable = main_file is None or os.path.exists(main_file)       # 'spawn' re-imports __main__ by path: a script read from stdin has none
    done = 0
    if jobs > 1 and spawnable and not (kind == "hmm" and n_models <= 400):           # (small worlds, as in the tests, are not worth starting interpreters for)
        try:
            with ProcessPoolExecutor(max_workers=jobs, mp_context=mp.get_context("spawn"), initializer=_worker_init,
                                     initargs=(root, seed, n_models)) as ex:
                for r in ex.map(_worker_task, [(kind, a) for a in args]):
                    done += 1
                    yield r
            return
        except Exception:            # a broken pool (no usable __main__, no process slots ...): finish in this process
            pass
    if _WORKER_WORLD is None or _WORKER_WORLD.n_models != n_models or _WORKER_WORLD.seed != seed:
        _worker_init(root, seed, n_models)
    for a in args[done:]:
        yield _worker_task((kind, a))
