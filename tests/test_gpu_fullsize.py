"""Full-shape checks (cfg2 of BASELINE.json: the 43-profile DB against 2000-ORF bins) through properties
that do not need the oracle to finish a full-size run: determinism, bin independence (sharding and
permutation invariance -- what the multi-GPU path relies on), recovery of every planted marker, and
oracle spot checks on sampled pairs."""
import numpy as np
import pytest

from checkm_amd import _lib, qa as cqa
from synthdata import synth
from oracle import p7
from tests import common

pytestmark = pytest.mark.gpu

NBINS = 12


@pytest.fixture(scope="module")
def world(gpu_ctx):
    profs = synth.cpr43_profiles()
    path = common.hmm_file("cpr43", profs)
    bins = [synth.make_bin(profs, 1000 + b, n_orfs=2000) for b in range(NBINS)]
    prof = _lib.Profiles(gpu_ctx, path)
    yield dict(ctx=gpu_ctx, profs=profs, path=path, bins=bins, prof=prof)
    prof.close()


def _rows(hits, b, seq_base=0):
    r = list(hits.rows(b))
    return [(int(hits.seq[i]) - seq_base, int(hits.model[i]), float(hits.full_evalue[i]), np.float32(hits.full_score[i]).view(np.uint32),
             np.float32(hits.dom_score[i]).view(np.uint32), int(hits.hmm_from[i]), int(hits.hmm_to[i]), int(hits.ali_from[i]), int(hits.ali_to[i]),
             int(hits.env_from[i]), int(hits.env_to[i]), float(hits.c_evalue[i])) for i in r]


def test_determinism_sharding_and_permutation(world):
    w = world
    ctx, prof, bins = w["ctx"], w["prof"], w["bins"]
    seqs = _lib.Seqs(ctx, bins)
    h1 = _lib.search(ctx, prof, seqs)
    h2 = _lib.search(ctx, prof, seqs)
    base = np.concatenate([[0], np.cumsum([len(b) for b in bins])])
    ref = [_rows(h1, b, int(base[b])) for b in range(NBINS)]
    assert ref == [_rows(h2, b, int(base[b])) for b in range(NBINS)]            # same call twice: identical, atomics notwithstanding
    assert sum(len(r) for r in ref) >= NBINS * 40
    h1.close(); h2.close(); seqs.close()
    # shard the bins over two "ranks": per-bin rows must not change (Z and domZ are per bin)
    for shard in ([0, 2, 4, 6, 8, 10], [1, 3, 5, 7, 9, 11]):
        s2 = _lib.Seqs(ctx, [bins[b] for b in shard])
        h = _lib.search(ctx, prof, s2)
        b2 = np.concatenate([[0], np.cumsum([len(bins[b]) for b in shard])])
        for k, b in enumerate(shard):
            assert _rows(h, k, int(b2[k])) == ref[b], b
        h.close(); s2.close()
    # reversed bin order
    order = list(range(NBINS))[::-1]
    s3 = _lib.Seqs(ctx, [bins[b] for b in order])
    h = _lib.search(ctx, prof, s3)
    b3 = np.concatenate([[0], np.cumsum([len(bins[b]) for b in order])])
    for k, b in enumerate(order):
        assert _rows(h, k, int(b3[k])) == ref[b], b
    h.close(); s3.close()


def test_planted_markers_are_recovered_and_reduced(world):
    w = world
    ctx, prof, bins = w["ctx"], w["prof"], w["bins"]
    seqs = _lib.Seqs(ctx, bins)
    hits = _lib.search(ctx, prof, seqs)
    plan = cqa.QAPlan.for_hmm_models(prof, [list(range(prof.n))] * NBINS)
    res = plan.reduce(ctx, hits, seqs)
    st = ctx.stats()
    assert st.pairs_ssv == NBINS * 2000 * 43 and st.residue_hmm == seqs.total_residues * 43
    for b in range(NBINS):
        found = set(int(hits.model[i]) for i in hits.rows(b))
        assert len(found) == 43, (b, sorted(set(range(43)) - found))                  # every model has its planted ORF reported
        assert res.hist[b].sum() == 43
        assert res.hist[b][0] <= 3 and res.completeness[b] >= 93.0                      # split/duplicated plants may miss a cutoff, never more
        assert 0.0 <= res.contamination[b] <= 15.0
        # the float64 division of the reference, redone here from the integer outputs (one set of 43 markers)
        nm = int(res.n_markers[b])
        s0 = int(res.set_off[b])
        assert res.completeness[b] == 100 * (float(int(res.set_present[s0])) / nm) / 1 and res.contamination[b] == 100 * (float(int(res.set_multi[s0])) / nm) / 1
    res.close(); hits.close(); seqs.close()


def test_sampled_pairs_against_the_oracle(world):
    w = world
    ctx, prof, bins = w["ctx"], w["prof"], w["bins"]
    seqs = _lib.Seqs(ctx, bins[:2])
    hs = p7.HmmSet(w["path"])
    recs = bins[0] + bins[1]
    rng = np.random.default_rng(11)
    pairs = [(int(m), int(s)) for m, s in zip(rng.integers(0, 43, 160), rng.integers(0, len(recs), 160))]
    got = _lib.debug_stages(ctx, prof, seqs, np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs]))
    for i, (m, s) in enumerate(pairs):
        o = hs.stages(m, p7.digitize(recs[s][2]))
        assert (o.msv_xJ, o.vit_xC, o.fwd_nscale) == (got[i].msv_xJ, got[i].vit_xC, got[i].fwd_nscale), (m, s)
        for f in ("msv_sc", "bias_sc", "vit_sc", "fwd_sc", "fwd_xC"):
            assert np.float32(getattr(o, f)).view(np.uint32) == np.float32(getattr(got[i], f)).view(np.uint32), (m, s, f)
    hs.close(); seqs.close()


def test_many_models_with_ragged_subsets(gpu_ctx):
    """lineage_wf breadth: a 300-profile DB, 16 bins of 400 ORFs, every bin with its own random subset of 60-140 models in its own
    order (~650 k pairs).  Properties: searching a bin alone gives the rows it gets in the batch (Z and domZ are per bin, the plan
    is per bin), every planted marker of a listed model is reported, and two bins are checked against the oracle on 12 of their
    models each."""
    rng = np.random.default_rng(77)
    lengths = rng.integers(40, 700, size=300)
    profs = []
    for i, M in enumerate(lengths):
        p = synth.random_profile(rng, int(M), "FAM%04d" % i, ("PF%05d.1" % (10000 + i)) if i % 3 else ("TIGR%05d" % (10000 + i)))
        p.stats = (-8.5 - 0.002 * int(M), 0.71, -9.5 - 0.002 * int(M), 0.71, -3.8, 0.71)
        profs.append(p)
    path = common.hmm_file("many300", profs)
    nb = 16
    bin_models = [[int(x) for x in rng.permutation(300)[:int(rng.integers(60, 141))]] for _ in range(nb)]
    bins = []
    for b in range(nb):
        sub = [profs[m] for m in bin_models[b][:25]]                      # plant 25 of the bin's models
        bins.append(synth.make_bin(sub, 7000 + b, n_orfs=400, dup_frac=0.1))
    prof = _lib.Profiles(gpu_ctx, path)
    seqs = _lib.Seqs(gpu_ctx, bins)
    hits = _lib.search(gpu_ctx, prof, seqs, bin_models)
    st = gpu_ctx.stats()
    assert st.pairs_ssv == sum(len(bins[b]) * len(bin_models[b]) for b in range(nb))
    base = np.concatenate([[0], np.cumsum([len(b) for b in bins])])
    ref = [_rows(hits, b, int(base[b])) for b in range(nb)]
    for b in range(nb):
        found = set(r[1] for r in ref[b])
        assert len(set(bin_models[b][:25]) - found) <= 2, (b, sorted(set(bin_models[b][:25]) - found))      # a planted ORF split in two halves can fall below E
        assert found <= set(bin_models[b])
    for b in (3, 11):                                                       # a bin searched alone
        s1 = _lib.Seqs(gpu_ctx, [bins[b]])
        h1 = _lib.search(gpu_ctx, prof, s1, [bin_models[b]])
        assert _rows(h1, 0, 0) == ref[b], b
        h1.close(); s1.close()
    hs = p7.HmmSet(path)
    for b in (0, 9):
        sub = bin_models[b][:12]
        rows = hs.search(sub, [p7.digitize(r[2]) for r in bins[b]], [r[0] for r in bins[b]])
        mine = [r for r in ref[b] if r[1] in set(sub)]
        assert len(rows) == len(mine) >= 12
        for o, g in zip(rows, mine):
            assert (o.seq_idx, o.model_idx, o.full_evalue, common.float_bits(o.full_score), common.float_bits(o.dom_score), o.hmm_from, o.hmm_to, o.ali_from, o.ali_to,
                    o.env_from, o.env_to, o.c_evalue) == g, (b, o.seq_idx, o.model_idx)
    hits.close(); seqs.close(); prof.close(); hs.close()


def test_whole_bins_row_for_row_against_the_oracle(world):
    """EVERY row of two complete cfg2 bins (2000 ORFs x 43 models each, 172 000 pairs) against the oracle, every column, floats as
    bit patterns; the oracle is threaded over the models."""
    w = world
    ctx, prof, bins = w["ctx"], w["prof"], w["bins"]
    seqs = _lib.Seqs(ctx, bins[:2])
    hits = _lib.search(ctx, prof, seqs)
    hs = p7.HmmSet(w["path"])
    base = 0
    for b in range(2):
        recs = bins[b]
        rows = common.oracle_search_threaded(hs, range(hs.n), [p7.digitize(r[2]) for r in recs], [r[0] for r in recs])
        mine = [common.hit_key(hits, i, base) for i in hits.rows(b)]
        assert len(rows) == len(mine) >= 43
        for o, g in zip(rows, mine):
            assert common.row_key(o) == g, (b, o.seq_idx, o.model_idx)
        base += len(recs)
    hs.close(); hits.close(); seqs.close()


def test_biased_composition_and_paralog_family(world):
    """A bin whose background is far from Swiss-Prot (skewed Dirichlet draw: the null models and the bias filter work on a
    composition they were not calibrated for) and which carries 20 paralogous copies of one family (the Forward / domain
    definition / null2 stages see a family-rich proteome).  All rows against the oracle."""
    from synthdata import synth_lineage as sl
    w = world
    ctx, prof, profs = w["ctx"], w["prof"], w["profs"]
    rng = np.random.default_rng(5)
    comp = rng.dirichlet(np.full(20, 0.35))
    recs = sl.make_lineage_bin(profs, list(range(len(profs))), 9191, 700, composition=comp, paralogs=(profs[7], 20), dup_frac=0.1)
    seqs = _lib.Seqs(ctx, [recs])
    hits = _lib.search(ctx, prof, seqs)
    st = ctx.stats()
    hs = p7.HmmSet(w["path"])
    rows = common.oracle_search_threaded(hs, range(hs.n), [p7.digitize(r[2]) for r in recs], [r[0] for r in recs])
    mine = [common.hit_key(hits, i) for i in hits.rows(0)]
    assert len(rows) == len(mine) >= 43 + 20
    for o, g in zip(rows, mine):
        assert common.row_key(o) == g, (o.seq_idx, o.model_idx)
    assert sum(1 for g in mine if g[1] == 7) >= 20                                   # the paralog family is reported copy by copy
    assert st.pairs_dom >= 43 + 20 and st.envelopes >= 43 + 20                       # the paralogs reach the domain stage one by one
    hs.close(); hits.close(); seqs.close()


def test_low_complexity_and_tandem_repeat_proteins(world):
    """What real proteomes hold and iid-background ORFs do not: homopolymer and dipeptide runs, proteins that are one short motif over
    and over, linkers of two or three residue types around real domains, and tandem arrays of one domain (2-8 copies: every array is a
    multi-domain region for the trace ensemble, the repeats' null2 correction sees a composition dominated by the domain itself).  The
    bias filter, null2 and the region heuristics do their real work here.  Every row against the oracle, as bit patterns."""
    w = world
    ctx, prof, profs = w["ctx"], w["prof"], w["profs"]
    rng = np.random.default_rng(77)
    AAS = synth.AMINO
    recs = []

    def add(seq):
        recs.append(("lc%04d_%d" % (len(recs) // 40 + 1, len(recs) % 40 + 1), "", seq + "*"))
    for k in range(30):                                               # pure low complexity
        a, b, c = (AAS[i] for i in rng.choice(20, 3, replace=False))
        kind = k % 5
        n = int(rng.integers(60, 500))
        if kind == 0:
            add(a * n)
        elif kind == 1:
            add((a + b) * (n // 2))
        elif kind == 2:
            add("".join(rng.choice([a, b, c], p=[0.6, 0.3, 0.1], size=n)))
        elif kind == 3:
            motif = "".join(rng.choice(list(AAS), size=int(rng.integers(3, 9))))
            add(motif * (n // len(motif)))
        else:
            add(synth.to_text(synth.random_residues(rng, 40)) + a * (n // 2) + synth.to_text(synth.random_residues(rng, 40)) + (b + c) * (n // 6))
    for k in range(40):                                               # domains inside low-complexity linkers, tandem arrays of one domain
        p = profs[int(rng.integers(0, len(profs)))]
        copies = int(rng.integers(1, 9)) if p.M < 200 else int(rng.integers(1, 4))
        link = lambda: "".join(rng.choice(list("QSGTPN"), p=[0.35, 0.25, 0.2, 0.1, 0.05, 0.05], size=int(rng.integers(0, 30))))
        s = link()
        for _ in range(copies):
            lo = 1 if rng.random() < 0.7 else int(rng.integers(1, max(2, p.M // 3)))
            s += synth.to_text(synth.sample_domain(rng, p, lo, p.M if rng.random() < 0.7 else int(rng.integers(max(lo + 10, 2 * p.M // 3), p.M + 1)))) + link()
        add(s)
    for _ in range(130):                                              # ordinary background around them (Z, the tail statistics)
        add(synth.to_text(synth.random_residues(rng, int(rng.integers(60, 600)))))
    seqs = _lib.Seqs(ctx, [recs])
    hits = _lib.search(ctx, prof, seqs)
    st = ctx.stats()
    hs = p7.HmmSet(w["path"])
    rows = common.oracle_search_threaded(hs, range(hs.n), [p7.digitize(r[2]) for r in recs], [r[0] for r in recs])
    mine = [common.hit_key(hits, i) for i in hits.rows(0)]
    assert len(rows) == len(mine) >= 60
    for o, g in zip(rows, mine):
        assert common.row_key(o) == g, (o.seq_idx, o.model_idx)
    assert st.regions_multi >= 1, st.regions_multi                                    # tandem arrays: most copies are separated by the region heuristics alone, some regions need the trace ensemble
    assert max(g[8] for g in mine) >= 2, max(g[8] for g in mine)                      # some target carries several reported domains of one model
    hs.close(); hits.close(); seqs.close()
