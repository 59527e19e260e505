"""GPU parity of the scan half: every kernel stage and the assembled search against the CPU oracle.

Bar: bit-exact.  Integer stages (SSV/MSV bytes, Viterbi words) by construction; float stages because
oracle and kernels evaluate in the same canonical order, so float32 results are
compared as bit patterns, not within a tolerance.
"""
import numpy as np
import pytest

from checkm_amd import _lib
from synthdata import synth
from oracle import p7
from tests import common

pytestmark = pytest.mark.gpu

STAGE_FIELDS = ["msv_xJ", "msv_sc", "null_sc", "bias_sc", "vit_xC", "vit_sc", "fwd_sc", "fwd_xC", "fwd_nscale"]


@pytest.fixture(scope="module")
def world(gpu_ctx):
    profs = common.mixed_profiles()
    path = common.hmm_file("mixed", profs)
    bins = [synth.make_bin(profs, 1000 + b, n_orfs=160, dup_frac=0.4) for b in range(2)]
    # edge cases: very short, all-X, stop-only tail, a long one
    rng = np.random.default_rng(5)
    bins[1] += [("edge_1", "", "M*"), ("edge_2", "", "XXXXXXXXXXXXXXXXXXXXXXXXXXXXXX*"), ("edge_3", "", "ACDEFGHIKLMNPQRSTVWY"),
                ("edge_4", "", synth.to_text(synth.random_residues(rng, 3100)) + "*"), ("edge_5", "", "BJZOUX*acdefghiklmnpqrstvwy")]
    prof = _lib.Profiles(gpu_ctx, path)
    seqs = _lib.Seqs(gpu_ctx, bins)
    hs = p7.HmmSet(path)
    recs = [r for b in bins for r in b]
    dsq = [p7.digitize(r[2]) for r in recs]
    yield dict(ctx=gpu_ctx, profs=profs, prof=prof, seqs=seqs, hs=hs, recs=recs, dsq=dsq, bins=bins)
    prof.close(); seqs.close(); hs.close()


def _cmp_stage(o, g, f):
    a, b = getattr(o, f), getattr(g, f)
    if isinstance(a, float):
        return np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32) or (np.isnan(a) and np.isnan(b))
    return a == b


def test_stage_scores_bit_exact(world):
    w = world
    rng = np.random.default_rng(3)
    nseq = len(w["recs"])
    pairs = [(m, s) for m in range(w["hs"].n) for s in rng.choice(nseq, size=40, replace=False)]
    pairs += [(m, nseq - k) for m in range(w["hs"].n) for k in range(1, 6)]          # edge sequences vs every model
    model = np.array([p[0] for p in pairs]); seq = np.array([p[1] for p in pairs])
    got = _lib.debug_stages(w["ctx"], w["prof"], w["seqs"], model, seq)
    bad = []
    for i, (m, s) in enumerate(pairs):
        o = w["hs"].stages(m, w["dsq"][s])
        for f in STAGE_FIELDS:
            if not _cmp_stage(o, got[i], f):
                bad.append((m, s, f, getattr(o, f), getattr(got[i], f)))
        # the packed exact-MSV kernel (the one the search runs on the pairs SSV cannot decide) against the same oracle bytes
        if got[i].msvp_xJ != o.msv_xJ or common.float_bits(got[i].msvp_sc) != common.float_bits(o.msv_sc):
            bad.append((m, s, "msv packed", o.msv_xJ, got[i].msvp_xJ, o.msv_sc, got[i].msvp_sc))
    assert not bad, bad[:10]


def test_planted_pairs_bit_exact(world):
    """Every pair the oracle carries beyond the MSV filter (true hits: overflow, rescaling, J state)."""
    w = world
    pairs = []
    for m in range(w["hs"].n):
        for s in range(len(w["recs"])):
            if len(pairs) < 4000 and w["hs"].stages(m, w["dsq"][s]).pass_msv:
                pairs.append((m, s))
    model = np.array([p[0] for p in pairs]); seq = np.array([p[1] for p in pairs])
    got = _lib.debug_stages(w["ctx"], w["prof"], w["seqs"], model, seq)
    bad = []
    for i, (m, s) in enumerate(pairs):
        o = w["hs"].stages(m, w["dsq"][s])
        for f in STAGE_FIELDS:
            if not _cmp_stage(o, got[i], f):
                bad.append((m, s, f, getattr(o, f), getattr(got[i], f)))
        if got[i].msvp_xJ != o.msv_xJ or common.float_bits(got[i].msvp_sc) != common.float_bits(o.msv_sc):
            bad.append((m, s, "msv packed", o.msv_xJ, got[i].msvp_xJ, o.msv_sc, got[i].msvp_sc))
    assert len(pairs) > 20
    assert not bad, bad[:10]


def test_envelopes_bit_exact(world):
    w = world
    rows = w["hs"].search(list(range(w["hs"].n)), w["dsq"][:160], [r[0] for r in w["recs"][:160]])
    assert len(rows) >= 10
    model = np.array([r.model_idx for r in rows]); seq = np.array([r.seq_idx for r in rows])
    ienv = np.array([r.env_from for r in rows]); jenv = np.array([r.env_to for r in rows])
    got = _lib.debug_envelopes(w["ctx"], w["prof"], w["seqs"], model, seq, ienv, jenv)
    for i, r in enumerate(rows):
        rc, envsc, oasc, null2, coords, xC, ns = w["hs"].envelope(r.model_idx, w["dsq"][r.seq_idx], r.env_from, r.env_to)
        g = got[i]
        assert g.ok == 1 and rc == 0
        assert common.float_bits(envsc) == common.float_bits(g.envsc), (i, envsc, g.envsc)
        assert common.float_bits(xC) == common.float_bits(g.fwd_xC) and ns == g.nscale
        assert (common.float_bits(null2) == common.float_bits(np.array(g.null2[:]))).all(), (i, null2, list(g.null2))
        assert common.float_bits(oasc) == common.float_bits(g.oasc), (i, oasc, g.oasc)
        assert list(coords) == [g.hmm_from, g.hmm_to, g.ali_from, g.ali_to], (i, coords, g.hmm_from, g.hmm_to, g.ali_from, g.ali_to)


ROW_INT = ["seq", "model", "tlen", "qlen", "dom_idx", "ndom", "hmm_from", "hmm_to", "ali_from", "ali_to", "env_from", "env_to"]
ROW_F32 = ["full_score", "full_bias", "dom_score", "dom_bias", "acc"]
ROW_F64 = ["full_evalue", "c_evalue", "i_evalue"]
ORACLE_NAME = {"seq": "seq_idx", "model": "model_idx"}


def _compare_search(w, hits, bin_models):
    off = 0
    for b, recs in enumerate(w["bins"]):
        names = [r[0] for r in recs]
        dsq = w["dsq"][off:off + len(recs)]
        models = bin_models[b] if bin_models else list(range(w["hs"].n))
        rows = w["hs"].search(models, dsq, names)
        got = list(hits.rows(b))
        assert len(rows) == len(got), (b, len(rows), len(got))
        for r, g in zip(rows, got):
            for f in ROW_INT:
                ov = getattr(r, ORACLE_NAME.get(f, f)) + (off if f == "seq" else 0)
                assert ov == getattr(hits, f)[g], (b, f, ov, getattr(hits, f)[g])
            for f in ROW_F32:
                assert common.float_bits(getattr(r, f)) == common.float_bits(getattr(hits, f)[g]), (b, f, getattr(r, f), getattr(hits, f)[g])
            for f in ROW_F64:
                assert getattr(r, f) == getattr(hits, f)[g], (b, f)
        off += len(recs)


def test_search_rows_identical(world):
    w = world
    hits = _lib.search(w["ctx"], w["prof"], w["seqs"])
    assert hits.n > 20
    _compare_search(w, hits, None)
    st = w["ctx"].stats()
    assert st.pairs_ssv == sum(1 for r in w["recs"] if len(r[2]) > 0) * w["hs"].n
    # the SSV-derived F1 decision must agree with the exact byte MSV on EVERY pair, not only on reported hits
    stages = [w["hs"].stages(m, d) for m in range(w["hs"].n) for d in w["dsq"]]
    want = sum(s.pass_msv for s in stages)
    assert st.pairs_bias == want, (st.pairs_bias, want)
    # ... and the two-pass Viterbi (J-free bound first, exact kernel only where the bound cannot decide) must send exactly the
    # oracle's F2 survivors on to the Forward parser
    want_fwd = sum(s.pass_vit for s in stages)
    assert st.pairs_fwd == want_fwd, (st.pairs_fwd, want_fwd)
    assert st.pairs_vit_exact < st.pairs_vit
    hits.close()


def test_search_per_bin_model_subsets(world):
    """lineage_wf shape: every bin has its own model list, in its own order (markerSets.py:337-339)."""
    w = world
    n = w["hs"].n
    bin_models = [[5, 2, 9, 0, 13], list(range(n - 1, -1, -2))]
    hits = _lib.search(w["ctx"], w["prof"], w["seqs"], bin_models)
    _compare_search(w, hits, bin_models)
    hits.close()


def test_domtblout_text_identical(world, tmp_path):
    w = world
    hits = _lib.search(w["ctx"], w["prof"], w["seqs"])
    off = 0
    for b, recs in enumerate(w["bins"]):
        names = [r[0] for r in recs]; descs = [r[1] for r in recs]
        rows = w["hs"].search(list(range(w["hs"].n)), w["dsq"][off:off + len(recs)], names)
        want = w["hs"].format_domtblout(rows, names, descs)
        path = str(tmp_path / ("b%d.txt" % b))
        hits.write_domtblout(w["prof"], w["seqs"], b, path)
        assert open(path).read() == want
        off += len(recs)
    hits.close()


def test_large_models_and_error_paths(gpu_ctx, tmp_path):
    """Register classes the mixed set does not reach (M up to 2048: SSV Q 48-64, Viterbi QH 12-16, Forward Q 24-32),
    plus the library's error codes for malformed input."""
    profs = synth.small_profiles(17, 2, 1300, 2048)
    path = common.hmm_file("large17", profs)
    recs = synth.make_bin(profs, 99, n_orfs=24, dup_frac=0.0)
    rng = np.random.default_rng(8)
    recs += [("long_%d" % i, "", synth.to_text(synth.random_residues(rng, 2500)) + "*") for i in range(2)]
    prof = _lib.Profiles(gpu_ctx, path)
    seqs = _lib.Seqs(gpu_ctx, [recs])
    hs = p7.HmmSet(path)
    dsq = [p7.digitize(r[2]) for r in recs]
    pairs = [(m, s) for m in range(hs.n) for s in range(len(recs))]
    got = _lib.debug_stages(gpu_ctx, prof, seqs, np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs]))
    for i, (m, s) in enumerate(pairs):
        o = hs.stages(m, dsq[s])
        for f in STAGE_FIELDS:
            assert _cmp_stage(o, got[i], f), (m, s, f, getattr(o, f), getattr(got[i], f))
    hits = _lib.search(gpu_ctx, prof, seqs)
    rows = hs.search(list(range(hs.n)), dsq, [r[0] for r in recs])
    assert hits.n == len(rows) >= 2
    for g, r in enumerate(rows):
        assert (r.seq_idx, r.model_idx, r.hmm_from, r.hmm_to, r.ali_from, r.ali_to, r.env_from, r.env_to) == \
            (hits.seq[g], hits.model[g], hits.hmm_from[g], hits.hmm_to[g], hits.ali_from[g], hits.ali_to[g], hits.env_from[g], hits.env_to[g])
        for f in ROW_F32:
            assert common.float_bits(getattr(r, f)) == common.float_bits(getattr(hits, f)[g]), f
        assert r.full_evalue == hits.full_evalue[g] and r.c_evalue == hits.c_evalue[g]
    hits.close(); prof.close(); seqs.close(); hs.close()
    # error paths
    bad = tmp_path / "bad.hmm"
    bad.write_text("HMMER3/f [x]\nNAME  broken\nLENG  5\nALPH  amino\nHMM     A\n")
    with pytest.raises(_lib.CkmError) as e:
        _lib.Profiles(gpu_ctx, str(bad))
    assert e.value.code == -3
    with pytest.raises(_lib.CkmError) as e:
        _lib.Profiles(gpu_ctx, str(tmp_path / "missing.hmm"))
    assert e.value.code == -2
    uncal = tmp_path / "uncal.hmm"
    p = synth.random_profile(rng, 30, "nocal", "PF77777.1")
    synth.write_hmm(str(uncal), [p])
    with pytest.raises(_lib.CkmError) as e:
        _lib.Profiles(gpu_ctx, str(uncal))
    assert e.value.code == -3 and "calibrated" in str(e.value)


def test_chunked_and_multi_worker_execution_equal_single_pass(world):
    """Tiny budgets force several SSV chunks and several envelope batches; several workers split the sequences by
    length class or the models; rows must not change."""
    import subprocess
    import sys
    import json
    import os
    w = world
    hits = _lib.search(w["ctx"], w["prof"], w["seqs"])
    ref = [[int(hits.seq[i]), int(hits.model[i]), int(hits.ali_from[i]), int(hits.ali_to[i]), int(np.float32(hits.full_score[i]).view(np.uint32)),
            float(hits.full_evalue[i])] for i in range(hits.n)]
    single_pass_launches = int(w["ctx"].stats().ssv_launches)
    hits.close()
    code = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
from checkm_amd import _lib
from synthdata import synth
from tests import common
profs = common.mixed_profiles(); path = common.hmm_file("mixed", profs)
bins = [synth.make_bin(profs, 1000 + b, n_orfs=160, dup_frac=0.4) for b in range(2)]
rng = np.random.default_rng(5)
bins[1] += [("edge_1", "", "M*"), ("edge_2", "", "XXXXXXXXXXXXXXXXXXXXXXXXXXXXXX*"), ("edge_3", "", "ACDEFGHIKLMNPQRSTVWY"),
            ("edge_4", "", synth.to_text(synth.random_residues(rng, 3100)) + "*"), ("edge_5", "", "BJZOUX*acdefghiklmnpqrstvwy")]
ctx = _lib.Context(0); prof = _lib.Profiles(ctx, path); seqs = _lib.Seqs(ctx, bins)
hits = _lib.search(ctx, prof, seqs)
st = ctx.stats()
print(json.dumps({"rows": [[int(hits.seq[i]), int(hits.model[i]), int(hits.ali_from[i]), int(hits.ali_to[i]), int(np.float32(hits.full_score[i]).view(np.uint32)),
      float(hits.full_evalue[i])] for i in range(hits.n)], "launches": int(st.ssv_launches)}))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CKM_PAIR_BUDGET="700", CKM_WS_BUDGET_MB="64")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    got = json.loads(out.stdout.strip().split("\n")[-1])
    assert got["rows"] == ref
    assert got["launches"] > single_pass_launches          # several model chunks, each with its own launches
    # the same search spread over workers by sequence-length class (2, 3, 4 lanes)
    for extra in (dict(CKM_WORKERS="2"), dict(CKM_WORKERS="3"), dict(CKM_WORKERS="4")):
        env = dict(os.environ, CKM_WORKER_MIN_PAIRS="1", **extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (extra, out.stderr[-2000:])
        got = json.loads(out.stdout.strip().split("\n")[-1])
        assert got["rows"] == ref, extra


def test_fasta_ingest_equals_packed_records(world, tmp_path):
    """ckm_seqs_from_fasta (library reads the genes.faa files) against ckm_seqs_pack (caller supplies residues)."""
    w = world
    paths = []
    for b, recs in enumerate(w["bins"]):
        p = tmp_path / ("bin%d.faa" % b)
        synth.write_fasta(str(p), recs)
        paths.append(str(p))
    s2 = _lib.Seqs.from_fasta(w["ctx"], paths)
    assert (s2.nseq, s2.nbins, s2.total_residues) == (w["seqs"].nseq, w["seqs"].nbins, w["seqs"].total_residues)
    assert list(s2.bin_off) == list(w["seqs"].bin_off)
    for i in (0, 7, s2.nseq - 1):
        assert s2.names[i] == w["seqs"].names[i] and s2.descs[i] == w["seqs"].descs[i]
    h1 = _lib.search(w["ctx"], w["prof"], w["seqs"]); h2 = _lib.search(w["ctx"], w["prof"], s2)
    assert h1.n == h2.n
    for f in _lib.HIT_FIELDS:
        assert (getattr(h1, f) == getattr(h2, f)).all(), f
    h1.close(); h2.close(); s2.close()
    with pytest.raises(_lib.CkmError) as e:
        _lib.Seqs.from_fasta(w["ctx"], [str(tmp_path / "nope.faa")])
    assert e.value.code == -2


def _tandem_records(profs, seed, n_per_model=2):
    """ORFs whose domain copies abut with uncertain boundaries -- a copy truncated at a random node followed at once by a
    copy that starts mid-model, or four fragments in a row: the posterior never drops between them, so the region
    fails the rt3 test and is resolved by the trace ensemble."""
    rng = np.random.default_rng(seed)
    recs = []
    for mi, pr in enumerate(profs):
        M = pr.M
        for r in range(n_per_model):
            parts = [synth.random_residues(rng, 10 + 7 * r)]
            if r % 2 == 0:
                a = int(rng.integers(M // 2, M)); b = int(rng.integers(1, M // 2))
                parts += [synth.sample_domain(rng, pr, 1, a), synth.sample_domain(rng, pr, b, M)]
            else:
                for c in range(4):
                    a = int(rng.integers(1, M // 2)); b = int(rng.integers(a + M // 4, M + 1))
                    parts.append(synth.sample_domain(rng, pr, a, b))
            parts.append(synth.random_residues(rng, 12))
            recs.append(("tandem%d_%d" % (mi, r + 1), "", synth.to_text(np.concatenate(parts)) + "*"))
    # thirty fragments in a row: the sampled traces hold ~30 domains, more than the 16 segment slots of the first attempt,
    # so the ensemble is repeated with a larger table (on both sides)
    for mi in (0, 1, 4):
        pr = profs[mi]; M = pr.M
        parts = [synth.random_residues(rng, 8)]
        for c in range(30):
            a = int(rng.integers(1, M // 2)); b = int(rng.integers(a + M // 4, M + 1))
            parts.append(synth.sample_domain(rng, pr, a, b))
        parts.append(synth.random_residues(rng, 8))
        recs.append(("tandem%d_%d" % (mi, n_per_model + 1), "", synth.to_text(np.concatenate(parts)) + "*"))
    return recs


@pytest.fixture(scope="module")
def tandem(gpu_ctx):
    profs = common.mixed_profiles()
    path = common.hmm_file("mixed", profs)
    recs = _tandem_records(profs, 77)
    prof = _lib.Profiles(gpu_ctx, path)
    seqs = _lib.Seqs(gpu_ctx, [recs])
    hs = p7.HmmSet(path)
    dsq = [p7.digitize(r[2]) for r in recs]
    yield dict(ctx=gpu_ctx, profs=profs, prof=prof, seqs=seqs, hs=hs, recs=recs, dsq=dsq, bins=[recs])
    prof.close(); seqs.close(); hs.close()


def test_trace_ensemble_identical(tandem):
    """Every sampled segment of every one of the 200 traces, the per-residue null2 odds sums (bit patterns) and the
    clustered envelopes of a multi-domain region equal the oracle's."""
    w = tandem
    nmulti = 0
    for s, rec in enumerate(w["recs"]):
        m = int(rec[0][len("tandem"):].split("_")[0])
        L = len(w["dsq"][s])
        for (ireg, jreg) in ((1, L), (5, L - 3)):
            rc, n2o, sego, nsego, envo = w["hs"].region_ensemble(m, w["dsq"][s], ireg, jreg)
            assert rc == 0
            n2g, segg, nsegg, envg = _lib.debug_region(w["ctx"], w["prof"], w["seqs"], m, s, ireg, jreg)
            assert (nsego == nsegg).all(), (rec[0], np.nonzero(nsego != nsegg)[0][:5])
            for t in range(200):
                assert (sego[t, :nsego[t]] == segg[t, :nsegg[t]]).all(), (rec[0], t)
            assert (common.float_bits(n2o) == common.float_bits(n2g)).all(), (rec[0], np.nonzero(n2o != n2g)[0][:5])
            assert envo.tolist() == envg.tolist(), (rec[0], envo.tolist(), envg.tolist())
            nmulti += len(envo) >= 2
    assert nmulti >= len(w["recs"])          # the planted copies are found as separate envelopes (two regions per record)


def test_search_multidomain_rows_identical(tandem):
    w = tandem
    hits = _lib.search(w["ctx"], w["prof"], w["seqs"])
    _compare_search(w, hits, None)
    st = w["ctx"].stats()
    assert st.regions_multi >= len(w["recs"]) // 2, st.regions_multi
    assert max(hits.ndom[g] for g in hits.rows(0)) >= 10        # the thirty-fragment targets
    hits.close()


def test_register_class_boundaries(gpu_ctx):
    """Model lengths on both sides of every register-class edge (SSV: 32 cells per register, Viterbi: 128 per packed register,
    Forward: 64 per slot) plus very short models, against planted, tandem, random and degenerate-symbol targets: every stage
    of every pair and the assembled rows."""
    rng = np.random.default_rng(31)
    lengths = [5, 9, 31, 32, 33, 63, 64, 65, 96, 127, 128, 129, 191, 192, 193, 255, 256, 257, 383, 384, 385, 511, 512, 513, 640, 641]
    profs = []
    for i, M in enumerate(lengths):
        p = synth.random_profile(rng, M, "EDGE%03d" % M, "PF%05d.1" % (90000 + i))
        p.stats = (-8.5 - 0.002 * M, 0.71, -9.5 - 0.002 * M, 0.71, -3.8, 0.71)
        profs.append(p)
    path = common.hmm_file("edges", profs)
    recs = []
    for i, p in enumerate(profs):
        dom = synth.sample_domain(rng, p)
        recs.append(("e%d_1" % i, "", synth.to_text(np.concatenate([synth.random_residues(rng, 7), dom, synth.random_residues(rng, 11)])) + "*"))
        if p.M >= 31:
            a = int(rng.integers(p.M // 2, p.M)); b = int(rng.integers(1, p.M // 2))
            recs.append(("e%d_2" % i, "", synth.to_text(np.concatenate([synth.sample_domain(rng, p, 1, a), synth.sample_domain(rng, p, b, p.M)]))))
    recs += [("r_%d" % k, "", synth.to_text(synth.random_residues(rng, int(L)))) for k, L in enumerate([1, 2, 15, 16, 17, 63, 64, 65, 300, 1023, 1024, 1025])]
    recs += [("d_1", "", "BJZOUX" * 20), ("d_2", "", "acdefghiklmnpqrstvwy" * 9 + "*")]
    prof = _lib.Profiles(gpu_ctx, path)
    seqs = _lib.Seqs(gpu_ctx, [recs])
    hs = p7.HmmSet(path)
    dsq = [p7.digitize(r[2]) for r in recs]
    pairs = [(m, s) for m in range(hs.n) for s in range(len(recs))]
    got = _lib.debug_stages(gpu_ctx, prof, seqs, np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs]))
    bad = []
    for i, (m, s) in enumerate(pairs):
        o = hs.stages(m, dsq[s])
        for f in STAGE_FIELDS:
            if not _cmp_stage(o, got[i], f):
                bad.append((lengths[m], recs[s][0], f, getattr(o, f), getattr(got[i], f)))
        if got[i].msvp_xJ != o.msv_xJ or common.float_bits(got[i].msvp_sc) != common.float_bits(o.msv_sc):
            bad.append((lengths[m], recs[s][0], "msv packed", o.msv_xJ, got[i].msvp_xJ))
    assert not bad, bad[:10]
    w = dict(bins=[recs], dsq=dsq, hs=hs)
    hits = _lib.search(gpu_ctx, prof, seqs)
    _compare_search(w, hits, None)
    assert len(set(int(hits.model[g]) for g in hits.rows(0))) >= len(lengths) - 2       # (nearly) every model finds its planted target
    hits.close(); prof.close(); seqs.close(); hs.close()


def test_ragged_model_subsets_single_and_multi_worker(gpu_ctx):
    """lineage_wf shape at some breadth: 8 bins, each with its own random subset of the models in its own order (one bin lists a
    model twice, one bin has no models), against the oracle bin by bin -- and the same search on 3 workers (sequence-length
    classes) in a fresh process must return the same rows."""
    import json
    import os
    import subprocess
    import sys
    code = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
from checkm_amd import _lib
from synthdata import synth
from tests import common
profs = common.mixed_profiles(); path = common.hmm_file("mixed", profs)
rng = np.random.default_rng(123)
bins = [synth.make_bin(profs, 3000 + b, n_orfs=70, dup_frac=0.3) for b in range(8)]
bin_models = []
for b in range(8):
    k = int(rng.integers(3, 13))
    bin_models.append([int(x) for x in rng.permutation(len(profs))[:k]])
bin_models[2] = bin_models[2] + [bin_models[2][0]]
bin_models[5] = []
ctx = _lib.Context(0); prof = _lib.Profiles(ctx, path); seqs = _lib.Seqs(ctx, bins)
hits = _lib.search(ctx, prof, seqs, bin_models)
rows = [[b, int(hits.seq[i]), int(hits.model[i]), int(hits.dom_idx[i]), int(hits.ndom[i]), int(hits.env_from[i]), int(hits.env_to[i]), int(hits.ali_from[i]), int(hits.ali_to[i]),
         int(np.float32(hits.full_score[i]).view(np.uint32)), int(np.float32(hits.dom_score[i]).view(np.uint32)), float(hits.full_evalue[i]), float(hits.i_evalue[i])]
        for b in range(8) for i in hits.rows(b)]
print(json.dumps({"rows": rows, "bin_models": bin_models}))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    results = []
    for extra in (dict(CKM_WORKERS="1"), dict(CKM_WORKERS="3", CKM_WORKER_MIN_PAIRS="1")):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **extra), capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, (extra, out.stderr[-2000:])
        results.append(json.loads(out.stdout.strip().split("\n")[-1]))
    assert results[0]["rows"] == results[1]["rows"] and len(results[0]["rows"]) > 30
    # the oracle, bin by bin, with the bin's own model list and order
    profs = common.mixed_profiles()
    hs = p7.HmmSet(common.hmm_file("mixed", profs))
    bins = [synth.make_bin(profs, 3000 + b, n_orfs=70, dup_frac=0.3) for b in range(8)]
    got = results[0]["rows"]; bin_models = results[0]["bin_models"]
    off = 0; at = 0
    for b, recs in enumerate(bins):
        rows = hs.search(bin_models[b], [p7.digitize(r[2]) for r in recs], [r[0] for r in recs]) if bin_models[b] else []
        mine = [g for g in got if g[0] == b]
        assert len(mine) == len(rows), (b, len(mine), len(rows))
        for g, r in zip(mine, rows):
            assert g[1:9] == [r.seq_idx + off, r.model_idx, r.dom_idx, r.ndom, r.env_from, r.env_to, r.ali_from, r.ali_to], (b, g, r.seq_idx)
            assert g[9] == int(common.float_bits(r.full_score)) and g[10] == int(common.float_bits(r.dom_score)) and g[11] == r.full_evalue and g[12] == r.i_evalue
        off += len(recs)
    assert sum(1 for g in got if g[0] == 5) == 0
    hs.close()


def test_empty_bins_and_empty_records(gpu_ctx, tmp_path):
    """Bins without sequences, a record without residues and a stop-only record (prodigal can emit both for a tiny contig): no rows,
    no pairs, a domtblout that is header and trailer only -- and the neighbouring bin is scored as if it were alone."""
    from checkm_amd import qa as cqa
    from checkm_amd.hmmer import read_domtblout
    profs = synth.small_profiles(7, 6, 20, 150)            # (a calibrated set: checkm_amd/synth_stats.json)
    path = common.hmm_file("empties", profs)
    full = synth.make_bin(profs, 77, n_orfs=40)
    bins = [[], full, [("only_empty_1", "", ""), ("tiny_2", "", "*")], []]
    prof = _lib.Profiles(gpu_ctx, path)
    seqs = _lib.Seqs(gpu_ctx, bins)
    alone = _lib.Seqs(gpu_ctx, [full])
    hs = p7.HmmSet(path)
    try:
        hits = _lib.search(gpu_ctx, prof, seqs)
        ref = _lib.search(gpu_ctx, prof, alone)
        assert ref.n > 0 and hits.n == ref.n and list(hits.bin_row_off) == [0, 0, hits.n, hits.n, hits.n]
        for f in ("model", "hmm_from", "hmm_to", "ali_from", "ali_to", "env_from", "env_to", "full_evalue", "i_evalue", "dom_score"):
            assert list(getattr(hits, f)) == list(getattr(ref, f)), f
        assert [int(x) for x in hits.seq] == [int(x) for x in ref.seq]        # bin 0 holds no sequences, so the indices coincide
        for b in (0, 2, 3):
            out = str(tmp_path / ("empty%d.txt" % b))
            hits.write_domtblout(prof, seqs, b, out)
            names = [r[0] for r in bins[b]]
            assert open(out).read() == hs.format_domtblout([], names, [r[1] for r in bins[b]])
            assert read_domtblout(out) == []
        plan = cqa.QAPlan.for_hmm_models(prof, [list(range(prof.n))] * 4)
        q = plan.reduce(gpu_ctx, hits, seqs)
        assert [float(q.completeness[b]) for b in (0, 2, 3)] == [0.0, 0.0, 0.0] and float(q.completeness[1]) > 50.0
        q.close(); hits.close(); ref.close()
    finally:
        prof.close(); seqs.close(); alone.close(); hs.close()


def test_trace_ensemble_substream_mode():
    """CKM_ENS_STREAM=substream (opt-in; rounds 1-3's default): every one of the 200 tracebacks draws from its own sub-stream of the
    generator.  The DEFAULT -- one stream per region, carried from trace to trace as hmmsearch carries its generator -- is what every
    other test of this file runs.  In a fresh process, against the oracle in the same mode: same segments, same null2 sums, same
    envelopes for a few regions, and the rows of a whole search."""
    import json
    import os
    import subprocess
    import sys
    code = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
from checkm_amd import _lib
from oracle import p7
from tests import common
from tests.test_gpu_scan import _tandem_records
profs = common.mixed_profiles(); path = common.hmm_file("mixed", profs)
recs = _tandem_records(profs, 77)[:6]
p7.lib().p7o_set_ensemble_stream(0)
ctx = _lib.Context(0); prof = _lib.Profiles(ctx, path); seqs = _lib.Seqs(ctx, [recs])
hs = p7.HmmSet(path)
dsq = [p7.digitize(r[2]) for r in recs]
out = []
for si in (0, 3):
    m = int(recs[si][0][len("tandem"):].split("_")[0]); L = len(dsq[si])
    rc, n2, segs, nseg, env = hs.region_ensemble(m, dsq[si], 1, L)
    g_n2, g_segs, g_nseg, g_env = _lib.debug_region(ctx, prof, seqs, m, si, 1, L)
    same_segs = all((segs[t, :nseg[t]] == g_segs[t, :g_nseg[t]]).all() for t in range(200)) if (nseg == g_nseg).all() else False
    out.append(dict(rc=int(rc), n2=bool((n2.view(np.uint32) == g_n2.view(np.uint32)).all()), segs=bool(same_segs), env=bool(env.tolist() == g_env.tolist()), nenv=int(len(g_env))))
hits = _lib.search(ctx, prof, seqs)
rows = hs.search(list(range(hs.n)), dsq, [r[0] for r in recs])
same = hits.n == len(rows) and all((r.seq_idx, r.model_idx, r.hmm_from, r.hmm_to, r.ali_from, r.ali_to, r.env_from, r.env_to, r.ndom) ==
                                  (hits.seq[i], hits.model[i], hits.hmm_from[i], hits.hmm_to[i], hits.ali_from[i], hits.ali_to[i], hits.env_from[i], hits.env_to[i], hits.ndom[i]) and
                                  np.float32(r.dom_score).view(np.uint32) == np.float32(hits.dom_score[i]).view(np.uint32) for i, r in enumerate(rows))
print(json.dumps(dict(regions=out, rows=int(hits.n), same=bool(same), multi=int(ctx.stats().regions_multi))))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, CKM_ENS_STREAM="substream"), capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    out = json.loads(res.stdout.strip().split("\n")[-1])
    for o in out["regions"]:
        assert o["rc"] == 0 and o["n2"] and o["segs"] and o["env"] and o["nenv"] >= 1, o
    assert out["same"] and out["rows"] > 6 and out["multi"] >= 2, out


def test_real_format_hmm_file_searched(gpu_ctx, tmp_path):
    """The library's reader on an HMM file laid out like CheckM's data bundle (tests/common.py:real_format_hmm_text: 3/b and 3/f headers, DATE /
    NSEQ / EFFN / CKSUM / BM / SM / COM, DESC with spaces, annotation columns, records without COMPO, records without ACC): the headers the
    mirror of checkm/hmmerModelParser.py sees, and a search whose domtblout text equals the oracle's on the same file."""
    profs = synth.small_profiles(11, 12, 40, 300)[:8]
    path = common.real_format_hmm_file("real8", profs)
    prof = _lib.Profiles(gpu_ctx, path)
    assert prof.n == len(profs)
    assert [h["name"] for h in prof.headers] == [p.name for p in profs] and [h["leng"] for h in prof.headers] == [p.M for p in profs]
    assert not prof.headers[3]["acc"] and not prof.headers[7]["acc"] and prof.headers[0]["acc"] == profs[0].acc
    recs = synth.make_bin(profs, 5, n_orfs=60, dup_frac=0.3)
    seqs = _lib.Seqs(gpu_ctx, [recs])
    hits = _lib.search(gpu_ctx, prof, seqs)
    hs = p7.HmmSet(path)
    names, descs = [r[0] for r in recs], [r[1] for r in recs]
    rows = hs.search(list(range(hs.n)), [p7.digitize(r[2]) for r in recs], names)
    assert hits.n == len(rows) >= len(profs)
    out = str(tmp_path / "real.txt")
    hits.write_domtblout(prof, seqs, 0, out)
    assert open(out).read() == hs.format_domtblout(rows, names, descs)
    hits.close(); seqs.close(); prof.close(); hs.close()
