"""The device-driven cascade (stages chained on the device, conservative decisions, ONE host synchronisation per lane, exact decisions
again on the host) against the host-driven one (CKM_CASCADE=host: a device phase, a copy and a host decision per stage): same rows,
bit for bit, on one lane and on three, with the long/short split of a lane,
with more pairs than one SSV pass holds (host-driven, chunk by chunk) -- and when the device-side workspace is too small the regions
that find no room are rescored by the host-driven rounds, and when the device-side tables are too small the lane is handed to the
host-driven cascade (counted in cascade_fallback_lanes); the rows never change."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CODE = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
from checkm_amd import _lib
from synthdata import synth
from tests import common
import os
if os.environ.get("CKM_TEST_SHAPE") == "classes":
    # one model per SSV launch class and more (lengths 24, 48, ... 2040, and 2100 beyond the SSV image): with the long / short split
    # of the lane that is > 100 groups of queues and streams in ONE search
    prng = np.random.default_rng(77)
    profs = [synth.random_profile(prng, M, "c%%04d" %% M, "PF%%05d.1" %% (50000 + M)) for M in list(range(24, 2049, 24)) + [2100]]
    for p in profs: p.stats = (-8.5 - 0.002 * p.M, 0.71, -9.5 - 0.002 * p.M, 0.71, -3.8, 0.71); p.ga = (25.0, 25.0)
    path = common.hmm_file("classes", profs)
else:
    profs = common.mixed_profiles(); path = common.hmm_file("mixed", profs)
rng = np.random.default_rng(5)
bins = [synth.make_bin(profs, 8800 + b, n_orfs=260, dup_frac=0.4) for b in range(6)]
# a tandem repeat (multi-domain region -> trace ensemble) in every other bin
for b in range(0, 6, 2):
    p = profs[3 + b]
    a, c = p.M * 3 // 4, p.M // 3
    t = np.concatenate([synth.random_residues(rng, 9), synth.sample_domain(rng, p, 1, a), synth.sample_domain(rng, p, c, p.M), synth.random_residues(rng, 8)])
    bins[b].append(("tandem%%d_1" %% b, "", synth.to_text(t) + "*"))
ctx = _lib.Context(0); prof = _lib.Profiles(ctx, path); seqs = _lib.Seqs(ctx, bins)
hits = _lib.search(ctx, prof, seqs)
st = ctx.stats()
f32 = lambda v: int(np.float32(v).view(np.uint32))
rows = [[b, int(hits.seq[i]), int(hits.model[i]), int(hits.dom_idx[i]), int(hits.ndom[i]), int(hits.hmm_from[i]), int(hits.hmm_to[i]), int(hits.ali_from[i]), int(hits.ali_to[i]),
         int(hits.env_from[i]), int(hits.env_to[i]), f32(hits.full_score[i]), f32(hits.full_bias[i]), f32(hits.dom_score[i]), f32(hits.dom_bias[i]), f32(hits.acc[i]),
         float(hits.full_evalue[i]), float(hits.c_evalue[i]), float(hits.i_evalue[i])] for b in range(6) for i in hits.rows(b)]
print(json.dumps({"rows": rows, "fallback": int(st.cascade_fallback_lanes), "pairs": [int(st.pairs_ssv), int(st.pairs_bias), int(st.pairs_vit), int(st.pairs_fwd), int(st.pairs_dom), int(st.envelopes), int(st.regions_multi)]}))
''' % ROOT


def _run(**extra):
    env = dict(os.environ, **extra)
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (extra, out.stderr[-3000:])
    return json.loads(out.stdout.strip().split("\n")[-1])


def test_device_cascade_equals_host_cascade():
    host = _run(CKM_CASCADE="host", CKM_WORKERS="1")
    assert len(host["rows"]) > 100 and host["pairs"][6] >= 2           # multi-domain regions are present
    for extra in (dict(CKM_WORKERS="1"), dict(CKM_WORKERS="1", CKM_SPLIT_MIN_PAIRS="1"),
                  dict(CKM_WORKERS="1", CKM_SPLIT_MIN_PAIRS="1", CKM_LONG_SHARE="0.2,0.4"),          # three parts by sequence length
                  dict(CKM_WORKERS="3", CKM_WORKER_MIN_PAIRS="1"), dict(CKM_WORKERS="2", CKM_WORKER_MIN_PAIRS="1", CKM_PAIR_BUDGET="9000")):
        dev = _run(**extra)
        if "CKM_PAIR_BUDGET" in extra:
            assert dev["fallback"] >= 1 and dev["rows"] == host["rows"]         # more pairs than one SSV pass holds: host-driven, chunk by chunk
            continue
        assert dev["fallback"] == 0, extra
        assert dev["rows"] == host["rows"], extra
        assert dev["pairs"][0] == host["pairs"][0]
        # the device decides conservatively: it may let a few more pairs through a filter than the exact test, never fewer
        assert dev["pairs"][1] >= host["pairs"][1] and dev["pairs"][3] >= host["pairs"][3] and dev["pairs"][4] >= host["pairs"][4]
        assert dev["pairs"][4] <= host["pairs"][4] + 5 and dev["pairs"][5] == host["pairs"][5]
    # workspace too small for the matrices of all envelopes at once: the regions that find no room are rescored by the host-driven
    # rounds (workspace-sized batches) -- no fallback of the lane, same rows
    small = _run(CKM_WORKERS="1", CKM_WS_BUDGET_MB="64", CKM_TRACE="1")
    assert small["fallback"] == 0 and small["rows"] == host["rows"]
    # device-side tables far too small: the lane is handed to the host-driven cascade (and the tables grow for the next call)
    tiny = _run(CKM_WORKERS="1", CKM_CAP_SHRINK="64")
    assert tiny["fallback"] >= 1 and tiny["rows"] == host["rows"]


def test_one_search_with_every_model_length_class():
    """A database whose models fall into every SSV launch class (the 8-lane classes, the 16-lane classes, and the no-SSV class beyond 2048
    nodes): more than a hundred groups in one device-driven search -- no fallback (round 3's first 8-lane build overran a 64-group limit
    and quietly ran configs[4]-shaped searches on the host-driven cascade), rows equal to the host-driven cascade's."""
    host = _run(CKM_CASCADE="host", CKM_WORKERS="1", CKM_TEST_SHAPE="classes")
    dev = _run(CKM_WORKERS="1", CKM_SPLIT_MIN_PAIRS="1", CKM_TEST_SHAPE="classes")
    assert len(host["rows"]) >= 6 * 86
    assert dev["fallback"] == 0 and dev["rows"] == host["rows"] and dev["pairs"][0] == host["pairs"][0]
