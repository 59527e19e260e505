#!/usr/bin/env python
"""Developer probe of the device-driven cascade: one small search (16 profiles x 3 bins x 200 ORFs + a tandem repeat), prints rows and
stage counters.  Used with CKM_CHAIN_STOP / CKM_CASCADE / CKM_TRACE to find the stage a change broke."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from checkm_amd import _lib  # noqa: E402
from synthdata import synth  # noqa: E402
from tests import common  # noqa: E402

profs = common.mixed_profiles()
path = common.hmm_file("mixed", profs)
rng = np.random.default_rng(5)
nb = int(os.environ.get("PROBE_BINS", "3")); no = int(os.environ.get("PROBE_ORFS", "200"))
bins = [synth.make_bin(profs, 8800 + b, n_orfs=no, dup_frac=0.4) for b in range(nb)]
p = profs[3]
t = np.concatenate([synth.random_residues(rng, 9), synth.sample_domain(rng, p, 1, p.M * 3 // 4), synth.sample_domain(rng, p, p.M // 3, p.M), synth.random_residues(rng, 8)])
bins[0].append(("tandem_1", "", synth.to_text(t) + "*"))
ctx = _lib.Context(0); prof = _lib.Profiles(ctx, path); seqs = _lib.Seqs(ctx, bins)
hits = _lib.search(ctx, prof, seqs)
st = ctx.stats()
import hashlib
h = hashlib.sha256()
for f in _lib.HIT_FIELDS:
    h.update(np.ascontiguousarray(getattr(hits, f)).tobytes())
print("rows", hits.n, "sha", h.hexdigest()[:16], "fallback", st.cascade_fallback_lanes, "pairs", st.pairs_ssv, st.pairs_msv_full, st.pairs_bias, st.pairs_vit, st.pairs_vit_exact,
      st.pairs_fwd, st.pairs_dom, st.envelopes, st.regions_multi, "ms", round(st.ms_total, 2))
