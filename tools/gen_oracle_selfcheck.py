#!/usr/bin/env python
"""Regression pin of the scan-half oracle: runs oracle/p7oracle.c on fixed synthetic inputs and stores what it returns
(domtblout text, stage values as float32 bit patterns, a trace ensemble).  NOT reference vectors -- HMMER does not exist in
/root/reference or this image (the scan half stays PARITY UNPINNED); the file only guards the restatement against
unnoticed changes, because the HIP path is tested against the oracle and would follow it silently.
usage: python tools/gen_oracle_selfcheck.py > tests/golden/oracle_selfcheck.json"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from synthdata import synth          # noqa: E402
from oracle import p7                 # noqa: E402
from tests import common             # noqa: E402


def build():
    profs = synth.small_profiles(7, 6, 20, 150) + synth.small_profiles(13, 1, 300, 400)
    path = common.hmm_file("selfcheck", profs)
    hs = p7.HmmSet(path)
    rng = np.random.default_rng(2024)
    recs = synth.make_bin(profs, 4242, n_orfs=60, dup_frac=0.5)
    p = profs[2]
    a, b = p.M * 3 // 4, p.M // 3
    tandem = np.concatenate([synth.random_residues(rng, 9), synth.sample_domain(rng, p, 1, a), synth.sample_domain(rng, p, b, p.M), synth.random_residues(rng, 8)])
    recs.append(("tandem_1", "", synth.to_text(tandem) + "*"))
    dsq = [p7.digitize(r[2]) for r in recs]
    names = [r[0] for r in recs]
    rows = hs.search(list(range(hs.n)), dsq, names)
    text = hs.format_domtblout(rows, names, [r[1] for r in recs])
    bits = lambda v: int(np.float32(v).view(np.uint32))
    stages = []
    for m in range(hs.n):
        for s in (0, 7, 23, len(recs) - 1):
            st = hs.stages(m, dsq[s])
            stages.append([m, s, st.msv_xJ, bits(st.msv_sc), bits(st.bias_sc), st.vit_xC, bits(st.vit_sc), bits(st.fwd_sc), bits(st.fwd_xC), st.fwd_nscale,
                           st.pass_msv, st.pass_bias, st.pass_vit, st.pass_fwd])
    L = len(dsq[-1])
    rc, n2, segs, nseg, env = hs.region_ensemble(2, dsq[-1], 1, L)
    out = {"note": "oracle regression pin, NOT reference-derived (see tools/gen_oracle_selfcheck.py)",
           "nrows": len(rows), "domtblout_sha256": hashlib.sha256(text.encode()).hexdigest(), "first_rows": text.split("\n")[3:9],
           "stages": stages, "ensemble": {"rc": rc, "n2_bits": [bits(v) for v in n2[:40]], "nseg_hist": np.bincount(nseg).tolist(),
                                          "first_traces": segs[:5, :3].tolist(), "env": env.tolist()},
           "seeds": [int(p7.lib().p7o_ensemble_seed(t)) for t in (0, 1, 2, 199)]}
    hs.close()
    return out


if __name__ == "__main__":
    json.dump(build(), sys.stdout, indent=1)
    sys.stdout.write("\n")
