#!/usr/bin/env python
"""Calibrate the synthetic profiles (STATS LOCAL MSV/VITERBI/FORWARD) the way hmmbuild would:
lambda from mean match relative entropy, mu/tau fitted on 200 random sequences of length 100.

TEST TOOLING: uses the CPU oracle to score the random sequences.  Output: checkm_amd/synth_stats.json
(committed), which checkm_amd/synth.py reads so that neither bench.py nor the product needs the oracle
to build its inputs.  Re-run only when synth.py's generators change.
"""
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from synthdata import synth  # noqa: E402
from oracle import p7  # noqa: E402

LN2 = float(np.log(2.0))


def gumbel_fit_loc(x, lam):
    return float(-np.log(np.mean(np.exp(-lam * np.asarray(x)))) / lam)


def gumbel_fit(x):
    """ML fit of both Gumbel parameters (Newton on lambda), as esl_gumbel_FitComplete."""
    x = np.asarray(x, dtype=np.float64)
    lam = np.pi / np.sqrt(6 * x.var())
    for _ in range(100):
        e = np.exp(-lam * x)
        fx = 1 / lam - x.mean() + (x * e).sum() / e.sum()
        dfx = ((x * e).sum() / e.sum()) ** 2 - (x * x * e).sum() / e.sum() - 1 / (lam * lam)
        step = fx / dfx
        lam -= step
        if abs(step) < 1e-9:
            break
    return gumbel_fit_loc(x, lam), float(lam)


def calibrate(profs, rng):
    for p in profs:
        p.stats = (-8.0, 0.7, -9.0, 0.7, -3.5, 0.7)
    tmp = tempfile.NamedTemporaryFile(suffix=".hmm", delete=False).name
    synth.write_hmm(tmp, profs)
    hs = p7.HmmSet(tmp)
    out = []
    for i, p in enumerate(profs):
        H = float(np.mean(np.sum(p.mat[1:] * np.log2(p.mat[1:] / synth.BGF), axis=1)))
        lam = LN2 + 1.44 / (p.M * H)
        msv, vit, fwd = [], [], []
        for _ in range(200):
            d = synth.random_residues(rng, 100).astype(np.uint8)
            st = hs.stages(i, d)
            if np.isfinite(st.msv_sc):
                msv.append((st.msv_sc - st.null_sc) / LN2)
            if np.isfinite(st.vit_sc):
                vit.append((st.vit_sc - st.null_sc) / LN2)
            fwd.append((st.fwd_sc - st.null_sc) / LN2)
        mmu = gumbel_fit_loc(msv, lam)
        vmu = gumbel_fit_loc(vit, lam)
        gmu, glam = gumbel_fit(fwd)
        tailp = 0.04
        tau = (gmu - np.log(-np.log(1 - tailp)) / glam) + np.log(tailp) / lam
        out.append([round(mmu, 4), round(lam, 5), round(vmu, 4), round(lam, 5), round(float(tau), 4), round(lam, 5)])
    hs.close()
    os.unlink(tmp)
    return out


def main_cfg3():
    """STATS LOCAL lines of the 2000-profile lineage world (checkm_amd/synth_lineage.py) -> checkm_amd/synth_stats_cfg3.json."""
    from synthdata import synth_lineage as sl
    rng = np.random.default_rng(20250925)
    profs = sl.lineage_profiles(with_stats=False)
    stats = {}
    for lo in range(0, len(profs), 250):
        chunk = profs[lo:lo + 250]
        for p, s in zip(chunk, calibrate(chunk, rng)):
            stats[p.acc] = {"stats": s, "M": p.M}
        print("calibrated %d / %d" % (min(lo + 250, len(profs)), len(profs)), flush=True)
    with open(os.path.join(ROOT, "checkm_amd", "synth_stats_cfg3.json"), "w") as f:
        json.dump(stats, f, indent=0, sort_keys=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cfg3":
        return main_cfg3()
    rng = np.random.default_rng(20250614)
    stats = {}
    profs = synth.cpr43_profiles(with_stats=False)
    for p, s in zip(profs, calibrate(profs, rng)):
        stats["cpr43/%s" % p.acc] = {"stats": s, "cut": 25.0, "M": p.M}
    for args in synth.SMALL_SETS:
        profs = synth.small_profiles(*args, with_stats=False)
        for p, s in zip(profs, calibrate(profs, rng)):
            stats["small/%d/%s" % (args[0], p.acc)] = {"stats": s, "cut": 20.0, "M": p.M}
    with open(os.path.join(ROOT, "checkm_amd", "synth_stats.json"), "w") as f:
        json.dump(stats, f, indent=0, sort_keys=True)
    print("calibrated %d profiles" % len(stats))


if __name__ == "__main__":
    main()
