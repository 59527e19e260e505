"""Shared helpers for the tests: synthetic inputs on disk, oracle and product side by side."""
import os
import tempfile

import numpy as np

from checkm_amd import synth

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
