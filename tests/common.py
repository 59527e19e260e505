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
