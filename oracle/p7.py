"""ctypes face of oracle/p7oracle.c (scan-half oracle).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libp7oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("p7oracle.c", "p7oracle.h", "p7simd.c", "p7simd.h", "Makefile")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


class Row(C.Structure):
    _fields_ = [("model_idx", C.c_int32), ("seq_idx", C.c_int32), ("tlen", C.c_int32), ("qlen", C.c_int32),
                ("full_evalue", C.c_double), ("full_score", C.c_float), ("full_bias", C.c_float),
                ("dom_idx", C.c_int32), ("ndom", C.c_int32), ("c_evalue", C.c_double), ("i_evalue", C.c_double),
                ("dom_score", C.c_float), ("dom_bias", C.c_float),
                ("hmm_from", C.c_int32), ("hmm_to", C.c_int32), ("ali_from", C.c_int32), ("ali_to", C.c_int32),
                ("env_from", C.c_int32), ("env_to", C.c_int32), ("acc", C.c_float),
                ("full_lnP", C.c_double), ("dom_lnP", C.c_double)]


class Stages(C.Structure):
    _fields_ = [("msv_xJ", C.c_int32), ("msv_sc", C.c_float), ("null_sc", C.c_float), ("bias_sc", C.c_float),
                ("vit_xC", C.c_int32), ("vit_sc", C.c_float), ("fwd_sc", C.c_float), ("fwd_xC", C.c_float),
                ("fwd_nscale", C.c_int32), ("pass_msv", C.c_int32), ("pass_bias", C.c_int32),
                ("pass_vit", C.c_int32), ("pass_fwd", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.p7o_hmmset_read.restype = C.c_void_p
        L.p7o_hmmset_read.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.p7o_hmmset_free.argtypes = [C.c_void_p]
        L.p7o_hmmset_n.argtypes = [C.c_void_p]
        L.p7o_hmmset_get.restype = C.c_void_p
        L.p7o_hmmset_get.argtypes = [C.c_void_p, C.c_int]
        L.p7o_hmm_M.argtypes = [C.c_void_p]
        L.p7o_hmm_name.restype = C.c_char_p
        L.p7o_hmm_name.argtypes = [C.c_void_p]
        L.p7o_hmm_acc.restype = C.c_char_p
        L.p7o_hmm_acc.argtypes = [C.c_void_p]
        L.p7o_digitize.argtypes = [C.c_char_p, C.c_int64, C.c_void_p]
        L.p7o_stages.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(Stages)]
        L.p7o_search.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                 C.POINTER(C.c_char_p), C.c_double, C.c_double,
                                 C.POINTER(C.POINTER(Row)), C.POINTER(C.c_int)]
        L.p7o_format_domtblout.restype = C.c_void_p
        L.p7o_format_domtblout.argtypes = [C.c_void_p, C.POINTER(Row), C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]
        L.p7o_free.argtypes = [C.c_void_p]
        L.p7o_envelope.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float),
                                   C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
        L.p7o_align.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.p7o_envelope_alignment.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.p7o_set_ensemble_stream.argtypes = [C.c_int]
        L.p7o_set_simd.argtypes = [C.c_int]
        L.p7o_simd_available.restype = C.c_int
        L.p7o_get_simd.restype = C.c_int
        L.p7o_msv_probe.restype = C.c_int64
        L.p7o_msv_probe.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int64)]
        L.p7o_ensemble_seed.restype = C.c_uint32
        L.p7o_ensemble_seed.argtypes = [C.c_int]
        L.p7o_region_ensemble.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32)]
        _lib = L
    return _lib


def set_simd(on):
    """The integer filters in their striped AVX2 form (oracle/p7simd.c) instead of the scalar loops -- same results (the tests check it);
    bench.py's cpu_baseline times the search that way (kind "port-simd").  Returns whether it is in force (False on a CPU without AVX2)."""
    lib().p7o_set_simd(1 if on else 0)
    return bool(lib().p7o_get_simd())


def digitize(text):
    b = text.encode() if isinstance(text, str) else text
    out = np.empty(len(b), dtype=np.uint8)
    lib().p7o_digitize(b, len(b), out.ctypes.data)
    return out


class HmmSet(object):
    def __init__(self, path):
        err = C.create_string_buffer(512)
        self.h = lib().p7o_hmmset_read(path.encode(), err, 512)
        if not self.h:
            raise IOError(err.value.decode())
        self.n = lib().p7o_hmmset_n(self.h)

    def model(self, i):
        return lib().p7o_hmmset_get(self.h, i)

    def M(self, i):
        return lib().p7o_hmm_M(self.model(i))

    def name(self, i):
        return lib().p7o_hmm_name(self.model(i)).decode()

    def acc(self, i):
        return lib().p7o_hmm_acc(self.model(i)).decode()

    def stages(self, i, dsq):
        st = Stages()
        d = np.ascontiguousarray(dsq, dtype=np.uint8)
        lib().p7o_stages(self.model(i), d.ctypes.data, len(d), C.byref(st))
        return st

    def envelope(self, i, dsq, ienv, jenv):
        d = np.ascontiguousarray(dsq, dtype=np.uint8)
        envsc, oasc, xC = C.c_float(), C.c_float(), C.c_float()
        ns = C.c_int32()
        null2 = np.zeros(20, dtype=np.float32)
        coords = np.zeros(4, dtype=np.int32)
        rc = lib().p7o_envelope(self.model(i), d.ctypes.data, len(d), ienv, jenv, C.byref(envsc), C.byref(oasc),
                                null2.ctypes.data, coords.ctypes.data, C.byref(xC), C.byref(ns))
        return rc, envsc.value, oasc.value, null2, coords, xC.value, ns.value

    def align(self, i, dsq):
        """hmmalign restated: residue (1-based) emitted by each match state of model i on the OA path, 0 = none."""
        d = np.ascontiguousarray(dsq, dtype=np.uint8)
        path = np.zeros(self.M(i), dtype=np.int32)
        rc = lib().p7o_align(self.model(i), d.ctypes.data, len(d), path.ctypes.data)
        return rc, path

    def envelope_alignment(self, i, dsq, ienv, jenv):
        """The domain alignment hmmsearch prints: (rc, path[M] -- envelope-local residue of each match state, 0 = none --, pp[Ld + 1] --
        posterior probability of each residue on the path in its emitting state)."""
        d = np.ascontiguousarray(dsq, dtype=np.uint8)
        path = np.zeros(self.M(i), dtype=np.int32)
        pp = np.zeros(jenv - ienv + 2, dtype=np.float32)
        rc = lib().p7o_envelope_alignment(self.model(i), d.ctypes.data, len(d), ienv, jenv, path.ctypes.data, pp.ctypes.data)
        return rc, path, pp

    def region_ensemble(self, i, dsq, ireg, jreg, cap=64):
        """200-trace ensemble of region ireg..jreg: (rc, n2sum[Lr], segs[200][cap][4], nseg[200], envelopes[n][4])."""
        d = np.ascontiguousarray(dsq, dtype=np.uint8)
        Lr = jreg - ireg + 1
        n2 = np.zeros(Lr, dtype=np.float32)
        segs = np.zeros((200, cap, 4), dtype=np.int32)
        nseg = np.zeros(200, dtype=np.int32)
        env = np.zeros((64, 4), dtype=np.int32)
        nenv = C.c_int32()
        rc = lib().p7o_region_ensemble(self.model(i), d.ctypes.data, len(d), ireg, jreg, n2.ctypes.data, segs.ctypes.data,
                                       nseg.ctypes.data, cap, env.ctypes.data, 64, C.byref(nenv))
        return rc, n2, segs, nseg, env[:nenv.value].copy()

    def search(self, model_idx, seqs, names, E=0.1, domE=0.1):
        """seqs: list of digitized uint8 arrays.  Returns list of Row (copied)."""
        offs = np.zeros(len(seqs) + 1, dtype=np.int64)
        for i, s in enumerate(seqs):
            offs[i + 1] = offs[i] + len(s)
        cat = np.concatenate(seqs) if seqs else np.zeros(0, dtype=np.uint8)
        cat = np.ascontiguousarray(cat, dtype=np.uint8)
        mi = np.ascontiguousarray(model_idx, dtype=np.int32)
        nm = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        rows = C.POINTER(Row)()
        n = C.c_int()
        lib().p7o_search(self.h, mi.ctypes.data, len(mi), cat.ctypes.data, offs.ctypes.data, len(seqs), nm,
                         E, domE, C.byref(rows), C.byref(n))
        out = []
        for i in range(n.value):
            r = Row()
            C.memmove(C.byref(r), C.byref(rows[i]), C.sizeof(Row))
            out.append(r)
        if n.value:
            lib().p7o_free(rows)
        return out

    def msv_probe(self, model_idx, seqs):
        """The MSV filter alone over every pair: (cells scored, sum of the final bytes)."""
        offs = np.zeros(len(seqs) + 1, dtype=np.int64)
        for i, s in enumerate(seqs):
            offs[i + 1] = offs[i] + len(s)
        cat = np.ascontiguousarray(np.concatenate(seqs) if seqs else np.zeros(0, dtype=np.uint8), dtype=np.uint8)
        mi = np.ascontiguousarray(model_idx, dtype=np.int32)
        chk = C.c_int64()
        cells = lib().p7o_msv_probe(self.h, mi.ctypes.data, len(mi), cat.ctypes.data, offs.ctypes.data, len(seqs), C.byref(chk))
        return int(cells), int(chk.value)

    def format_domtblout(self, rows, names, descs):
        arr = (Row * max(1, len(rows)))(*rows)
        nm = (C.c_char_p * max(1, len(names)))(*[n.encode() for n in names])
        ds = (C.c_char_p * max(1, len(descs)))(*[d.encode() for d in descs])
        p = lib().p7o_format_domtblout(self.h, arr, len(rows), nm, ds)
        s = C.string_at(p).decode()
        lib().p7o_free(p)
        return s

    def close(self):
        if self.h:
            lib().p7o_hmmset_free(self.h)
            self.h = None
