"""TEST INFRASTRUCTURE: ctypes face of tests/emu/libgene_emu.so -- the gene-calling pipeline source of libcheckm_hip.so
(checkm_amd/csrc/gene_pipe.h) compiled against a host executor.  Never imported by checkm_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libgene_emu.so")
_CSRC = os.path.join(_HERE, "..", "..", "checkm_amd", "csrc")
_lib = None


class Columns(C.Structure):
    _fields_ = [("n", C.c_uint64), ("bin", C.POINTER(C.c_uint32)), ("contig", C.POINTER(C.c_uint32)), ("begin", C.POINTER(C.c_int32)), ("end", C.POINTER(C.c_int32)),
                ("strand", C.POINTER(C.c_int8)), ("start_type", C.POINTER(C.c_uint8)), ("partial_left", C.POINTER(C.c_uint8)), ("partial_right", C.POINTER(C.c_uint8)),
                ("rbs_bin", C.POINTER(C.c_int32)), ("mot_len", C.POINTER(C.c_int32)), ("mot_ndx", C.POINTER(C.c_int32)), ("mot_spacer", C.POINTER(C.c_int32)),
                ("gc_cont", C.POINTER(C.c_double)), ("conf", C.POINTER(C.c_double)), ("score", C.POINTER(C.c_double)), ("cscore", C.POINTER(C.c_double)),
                ("sscore", C.POINTER(C.c_double)), ("rscore", C.POINTER(C.c_double)), ("uscore", C.POINTER(C.c_double)), ("tscore", C.POINTER(C.c_double)),
                ("prot_off", C.POINTER(C.c_uint64)), ("prot", C.c_char_p),
                ("nbins", C.c_uint64), ("bin_trained", C.POINTER(C.c_uint8)), ("bin_uses_sd", C.POINTER(C.c_uint8)), ("bin_gc", C.POINTER(C.c_double)),
                ("bin_bases", C.POINTER(C.c_uint64)), ("bin_coding", C.POINTER(C.c_uint64)), ("bin_nodes", C.POINTER(C.c_uint64))]


FIELDS = ("bin", "contig", "begin", "end", "strand", "start_type", "partial_left", "partial_right", "rbs_bin", "mot_len", "mot_ndx", "mot_spacer",
          "gc_cont", "conf", "score", "cscore", "sscore", "rscore", "uscore", "tscore")


def build(force=False):
    srcs = [os.path.join(_HERE, "gene_emu.cpp")] + [os.path.join(_CSRC, f) for f in ("gene_pipe.h", "gene_dev.h", "gene_exec.h")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function", "-Wno-unused-variable",
                               "-o", _LIB, os.path.join(_HERE, "gene_emu.cpp")])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.emu_genes_call.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.emu_genes_error.restype = C.c_char_p
        L.emu_genes_error.argtypes = [C.c_void_p]
        L.emu_genes_columns.argtypes = [C.c_void_p, C.POINTER(Columns)]
        L.emu_genes_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def call_genes(bins, trans_table=11, mask=True):
    """Same contract as checkm_amd._lib.call_genes: (columns, per-bin dict)."""
    parts, bin_first = [], [0]
    for contigs in bins:
        for c in contigs:
            parts.append(c.encode() if isinstance(c, str) else bytes(c))
        bin_first.append(len(parts))
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    if parts:
        np.cumsum([len(p) for p in parts], out=off[1:])
    bf = np.asarray(bin_first, dtype=np.uint32)
    text = b"".join(parts)
    h = C.c_void_p()
    rc = lib().emu_genes_call(text, off.ctypes.data, len(parts), bf.ctypes.data, len(bins), int(trans_table), 1 if mask else 0, C.byref(h))
    try:
        if rc != 0:
            raise RuntimeError(lib().emu_genes_error(h).decode())
        cols = Columns()
        lib().emu_genes_columns(h, C.byref(cols))
        n, nb = int(cols.n), int(cols.nbins)
        arr = np.ctypeslib.as_array
        out = {f: (arr(getattr(cols, f), shape=(n,)).copy() if n else np.zeros(0, dtype=np.int64)) for f in FIELDS}
        if n:
            out["contig"] = out["contig"] - bf[out["bin"]]
        po = arr(cols.prot_off, shape=(n + 1,)).copy() if n else np.zeros(1, dtype=np.uint64)
        blob = C.string_at(cols.prot, int(po[-1])) if n else b""
        txt, pl = blob.decode("ascii"), po.tolist()
        out["proteins"] = [txt[a:b] for a, b in zip(pl[:-1], pl[1:])]
        per_bin = {f: (arr(getattr(cols, "bin_" + f), shape=(nb,)).copy() if nb else np.zeros(0)) for f in ("trained", "uses_sd", "gc", "bases", "coding", "nodes")}
    finally:
        lib().emu_genes_free(h)
    return out, per_bin
