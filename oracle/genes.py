"""ctypes face of oracle/gene_oracle.c (TEST INFRASTRUCTURE: the restated node extraction of the gene finder, parity unpinned)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "libgeneoracle.so")
_lib = None
ATG, GTG, TTG, STOP = 0, 1, 2, 3


class Node(C.Structure):
    _fields_ = [("ndx", C.c_int32), ("type", C.c_int32), ("strand", C.c_int32), ("stop_val", C.c_int32), ("edge", C.c_int32)]


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("gene_oracle.c", "Makefile")]
    if force or not os.path.exists(_LIB) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _LIB


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.go_nodes.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Node), C.c_int]
        L.go_sort.argtypes = [C.POINTER(Node), C.c_int]
        L.go_digitize.argtypes = [C.c_char_p, C.c_int64, C.c_void_p]
        _lib = L
    return _lib


def digitize(text):
    b = text.encode() if isinstance(text, str) else text
    out = np.empty(len(b), dtype=np.uint8)
    lib().go_digitize(b, len(b), out.ctypes.data)
    return out


def nodes(seq, trans_table=11, closed=False, sort=True):
    """[(ndx, type, strand, stop_val, edge)] of one contig (digitized bases or text), in prodigal's working order when sort=True."""
    d = digitize(seq) if isinstance(seq, (str, bytes)) else np.ascontiguousarray(seq, dtype=np.uint8)
    cap = max(64, len(d))
    buf = (Node * cap)()
    n = lib().go_nodes(d.ctypes.data, len(d), trans_table, 1 if closed else 0, buf, cap)
    if n > cap:
        cap = n
        buf = (Node * cap)()
        n = lib().go_nodes(d.ctypes.data, len(d), trans_table, 1 if closed else 0, buf, cap)
    if sort:
        lib().go_sort(buf, n)
    return [(x.ndx, x.type, x.strand, x.stop_val, x.edge) for x in buf[:n]]
