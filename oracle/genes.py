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
    src = [os.path.join(_HERE, f) for f in ("gene_oracle.c", "gene_full.c", "Makefile")]
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


# ---- the whole gene finder (oracle/gene_full.c: single-genome mode as CheckM invokes it; parity unpinned) -----------------------------
class Training(C.Structure):
    _fields_ = [("gc", C.c_double), ("trans_table", C.c_int32), ("st_wt", C.c_double), ("bias", C.c_double * 3), ("type_wt", C.c_double * 3),
                ("uses_sd", C.c_int32), ("rbs_wt", C.c_double * 28), ("ups_comp", (C.c_double * 4) * 32), ("mot_wt", ((C.c_double * 4096) * 4) * 4),
                ("no_mot", C.c_double), ("gene_dc", C.c_double * 4096)]


class Gene(C.Structure):
    _fields_ = [("contig", C.c_int32), ("begin", C.c_int32), ("end", C.c_int32), ("strand", C.c_int32), ("start_type", C.c_int32),
                ("partial_left", C.c_int32), ("partial_right", C.c_int32), ("rbs_bin", C.c_int32), ("mot_len", C.c_int32), ("mot_ndx", C.c_int32),
                ("mot_spacer", C.c_int32), ("gc_cont", C.c_double), ("conf", C.c_double), ("score", C.c_double), ("cscore", C.c_double),
                ("sscore", C.c_double), ("rscore", C.c_double), ("uscore", C.c_double), ("tscore", C.c_double)]


def _full():
    L = lib()
    if not getattr(L, "_full_ready", False):
        L.pg_train.argtypes = [C.POINTER(C.c_void_p), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Training)]
        L.pg_find.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(Training), C.c_int, C.c_int, C.POINTER(Gene), C.c_int]
        L.pg_translate.argtypes = [C.c_void_p, C.c_int, C.POINTER(Gene), C.c_int, C.c_char_p]
        assert L.pg_sizeof_training() == C.sizeof(Training) and L.pg_sizeof_gene() == C.sizeof(Gene)
        L._full_ready = True
    return L


def train(contigs, trans_table=11, closed=False, mask=True):
    """Training pass over all contigs of a bin (digitized arrays or text).  Returns a Training, or None below 20000 bases."""
    L = _full()
    ds = [digitize(c) if isinstance(c, (str, bytes)) else np.ascontiguousarray(c, dtype=np.uint8) for c in contigs]
    ptrs = (C.c_void_p * len(ds))(*[d.ctypes.data for d in ds])
    lens = (C.c_int32 * len(ds))(*[len(d) for d in ds])
    t = Training()
    rc = L.pg_train(ptrs, lens, len(ds), trans_table, 1 if closed else 0, 1 if mask else 0, C.byref(t))
    return (t, ds) if rc == 0 else (None, ds)


def find_genes(contigs, trans_table=11, closed=False, mask=True):
    """(training, [Gene], [protein text]) of a bin: train on everything, then genes contig by contig."""
    L = _full()
    t, ds = train(contigs, trans_table, closed, mask)
    if t is None:
        return None, [], []
    genes, prots = [], []
    for ci, d in enumerate(ds):
        cap = max(16, len(d) // 90 + 16)
        buf = (Gene * cap)()
        n = L.pg_find(d.ctypes.data, len(d), ci, C.byref(t), 1 if closed else 0, 1 if mask else 0, buf, cap)
        assert n <= cap
        for k in range(n):
            g = Gene.from_buffer_copy(buf[k])
            out = C.create_string_buffer((g.end - g.begin + 1) // 3 + 2)
            L.pg_translate(d.ctypes.data, len(d), C.byref(g), trans_table, out)
            genes.append(g)
            prots.append(out.value.decode())
    return t, genes, prots
