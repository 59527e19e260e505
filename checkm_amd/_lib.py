"""ctypes binding of libcheckm_hip.so (include/checkm_hip.h).

The library is the only compute path: if it is missing or no gfx950 device is usable this
module raises -- there is no CPU fallback anywhere in checkm_amd.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcheckm_hip.so")

ABI_VERSION = 7


class CkmError(RuntimeError):
    def __init__(self, code, msg):
        RuntimeError.__init__(self, "libcheckm_hip error %d: %s" % (code, msg))
        self.code = code


class ModelHeader(C.Structure):
    _fields_ = [("name", C.c_char_p), ("acc", C.c_char_p), ("desc", C.c_char_p), ("leng", C.c_int32),
                ("has_ga", C.c_int32), ("has_tc", C.c_int32), ("has_nc", C.c_int32),
                ("ga", C.c_double * 2), ("tc", C.c_double * 2), ("nc", C.c_double * 2), ("evparam", C.c_float * 6), ("searchable", C.c_int32)]


class HitColumns(C.Structure):
    _fields_ = [("n", C.c_uint64), ("nbins", C.c_uint32), ("bin_row_off", C.POINTER(C.c_uint64)),
                ("seq", C.POINTER(C.c_uint32)), ("model", C.POINTER(C.c_uint32)),
                ("tlen", C.POINTER(C.c_int32)), ("qlen", C.POINTER(C.c_int32)),
                ("full_evalue", C.POINTER(C.c_double)), ("full_score", C.POINTER(C.c_float)), ("full_bias", C.POINTER(C.c_float)),
                ("dom_idx", C.POINTER(C.c_int32)), ("ndom", C.POINTER(C.c_int32)),
                ("c_evalue", C.POINTER(C.c_double)), ("i_evalue", C.POINTER(C.c_double)),
                ("dom_score", C.POINTER(C.c_float)), ("dom_bias", C.POINTER(C.c_float)),
                ("hmm_from", C.POINTER(C.c_int32)), ("hmm_to", C.POINTER(C.c_int32)),
                ("ali_from", C.POINTER(C.c_int32)), ("ali_to", C.POINTER(C.c_int32)),
                ("env_from", C.POINTER(C.c_int32)), ("env_to", C.POINTER(C.c_int32)), ("acc", C.POINTER(C.c_float)),
                ("target_name", C.POINTER(C.c_char_p)), ("full_score_d", C.POINTER(C.c_double)), ("dom_score_d", C.POINTER(C.c_double))]


HIT_FIELDS = ["seq", "model", "tlen", "qlen", "full_evalue", "full_score", "full_bias", "dom_idx", "ndom", "c_evalue",
              "i_evalue", "dom_score", "dom_bias", "hmm_from", "hmm_to", "ali_from", "ali_to", "env_from", "env_to", "acc"]


class SearchStats(C.Structure):
    _fields_ = [("pairs_ssv", C.c_uint64), ("pairs_msv_full", C.c_uint64), ("pairs_bias", C.c_uint64), ("pairs_vit", C.c_uint64),
                ("pairs_fwd", C.c_uint64), ("pairs_dom", C.c_uint64), ("envelopes", C.c_uint64), ("regions_multi", C.c_uint64), ("pairs_vit_exact", C.c_uint64), ("cells_ssv", C.c_uint64),
                ("residue_hmm", C.c_uint64), ("ms_ssv", C.c_double), ("ms_filters", C.c_double), ("ms_fwdbwd", C.c_double),
                ("ms_domains", C.c_double), ("ms_host", C.c_double), ("ms_total", C.c_double), ("ssv_launches", C.c_uint32), ("cascade_fallback_lanes", C.c_uint32),
                ("ws_cap_bytes", C.c_uint64), ("ws_used_bytes", C.c_uint64)]


class StageScores(C.Structure):
    _fields_ = [("msv_xJ", C.c_int32), ("msv_sc", C.c_float), ("null_sc", C.c_float), ("bias_sc", C.c_float),
                ("vit_xC", C.c_int32), ("vit_sc", C.c_float), ("fwd_sc", C.c_float), ("fwd_xC", C.c_float),
                ("fwd_nscale", C.c_int32), ("ssv_maxv", C.c_int32), ("msvp_xJ", C.c_int32), ("msvp_sc", C.c_float)]


class EnvelopeResult(C.Structure):
    _fields_ = [("envsc", C.c_float), ("oasc", C.c_float), ("fwd_xC", C.c_float), ("nscale", C.c_int32), ("null2", C.c_float * 20),
                ("hmm_from", C.c_int32), ("hmm_to", C.c_int32), ("ali_from", C.c_int32), ("ali_to", C.c_int32), ("ok", C.c_int32)]


class ModelInfo(C.Structure):
    _fields_ = [("nmodels", C.c_uint32), ("qlen", C.c_void_p), ("thr_kind", C.c_void_p), ("thr_full", C.c_void_p),
                ("thr_dom", C.c_void_p), ("is_pf", C.c_void_p), ("clan", C.c_void_p), ("nest_off", C.c_void_p),
                ("nest_idx", C.c_void_p), ("key", C.c_void_p)]


class ReduceFlags(C.Structure):
    _fields_ = [("ignore_thresholds", C.c_int32), ("skip_pseudogene_correction", C.c_int32), ("skip_adj_correction", C.c_int32),
                ("individual_markers", C.c_int32), ("evalue_threshold", C.c_double), ("length_threshold", C.c_double),
                ("bin_select", C.c_void_p), ("nvariants", C.c_uint32), ("bin_variant", C.c_void_p)]


class MarkerSetsCSR(C.Structure):
    _fields_ = [("nbins", C.c_uint32), ("set_off", C.c_void_p), ("marker_off", C.c_void_p), ("marker_key", C.c_void_p)]


class QAColumns(C.Structure):
    _fields_ = [("nbins", C.c_uint32), ("hist", C.POINTER(C.c_int32)), ("completeness", C.POINTER(C.c_double)),
                ("contamination", C.POINTER(C.c_double)), ("set_off", C.POINTER(C.c_uint32)),
                ("set_present", C.POINTER(C.c_int32)), ("set_multi", C.POINTER(C.c_int32)),
                ("nkept", C.c_uint64), ("kept_bin_off", C.POINTER(C.c_uint64)), ("kept_key", C.POINTER(C.c_uint32)),
                ("kept_row", C.POINTER(C.c_uint64)), ("kept_row2", C.POINTER(C.c_uint64)),
                ("kept_tlen", C.POINTER(C.c_int32)), ("kept_hmm_from", C.POINTER(C.c_int32)), ("kept_hmm_to", C.POINTER(C.c_int32)),
                ("kept_ali_from", C.POINTER(C.c_int32)), ("kept_ali_to", C.POINTER(C.c_int32)),
                ("kept_env_from", C.POINTER(C.c_int32)), ("kept_env_to", C.POINTER(C.c_int32))]


class OrfColumns(C.Structure):
    _fields_ = [("n", C.c_uint64), ("contig", C.POINTER(C.c_uint32)), ("ndx", C.POINTER(C.c_int32)), ("stop_val", C.POINTER(C.c_int32)),
                ("type", C.POINTER(C.c_uint8)), ("strand_rev", C.POINTER(C.c_uint8)), ("edge", C.POINTER(C.c_uint8)),
                ("ms_flags", C.c_double), ("ms_chain", C.c_double), ("bases", C.c_uint64), ("padded_bytes", C.c_uint64)]


class NucBatchView(C.Structure):
    _fields_ = [("text", C.c_void_p), ("contig_off", C.POINTER(C.c_uint64)), ("bin_first", C.POINTER(C.c_uint32)), ("contig_ids", C.c_void_p),
                ("bin_bases", C.POINTER(C.c_uint64)), ("ncontigs", C.c_uint32), ("nbins", C.c_uint32)]


class GeneColumns(C.Structure):
    _fields_ = [("n", C.c_uint64), ("bin", C.POINTER(C.c_uint32)), ("contig", C.POINTER(C.c_uint32)), ("begin", C.POINTER(C.c_int32)), ("end", C.POINTER(C.c_int32)),
                ("strand", C.POINTER(C.c_int8)), ("start_type", C.POINTER(C.c_uint8)), ("partial_left", C.POINTER(C.c_uint8)), ("partial_right", C.POINTER(C.c_uint8)),
                ("rbs_bin", C.POINTER(C.c_int32)), ("mot_len", C.POINTER(C.c_int32)), ("mot_ndx", C.POINTER(C.c_int32)), ("mot_spacer", C.POINTER(C.c_int32)),
                ("gc_cont", C.POINTER(C.c_double)), ("conf", C.POINTER(C.c_double)), ("score", C.POINTER(C.c_double)), ("cscore", C.POINTER(C.c_double)),
                ("sscore", C.POINTER(C.c_double)), ("rscore", C.POINTER(C.c_double)), ("uscore", C.POINTER(C.c_double)), ("tscore", C.POINTER(C.c_double)),
                ("prot_off", C.POINTER(C.c_uint64)), ("prot", C.POINTER(C.c_char)),
                ("nbins", C.c_uint64), ("bin_trained", C.POINTER(C.c_uint8)), ("bin_uses_sd", C.POINTER(C.c_uint8)), ("bin_gc", C.POINTER(C.c_double)),
                ("bin_bases", C.POINTER(C.c_uint64)), ("bin_coding", C.POINTER(C.c_uint64)), ("bin_nodes", C.POINTER(C.c_uint64)),
                ("ms_nodes", C.c_double), ("ms_dp_train", C.c_double), ("ms_score", C.c_double), ("ms_dp_find", C.c_double), ("ms_total", C.c_double)]


class TableColumns(C.Structure):
    _fields_ = [("cols", HitColumns), ("target_accession", C.POINTER(C.c_char_p)), ("query_name", C.POINTER(C.c_char_p)),
                ("query_accession", C.POINTER(C.c_char_p)), ("description", C.POINTER(C.c_char_p)),
                ("full_bias_d", C.POINTER(C.c_double)), ("dom_bias_d", C.POINTER(C.c_double)), ("acc_d", C.POINTER(C.c_double)),
                ("bin_missing", C.POINTER(C.c_uint8))]


# every symbol include/checkm_hip.h declares
EXPORTS = ["ckm_last_error", "ckm_abi_version", "ckm_device_count", "ckm_ctx_create", "ckm_ctx_destroy", "ckm_ctx_reserve",
           "ckm_profiles_load", "ckm_profiles_count", "ckm_profiles_header", "ckm_profiles_free",
           "ckm_seqs_pack", "ckm_seqs_from_fasta", "ckm_seqs_count", "ckm_seqs_bin_offsets", "ckm_seqs_name", "ckm_seqs_residues", "ckm_seqs_free", "ckm_search", "ckm_hits_columns", "ckm_hits_free",
           "ckm_hits_write_domtblout", "ckm_hits_write_alignments", "ckm_last_search_stats", "ckm_reduce", "ckm_qa_columns_get", "ckm_qa_free", "ckm_count_sets",
           "ckm_align", "ckm_tables_read", "ckm_tables_assign_models", "ckm_tables_get", "ckm_tables_free",
           "ckm_orf_scan", "ckm_orf_columns_get", "ckm_orf_free", "ckm_debug_orf_flags", "ckm_genes_call", "ckm_genes_columns_get", "ckm_genes_free", "ckm_genes_coding_union", "ckm_genes_write_bin",
           "ckm_nuc_batch_read", "ckm_nuc_batch_view_get", "ckm_nuc_batch_free",
           "ckm_debug_stages", "ckm_debug_envelopes", "ckm_debug_region"]

_lib = None


def load():
    """Load the shared library (no device needed for this); raises if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback)" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.ckm_last_error.restype = C.c_char_p
    L.ckm_device_count.argtypes = [C.POINTER(C.c_int)]
    L.ckm_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.ckm_ctx_destroy.argtypes = [C.c_void_p]
    L.ckm_ctx_destroy.restype = None
    L.ckm_ctx_reserve.argtypes = [C.c_void_p, C.c_uint64, C.c_double]
    L.ckm_profiles_load.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p)]
    L.ckm_profiles_count.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    L.ckm_profiles_header.argtypes = [C.c_void_p, C.c_int32, C.POINTER(ModelHeader)]
    L.ckm_profiles_free.argtypes = [C.c_void_p]
    L.ckm_profiles_free.restype = None
    L.ckm_seqs_pack.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32,
                                C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_void_p)]
    L.ckm_seqs_from_fasta.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_void_p)]
    L.ckm_seqs_count.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.ckm_seqs_bin_offsets.argtypes = [C.c_void_p, C.POINTER(C.POINTER(C.c_uint32))]
    L.ckm_seqs_name.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.POINTER(C.c_int32)]
    L.ckm_seqs_residues.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.ckm_seqs_free.argtypes = [C.c_void_p]
    L.ckm_seqs_free.restype = None
    L.ckm_search.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                             C.POINTER(C.c_void_p)]
    L.ckm_hits_columns.argtypes = [C.c_void_p, C.POINTER(HitColumns)]
    L.ckm_hits_free.argtypes = [C.c_void_p]
    L.ckm_hits_free.restype = None
    L.ckm_hits_write_domtblout.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p]
    L.ckm_hits_write_alignments.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p]
    L.ckm_last_search_stats.argtypes = [C.c_void_p, C.POINTER(SearchStats)]
    L.ckm_reduce.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(HitColumns), C.c_void_p, C.POINTER(ModelInfo),
                             C.POINTER(ReduceFlags), C.POINTER(MarkerSetsCSR), C.POINTER(C.c_void_p)]
    L.ckm_qa_columns_get.argtypes = [C.c_void_p, C.POINTER(QAColumns)]
    L.ckm_qa_free.argtypes = [C.c_void_p]
    L.ckm_qa_free.restype = None
    L.ckm_count_sets.argtypes = [C.c_void_p, C.POINTER(MarkerSetsCSR)] + [C.c_void_p] * 7
    L.ckm_align.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.ckm_tables_read.argtypes = [C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_void_p)]
    L.ckm_tables_assign_models.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_uint64)]
    L.ckm_tables_get.argtypes = [C.c_void_p, C.POINTER(TableColumns)]
    L.ckm_tables_free.argtypes = [C.c_void_p]
    L.ckm_tables_free.restype = None
    L.ckm_orf_scan.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.ckm_orf_columns_get.argtypes = [C.c_void_p, C.POINTER(OrfColumns)]
    L.ckm_orf_free.argtypes = [C.c_void_p]
    L.ckm_orf_free.restype = None
    L.ckm_genes_call.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    L.ckm_genes_columns_get.argtypes = [C.c_void_p, C.POINTER(GeneColumns)]
    L.ckm_genes_free.argtypes = [C.c_void_p]
    L.ckm_genes_free.restype = None
    L.ckm_genes_coding_union.argtypes = [C.c_void_p, C.c_void_p]
    L.ckm_genes_write_bin.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p]
    L.ckm_nuc_batch_read.argtypes = [C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(C.c_void_p)]
    L.ckm_nuc_batch_view_get.argtypes = [C.c_void_p, C.POINTER(NucBatchView)]
    L.ckm_nuc_batch_free.argtypes = [C.c_void_p]
    L.ckm_nuc_batch_free.restype = None
    L.ckm_debug_orf_flags.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
    L.ckm_debug_stages.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
    L.ckm_debug_envelopes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_uint32, C.c_void_p]
    L.ckm_debug_region.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    if L.ckm_abi_version() != ABI_VERSION:
        raise ImportError("libcheckm_hip ABI version mismatch")
    _lib = L
    return L


def _chk(rc):
    if rc != 0:
        raise CkmError(rc, load().ckm_last_error().decode(errors="replace"))


def device_count():
    n = C.c_int(0)
    rc = load().ckm_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class Context(object):
    def __init__(self, device=0):
        self.h = C.c_void_p()
        _chk(load().ckm_ctx_create(device, C.byref(self.h)))
        self.device = device

    def close(self):
        if self.h:
            load().ckm_ctx_destroy(self.h)
            self.h = C.c_void_p()

    def reserve(self, pairs, cells):
        """Start allocating the workspace a search of that size will ask for, in the background (ckm_ctx_reserve: `cells` = sum over
        the bins, over the models a bin is scanned against, of (Mp + 64) * Mp, Mp the model length padded to 64)."""
        _chk(load().ckm_ctx_reserve(self.h, int(pairs), float(cells)))

    def stats(self):
        st = SearchStats()
        _chk(load().ckm_last_search_stats(self.h, C.byref(st)))
        return st


class Profiles(object):
    def __init__(self, ctx, path):
        self.ctx = ctx
        self.h = C.c_void_p()
        _chk(load().ckm_profiles_load(ctx.h, path.encode(), C.byref(self.h)))
        n = C.c_int32()
        _chk(load().ckm_profiles_count(self.h, C.byref(n)))
        self.n = n.value
        self.headers = []
        for i in range(self.n):
            hd = ModelHeader()
            _chk(load().ckm_profiles_header(self.h, i, C.byref(hd)))
            self.headers.append({"name": hd.name.decode(), "acc": hd.acc.decode() if hd.acc else None,
                                 "desc": hd.desc.decode() if hd.desc else None, "leng": hd.leng,
                                 "ga": tuple(hd.ga) if hd.has_ga else None, "tc": tuple(hd.tc) if hd.has_tc else None,
                                 "nc": tuple(hd.nc) if hd.has_nc else None, "evparam": tuple(hd.evparam), "searchable": bool(hd.searchable)})

    def close(self):
        if self.h:
            load().ckm_profiles_free(self.h)
            self.h = C.c_void_p()


class _LazyStrings(object):
    """names[i] / descs[i] fetched from the library on demand (millions of ORFs need no Python list)."""

    def __init__(self, seqs, which):
        self.seqs, self.which, self.cache = seqs, which, {}

    def __len__(self):
        return self.seqs.nseq

    def __getitem__(self, i):
        v = self.cache.get(i)
        if v is None:
            name, desc = C.c_char_p(), C.c_char_p()
            _chk(load().ckm_seqs_name(self.seqs.h, int(i), C.byref(name), C.byref(desc), None))
            v = (name.value if self.which == 0 else desc.value).decode()
            self.cache[i] = v
        return v


class Seqs(object):
    """All sequences of all bins, packed and resident in HBM."""

    @classmethod
    def from_fasta(cls, ctx, paths):
        """One protein FASTA file per bin, read, digitized and packed by the library (ckm_seqs_from_fasta)."""
        self = cls.__new__(cls)
        self.ctx = ctx
        arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
        self.h = C.c_void_p()
        _chk(load().ckm_seqs_from_fasta(ctx.h, arr, len(paths), C.byref(self.h)))
        n, nb = C.c_uint32(), C.c_uint32()
        _chk(load().ckm_seqs_count(self.h, C.byref(n), C.byref(nb)))
        self.nseq, self.nbins = n.value, nb.value
        bo = C.POINTER(C.c_uint32)()
        _chk(load().ckm_seqs_bin_offsets(self.h, C.byref(bo)))
        self.bin_off = np.ctypeslib.as_array(bo, shape=(self.nbins + 1,)).copy()
        tot = C.c_uint64()
        _chk(load().ckm_seqs_residues(self.h, C.byref(tot)))
        self.total_residues = int(tot.value)
        self.names = _LazyStrings(self, 0)
        self.descs = _LazyStrings(self, 1)
        return self

    def __init__(self, ctx, bins):
        """bins: list of lists of (name, desc, residues) records."""
        self.ctx = ctx
        names, descs, parts = [], [], []
        bin_off = [0]
        for recs in bins:
            for name, desc, seq in recs:
                names.append(name.encode())
                descs.append((desc or "").encode())
                parts.append(seq.encode() if isinstance(seq, str) else seq)
            bin_off.append(len(names))
        self.nseq = len(names)
        self.nbins = len(bins)
        self.names = [n.decode() for n in names]
        self.descs = [d.decode() for d in descs]
        self.lengths = np.array([len(p) for p in parts], dtype=np.int64)
        off = np.zeros(self.nseq + 1, dtype=np.uint64)
        np.cumsum(self.lengths, out=off[1:])
        text = b"".join(parts)
        self.bin_off = np.array(bin_off, dtype=np.uint32)
        na = (C.c_char_p * max(1, self.nseq))(*names)
        da = (C.c_char_p * max(1, self.nseq))(*descs)
        self.h = C.c_void_p()
        _chk(load().ckm_seqs_pack(ctx.h, text, off.ctypes.data, self.nseq, self.bin_off.ctypes.data, self.nbins, na, da,
                                  C.byref(self.h)))
        self.total_residues = int(self.lengths.sum())

    def close(self):
        if self.h:
            load().ckm_seqs_free(self.h)
            self.h = C.c_void_p()


class Hits(object):
    def __init__(self, h):
        self.h = h
        cols = HitColumns()
        _chk(load().ckm_hits_columns(h, C.byref(cols)))
        self.cols = cols
        self.n = int(cols.n)
        self.nbins = int(cols.nbins)
        self.bin_row_off = np.ctypeslib.as_array(cols.bin_row_off, shape=(self.nbins + 1,)).copy()
        for f in HIT_FIELDS:
            ptr = getattr(cols, f)
            setattr(self, f, np.ctypeslib.as_array(ptr, shape=(self.n,)).copy() if self.n else np.zeros(0))

    def rows(self, b):
        return range(int(self.bin_row_off[b]), int(self.bin_row_off[b + 1]))

    def write_domtblout(self, profiles, seqs, b, path):
        _chk(load().ckm_hits_write_domtblout(self.h, profiles.h, seqs.h, b, path.encode()))

    def write_alignments(self, ctx, profiles, seqs, b, path):
        """hmmsearch-style report of bin b with the domain alignments (what `-o` holds when CheckM keeps alignments)."""
        _chk(load().ckm_hits_write_alignments(ctx.h, self.h, profiles.h, seqs.h, b, path.encode()))

    def close(self):
        if self.h:
            load().ckm_hits_free(self.h)
            self.h = None


class Tables(object):
    """domtblout text of many bins, parsed by the library (ckm_tables_read); column views + the ext form ckm_reduce takes."""

    def __init__(self, paths):
        arr = (C.c_char_p * max(1, len(paths)))(*[p.encode() for p in paths])
        h = C.c_void_p()
        _chk(load().ckm_tables_read(arr, len(paths), C.byref(h)))
        self.h = h
        self._refresh()

    def _refresh(self):
        tc = TableColumns()
        _chk(load().ckm_tables_get(self.h, C.byref(tc)))
        self.tc = tc
        cols = tc.cols
        self.n = int(cols.n)
        self.nbins = int(cols.nbins)
        self.bin_row_off = np.ctypeslib.as_array(cols.bin_row_off, shape=(self.nbins + 1,)).copy()
        self.missing = np.ctypeslib.as_array(tc.bin_missing, shape=(max(1, self.nbins),)).copy()[:self.nbins].astype(bool)

    def assign_models(self, keys):
        """keys: markerHits key of every model slot; returns the number of rows whose accession is not among them."""
        arr = (C.c_char_p * max(1, len(keys)))(*[k.encode() for k in keys])
        miss = C.c_uint64()
        _chk(load().ckm_tables_assign_models(self.h, arr, len(keys), C.byref(miss)))
        self._refresh()
        return int(miss.value)

    def column(self, name):
        ptr = getattr(self.tc, name) if name in ("full_bias_d", "dom_bias_d", "acc_d") else getattr(self.tc.cols, name)
        return np.ctypeslib.as_array(ptr, shape=(self.n,)) if self.n else np.zeros(0)

    def hit(self, r):
        """Row r as the reference's HmmerHitDOM would hold it (checkm/hmmer.py:255-285): Python ints, floats and strings."""
        c, t = self.tc.cols, self.tc
        return dict(target_name=c.target_name[r].decode(), target_accession=t.target_accession[r].decode(), target_length=int(c.tlen[r]),
                    query_name=t.query_name[r].decode(), query_accession=t.query_accession[r].decode(), query_length=int(c.qlen[r]),
                    full_e_value=float(c.full_evalue[r]), full_score=float(c.full_score_d[r]), full_bias=float(t.full_bias_d[r]),
                    dom=int(c.dom_idx[r]), ndom=int(c.ndom[r]), c_evalue=float(c.c_evalue[r]), i_evalue=float(c.i_evalue[r]),
                    dom_score=float(c.dom_score_d[r]), dom_bias=float(t.dom_bias_d[r]), hmm_from=int(c.hmm_from[r]), hmm_to=int(c.hmm_to[r]),
                    ali_from=int(c.ali_from[r]), ali_to=int(c.ali_to[r]), env_from=int(c.env_from[r]), env_to=int(c.env_to[r]),
                    acc=float(t.acc_d[r]), target_description=t.description[r].decode())

    def text(self, name, r):
        v = getattr(self.tc, name)[r] if name != "target_name" else self.tc.cols.target_name[r]
        return v.decode()

    def ext(self):
        """(HitColumns, keepalive) for QAPlan.reduce(ext=...)."""
        return self.tc.cols, [self]

    def close(self):
        if self.h:
            load().ckm_tables_free(self.h)
            self.h = None


def search(ctx, profiles, seqs, bin_models=None, E=0.1, domE=0.1):
    """bin_models: None (all models for every bin) or a list (per bin) of model-index lists."""
    out = C.c_void_p()
    if bin_models is None:
        _chk(load().ckm_search(ctx.h, profiles.h, seqs.h, None, None, E, domE, C.byref(out)))
    else:
        off = np.zeros(len(bin_models) + 1, dtype=np.uint32)
        for i, m in enumerate(bin_models):
            off[i + 1] = off[i] + len(m)
        flat = [x for m in bin_models for x in m]
        idx = np.array(flat if flat else [0], dtype=np.uint32)
        _chk(load().ckm_search(ctx.h, profiles.h, seqs.h, off.ctypes.data, idx.ctypes.data, E, domE, C.byref(out)))
    return Hits(out)


def debug_stages(ctx, profiles, seqs, model, seq):
    model = np.ascontiguousarray(model, dtype=np.uint32)
    seq = np.ascontiguousarray(seq, dtype=np.uint32)
    out = (StageScores * len(model))()
    _chk(load().ckm_debug_stages(ctx.h, profiles.h, seqs.h, model.ctypes.data, seq.ctypes.data, len(model), out))
    return out


def debug_envelopes(ctx, profiles, seqs, model, seq, ienv, jenv):
    model = np.ascontiguousarray(model, dtype=np.uint32)
    seq = np.ascontiguousarray(seq, dtype=np.uint32)
    ienv = np.ascontiguousarray(ienv, dtype=np.int32)
    jenv = np.ascontiguousarray(jenv, dtype=np.int32)
    out = (EnvelopeResult * len(model))()
    _chk(load().ckm_debug_envelopes(ctx.h, profiles.h, seqs.h, model.ctypes.data, seq.ctypes.data, ienv.ctypes.data,
                                    jenv.ctypes.data, len(model), out))
    return out


def debug_region(ctx, profiles, seqs, model, seq, ireg, jreg, cap=64):
    """Trace ensemble of one region: (n2sum[Lr], segs[200][cap][4], nseg[200], envelopes[n][4]), region-local coordinates."""
    n2 = np.zeros(jreg - ireg + 1, dtype=np.float32)
    segs = np.zeros((200, cap, 4), dtype=np.int32)
    nseg = np.zeros(200, dtype=np.int32)
    env = np.zeros((64, 4), dtype=np.int32)
    nenv = C.c_int32()
    _chk(load().ckm_debug_region(ctx.h, profiles.h, seqs.h, model, seq, ireg, jreg, n2.ctypes.data, segs.ctypes.data, nseg.ctypes.data, cap,
                                 env.ctypes.data, 64, C.byref(nenv)))
    return n2, segs, nseg, env[:nenv.value].copy()


def align(ctx, profiles, seqs, model, seq):
    """Optimal-accuracy alignment of whole sequences to models (what hmmalign computes per sequence): list of int32 arrays, one per
    pair, of length M: 1-based residue emitted by each match state, 0 = none."""
    model = np.ascontiguousarray(model, dtype=np.uint32)
    seq = np.ascontiguousarray(seq, dtype=np.uint32)
    off = np.zeros(len(model) + 1, dtype=np.uint64)
    for j, m in enumerate(model):
        off[j + 1] = off[j] + profiles.headers[int(m)]["leng"]
    out = np.zeros(max(1, int(off[-1])), dtype=np.int32)
    _chk(load().ckm_align(ctx.h, profiles.h, seqs.h, model.ctypes.data, seq.ctypes.data, len(model), off.ctypes.data, out.ctypes.data))
    return [out[int(off[j]):int(off[j + 1])].copy() for j in range(len(model))]


def orf_nodes(ctx, contigs, trans_table=11, closed=False):
    """Start / stop nodes of all six frames of a bin's contigs (ckm_orf_scan): contigs = list of nucleotide strings (or bytes).
    Returns (columns dict of numpy arrays: contig, ndx, stop_val, type, strand_rev, edge; stats dict)."""
    parts = [c.encode() if isinstance(c, str) else bytes(c) for c in contigs]
    off = np.zeros(len(parts) + 1, dtype=np.uint64)
    np.cumsum([len(p) for p in parts], out=off[1:])
    text = b"".join(parts)
    h = C.c_void_p()
    _chk(load().ckm_orf_scan(ctx.h, text, off.ctypes.data, len(parts), int(trans_table), 1 if closed else 0, C.byref(h)))
    try:
        cols = OrfColumns()
        _chk(load().ckm_orf_columns_get(h, C.byref(cols)))
        n = int(cols.n)
        arr = np.ctypeslib.as_array
        out = {f: (arr(getattr(cols, f), shape=(n,)).copy() if n else np.zeros(0, dtype=np.int64)) for f in ("contig", "ndx", "stop_val", "type", "strand_rev", "edge")}
        stats = dict(ms_flags=cols.ms_flags, ms_chain=cols.ms_chain, bases=int(cols.bases), padded_bytes=int(cols.padded_bytes))
    finally:
        load().ckm_orf_free(h)
    return out, stats


GENE_FIELDS = ("bin", "contig", "begin", "end", "strand", "start_type", "partial_left", "partial_right", "rbs_bin", "mot_len", "mot_ndx", "mot_spacer",
               "gc_cont", "conf", "score", "cscore", "sscore", "rscore", "uscore", "tscore")


class GeneBatch(object):
    """The nucleotides of a batch of bins laid out for ckm_genes_call: bins = [[(contig id, sequence as str or bytes), ...], ...]
    (or plain sequences).  One batch serves both translation tables and the writers."""

    def __init__(self, bins):
        parts, ids, bin_first = [], [], [0]
        for contigs in bins:
            for c in contigs:
                cid, seq = c if isinstance(c, tuple) else ("", c)
                parts.append(seq.encode() if isinstance(seq, str) else bytes(seq))
                ids.append(cid.encode() if isinstance(cid, str) else bytes(cid))
            bin_first.append(len(parts))
        self.nbins, self.ncontigs = len(bins), len(parts)
        self.off = np.zeros(len(parts) + 1, dtype=np.uint64)
        if parts:
            np.cumsum([len(p) for p in parts], out=self.off[1:])
        self.bin_first = np.asarray(bin_first, dtype=np.uint32)
        self.text = b"".join(parts)
        self.ids = (C.c_char_p * max(1, len(ids)))(*ids)
        self.bases = [int(self.off[bin_first[b + 1]] - self.off[bin_first[b]]) for b in range(self.nbins)]
        self._h = None

    @classmethod
    def from_files(cls, paths):
        """The batch of the plain nucleotide FASTA files `paths`, a bin each, read and laid out by the library's host threads
        (ckm_nuc_batch_read: the record rules of geneFinder.read_contigs_bytes) -- no Python object per contig, no interpreter lock held."""
        self = cls.__new__(cls)
        arr = (C.c_char_p * max(1, len(paths)))(*[p.encode() for p in paths])
        h = C.c_void_p()
        _chk(load().ckm_nuc_batch_read(arr, len(paths), C.byref(h)))
        self._h = h
        v = NucBatchView()
        _chk(load().ckm_nuc_batch_view_get(h, C.byref(v)))
        self.nbins, self.ncontigs = int(v.nbins), int(v.ncontigs)
        self.off = np.ctypeslib.as_array(v.contig_off, shape=(self.ncontigs + 1,))
        total = int(self.off[-1])
        # views of the library's batch (alive until close): the text as bytes-like uint8, the ids as an address
        self.text = np.ctypeslib.as_array((C.c_uint8 * total).from_address(v.text)) if total else np.zeros(0, dtype=np.uint8)
        self.ids = v.contig_ids
        self.bin_first = np.ctypeslib.as_array(v.bin_first, shape=(self.nbins + 1,))
        self.bases = [int(x) for x in np.ctypeslib.as_array(v.bin_bases, shape=(self.nbins,))] if self.nbins else []
        return self

    def text_arg(self):
        """The text as ckm_genes_call / ckm_genes_write_bin take it."""
        return self.text if isinstance(self.text, bytes) else self.text.ctypes.data

    def contigs(self, b):
        """[(id bytes, sequence bytes)] of bin b (tests, diagnostics)."""
        out = []
        for c in range(int(self.bin_first[b]), int(self.bin_first[b + 1])):
            a, z = int(self.off[c]), int(self.off[c + 1])
            if self._h is None:
                out.append((self.ids[c], self.text[a:z]))
            else:
                out.append((C.cast(self.ids + c * C.sizeof(C.c_void_p), C.POINTER(C.c_char_p))[0], self.text[a:z].tobytes()))
        return out

    def close(self):
        if getattr(self, "_h", None):
            self.off = self.bin_first = self.text = None
            load().ckm_nuc_batch_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class GeneCall(object):
    """Result of one ckm_genes_call (a batch of bins, one translation table); close() frees it."""

    def __init__(self, ctx, batch, trans_table=11, closed=False, mask=True):
        self.batch, self.table = batch, int(trans_table)
        self.h = C.c_void_p()
        _chk(load().ckm_genes_call(ctx.h, batch.text_arg(), batch.off.ctypes.data, batch.ncontigs, batch.bin_first.ctypes.data, batch.nbins, self.table, 1 if closed else 0, 1 if mask else 0, C.byref(self.h)))
        cols = GeneColumns()
        _chk(load().ckm_genes_columns_get(self.h, C.byref(cols)))
        self._cols = cols
        nb = int(cols.nbins)
        arr = np.ctypeslib.as_array
        self.per_bin = {f: (arr(getattr(cols, "bin_" + f), shape=(nb,)).copy() if nb else np.zeros(0)) for f in ("trained", "uses_sd", "gc", "bases", "coding", "nodes")}
        self.stats = dict(ms_nodes=cols.ms_nodes, ms_dp_train=cols.ms_dp_train, ms_score=cols.ms_score, ms_dp_find=cols.ms_dp_find, ms_total=cols.ms_total)
        self.ngenes = int(cols.n)

    def columns(self):
        """Columns dict of numpy arrays over all genes + 'proteins' list of str; `contig` is the contig's index inside its bin."""
        cols, n = self._cols, self.ngenes
        arr = np.ctypeslib.as_array
        out = {f: (arr(getattr(cols, f), shape=(n,)).copy() if n else np.zeros(0, dtype=np.int64)) for f in GENE_FIELDS}
        if n:
            out["contig"] = out["contig"] - self.batch.bin_first[out["bin"]]
        po = arr(cols.prot_off, shape=(n + 1,)).copy() if n else np.zeros(1, dtype=np.uint64)
        blob = C.string_at(cols.prot, int(po[-1])) if n else b""
        txt, pl = blob.decode("ascii"), po.tolist()
        out["proteins"] = [txt[a:b] for a, b in zip(pl[:-1], pl[1:])]
        return out

    def genes_per_bin(self):
        if not self.ngenes:
            return np.zeros(self.batch.nbins, dtype=np.int64)
        b = np.ctypeslib.as_array(self._cols.bin, shape=(self.ngenes,))
        return np.bincount(b, minlength=self.batch.nbins)

    def coding_union(self):
        """Bases of every bin covered by at least one gene (checkm/prodigal.py:246-274)."""
        out = np.zeros(max(1, self.batch.nbins), dtype=np.uint64)
        _chk(load().ckm_genes_coding_union(self.h, out.ctypes.data))
        return out[:self.batch.nbins]

    def write_bin(self, b, aaFile, gffFile, ntFile=None):
        bt = self.batch
        _chk(load().ckm_genes_write_bin(self.h, int(b), self.table, bt.ids, bt.text_arg(), bt.off.ctypes.data, bt.bin_first.ctypes.data,
                                        aaFile.encode(), gffFile.encode(), ntFile.encode() if ntFile else None))

    def close(self):
        if self.h:
            load().ckm_genes_free(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def call_genes(ctx, bins, trans_table=11, closed=False, mask=True):
    """Gene calling for a batch of bins (ckm_genes_call): bins = list of lists of nucleotide strings / bytes (the contigs of each bin).
    Returns (columns dict of numpy arrays over all genes + 'proteins' list of str, per-bin dict, stats dict); `contig` is the contig's
    index inside its bin."""
    call = GeneCall(ctx, GeneBatch(bins), trans_table, closed, mask)
    try:
        return call.columns(), call.per_bin, call.stats
    finally:
        call.close()


def debug_orf_flags(ctx, nbytes, reps=10):
    """Average duration (ms) of one launch of the streaming codon-flag kernel over nbytes of device-generated nucleotides."""
    ms = C.c_double()
    _chk(load().ckm_debug_orf_flags(ctx.h, int(nbytes), int(reps), C.byref(ms)))
    return ms.value
