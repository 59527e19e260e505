"""Packed-array face of the reduce half: builds the CSR marker-set / model-info arguments of
ckm_reduce (include/checkm_hip.h) and wraps its result.

Reference semantics reproduced by the library: ResultsManager.vetHit/addHit
(checkm/resultsParser.py:340-399), PFAM.filterHitsFromSameClan (checkm/util/pfam.py:86-147),
identifyAdjacentMarkerGenes (resultsParser.py:401-479), geneCounts (:513-537) and
MarkerSet.genomeCheck (checkm/markerSets.py:206-238).
"""
import ctypes as C

import numpy as np

from checkm_amd import _lib

THR_NONE, THR_NC_TIGR, THR_GA, THR_TC, THR_NC = 0, 1, 2, 3, 4
E_VAL, LENGTH = 1e-10, 0.7            # checkm/defaultValues.py:36-37


def resolve_threshold(acc, ga, tc, nc):
    """The cascade of resultsParser.py:356-367 for one model (before --ignore_thresholds)."""
    if nc is not None and 'TIGR' in acc:
        return THR_NC_TIGR, nc
    if ga is not None:
        return THR_GA, ga
    if tc is not None:
        return THR_TC, tc
    if nc is not None:
        return THR_NC, nc
    return THR_NONE, (0.0, 0.0)


class KeyTable(object):
    """String <-> dense id for markerHits dict keys (model accessions and marker ids)."""

    def __init__(self):
        self.ids = {}
        self.names = []

    def get(self, s):
        i = self.ids.get(s)
        if i is None:
            i = len(self.names)
            self.ids[s] = i
            self.names.append(s)
        return i


class QAResult(object):
    def __init__(self, handle, plan):
        self.h = handle
        self.plan = plan
        cols = _lib.QAColumns()
        _lib._chk(_lib.load().ckm_qa_columns_get(handle, C.byref(cols)))
        nb = int(cols.nbins)
        arr = np.ctypeslib.as_array
        self.nbins = nb
        self.hist = arr(cols.hist, shape=(nb * 6,)).copy().reshape(nb, 6) if nb else np.zeros((0, 6), dtype=np.int32)
        self.completeness = arr(cols.completeness, shape=(nb,)).copy() if nb else np.zeros(0)
        self.contamination = arr(cols.contamination, shape=(nb,)).copy() if nb else np.zeros(0)
        self.set_off = arr(cols.set_off, shape=(nb + 1,)).copy()
        ns = int(self.set_off[-1])
        self.set_present = arr(cols.set_present, shape=(ns,)).copy() if ns else np.zeros(0, dtype=np.int32)
        self.set_multi = arr(cols.set_multi, shape=(ns,)).copy() if ns else np.zeros(0, dtype=np.int32)
        nk = int(cols.nkept)
        self.kept_bin_off = arr(cols.kept_bin_off, shape=(nb + 1,)).copy()
        for f in ("kept_key", "kept_row", "kept_row2", "kept_tlen", "kept_hmm_from", "kept_hmm_to", "kept_ali_from",
                  "kept_ali_to", "kept_env_from", "kept_env_to"):
            setattr(self, f, arr(getattr(cols, f), shape=(nk,)).copy() if nk else np.zeros(0, dtype=np.int64))
        self.n_markers = plan.n_markers
        self.n_sets = plan.n_sets

    def close(self):
        if self.h:
            _lib.load().ckm_qa_free(self.h)
            self.h = None


class QAPlan(object):
    """Everything ckm_reduce needs besides the hits: model info + per-bin marker sets in CSR form."""

    def __init__(self, keys, model_acc, model_qlen, model_thr, bin_sets, clans=None, nested=None):
        """
        keys       KeyTable shared with the caller
        model_acc  per model: the markerHits key string (query_accession, or name when ACC is '-')
        model_qlen per model: LENG
        model_thr  per model: (kind, (full, dom))
        bin_sets   per bin: list of collocated sets, each an ordered list of marker id strings
        clans      dict pfam accession without version -> clan id (checkm/util/pfam.py:34-56) or None
        nested     dict pfam accession without version -> set of nested accessions, or None
        """
        self.keys = keys
        n = len(model_acc)
        self.nmodels = n
        self.qlen = np.asarray(model_qlen, dtype=np.int32)
        self.thr_kind = np.asarray([t[0] for t in model_thr], dtype=np.uint8)
        self.thr_full = np.asarray([t[1][0] for t in model_thr], dtype=np.float64)
        self.thr_dom = np.asarray([t[1][1] for t in model_thr], dtype=np.float64)
        self.is_pf = np.asarray([1 if a.startswith('PF') else 0 for a in model_acc], dtype=np.uint8)
        self.key = np.asarray([keys.get(a) for a in model_acc], dtype=np.uint32)
        clans = clans or {}
        nested = nested or {}
        stripped = [a[0:a.rfind('.')] for a in model_acc]      # pfam.py:111-112 (rfind -1 drops the last char, as Python does)
        clan_ids = {}
        self.clan = np.asarray([clan_ids.setdefault(clans[s], len(clan_ids)) if s in clans else -1 for s in stripped], dtype=np.int32)
        by_stripped = {}
        for i, s in enumerate(stripped):
            by_stripped.setdefault(s, []).append(i)
        off, idx = [0], []
        for s in stripped:
            for other in sorted(nested.get(s, ())):
                idx.extend(by_stripped.get(other, ()))
            off.append(len(idx))
        self.nest_off = np.asarray(off, dtype=np.uint32)
        self.nest_idx = np.asarray(idx if idx else [0], dtype=np.uint32)
        set_off, marker_off, marker_key = [0], [0], []
        self.n_markers, self.n_sets = [], []
        for sets in bin_sets:
            nm = 0
            for st in sets:
                for m in st:
                    marker_key.append(keys.get(m))
                nm += len(st)
                marker_off.append(len(marker_key))
            set_off.append(len(marker_off) - 1)
            self.n_markers.append(nm)
            self.n_sets.append(len(sets))
        self.nbins = len(bin_sets)
        self.set_off = np.asarray(set_off, dtype=np.uint32)
        self.marker_off = np.asarray(marker_off, dtype=np.uint32)
        self.marker_key = np.asarray(marker_key if marker_key else [0], dtype=np.uint32)
        self.n_markers = np.asarray(self.n_markers, dtype=np.int64)
        self.n_sets = np.asarray(self.n_sets, dtype=np.int64)

    def with_empty_bins(self, nbins):
        """The same model tables for `nbins` bins without marker sets (the reduction of ResultsParser: sets are counted later)."""
        import copy
        p = copy.copy(self)
        p.nbins = nbins
        p.set_off = np.zeros(nbins + 1, dtype=np.uint32)
        p.marker_off = np.zeros(1, dtype=np.uint32)
        p.marker_key = np.zeros(1, dtype=np.uint32)
        p.n_markers = np.zeros(nbins, dtype=np.int64)
        p.n_sets = np.zeros(nbins, dtype=np.int64)
        return p

    @classmethod
    def for_hmm_models(cls, profiles, bin_models, clans=None, nested=None):
        """HMM_MODELS_SET shape (markerSets.py:265-274): one set holding every accession of the bin's models."""
        keys = KeyTable()
        acc, qlen, thr = [], [], []
        for hd in profiles.headers:
            a = hd["acc"] if hd["acc"] else hd["name"]
            acc.append(a)
            qlen.append(hd["leng"])
            thr.append(resolve_threshold(a, hd["ga"], hd["tc"], hd["nc"]))
        sets = [[sorted(set(acc[m] for m in models))] for models in bin_models]
        return cls(keys, acc, qlen, thr, sets, clans, nested)

    def _structs(self, flags):
        mi = _lib.ModelInfo(self.nmodels, self.qlen.ctypes.data, self.thr_kind.ctypes.data, self.thr_full.ctypes.data,
                            self.thr_dom.ctypes.data, self.is_pf.ctypes.data, self.clan.ctypes.data, self.nest_off.ctypes.data,
                            self.nest_idx.ctypes.data, self.key.ctypes.data)
        ms = _lib.MarkerSetsCSR(self.nbins, self.set_off.ctypes.data, self.marker_off.ctypes.data, self.marker_key.ctypes.data)
        return mi, ms

    def with_threshold_variants(self, thr_lists):
        """The same plan with one threshold table per variant (thr_lists: per variant, per model (kind, (full, dom))): bins pick
        their variant through reduce(bin_variant=...) -- the sticky header view of a bin's own model subset can give a model
        other cutoffs than another subset does."""
        import copy
        p = copy.copy(self)
        p.nvariants = len(thr_lists)
        p.thr_kind = np.ascontiguousarray([[t[0] for t in thr] for thr in thr_lists], dtype=np.uint8).reshape(-1)
        p.thr_full = np.ascontiguousarray([[t[1][0] for t in thr] for thr in thr_lists], dtype=np.float64).reshape(-1)
        p.thr_dom = np.ascontiguousarray([[t[1][1] for t in thr] for thr in thr_lists], dtype=np.float64).reshape(-1)
        return p

    def reduce(self, ctx, hits, seqs, ignore_thresholds=False, evalue=E_VAL, length=LENGTH, skip_pseudogene=False,
               skip_adj=False, individual_markers=False, bin_select=None, ext=None, bin_variant=None):
        """hits: _lib.Hits from a search (then seqs is required), or None with ext = (HitColumns, keepalive)."""
        sel = var = None
        fl = _lib.ReduceFlags(int(ignore_thresholds), int(skip_pseudogene), int(skip_adj), int(individual_markers), float(evalue), float(length), None, 0, None)
        if bin_select is not None:
            sel = np.ascontiguousarray(bin_select, dtype=np.uint8)
            fl.bin_select = sel.ctypes.data
        if bin_variant is not None and getattr(self, "nvariants", 1) > 1:
            var = np.ascontiguousarray(bin_variant, dtype=np.uint32)
            fl.nvariants = self.nvariants
            fl.bin_variant = var.ctypes.data
        mi, ms = self._structs(fl)
        out = C.c_void_p()
        if ext is None:
            _lib._chk(_lib.load().ckm_reduce(ctx.h, hits.h, None, seqs.h, C.byref(mi), C.byref(fl), C.byref(ms), C.byref(out)))
        else:
            _lib._chk(_lib.load().ckm_reduce(ctx.h, None, C.byref(ext[0]), None, C.byref(mi), C.byref(fl), C.byref(ms), C.byref(out)))
        return QAResult(out, self)


def ext_columns(bins_rows, keys_model_index):
    """Build a HitColumns for tables parsed from domtblout text.

    bins_rows: per bin, list of dict rows with the HmmerHitDOM field names (checkm/hmmer.py:255-285).
    keys_model_index: callable(row) -> model index in the plan.
    Returns (HitColumns, keepalive list)."""
    n = sum(len(r) for r in bins_rows)
    off = np.zeros(len(bins_rows) + 1, dtype=np.uint64)
    cols = {f: np.zeros(max(n, 1), dtype=t) for f, t in (
        ("seq", np.uint32), ("model", np.uint32), ("tlen", np.int32), ("qlen", np.int32), ("full_evalue", np.float64),
        ("full_score", np.float32), ("full_bias", np.float32), ("dom_idx", np.int32), ("ndom", np.int32),
        ("c_evalue", np.float64), ("i_evalue", np.float64), ("dom_score", np.float32), ("dom_bias", np.float32),
        ("hmm_from", np.int32), ("hmm_to", np.int32), ("ali_from", np.int32), ("ali_to", np.int32),
        ("env_from", np.int32), ("env_to", np.int32), ("acc", np.float32))}
    fs_d = np.zeros(max(n, 1), dtype=np.float64)      # scores as the float64 the text parses to
    ds_d = np.zeros(max(n, 1), dtype=np.float64)
    names = []
    i = 0
    for b, rows in enumerate(bins_rows):
        for r in rows:
            cols["seq"][i] = i
            cols["model"][i] = keys_model_index(r)
            cols["tlen"][i] = r["target_length"]; cols["qlen"][i] = r["query_length"]
            cols["full_evalue"][i] = r["full_e_value"]; cols["full_score"][i] = r["full_score"]; cols["full_bias"][i] = r["full_bias"]
            cols["dom_idx"][i] = r["dom"]; cols["ndom"][i] = r["ndom"]; cols["c_evalue"][i] = r["c_evalue"]; cols["i_evalue"][i] = r["i_evalue"]
            cols["dom_score"][i] = r["dom_score"]; cols["dom_bias"][i] = r["dom_bias"]
            fs_d[i] = r["full_score"]; ds_d[i] = r["dom_score"]
            cols["hmm_from"][i] = r["hmm_from"]; cols["hmm_to"][i] = r["hmm_to"]; cols["ali_from"][i] = r["ali_from"]; cols["ali_to"][i] = r["ali_to"]
            cols["env_from"][i] = r["env_from"]; cols["env_to"][i] = r["env_to"]; cols["acc"][i] = r["acc"]
            names.append(r["target_name"].encode())
            i += 1
        off[b + 1] = i
    name_arr = (C.c_char_p * max(1, n))(*names)
    hc = _lib.HitColumns()
    hc.n = n
    hc.nbins = len(bins_rows)
    hc.bin_row_off = off.ctypes.data_as(C.POINTER(C.c_uint64))
    for f, a in cols.items():
        setattr(hc, f, a.ctypes.data_as(type(getattr(hc, f))))
    hc.target_name = name_arr
    hc.full_score_d = fs_d.ctypes.data_as(C.POINTER(C.c_double))
    hc.dom_score_d = ds_d.ctypes.data_as(C.POINTER(C.c_double))
    return hc, [off, cols, name_arr, names, fs_d, ds_d]
