"""Marker-set containers and marker-file parsing, API-compatible with checkm/markerSets.py.

MarkerSet.genomeCheck (markerSets.py:206-238) is computed by the set-counting kernel
(ckm_count_sets); the float64 division is finished here in the reference's accumulation order.
The per-bin temporary HMM file of the reference (hmmfetch -f, markerSets.py:326-343,443-476) is
replaced by an index list into ONE resident profile database, in database order (deterministic,
unlike the reference's set-iteration order).
"""
import ast
import functools
import ctypes as C
import gzip
import logging
import threading
from collections.abc import MutableMapping
import os
import pickle
import sys

import numpy as np

from checkm_amd.defaultValues import DefaultValues
from checkm_amd.hmmerModelParser import HmmModelParser, models_dict, read_headers
from checkm_amd.pfam import PFAM


class BinMarkerSets(object):
    """One or more marker sets associated with a bin (markerSets.py:39-154)."""

    TAXONOMIC_MARKER_SET = 1
    TREE_MARKER_SET = 2
    HMM_MODELS_SET = 3

    def __init__(self, binId, markerSetType):
        self.logger = logging.getLogger('timestamp')
        self.markerSets = []
        self.binId = binId
        self.markerSetType = markerSetType
        self.selectedLinageSpecificMarkerSet = None

    def numMarkerSets(self):
        return len(self.markerSets)

    def addMarkerSet(self, markerSet):
        self.markerSets.append(markerSet)

    def markerSetIter(self):
        return iter(self.markerSets)

    def getMarkerGenes(self):
        genes = set()
        for ms in self.markerSets:
            genes |= ms.getMarkerGenes()
        return genes

    def mostSpecificMarkerSet(self):
        return self.markerSets[0]

    def treeMarkerSet(self):
        pass        # a stub in the reference too (markerSets.py:78-79)

    def selectedMarkerSet(self):
        if self.markerSetType == self.TAXONOMIC_MARKER_SET:
            return self.mostSpecificMarkerSet()
        if self.markerSetType == self.TREE_MARKER_SET:
            return self.selectedLinageSpecificMarkerSet
        if len(self.markerSets) == 1:
            return self.markerSets[0]
        self.logger.error('Expect a single marker set to be associated with each bin.')
        sys.exit(1)

    def setLineageSpecificSelectedMarkerSet(self, selectedMarkerSetMap):
        """Follow the selected-set map until a set this bin actually carries is reached (markerSets.py:95-121)."""
        wanted = selectedMarkerSetMap[self.mostSpecificMarkerSet().UID]
        self.selectedLinageSpecificMarkerSet = None
        while self.selectedLinageSpecificMarkerSet is None:
            for ms in self.markerSets:
                if ms.UID == wanted:
                    self.selectedLinageSpecificMarkerSet = ms
                    break
            else:
                wanted = selectedMarkerSetMap[wanted]
        if self.selectedLinageSpecificMarkerSet is None:
            self.logger.error('Failed to set a selected lineage-specific marker set.')
            sys.exit(1)

    def removeMarkers(self, markersToRemove):
        for ms in self.markerSets:
            ms.removeMarkers(markersToRemove)

    def write(self, fout):
        fout.write(self.binId + '\t' + str(len(self.markerSets)))
        for ms in self.markerSets:
            fout.write('\t' + str(ms))
        fout.write('\n')

    def read(self, line):
        f = line.split('\t')
        for i in range(int(f[1])):
            uid, lineage, ngen, sets = f[4 * i + 2], f[4 * i + 3], int(f[4 * i + 4]), _parse_marker_sets(f[4 * i + 5].strip())
            self.markerSets.append(MarkerSet(uid, lineage, ngen, None, sets))     # sets: the cached parsed literal; Python sets on demand


@functools.lru_cache(maxsize=8192)
def _parse_marker_sets(text):
    """The marker-set literal of one lineage node (`[set([...]), ...]`, kilobytes long), parsed once per distinct string: the bins of a
    run share a few dozen lineage nodes, and a 1000-bin Lineage marker file holds 3000 of these literals -- 4.4 of the 4.8 s that
    reading it took went into compiling the same strings again (the reference evals every one, markerSets.py:151).  Returns tuples: the
    callers build their own sets."""
    return tuple(tuple(s) for s in ast.literal_eval(text))


def count_sets(list_of_sets, hits):
    """Device counting for ONE marker set: returns (set_present, set_multi, hist6, present_total, multi_total)."""
    from checkm_amd import _lib, runtime
    markers, marker_off, first = [], [0], []
    seen = set()
    for s in list_of_sets:
        for m in s:
            markers.append(m)
            first.append(0 if m in seen else 1)
            seen.add(m)
        marker_off.append(len(markers))
    counts = np.array([len(hits[m]) if m in hits else 0 for m in markers] or [0], dtype=np.int32)
    # --individual_markers counts dict membership, even with an empty list
    member = np.array([1 if m in hits else 0 for m in markers] or [0], dtype=np.int32)
    nsets = len(list_of_sets)
    set_off = np.array([0, nsets], dtype=np.uint32)
    moff = np.array(marker_off, dtype=np.uint32)
    mkey = np.arange(max(1, len(markers)), dtype=np.uint32)
    firsta = np.array(first or [0], dtype=np.uint8)
    ms = _lib.MarkerSetsCSR(1, set_off.ctypes.data, moff.ctypes.data, mkey.ctypes.data)
    pres = np.zeros(max(1, nsets), dtype=np.int32); mult = np.zeros(max(1, nsets), dtype=np.int32)
    hist = np.zeros(6, dtype=np.int32); pt = np.zeros(1, dtype=np.int32); mt = np.zeros(1, dtype=np.int32)
    _lib._chk(_lib.load().ckm_count_sets(runtime.get_ctx().h, C.byref(ms), counts.ctypes.data, firsta.ctypes.data, pres.ctypes.data,
                                         mult.ctypes.data, hist.ctypes.data, pt.ctypes.data, mt.ctypes.data))
    n_member = int((member * firsta[:len(member)]).sum())
    empty_members = int(((member == 1) & (counts == 0) & (firsta[:len(member)] == 1)).sum())
    return pres[:nsets], mult[:nsets], hist, int(pt[0]), int(mt[0]), n_member, empty_members


_SRC_INFO = {}       # id(parsed literal) -> (literal, n markers, frozenset of marker genes): the literals live in _parse_marker_sets' cache


def _src_info(src):
    ent = _SRC_INFO.get(id(src))
    if ent is None or ent[0] is not src:
        if len(_SRC_INFO) > 16384:
            _SRC_INFO.clear()
        genes = set()
        for st in src:
            genes.update(st)
        ent = _SRC_INFO[id(src)] = (src, sum(len(st) for st in src), frozenset(genes))
    return ent


class MarkerSet(object):
    """Marker genes organised into collocated sets (markerSets.py:157-238).

    Read from a marker file, the list of Python sets (`markerSet`) is built the first time somebody asks for it: a Lineage marker file
    of 1000 bins holds 3000 literals of ~500 sets each, of which the QA table needs sizes and a flattened form only (both come from the
    parsed literal, shared by the bins of a lineage)."""

    def __init__(self, UID, lineageStr, numGenomes, markerSet=None, _src=None):
        self.logger = logging.getLogger('timestamp')
        self.UID = UID
        self.lineageStr = lineageStr
        self.numGenomes = numGenomes
        self._ms = markerSet
        self._src = _src            # the parsed literal (tuple of tuples) while the sets are as the file holds them
        self._flat = None

    @property
    def markerSet(self):
        if self._ms is None:
            self._ms = [set(s) for s in self._src] if self._src is not None else []
        return self._ms

    @markerSet.setter
    def markerSet(self, value):
        self._ms, self._src, self._flat = value, None, None

    def flat(self, keys):
        """(key id of every marker in set order, 1 where a marker occurs for the first time, length of every set) as numpy arrays,
        for the batched counting of ResultsParser.batchedGeneCounts.  `keys`: the qa.KeyTable of the reduction the ids refer to."""
        cache = keys.__dict__.setdefault("_flat_cache", {})
        ent = cache.get(id(self._src)) if self._src is not None else None
        if ent is not None and ent[0] is self._src:
            return ent[1]
        if self._src is None and self._flat is not None and self._flat[0] is keys and self._flat[1] == self._version():
            return self._flat[2]
        ids, first, lens, seen = [], [], [], set()
        for st in (self._src if self._src is not None else self.markerSet):
            for m in st:
                ids.append(keys.get(m))
                first.append(0 if m in seen else 1)
                seen.add(m)
            lens.append(len(st))
        out = (np.asarray(ids, dtype=np.int64), np.asarray(first, dtype=np.uint8), np.asarray(lens, dtype=np.int64))
        if self._src is not None:
            cache[id(self._src)] = (self._src, out)
        else:
            self._flat = (keys, self._version(), out)
        return out

    def _version(self):
        return (id(self._ms), len(self._ms), sum(len(s) for s in self._ms))

    def __repr__(self):
        return str(self.UID) + '\t' + self.lineageStr + '\t' + str(self.numGenomes) + '\t' + str(self.markerSet)

    def size(self):
        if self._src is not None:
            return _src_info(self._src)[1], len(self._src)
        return sum(len(m) for m in self.markerSet), len(self.markerSet)

    def numMarkers(self):
        return self.size()[0]

    def numSets(self):
        return len(self._src) if self._src is not None else len(self.markerSet)

    def getMarkerGenes(self):
        if self._src is not None:
            return set(_src_info(self._src)[2])
        genes = set()
        for m in self.markerSet:
            genes |= set(m)
        return genes

    def removeMarkers(self, markersToRemove):
        if self._src is not None and _src_info(self._src)[2].isdisjoint(markersToRemove):
            return                       # nothing of it is here: the sets stay as the file holds them
        kept = []
        for ms in self.markerSet:
            rest = ms - markersToRemove
            if rest:
                kept.append(rest)
        self.markerSet = kept            # (drops the literal: sizes and the flattened form now come from the sets)

    def genomeCheck(self, hits, bIndividualMarkers):
        """Completeness / contamination; counting on the device, float64 division in the reference's order."""
        pres, mult, _hist, _pt, mt, n_member, empty_members = count_sets(self.markerSet, hits)
        if bIndividualMarkers:
            # `marker in hits` counts a key with an empty list as present and contributes len-1 = -1
            present = n_member
            multi = mt - empty_members
            return 100 * float(present) / self.numMarkers(), 100 * float(multi) / self.numMarkers()
        comp = 0.0
        cont = 0.0
        for i, ms in enumerate(self.markerSet):
            comp += float(int(pres[i])) / len(ms)
            cont += float(int(mult[i])) / len(ms)
        return 100 * comp / len(self.markerSet), 100 * cont / len(self.markerSet)


def wanted_model(name, acc, keys):
    """`hmmfetch -f <keyfile>` (checkm/markerSets.py:326-343 -> checkm/hmmer.py:97-110) takes a record when a key equals its NAME
    or its ACC."""
    return (acc is not None and acc in keys) or (name is not None and name in keys)


class _LazyLineageSets(MutableMapping):
    """{binId: BinMarkerSets} whose values are built from their marker-file line the first time they are looked at.  NOT a dict subclass
    (round 5's was, and dict(d) / d.copy() / d.pop(b) / {**d} read the raw storage and handed out None for bins not yet built): a
    Mapping whose every access path goes through __getitem__; copies and pickles are plain dicts of built values, as the reference
    returns (checkm/markerSets.py:478-511)."""

    def __init__(self, parser, lines):
        self._parser, self._lines, self._built, self._selected, self._exclude = parser, dict(lines), {}, None, None
        self._order = list(lines)                         # the file's order, whatever has been built or assigned since
        self._lock = threading.RLock()

    def _make(self, binId):
        with self._lock:
            if binId in self._built:
                return self._built[binId]
            line = self._lines.pop(binId)
            bms = BinMarkerSets(binId, BinMarkerSets.TREE_MARKER_SET)
            bms.read(line)
            if self._selected is None:
                self._selected = self._parser.parseSelectedMarkerSetMap()     # parsed once, not once per line (markerSets.py:506)
            bms.setLineageSpecificSelectedMarkerSet(self._selected)
            if self._exclude:
                bms.removeMarkers(self._exclude)
            self._built[binId] = bms
            return bms

    def exclude(self, markers):
        """removeMarkers for every bin: applied to the bins already built now, to the others when they are built."""
        with self._lock:
            self._exclude = set(markers)
            for v in self._built.values():
                v.removeMarkers(self._exclude)

    def __getitem__(self, binId):
        with self._lock:
            if binId in self._built:
                return self._built[binId]
            if binId in self._lines:
                return self._make(binId)
        raise KeyError(binId)

    def __setitem__(self, binId, value):
        with self._lock:
            if binId not in self._built and binId not in self._lines:
                self._order.append(binId)
            self._lines.pop(binId, None)
            self._built[binId] = value

    def __delitem__(self, binId):
        with self._lock:
            if binId in self._built:
                del self._built[binId]
                self._lines.pop(binId, None)
            elif binId in self._lines:
                del self._lines[binId]
            else:
                raise KeyError(binId)
            self._order.remove(binId)

    def __contains__(self, binId):
        return binId in self._built or binId in self._lines

    def __iter__(self):
        return iter(list(self._order))

    def __len__(self):
        return len(self._order)

    def copy(self):
        return dict(self.items())

    def __reduce__(self):
        return (dict, (dict(self.items()),))

    def __repr__(self):
        return '_LazyLineageSets(%d bins, %d built)' % (len(self._order), len(self._built))


class MarkerSetParser(object):
    """Marker-file parsing (markerSets.py:241-540)."""

    def __init__(self, threads=1):
        self.logger = logging.getLogger('timestamp')
        self.numThreads = threads

    # ---- marker sets per bin ------------------------------------------------------------------
    def getMarkerSets(self, outDir, binIds, markerFile, excludeMarkersFile=None):
        kind = self.markerFileType(markerFile)
        out = {}
        if kind == BinMarkerSets.TAXONOMIC_MARKER_SET:
            bms = self.parseTaxonomicMarkerSetFile(markerFile)
            for binId in binIds:
                out[binId] = bms
        elif kind == BinMarkerSets.TREE_MARKER_SET:
            out = self.parseLineageMarkerSetFile(markerFile)
        else:
            accs = set(m.acc for m in HmmModelParser(markerFile).parse())
            ms = MarkerSet(0, "N/A", -1, [accs])
            for binId in binIds:
                b = BinMarkerSets(binId, BinMarkerSets.HMM_MODELS_SET)
                b.addMarkerSet(ms)
                out[binId] = b
        exclude = set()
        if excludeMarkersFile:
            exclude = self.readExcludeMarkersFile(excludeMarkersFile)
        exclude.update(DefaultValues.MARKERS_TO_EXCLUDE)
        if isinstance(out, _LazyLineageSets):
            out.exclude(exclude)
            return out
        for b in out.values():
            b.removeMarkers(exclude)
        return out

    def readExcludeMarkersFile(self, excludeMarkersFile):
        out = set()
        with open(excludeMarkersFile) as f:
            for line in f:
                if line[0] == '#':
                    continue
                out.add(line.strip())
        return out

    def markerFileType(self, markerFile):
        with open(markerFile, 'r') as f:
            header = f.readline()
        if DefaultValues.TAXON_MARKER_FILE_HEADER in header:
            return BinMarkerSets.TAXONOMIC_MARKER_SET
        if DefaultValues.LINEAGE_MARKER_FILE_HEADER in header:
            return BinMarkerSets.TREE_MARKER_SET
        if 'HMMER3' in header:
            return BinMarkerSets.HMM_MODELS_SET
        self.logger.error('Unrecognized file type: ' + markerFile)
        sys.exit(1)

    def parseTaxonomicMarkerSetFile(self, markerSetFile):
        with open(markerSetFile) as f:
            f.readline()
            line = f.readline()
        bms = BinMarkerSets(line.split('\t')[0], BinMarkerSets.TAXONOMIC_MARKER_SET)
        bms.read(line)
        return bms

    def parseLineageMarkerSetFile(self, markerSetFile):
        """{binId: BinMarkerSets} of a Lineage marker file (markerSets.py:478-511).  The lines are split off here and PARSED WHEN A BIN
        IS ASKED FOR: one process per GPU reduces an eighth of the bins and has no use for the other lines' sets (rank 0 asks for all
        of them when it prints the table)."""
        lines = {}
        with open(markerSetFile) as f:
            f.readline()
            for line in f:
                lines[line[:line.find('\t')] if '\t' in line else line.split('\t')[0]] = line
        return _LazyLineageSets(self, lines)

    def parseSelectedMarkerSetMap(self):
        m = {}
        with open(DefaultValues.SELECTED_MARKER_SETS) as f:
            for line in f:
                p = line.split('\t')
                m[p[0]] = p[1].rstrip()
        return m

    # ---- which models does a bin need ------------------------------------------------------------
    def hmmDatabaseFor(self, markerFile):
        """The profile database a marker file refers to: itself for raw HMM files, else checkm.hmm."""
        return markerFile if self.markerFileType(markerFile) == BinMarkerSets.HMM_MODELS_SET else DefaultValues.HMM_MODELS

    def markerAccessionsForBins(self, binIds, markerFile):
        """{binId: set(accession)} -- marker genes plus every Pfam family of the clans they span
        (markerSets.py:443-457); None for raw HMM files (= every model)."""
        kind = self.markerFileType(markerFile)
        if kind == BinMarkerSets.HMM_MODELS_SET:
            return {b: None for b in binIds}
        pfam = PFAM(DefaultValues.PFAM_CLAN_FILE)
        cache = {}

        def expand(bms):
            genes = bms.getMarkerGenes()
            key = frozenset(genes)
            if key not in cache:
                cache[key] = genes | pfam.genesInSameClan(genes)
            return cache[key]
        if kind == BinMarkerSets.TAXONOMIC_MARKER_SET:
            accs = expand(self.parseTaxonomicMarkerSetFile(markerFile))
            return {b: accs for b in binIds}
        # Lineage file: only the union of the marker genes of a bin's sets is needed here, and bins of one lineage carry the same
        # literals -- key the expansion by the literals themselves instead of building the set objects of every bin
        by_text, out = {}, {}
        want = set(binIds)
        with open(markerFile) as f:
            f.readline()
            for line in f:
                p = line.rstrip('\n').split('\t')
                if p[0] not in want:
                    continue
                key = tuple(p[4 * i + 5].strip() for i in range(int(p[1])))
                accs = by_text.get(key)
                if accs is None:
                    genes = set()
                    for text in key:
                        for s in _parse_marker_sets(text):
                            genes.update(s)
                    gk = frozenset(genes)
                    if gk not in cache:
                        cache[gk] = genes | pfam.genesInSameClan(genes)
                    accs = by_text[key] = cache[gk]
                out[p[0]] = accs
        missing = [b for b in binIds if b not in out]
        if missing:
            per_bin = self.parseLineageMarkerSetFile(markerFile)      # (raises the reference's KeyError for a bin the file does not list)
            for b in missing:
                out[b] = expand(per_bin[b])
        return {b: out[b] for b in binIds}

    def createHmmModels(self, outDir, binIds, markerFile):
        """{binId: {acc: HmmModel}} without launching anything (markerSets.py:299-324)."""
        db = self.hmmDatabaseFor(markerFile)
        headers = read_headers(db)
        wanted = self.markerAccessionsForBins(list(binIds), markerFile)
        out = {}
        for b in binIds:
            sel = [h for h in headers if wanted[b] is None or wanted_model(h['name'], h['acc'], wanted[b])]
            out[b] = models_dict(sel)
        return out

    def createHmmModelFile(self, binId, markerFile):
        """Kept for API compatibility: writes the bin's model subset as a HMMER3 text file (database order)."""
        import tempfile
        import uuid
        db = self.hmmDatabaseFor(markerFile)
        wanted = self.markerAccessionsForBins([binId], markerFile)[binId]
        out = os.path.join(tempfile.gettempdir(), str(uuid.uuid4()))
        with open(db) as fin, open(out, 'w') as fout:
            rec, acc, name = [], None, None
            for line in fin:
                rec.append(line)
                if line.startswith('NAME'):
                    name = line.split(None, 1)[1].strip()
                elif line.startswith('ACC'):
                    acc = line.split(None, 1)[1].strip()
                elif line.startswith('//'):
                    if wanted is None or wanted_model(name, acc, wanted):
                        fout.writelines(rec)
                    rec, acc, name = [], None, None
        return out

    # ---- model-info cache between `analyze` and `qa` ----------------------------------------------
    def writeBinModels(self, binIdToModels, filename):
        self.logger.info('Saving HMM info to file.')
        with gzip.open(filename, 'wb') as output:
            pickle.dump(binIdToModels, output, pickle.HIGHEST_PROTOCOL)

    def loadBinModels(self, filename):
        self.logger.info('Reading HMM info from file.')
        with gzip.open(filename, 'rb') as f:
            return pickle.load(f)
