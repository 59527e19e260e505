"""ResultsParser / ResultsManager, API-compatible with checkm/resultsParser.py:41-565.

All hit filtering (vetHit, addHit, clan filter, adjacent-ORF merging) runs inside libcheckm_hip
(ckm_reduce); marker-set counting runs in its kernels (ckm_count_sets via MarkerSet.genomeCheck).
When the scan ran in this process its packed hits are reduced directly; otherwise the domtblout
text written earlier is parsed and handed to the same library entry (column form).
"""
from collections import defaultdict
import ast
import logging
import os
import sys
import weakref

import numpy as np

from checkm_amd import qa as cqa
from checkm_amd import runtime
from checkm_amd.common import checkFileExists
from checkm_amd.defaultValues import DefaultValues
from checkm_amd.hmmer import HmmerHitDOM, read_domtblout
from checkm_amd.markerSets import count_sets
from checkm_amd.pfam import PFAM


_PFAM_CACHE = {}


def _pfam_tables():
    """Clan and nesting maps of Pfam-A.hmm.dat, parsed once per file state (the reference re-reads the file for every bin,
    checkm/resultsParser.py:208 -> util/pfam.py:34-56)."""
    path = DefaultValues.PFAM_CLAN_FILE
    if not os.path.exists(path):
        return {}, {}
    st = os.stat(path)
    key = (os.path.abspath(path), st.st_mtime_ns, st.st_size)
    if key not in _PFAM_CACHE:
        _PFAM_CACHE.clear()
        pf = PFAM(path)
        pf.readClansAndNesting()
        _PFAM_CACHE[key] = (pf.clan, pf.nested)
    return _PFAM_CACHE[key]


def _plan_for_models(models_list):
    """models_list: list of HmmModel (one per model slot).  Builds the ckm_model_info part of a QAPlan."""
    clans, nested = _pfam_tables()
    keys = cqa.KeyTable()
    acc = [m.acc for m in models_list]
    qlen = [m.leng for m in models_list]
    thr = [cqa.resolve_threshold(m.acc, m.ga, m.tc, m.nc) for m in models_list]
    return keys, acc, qlen, thr, clans, nested


class ResultsManager(object):
    """Results of a single bin (resultsParser.py:322-565)."""

    def __init__(self, binId, models, bIgnoreThresholds=False, evalueThreshold=DefaultValues.E_VAL,
                 lengthThreshold=DefaultValues.LENGTH, bSkipPseudoGeneCorrection=False, binStats=None):
        self.binId = binId
        self._mh = {}              # markerHits once somebody looked at it (or assigned it)
        self._lazy = None          # (QAResult, local bin, KeyTable, row -> HmmerHitDOM) while nobody has: see the markerHits property
        self.bIgnoreThresholds = bIgnoreThresholds
        self.evalueThreshold = evalueThreshold
        self.lengthThreshold = lengthThreshold
        self.bSkipPseudoGeneCorrection = bSkipPseudoGeneCorrection
        self.models = models
        self.binStats = binStats
        self._raw = []
        self._counts = {}          # bIndividualMarkers -> (markerSet object, [n0..n5+, comp, cont]) from the batched count; only trusted
                                   # while markerHits has never been handed out (nobody can have changed a hit list)
        self.remote = False        # True on rank 0 for a bin another rank scanned: only its gathered QA row is known here
        self.het = None            # strain heterogeneity gathered from the owning rank

    @property
    def markerHits(self):
        """{marker id: [HmmerHitDOM]} as the reference holds it (resultsParser.py:330).  After a batched reduction the dict is built
        the first time somebody asks for it: the QA table (formats 1 and 2) never does -- it is counted from the kept rows' key ids."""
        if self._lazy is not None:
            res, lb, keys, mk = self._lazy
            self._lazy = None
            self._counts = {}
            self._mh = _marker_hits_from(res, lb, keys, mk, lazy=True)
        return self._mh

    @markerHits.setter
    def markerHits(self, value):
        self._lazy = None
        self._counts = {}
        self._mh = value

    def _set_lazy(self, res, lb, keys, row_to_hit):
        self._lazy = (res, lb, keys, row_to_hit)
        self._counts = {}
        _LIVE_LAZY.append(weakref.ref(self))

    def materialize(self):
        return self.markerHits

    def _cached_counts(self, markerSet, bIndividualMarkers):
        hit = self._counts.get(bool(bIndividualMarkers))
        if hit is not None and hit[0] is markerSet and (self._lazy is not None or self.remote):
            return list(hit[1])
        return None

    def vetHit(self, hit):
        """One row against its model's cutoffs (resultsParser.py:340-377); the batched form of this test runs inside ckm_reduce.
        A model that has a cutoff and fails it is rejected outright -- the E-value/length rule is only for models without cutoffs."""
        model = self.models[hit.query_accession]
        span = float(hit.ali_to - hit.ali_from) / float(hit.query_length)          # no +1, as in the reference
        if not self.bSkipPseudoGeneCorrection and span < DefaultValues.PSEUDOGENE_LENGTH:
            return False
        cutoff = None
        if not self.bIgnoreThresholds:
            if model.nc is not None and 'TIGR' in model.acc:
                cutoff = model.nc
            elif model.ga is not None:
                cutoff = model.ga
            elif model.tc is not None:
                cutoff = model.tc
            elif model.nc is not None:
                cutoff = model.nc
        if cutoff is not None:
            return bool(cutoff[0] <= hit.full_score and cutoff[1] <= hit.dom_score)
        return bool(hit.full_e_value <= self.evalueThreshold and span >= self.lengthThreshold)

    # ---- hit intake: rows are collected, filtering happens in the library -------------------------
    def addHit(self, hit):
        self._raw.append(hit)

    def _reduce_raw(self, bSkipAdjCorrection):
        """Run vetHit/addHit/clan filter/adjacent merge for the collected rows (ext form of ckm_reduce)."""
        accs = list(self.models.keys())
        slot = {a: i for i, a in enumerate(accs)}
        keys, acc, qlen, thr, clans, nested = _plan_for_models([self.models[a] for a in accs])
        plan = cqa.QAPlan(keys, acc, qlen, thr, [[]], clans, nested)
        rows = [h.as_dict() for h in self._raw]
        hc = cqa.ext_columns([rows], lambda r: slot[r["query_accession"]])
        res = plan.reduce(runtime.get_ctx(), None, None, self.bIgnoreThresholds, self.evalueThreshold, self.lengthThreshold,
                          self.bSkipPseudoGeneCorrection, bSkipAdjCorrection, False, None, hc)
        self.markerHits = _marker_hits_from(res, 0, keys, lambda r: self._raw[r])
        res.close()

    def identifyAdjacentMarkerGenes(self):
        self._reduce_raw(False)

    # ---- counting ----------------------------------------------------------------------------------
    def countUniqueHits(self):
        uniq = multi = 0
        for hits in self.markerHits.values():
            if len(hits) == 1:
                uniq += 1
            elif len(hits) > 1:
                multi += 1
        return uniq, multi

    def hitsToMarkerGene(self, markerSet):
        ret = {}
        for marker in markerSet.getMarkerGenes():
            try:
                ret[marker] = len(self.markerHits[marker])
            except KeyError:
                ret[marker] = 0
        return ret

    def geneCountsForSelectedMarkerSet(self, binMarkerSets, bIndividualMarkers):
        sel = binMarkerSets.selectedMarkerSet()
        row = self._cached_counts(sel, bIndividualMarkers)
        if row is not None:
            return row
        return self.geneCounts(sel, self.markerHits, bIndividualMarkers)

    def geneCounts(self, markerSet, markerHits, bIndividualMarkers):
        """[n0, n1, n2, n3, n4, n5+, completeness, contamination] (resultsParser.py:513-537); counted on the device.  When
        ResultsParser.batchedGeneCounts has already counted this marker set for all bins in one launch, that row is returned."""
        _pres, _mult, hist, _pt, _mt, _nm, empty_members = count_sets(markerSet.markerSet, markerHits)
        counts = [int(x) for x in hist]
        comp, cont = markerSet.genomeCheck(markerHits, bIndividualMarkers)
        return counts + [comp, cont]

    def geneCopyNumber(self, binMarkerSets):
        out = {'GCN0': [], 'GCN1': [], 'GCN2': [], 'GCN3': [], 'GCN4': [], 'GCN5+': []}
        genes = binMarkerSets.selectedMarkerSet().getMarkerGenes()
        for marker in self.models:
            if marker not in genes:
                continue
            mid = os.path.splitext(marker)[0]
            if marker in self.markerHits:
                n = len(self.markerHits[marker])
                out['GCN5+' if n >= 5 else 'GCN' + str(n)].append(mid)
            else:
                out['GCN0'].append(mid)
        return out

    def getSummary(self, binMarkerSets, bIndividualMarkers, outputFormat=1):
        summary = {}
        if outputFormat in (1, 2):
            sel = binMarkerSets.selectedMarkerSet()
            data = self.geneCountsForSelectedMarkerSet(binMarkerSets, bIndividualMarkers)
            summary['marker lineage'] = sel.lineageStr
            summary['# genomes'] = sel.numGenomes
            summary['# markers'] = sel.numMarkers()
            summary['# marker sets'] = sel.numSets()
            for i, k in enumerate(['0', '1', '2', '3', '4', '5+']):
                summary[k] = data[i]
            summary['Completeness'] = data[6]
            summary['Contamination'] = data[7]
            if outputFormat == 2:
                summary.update(self.binStats)
        elif outputFormat == 5:
            genes = binMarkerSets.selectedMarkerSet().getMarkerGenes()
            for marker, hl in self.markerHits.items():
                if marker in genes:
                    summary[marker] = [h.target_name for h in hl]
        elif outputFormat == 6:
            genes = binMarkerSets.selectedMarkerSet().getMarkerGenes()
            for marker, hl in self.markerHits.items():
                if marker in genes and len(hl) >= 2:
                    summary[marker] = [h.target_name for h in hl]
        elif outputFormat == 7:
            # genes that carry more than one copy of the same marker (resultsParser.py:633-653)
            genes = binMarkerSets.selectedMarkerSet().getMarkerGenes()
            per_gene = defaultdict(dict)
            for marker, hl in self.markerHits.items():
                if marker not in genes:
                    continue
                for h in hl:
                    per_gene[h.target_name][marker] = per_gene[h.target_name].get(marker, 0) + 1
            for gene, counts in per_gene.items():
                for marker, n in counts.items():
                    if n > 1:
                        summary.setdefault(gene, {})[marker] = n
        elif outputFormat == 8:
            genes = binMarkerSets.selectedMarkerSet().getMarkerGenes()
            per_gene = {}
            for marker, hl in self.markerHits.items():
                if marker not in genes:
                    continue
                for h in hl:
                    per_gene.setdefault(h.target_name, []).append(h)
            for gene, hl in per_gene.items():
                summary[gene] = {}
                for h in hl:
                    summary[gene].setdefault(h.query_accession, []).append([h.ali_from, h.ali_to])
        else:
            print("Unknown output format: ", outputFormat)
        return summary

    def printSummary(self, outputFormat, aai, binMarkerSets, bIndividualMarkers, coverageBinProfiles=None, table=None, anaFolder=None):
        """One bin in output formats 1-9 (resultsParser.py:678-966); returns the number of reported rows for formats 6 and 7, else 0."""
        het = aai.aaiMeanBinHetero.get(self.binId, 0.0) if aai is not None else 0.0
        if self.het is not None:
            het = self.het
        if outputFormat in (1, 2):
            sel = binMarkerSets.selectedMarkerSet()
            lineage = sel.lineageStr
            if sel.UID != '0':
                lineage += ' (' + str(sel.UID) + ')'
            data = self.geneCountsForSelectedMarkerSet(binMarkerSets, bIndividualMarkers)
        if outputFormat == 1:
            if table is None:
                print("%s\t%s\t%d\t%d\t%d\t%s\t%0.2f\t%0.2f\t%0.2f" % (self.binId, lineage, sel.numGenomes, sel.numMarkers(), sel.numSets(),
                                                                           "\t".join(str(data[i]) for i in range(6)), data[6], data[7], het))
            else:
                table.add_row([self.binId, lineage, sel.numGenomes, sel.numMarkers(), sel.numSets()] + data + [het])
        elif outputFormat == 2:
            bs = self.binStats
            if table is None:
                row = self.binId
                row += '\t%s\t%d\t%d\t%d' % (lineage, sel.numGenomes, sel.numMarkers(), sel.numSets())
                row += '\t%0.2f\t%0.2f\t%0.2f' % (data[6], data[7], het)
                row += '\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d' % (bs['Genome size'], bs['# ambiguous bases'], bs['# scaffolds'], bs['# contigs'],
                                                                     bs['N50 (scaffolds)'], bs['N50 (contigs)'], bs['Mean scaffold length'],
                                                                     bs['Mean contig length'], bs['Longest scaffold'], bs['Longest contig'])
                row += '\t%.1f\t%.2f' % (bs['GC'] * 100, bs['GC std'] * 100)
                row += '\t%.2f\t%d\t%d' % (bs['Coding density'] * 100, bs['Translation table'], bs['# predicted genes'])
                row += '\t' + '\t'.join(str(data[i]) for i in range(6))
                if coverageBinProfiles:
                    if self.binId in coverageBinProfiles:
                        for _, cov in coverageBinProfiles[self.binId].items():
                            row += '\t%.2f\t%.2f' % (cov[0], cov[1])
                    else:
                        for _bam in coverageBinProfiles[list(coverageBinProfiles.keys())[0]]:
                            row += '\t%.2f\t%.2f' % (0, 0)
                print(row)
            else:
                row = [self.binId, lineage, sel.numGenomes, sel.numMarkers(), sel.numSets(), data[6], data[7], het]
                row.extend([bs['Genome size'], bs['# ambiguous bases'], bs['# scaffolds'], bs['# contigs'], bs['N50 (scaffolds)'], bs['N50 (contigs)'],
                            int(bs['Mean scaffold length']), int(bs['Mean contig length']), bs['Longest scaffold'], bs['Longest contig']])
                row.extend([bs['GC'] * 100, bs['GC std'] * 100, bs['Coding density'] * 100, bs['Translation table'], bs['# predicted genes']])
                row.extend(data[0:6])
                if coverageBinProfiles:
                    if self.binId in coverageBinProfiles:
                        for _, cov in coverageBinProfiles[self.binId].items():
                            row.extend(cov)
                    else:
                        for _bam in coverageBinProfiles[list(coverageBinProfiles.keys())[0]]:
                            row.extend([0, 0])
                table.add_row(row)
        elif outputFormat == 3:
            for ms in binMarkerSets.markerSetIter():
                data = self.geneCounts(ms, self.markerHits, bIndividualMarkers)
                if table is None:
                    print("%s\t%s\t%s\t%d\t%d\t%d\t%s\t%0.2f\t%0.2f\t%0.2f" % (self.binId, ms.UID, ms.lineageStr, ms.numGenomes, ms.numMarkers(), ms.numSets(),
                                                                                   "\t".join(str(data[i]) for i in range(6)), data[6], data[7], het))
                else:
                    table.add_row([self.binId, ms.UID, ms.lineageStr, ms.numGenomes, ms.numMarkers(), ms.numSets()] + data + [het])
        elif outputFormat == 4:
            sel = binMarkerSets.selectedMarkerSet()
            data = self.hitsToMarkerGene(sel)
            row = "Node Id: %s; Marker lineage: %s" % (sel.UID, sel.lineageStr)
            for marker in data:
                row += '\t' + marker
            print(row)
            row = self.binId
            for count in data.values():
                row += '\t' + str(count)
            print(row)
            print()
        elif outputFormat == 5:
            genes = binMarkerSets.selectedMarkerSet().getMarkerGenes()
            for marker, hit_list in self.markerHits.items():
                if marker not in genes:
                    continue
                for hit in hit_list:
                    print(self.binId, marker, hit.target_name, sep='\t', end='\n')
        elif outputFormat == 6:
            genes = binMarkerSets.selectedMarkerSet().getMarkerGenes()
            reported = 0
            for marker, hit_list in self.markerHits.items():
                if marker not in genes:
                    continue
                if len(hit_list) >= 2:
                    print(self.binId, marker, sep='\t', end='\t')
                    print(','.join(sorted(h.target_name for h in hit_list)), end='\n')
                    reported += 1
            return reported
        elif outputFormat == 7:
            genes = binMarkerSets.selectedMarkerSet().getMarkerGenes()
            reported = 0
            for marker, hit_list in self.markerHits.items():
                if marker not in genes:
                    continue
                if len(hit_list) >= 2:
                    same = set()
                    for i in range(len(hit_list)):
                        scaffold = hit_list[i].target_name[0:hit_list[i].target_name.rfind('_')]
                        for j in range(i + 1, len(hit_list)):
                            if scaffold == hit_list[j].target_name[0:hit_list[j].target_name.rfind('_')]:
                                same.add(hit_list[i].target_name)
                                same.add(hit_list[j].target_name)
                    if len(same) >= 2:
                        print(self.binId, marker, sep='\t', end='\t')
                        print(','.join(sorted(list(same))), end='\n')
                        reported += 1
            return reported
        elif outputFormat == 8:
            genes = binMarkerSets.selectedMarkerSet().getMarkerGenes()
            per_gene = {}
            for marker, hit_list in self.markerHits.items():
                if marker not in genes:
                    continue
                for hit in hit_list:
                    per_gene[hit.target_name] = per_gene.get(hit.target_name, []) + [hit]
            for gene, hits in per_gene.items():
                row = self.binId + '\t' + gene
                for hit in hits:
                    row += '\t' + hit.query_accession + ',' + str(hit.ali_from) + ',' + str(hit.ali_to)
                print(row)
        elif outputFormat == 9:
            if anaFolder is None:
                raise ValueError("AnaFolder must not be None for outputFormat 9")
            genes = binMarkerSets.selectedMarkerSet().getMarkerGenes()
            info = {}
            for marker, hit_list in self.markerHits.items():
                if marker not in genes:
                    continue
                for hit in hit_list:
                    info[hit.target_name] = {"marker": marker, "ali_from": str(hit.ali_from), "ali_to": str(hit.ali_to)}
            seqs = _read_fasta_full_headers("/".join([anaFolder, "bins", self.binId, "genes.faa"]))
            kept = [h for h in seqs.keys() if h.split(" # ")[0] in info]

            def by_contig_and_gene(header):
                ctg, num = header.split(" # ")[0].rsplit("_", 1)
                return ctg, int(num)
            for header in sorted(kept, key=by_contig_and_gene):
                elems = header.split(" # ")
                gene = elems[0]
                contig, num = gene.rsplit("_", 1)
                start, end, strand = elems[1], elems[2], elems[3]
                if table is not None:       # (sic) the reference prints FASTA when a table object is given and a tab row when it is not
                    gene_info = "geneId={};start={};end={};strand={};protlen={}".format(num, start, end, strand, str(len(seqs[header])))
                    marker_info = "marker={};mstart={};mend={}".format(info[gene]["marker"], info[gene]["ali_from"], info[gene]["ali_to"])
                    print(">" + " ".join([self.binId, contig, gene_info, marker_info]), seqs[header], sep="\n")
                else:
                    print("\t".join([self.binId, contig, num, start, end, strand, str(len(seqs[header])), info[gene]["marker"], info[gene]["ali_from"],
                                     info[gene]["ali_to"], seqs[header]]))
        else:
            logging.getLogger('timestamp').error("Unknown output format: %d", outputFormat)
        return 0


def _read_fasta_full_headers(path):
    """header line (without '>') -> residues; every sequence line loses its last character, i.e. the newline
    (checkm/util/seqUtils.py:180-211 with trimHeader=False)."""
    seqs, cur = {}, None
    with open(path) as f:
        for line in f:
            if not line.strip():
                continue
            if line[0] == '>':
                cur = line[1:].rstrip()
                seqs[cur] = []
            else:
                seqs[cur].append(line[0:-1])
    return {k: ''.join(v) for k, v in seqs.items()}


def _kept_hit(res, i, row_to_hit):
    """HmmerHitDOM of kept row i of a QAResult (the merged coordinates of an adjacent pair come from the library)."""
    base = row_to_hit(int(res.kept_row[i]))
    h = HmmerHitDOM.from_fields(**base.as_dict())
    if int(res.kept_row2[i]) != 0xFFFFFFFFFFFFFFFF:
        other = row_to_hit(int(res.kept_row2[i]))
        h.target_name = DefaultValues.SEQ_CONCAT_CHAR.join(sorted([base.target_name, other.target_name]))
    h.target_length = int(res.kept_tlen[i])
    h.hmm_from, h.hmm_to = int(res.kept_hmm_from[i]), int(res.kept_hmm_to[i])
    h.ali_from, h.ali_to = int(res.kept_ali_from[i]), int(res.kept_ali_to[i])
    h.env_from, h.env_to = int(res.kept_env_from[i]), int(res.kept_env_to[i])
    return h


class LazyHits(object):
    """The hit list of one marker, as ResultsManager.markerHits holds it: a sequence whose HmmerHitDOM objects are built the first time
    one of them is looked at.  The QA table (formats 1 and 2), geneCounts, countUniqueHits ... only ask for len(); a thousand bins carry
    ~10^6 kept hits, and building a Python object for each of them used to cost as much as the scan."""

    __slots__ = ("_res", "_idx", "_mk", "_items", "__weakref__")

    def __init__(self, res, row_to_hit):
        self._res, self._idx, self._mk, self._items = res, [], row_to_hit, None

    def materialize(self):
        if self._items is None:
            self._items = [_kept_hit(self._res, i, self._mk) for i in self._idx]
            self._res = self._mk = None
        return self._items

    def __len__(self):
        return len(self._idx) if self._items is None else len(self._items)

    def __iter__(self):
        return iter(self.materialize())

    def __getitem__(self, k):
        return self.materialize()[k]

    def __setitem__(self, k, v):
        self.materialize()[k] = v

    def __delitem__(self, k):
        del self.materialize()[k]

    def append(self, h):
        self.materialize().append(h)

    def __add__(self, other):
        return self.materialize() + list(other)

    def __eq__(self, other):
        return self.materialize() == (other.materialize() if isinstance(other, LazyHits) else other)

    def __repr__(self):
        return repr(self.materialize())

    def __contains__(self, h):
        return h in self.materialize()

    def __getattr__(self, name):          # sort, extend, insert, pop, index, count, remove, reverse ...: the list's own
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)


_LIVE_LAZY = []          # weak references to lazy lists whose rows still live in a resident scan


def materialize_lazy_hits():
    """Called before a resident scan is released: lists that were never looked at take their hits now."""
    live = list(_LIVE_LAZY)
    del _LIVE_LAZY[:]
    for ref in live:                       # (a ResultsManager's dict first: that registers its lists, which are filled in next)
        lz = ref()
        if lz is not None:
            lz.materialize()
    live = list(_LIVE_LAZY)
    del _LIVE_LAZY[:]
    for ref in live:
        lz = ref()
        if lz is not None:
            lz.materialize()


def _marker_hits_from(res, b, keys, row_to_hit, lazy=False):
    """Rebuild ResultsManager.markerHits ({acc: [HmmerHitDOM]}) of bin b from the library's kept rows.
    A defaultdict, as PFAM.filterHitsFromSameClan returns one (pfam.py:94)."""
    mh = defaultdict(list)
    if len(_LIVE_LAZY) > 4096:
        _LIVE_LAZY[:] = [r for r in _LIVE_LAZY if r() is not None]
    for i in range(int(res.kept_bin_off[b]), int(res.kept_bin_off[b + 1])):
        k = keys.names[int(res.kept_key[i])]
        if lazy:
            lst = mh.get(k)
            if lst is None:
                lst = mh[k] = LazyHits(res, row_to_hit)
                _LIVE_LAZY.append(weakref.ref(lst))
            lst._idx.append(i)
        else:
            mh[k].append(_kept_hit(res, i, row_to_hit))
    return mh


class _Table(object):
    """Minimal frame-ruled table for the non-tab output mode."""

    def __init__(self, header):
        self.header = header
        self.rows = []

    def add_row(self, row):
        self.rows.append(row)

    def render(self, sort_col=None, reverse=False):
        rows = list(self.rows)
        if sort_col is not None:
            i = self.header.index(sort_col)
            rows.sort(key=lambda r: r[i], reverse=reverse)
        txt = [[("%.2f" % c) if isinstance(c, float) else str(c) for c in r] for r in rows]
        w = [max([len(self.header[i])] + [len(r[i]) for r in txt]) for i in range(len(self.header))]
        rule = '-' * (sum(w) + 2 * len(w) + len(w) - 1)
        out = [rule, '  '.join((self.header[i].ljust(w[i]) if i == 0 else self.header[i].center(w[i])) for i in range(len(w))), rule]
        for r in txt:
            out.append('  '.join((r[i].ljust(w[i]) if i == 0 else r[i].center(w[i])) for i in range(len(w))))
        out.append(rule)
        return '\n'.join(out)


class ResultsParser(object):
    """Parse the scan output for every bin and derive QA statistics (resultsParser.py:41-319)."""

    def __init__(self, binIdToModels):
        self.logger = logging.getLogger('timestamp')
        self.results = {}
        self.models = binIdToModels
        self._binStatsFile = DefaultValues.BIN_STATS_OUT
        self._pool = None

    def analyseResults(self, outDir, binStatsFile, hmmTableFile, bIgnoreThresholds=False, evalueThreshold=DefaultValues.E_VAL,
                       lengthThreshold=DefaultValues.LENGTH, bSkipPseudoGeneCorrection=False, bSkipAdjCorrection=False):
        binStats = self.parseBinStats(outDir, binStatsFile)
        self._binStatsFile = binStatsFile
        self.parseBinHits(outDir, hmmTableFile, bSkipAdjCorrection, bIgnoreThresholds, evalueThreshold, lengthThreshold,
                          bSkipPseudoGeneCorrection, binStats)
        return binStats

    def parseBinStats(self, resultsFolder, binStatsFile):
        path = os.path.join(resultsFolder, 'storage', binStatsFile)
        checkFileExists(path)
        stats = {}
        with open(path) as f:
            for line in f:
                p = line.split('\t')
                stats[p[0]] = ast.literal_eval(p[1])
        return stats

    def _parseCache(self, resultsFolder, name):
        path = os.path.join(resultsFolder, 'storage', name)
        checkFileExists(path)
        table = {}
        with open(path) as f:
            for line in f:
                p = line.split('\t')
                table[p[0]] = ast.literal_eval(p[1])
        return table

    def parseBinStatsExt(self, resultsFolder):
        """storage/bin_stats_ext.tsv back into {binId: dict} (resultsParser.py:161-174; read by the plots at main.py:685-687)."""
        return self._parseCache(resultsFolder, DefaultValues.BIN_STATS_EXT_OUT)

    def parseMarkerGeneStats(self, resultsFolder):
        """storage/marker_gene_stats.tsv back into {binId: {gene: {marker: [[start, end], ...]}}} (resultsParser.py:176-189)."""
        return self._parseCache(resultsFolder, DefaultValues.MARKER_GENE_STATS)

    def parseBinHits(self, outDir, hmmTableFile, bSkipAdjCorrection=False, bIgnoreThresholds=False, evalueThreshold=DefaultValues.E_VAL,
                     lengthThreshold=DefaultValues.LENGTH, bSkipPseudoGeneCorrection=False, binStats=None):
        if not self.models:
            self.logger.error('Models must be parsed before identifying HMM hits.')
            sys.exit(1)
        self.logger.info('Parsing HMM hits to marker genes:')
        from checkm_amd.markerGeneFinder import SCAN_CACHE
        binIds = list(self.models.keys())
        ent = SCAN_CACHE.get((os.path.abspath(outDir), hmmTableFile))
        models = self.models           # (not `self` in the closure: the parser keeps it, and a cycle would keep a thousand bins' results alive until the collector runs)
        mk = lambda b: ResultsManager(b, models[b], bIgnoreThresholds, evalueThreshold, lengthThreshold, bSkipPseudoGeneCorrection,
                                      binStats[b] if (binStats is not None and b in binStats) else None)
        self._mk, self._text_args = mk, (outDir, hmmTableFile, bSkipAdjCorrection, bIgnoreThresholds, evalueThreshold, lengthThreshold, bSkipPseudoGeneCorrection)
        self.remote_bins = []
        self._pool = None
        if ent is not None and ent.get("pool") is not None:
            # find() fanned out over worker processes (checkm_amd/workers.py): they reduce the bins they scanned; this process learns the
            # QA rows in printSummary (one gather among the workers) and reads the workers' tables for anything that needs the hits
            from checkm_amd import workers
            self._pool = (ent["pool"], outDir, hmmTableFile)
            try:
                ent["pool"].call("analyse", dict(outDir=outDir, binStatsFile=self._binStatsFile, hmmTableFile=hmmTableFile, bIgnoreThresholds=bIgnoreThresholds,
                                                 evalueThreshold=evalueThreshold, lengthThreshold=lengthThreshold,
                                                 bSkipPseudoGeneCorrection=bSkipPseudoGeneCorrection, bSkipAdjCorrection=bSkipAdjCorrection))
            except workers.WorkerError as e:
                self.logger.error('reduction of the marker-gene hits failed: %s' % e)
                sys.exit(1)
            self.remote_bins = list(binIds)
            self.logger.info('    Finished parsing hits for %d of %d (100.00%%) bins.' % (len(binIds), len(binIds)))
            return
        if ent is not None and ent.get("world", 1) > 1:
            # one process per GPU: this rank reduces the bins it scanned; the QA rows of the others arrive in printSummary's gather
            self.remote_bins = [b for b in binIds if b not in ent["owned"]]
            binIds = [b for b in binIds if b in ent["owned"]]
        resident = [b for b in binIds if ent is not None and b in ent["where"]]
        rest = [b for b in binIds if ent is None or b not in ent["where"]]
        if resident:
            self._reduce_resident(ent, resident, mk, bSkipAdjCorrection, bIgnoreThresholds, evalueThreshold, lengthThreshold, bSkipPseudoGeneCorrection)
        if rest:
            self._reduce_text(outDir, hmmTableFile, rest, mk, bSkipAdjCorrection, bIgnoreThresholds, evalueThreshold, lengthThreshold,
                              bSkipPseudoGeneCorrection)
        self._order_results()
        self.logger.info('    Finished parsing hits for %d of %d (100.00%%) bins.' % (len(binIds), len(binIds)))

    # the packed hits of a scan run in this process: no text round-trip
    def _reduce_resident(self, ent, binIds, mk, skip_adj, ignore, evalue, length, skip_pseudo):
        profiles = ent["profiles"]
        slot_acc = [hd["acc"] if hd["acc"] else hd["name"] for hd in profiles.headers]
        # Bins whose (possibly sticky) header views agree on every model they share are reduced in one library call: a view is
        # {acc: (acc, leng, ga, tc, nc)}; a bin joins the first group it does not contradict (lineage_wf: every bin has its own
        # model subset, but the thresholds of a model rarely depend on the subset).
        groups = []          # [merged view, members, last dict object seen]
        group_of = {}        # id(model dict) -> its group: find() hands the SAME dict to every bin with the same model subset
        for b in binIds:
            mb = self.models[b]
            g = group_of.get(id(mb))
            if g is not None:
                g[1].append(b)
                continue
            placed = False
            view = {a: (m.acc, m.leng, m.ga, m.tc, m.nc) for a, m in mb.items()}
            for g in groups:
                merged = g[0]
                if all(merged.get(a, v) == v for a, v in view.items()):
                    merged.update(view); g[1].append(b); g[2] = mb; placed = True
                    group_of[id(mb)] = g
                    break
            if not placed:
                groups.append([view, [b], mb])
                group_of[id(mb)] = groups[-1]

        # one plan per group: the model slots of the whole database, a slot's cutoffs from the group's (sticky) view where the group knows the
        # model, none elsewhere.  Built from two shared base lists with only the known slots written over (a lineage_wf run has ~120 groups of
        # ~600 known models among 2000 slots: a Python object per slot and group took 0.1 s of every pass), cutoff cascades resolved once
        # per distinct (acc, ga, tc, nc).
        clans, nested = _pfam_tables()
        base_acc, base_len = list(slot_acc), [hd["leng"] for hd in profiles.headers]
        none_thr = cqa.resolve_threshold("", None, None, None)
        slots_of = {}
        for k, a in enumerate(slot_acc):
            slots_of.setdefault(a, []).append(k)
        thr_cache = {}
        plans = []
        for merged, members, _last in groups:
            acc_l, len_l, thr_l = list(base_acc), list(base_len), [none_thr] * len(base_acc)
            for a, v in merged.items():
                t = thr_cache.get(v)
                if t is None:
                    t = thr_cache[v] = cqa.resolve_threshold(v[0], v[2], v[3], v[4])
                for k in slots_of.get(a, ()):
                    acc_l[k], len_l[k], thr_l[k] = v[0], v[1], t
            plans.append(((cqa.KeyTable(), acc_l, len_l, thr_l, clans, nested), members))
        # Groups that differ in nothing but cutoffs (the sticky header view of another model subset) become threshold VARIANTS of
        # one plan: every part of the scan is then reduced in ONE library call, each bin under its own variant.
        (keys, acc, qlen, thr, clans, nested), _m = plans[0]
        if all(pl[0][1] == acc and pl[0][2] == qlen for pl in plans):
            plan0 = cqa.QAPlan(keys, acc, qlen, thr, [], clans, nested)
            if len(plans) > 1:
                plan0 = plan0.with_threshold_variants([pl[0][3] for pl in plans])
            jobs = [(plan0, keys, [(b, v) for v, pl in enumerate(plans) for b in pl[1]])]
        else:
            jobs = [(cqa.QAPlan(k, a, q, t, [], c, n), k, [(b, 0) for b in members]) for (k, a, q, t, c, n), members in plans]
        for plan0, keys, members in jobs:
            by_part = {}
            for b, v in members:
                pi, lb = ent["where"][b]
                by_part.setdefault(pi, []).append((b, lb, v))
            for pi, lst in by_part.items():
                part = ent["parts"][pi]
                hits, seqs = part["hits"], part["seqs"]
                nb = hits.nbins
                plan = plan0.with_empty_bins(nb)
                sel = np.zeros(nb, dtype=np.uint8)
                var = np.zeros(nb, dtype=np.uint32)
                for _b, lb, v in lst:
                    sel[lb] = 1; var[lb] = v
                res = plan.reduce(runtime.get_ctx(), hits, seqs, ignore, evalue, length, skip_pseudo, skip_adj, False, sel, None, var)
                to_hit = lambda r, h=hits, q=seqs: _hit_from_columns(h, q, profiles, r)
                for b, lb, _v in lst:
                    rm = mk(b)
                    rm._set_lazy(res, lb, keys, to_hit)
                    self.results[b] = rm
                res.close()                  # (the columns were copied out: QAResult keeps numpy arrays)

    # tables written by an earlier command: the library parses the text of all bins at once (ckm_tables_read) and reduces bins
    # that share their model view in one call; rows become HmmerHitDOM objects only for the hits that are kept
    def _reduce_text(self, outDir, hmmTableFile, binIds, mk, skip_adj, ignore, evalue, length, skip_pseudo):
        from checkm_amd import _lib
        paths = [os.path.join(outDir, 'bins', b, hmmTableFile) for b in binIds]
        tables = _lib.Tables(paths)
        try:
            for i, b in enumerate(binIds):
                if tables.missing[i]:
                    sys.stderr.write("[Errno 2] No such file or directory: '%s'\n" % paths[i])
            groups = {}
            for i, b in enumerate(binIds):
                sig = tuple((a, m.acc, m.leng, m.ga, m.tc, m.nc) for a, m in self.models[b].items())
                groups.setdefault(sig, []).append(i)
            nb = len(binIds)
            for sig, members in groups.items():
                first = self.models[binIds[members[0]]]
                accs = list(first.keys())
                keys, acc, qlen, thr, clans, nested = _plan_for_models([first[a] for a in accs])
                plan = cqa.QAPlan(keys, acc, qlen, thr, [[] for _ in range(nb)], clans, nested)
                tables.assign_models(accs)            # a row whose accession is not a model of the bin is an error, as the reference's KeyError is
                sel = np.zeros(nb, dtype=np.uint8)
                sel[members] = 1
                res = plan.reduce(runtime.get_ctx(), None, None, ignore, evalue, length, skip_pseudo, skip_adj, False, sel, tables.ext())
                for i in members:
                    rm = mk(binIds[i])
                    rm.markerHits = _marker_hits_from(res, i, keys, lambda r: HmmerHitDOM.from_fields(**tables.hit(r)))
                    self.results[binIds[i]] = rm
                res.close()
        finally:
            tables.close()

    def parseHmmerResults(self, fileName, resultsManager, bSkipAdjCorrection):
        try:
            for h in read_domtblout(fileName):
                resultsManager.addHit(h)
            resultsManager._reduce_raw(bSkipAdjCorrection)
        except IOError as detail:
            sys.stderr.write(str(detail) + "\n")

    # ---- output ------------------------------------------------------------------------------------
    def _getHeader(self, outputFormat, binMarkerSets=None, coverageBinProfiles=None, table=None):
        """Column names of every output format (resultsParser.py:219-273)."""
        if outputFormat == 1:
            return ['Bin Id', 'Marker lineage', '# genomes', '# markers', '# marker sets', '0', '1', '2', '3', '4', '5+',
                    'Completeness', 'Contamination', 'Strain heterogeneity']
        if outputFormat == 2:
            header = ['Bin Id', 'Marker lineage', '# genomes', '# markers', '# marker sets', 'Completeness', 'Contamination',
                      'Strain heterogeneity', 'Genome size (bp)', '# ambiguous bases', '# scaffolds', '# contigs', 'N50 (scaffolds)',
                      'N50 (contigs)', 'Mean scaffold length (bp)', 'Mean contig length (bp)', 'Longest scaffold (bp)',
                      'Longest contig (bp)', 'GC', 'GC std (scaffolds > 1kbp)', 'Coding density', 'Translation table',
                      '# predicted genes', '0', '1', '2', '3', '4', '5+']
            if coverageBinProfiles is not None:
                for bamId in coverageBinProfiles[list(coverageBinProfiles.keys())[0]]:
                    header += ['Coverage (' + bamId + ')', 'Coverage std (' + bamId + ')']
            return header
        if outputFormat == 3:
            return ['Bin Id', 'Node Id', 'Marker lineage', '# genomes', '# markers', '# marker sets', '0', '1', '2', '3', '4', '5+',
                    'Completeness', 'Contamination', 'Strain heterogeneity']
        if outputFormat == 4:
            return None
        if outputFormat == 5:
            return ['Bin Id', 'Marker Id', 'Gene Id']
        if outputFormat in (6, 7):
            return ['Bin Id', 'Marker Id', 'Gene Ids']
        if outputFormat == 8:
            return ['Bin Id', 'Gene Id', '{Marker Id, Start position, End position}']
        if outputFormat == 9:
            if table is not None:
                return ['Bin Id', 'Contig', 'Gene Number', 'Gene Start', 'Gene End', 'Gene Strand', 'Prot Length', 'Marker Id', 'Align Start', 'Align End',
                        'Sequence']
            return " "
        if outputFormat == 10:
            return ['Scaffold Id', 'Bin Id', 'Length', '# contigs', 'GC', '# ORFs', 'Coding density', 'Marker Ids']

    def batchedGeneCounts(self, binIdToBinMarkerSets, bIndividualMarkers, binIds=None):
        """geneCounts of the SELECTED marker set of every bin in ONE ckm_count_sets launch (the reference counts bin by bin in Python,
        resultsParser.py:513-537 + markerSets.py:206-238); the float64 division is finished here in the reference's accumulation
        order.  The rows are left with each ResultsManager, whose geneCounts() returns them.  Returns {binId: [n0..n5+, comp, cont]}.
        A bin whose hit dict nobody has looked at yet is counted from the key ids of its kept rows (numpy gathers over the marker
        set's flattened form, MarkerSet.flat); a bin whose dict exists is counted from the dict, whatever its owner did to it."""
        import ctypes as C
        from checkm_amd import _lib
        bins = [b for b in (binIds if binIds is not None else sorted(self.results)) if not self.results[b].remote]
        if not bins:
            return {}
        c_parts, f_parts, m_parts, len_parts, nset, nmark, sets_of = [], [], [], [], [], [], []
        for b in bins:
            ms = binIdToBinMarkerSets[b].selectedMarkerSet()
            rm = self.results[b]
            sets_of.append(ms)
            if rm._lazy is not None:
                res, lb, keys, _mk = rm._lazy
                ids, first, lens = ms.flat(keys)
                o0, o1 = int(res.kept_bin_off[lb]), int(res.kept_bin_off[lb + 1])
                kc = np.bincount(res.kept_key[o0:o1], minlength=len(keys.names)) if o1 > o0 else np.zeros(len(keys.names), dtype=np.int64)
                cnt = kc[ids].astype(np.int32)
                c_parts.append(cnt); m_parts.append((cnt > 0).astype(np.int32)); f_parts.append(first); len_parts.append(lens)
            else:
                hits = rm.markerHits
                counts, member, first, seen, lens = [], [], [], set(), []
                for st in ms.markerSet:
                    for m in st:
                        present = m in hits
                        counts.append(len(hits[m]) if present else 0)
                        member.append(1 if present else 0)          # (a key with an empty list counts as present: quirk Q11)
                        first.append(0 if m in seen else 1)
                        seen.add(m)
                    lens.append(len(st))
                c_parts.append(np.asarray(counts, dtype=np.int32)); m_parts.append(np.asarray(member, dtype=np.int32))
                f_parts.append(np.asarray(first, dtype=np.uint8)); len_parts.append(np.asarray(lens, dtype=np.int64))
            nset.append(len(len_parts[-1])); nmark.append(len(c_parts[-1]))
        nb = len(bins)
        cat = lambda parts, dt: np.ascontiguousarray(np.concatenate(parts + [np.zeros(1, dtype=dt)]).astype(dt, copy=False))
        c, fa, mem, lens_all = cat(c_parts, np.int32), cat(f_parts, np.uint8), cat(m_parts, np.int32), cat(len_parts, np.int64)
        so = np.zeros(nb + 1, dtype=np.uint32); np.cumsum(nset, out=so[1:])
        nsets = int(so[-1])
        mo = np.zeros(nsets + 1, dtype=np.uint32); np.cumsum(lens_all[:nsets], out=mo[1:])
        mk = np.arange(max(1, int(mo[-1])), dtype=np.uint32)
        csr = _lib.MarkerSetsCSR(nb, so.ctypes.data, mo.ctypes.data, mk.ctypes.data)
        pres = np.zeros(max(1, nsets), dtype=np.int32); mult = np.zeros(max(1, nsets), dtype=np.int32)
        hist = np.zeros(nb * 6, dtype=np.int32); pt = np.zeros(nb, dtype=np.int32); mt = np.zeros(nb, dtype=np.int32)
        _lib._chk(_lib.load().ckm_count_sets(runtime.get_ctx().h, C.byref(csr), c.ctypes.data, fa.ctypes.data, pres.ctypes.data, mult.ctypes.data,
                                             hist.ctypes.data, pt.ctypes.data, mt.ctypes.data))
        out = {}
        flen = lens_all.astype(np.float64)
        for k, b in enumerate(bins):
            ms = sets_of[k]
            s0, s1 = int(so[k]), int(so[k + 1])
            m0, m1 = int(mo[s0]), int(mo[s1])
            if bIndividualMarkers:
                fk = fa[m0:m1].astype(bool)
                n_member = int(mem[m0:m1][fk].sum())
                empty = int(((mem[m0:m1] == 1) & (c[m0:m1] == 0) & fk).sum())
                nmk = int(mo[s1] - mo[s0])
                comp, cont = 100 * float(n_member) / nmk, 100 * float(int(mt[k]) - empty) / nmk
            elif s1 > s0:
                if (lens_all[s0:s1] == 0).any():
                    raise ZeroDivisionError("float division by zero")        # an empty collocated set: present / len(ms) in the reference's loop (markerSets.py:219-236)
                # comp += present/len(set) over the sets IN ORDER, in float64 (markerSets.py:219-236): cumsum adds left to right
                comp = float(np.cumsum(pres[s0:s1] / flen[s0:s1])[-1])
                cont = float(np.cumsum(mult[s0:s1] / flen[s0:s1])[-1])
                comp, cont = 100 * comp / (s1 - s0), 100 * cont / (s1 - s0)
            else:
                comp, cont = 100 * 0.0 / len(ms.markerSet), 100 * 0.0 / len(ms.markerSet)       # (ZeroDivisionError, as the reference raises)
            row = [int(x) for x in hist[k * 6:k * 6 + 6]] + [comp, cont]
            self.results[b]._counts[bool(bIndividualMarkers)] = (ms, row)
            out[b] = row
        return out

    def _gather_rows(self, aai, binIdToBinMarkerSets, bIndividualMarkers, order):
        """The QA rows of the bins THIS process reduced, packed to fixed width, exchanged by ONE all_gather (checkm_amd/dist.py);
        returns the table of all ranks, sorted by bin index into `order`."""
        from checkm_amd import dist as cdist
        index = {b: i for i, b in enumerate(order)}
        own = [b for b in sorted(self.results) if not self.results[b].remote]
        rows = self.batchedGeneCounts(binIdToBinMarkerSets, bIndividualMarkers, own)
        packed = cdist.pack_qa_rows([index[b] for b in own], [binIdToBinMarkerSets[b].selectedMarkerSet().numMarkers() for b in own],
                                    [binIdToBinMarkerSets[b].selectedMarkerSet().numSets() for b in own],
                                    [rows[b][0:6] for b in own] if own else np.zeros((0, 6)), [rows[b][6] for b in own], [rows[b][7] for b in own])
        for k, b in enumerate(own):
            packed[k, 11] = aai.aaiMeanBinHetero.get(b, 0.0) if aai is not None else 0.0
        return cdist.gather_qa_rows(packed, len(order), cdist.collective_device())

    def _order_results(self):
        """self.results in the key order of self.models, as the reference fills it (resultsParser.py:191-217): cacheResults and the
        per-bin output formats iterate the dict."""
        self.results = {b: self.results[b] for b in list(self.models) + [b for b in self.results if b not in self.models] if b in self.results}

    def _adopt_rows(self, table, order, binIdToBinMarkerSets, bIndividualMarkers):
        """Rows of bins reduced elsewhere become ResultsManagers that only know their row (remote = True)."""
        for r in table:
            b = order[int(r[0])]
            if b in self.results and not self.results[b].remote:
                continue
            rm = self._mk(b) if hasattr(self, "_mk") else ResultsManager(b, self.models[b])
            rm.remote = True
            ms = binIdToBinMarkerSets[b].selectedMarkerSet()
            rm._counts[bool(bIndividualMarkers)] = (ms, [int(x) for x in r[3:9]] + [float(r[9]), float(r[10])])
            rm.het = float(r[11])
            self.results[b] = rm
        self._order_results()

    def _gather_remote_rows(self, aai, binIdToBinMarkerSets, bIndividualMarkers):
        """One process per GPU: ONE all_gather of fixed-width QA rows brings the rows of the bins the other ranks scanned."""
        order = sorted(self.models.keys())
        self._adopt_rows(self._gather_rows(aai, binIdToBinMarkerSets, bIndividualMarkers, order), order, binIdToBinMarkerSets, bIndividualMarkers)

    def _rows_from_workers(self, aai, binIdToBinMarkerSets, bIndividualMarkers):
        """find() ran on worker processes: each reduces and counts its bins, the workers exchange the rows (one all_gather) and worker
        0 hands the table over."""
        from checkm_amd import workers
        from checkm_amd.markerGeneFinder import SCAN_CACHE
        pool, outDir, tbl = self._pool
        order = sorted(self.models.keys())
        ent = SCAN_CACHE.get((os.path.abspath(outDir), tbl))
        if ent is None or not pool.conns:
            # the scan was released (release_scan(outDir)) or the workers are gone: read the tables they wrote, as a later `checkm qa` would
            self._localize_remote()
            return
        owners = ent["owners"]
        per = [dict(sets={}) for _ in pool.devs]
        for b in order:
            per[owners[b]]["sets"][b] = binIdToBinMarkerSets[b].selectedMarkerSet()
        het = dict(aai.aaiMeanBinHetero) if aai is not None else {}
        try:
            replies = pool.call("summary", dict(outDir=outDir, hmmTableFile=tbl, bIndividualMarkers=bIndividualMarkers, order=order, het=het), per)
        except workers.WorkerError as e:
            self.logger.error('gathering the QA rows failed: %s' % e)
            sys.exit(1)
        self._adopt_rows(replies[0], order, binIdToBinMarkerSets, bIndividualMarkers)

    def _localize_remote(self):
        """Bins whose hits live in another process (remote rows, or not reduced yet): read the tables their owners wrote."""
        remote = [b for b in self.models if b not in self.results or self.results[b].remote]
        if remote and hasattr(self, "_text_args"):
            outDir, tbl, skip_adj, ignore, ev, ln, skip_ps = self._text_args
            for b in remote:
                self.results.pop(b, None)
            self._reduce_text(outDir, tbl, remote, self._mk, skip_adj, ignore, ev, ln, skip_ps)
            self.remote_bins = []
            self._order_results()

    def printSummary(self, outputFormat, aai, binIdToBinMarkerSets, bIndividualMarkers, coverageFile, bTabTable, outFile, anaFolder):
        """The QA table in any of the output formats (resultsParser.py:275-319).  Tab mode is byte-compatible with the reference; the framed
        table of the non-tab mode (formats 1, 2, 3, 9) is drawn by _Table (prettytable is not a dependency here).
        One process per GPU: formats 1 and 2 are completed by the single gather of QA rows and printed by rank 0; for the other
        formats rank 0 reads the tables the other ranks wrote."""
        if coverageFile:
            self.logger.error('Coverage profiles are not part of this path.')
            sys.exit(1)
        from checkm_amd import dist as cdist
        if self._pool is not None:
            if outputFormat in (1, 2):
                self._rows_from_workers(aai, binIdToBinMarkerSets, bIndividualMarkers)
            else:
                self._localize_remote()
        elif cdist.world_size() > 1:
            if outputFormat in (1, 2):
                self._gather_remote_rows(aai, binIdToBinMarkerSets, bIndividualMarkers)
            elif cdist.env_rank()[0] == 0 and getattr(self, "remote_bins", None):
                outDir, tbl, skip_adj, ignore, ev, ln, skip_ps = self._text_args
                self._reduce_text(outDir, tbl, self.remote_bins, self._mk, skip_adj, ignore, ev, ln, skip_ps)
                self.remote_bins = []
            if cdist.env_rank()[0] != 0:
                return
        if outputFormat in (1, 2):
            self.batchedGeneCounts(binIdToBinMarkerSets, bIndividualMarkers)
        old = sys.stdout
        if outFile:
            sys.stdout = open(outFile, 'w')
        try:
            header = self._getHeader(outputFormat, binIdToBinMarkerSets[list(binIdToBinMarkerSets.keys())[0]], None, bTabTable)
            table = None
            if bTabTable or outputFormat not in (1, 2, 3, 9):
                bTabTable = True
                if header is not None:
                    print('\t'.join(header))
            else:
                table = _Table(header)
            reported = 0
            for binId in sorted(self.results.keys()):
                reported += self.results[binId].printSummary(outputFormat, aai, binIdToBinMarkerSets[binId], bIndividualMarkers, None, table, anaFolder)
            if outputFormat in (6, 7) and reported == 0:
                print('[No marker genes satisfied the reporting criteria.]')
            if not bTabTable:
                if outputFormat in (1, 2):
                    print(table.render('Completeness', True))
                elif table.rows:
                    print(table.render())
        finally:
            if outFile:
                sys.stdout.close()
            sys.stdout = old

    def cacheResults(self, outDir, binIdToBinMarkerSets, bIndividualMarkers):
        """storage/bin_stats_ext.tsv and storage/marker_gene_stats.tsv (resultsParser.py:121-143)."""
        if self._pool is not None:
            self._localize_remote()            # the copy numbers and the per-gene table need the hits themselves
        with open(os.path.join(outDir, 'storage', DefaultValues.BIN_STATS_EXT_OUT), 'w') as fout:
            for binId in self.results:
                ext = self.results[binId].getSummary(binIdToBinMarkerSets[binId], bIndividualMarkers, outputFormat=2)
                ext.update(self.results[binId].geneCopyNumber(binIdToBinMarkerSets[binId]))
                fout.write(binId + '\t' + str(ext) + '\n')
        with open(os.path.join(outDir, 'storage', DefaultValues.MARKER_GENE_STATS), 'w') as fout:
            for binId in self.results:
                fout.write(binId + '\t' + str(self.results[binId].getSummary(binIdToBinMarkerSets[binId], bIndividualMarkers, outputFormat=8)) + '\n')


def _hit_from_columns(hits, seqs, profiles, r):
    """HmmerHitDOM of packed row r, with the values the domtblout TEXT carries (%9.2g / %6.1f / %5.1f / %4.2f)."""
    hd = profiles.headers[int(hits.model[r])]
    s = int(hits.seq[r])
    g2 = lambda v: float("%9.2g" % v)
    f1 = lambda v: float("%.1f" % v)
    return HmmerHitDOM.from_fields(
        target_name=seqs.names[s], target_accession='-', target_length=int(hits.tlen[r]), query_name=hd["name"],
        query_accession=hd["acc"] if hd["acc"] else hd["name"], query_length=int(hits.qlen[r]),
        full_e_value=g2(hits.full_evalue[r]), full_score=f1(hits.full_score[r]), full_bias=f1(hits.full_bias[r]),
        dom=int(hits.dom_idx[r]), ndom=int(hits.ndom[r]), c_evalue=g2(hits.c_evalue[r]), i_evalue=g2(hits.i_evalue[r]),
        dom_score=f1(hits.dom_score[r]), dom_bias=f1(hits.dom_bias[r]), hmm_from=int(hits.hmm_from[r]), hmm_to=int(hits.hmm_to[r]),
        ali_from=int(hits.ali_from[r]), ali_to=int(hits.ali_to[r]), env_from=int(hits.env_from[r]), env_to=int(hits.env_to[r]),
        acc=float("%.2f" % hits.acc[r]), target_description=seqs.descs[s] if seqs.descs[s] else '-')
