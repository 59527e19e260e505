"""HmmerAligner, API-compatible with checkm/hmmerAligner.py:36-451 (the three makeAlignment* entry points).

The reference writes the copies of every marker with >= 2 hits in a bin to a FASTA file, runs `hmmalign --outformat Pfam`
against the marker's model (fetched with hmmfetch into a temporary file), and keeps of the Stockholm output only the match
columns ('#=GC RF' x), upper-cased, as <alignOutputDir>/<binId>/<markerId>.masked.faa.  Here all copies of all markers of all
bins are aligned in ONE library call (ckm_align: the optimal-accuracy alignment hmmalign computes per sequence) and the masked
files are written directly; the intermediate .unaligned.faa / .aligned.faa files, which the reference deletes, never exist."""
from collections import defaultdict
import logging
import os
import sys

from checkm_amd import _lib, runtime
from checkm_amd.common import makeSurePathExists, read_fasta
from checkm_amd.defaultValues import DefaultValues
from checkm_amd.resultsParser import ResultsParser


class HmmerAligner(object):
    def __init__(self, threads):
        self.logger = logging.getLogger('timestamp')
        self.totalThreads = threads          # accepted for compatibility: one batched device call replaces the worker processes
        self.outputFormat = 'Pfam'

    def makeAlignmentTopHit(self, outDir, hmmModelFile, hmmTableFile, binIdToModels, bIgnoreThresholds, evalueThreshold, lengthThreshold,
                            bReportHitStats, alignOutputDir, bKeepUnmaskedAlign=False):
        """Align the top hit of every marker in every bin (hmmerAligner.py:44-83); one <marker>.masked.faa per marker."""
        return self._align_across_bins(self._extractMarkerSeqsTopHits, outDir, hmmModelFile, hmmTableFile, binIdToModels, bIgnoreThresholds,
                                       evalueThreshold, lengthThreshold, bReportHitStats, alignOutputDir, bKeepUnmaskedAlign)

    def makeAlignmentToPhyloMarkers(self, outDir, hmmModelFile, hmmTableFile, binIdToModels, bIgnoreThresholds, evalueThreshold, lengthThreshold,
                                    bReportHitStats, alignOutputDir, bKeepUnmaskedAlign=False):
        """Align the unique hits to a set of common marker genes (hmmerAligner.py:85-123)."""
        return self._align_across_bins(self._extractMarkerSeqsUnique, outDir, hmmModelFile, hmmTableFile, binIdToModels, bIgnoreThresholds,
                                       evalueThreshold, lengthThreshold, bReportHitStats, alignOutputDir, bKeepUnmaskedAlign)

    def _align_across_bins(self, extract, outDir, hmmModelFile, hmmTableFile, binIdToModels, bIgnoreThresholds, evalueThreshold, lengthThreshold,
                           bReportHitStats, alignOutputDir, bKeepUnmaskedAlign):
        if bKeepUnmaskedAlign:
            self.logger.error('The unmasked Stockholm alignment is not produced on this path (only the masked match columns are).')
            sys.exit(1)
        self.logger.info("Extracting marker genes to align.")
        resultsParser = ResultsParser(binIdToModels)
        resultsParser.parseBinHits(outDir, hmmTableFile, False, bIgnoreThresholds, evalueThreshold, lengthThreshold)
        markerSeqs, markerStats = extract(outDir, resultsParser)
        firstBin = list(binIdToModels.keys())[0]
        makeSurePathExists(alignOutputDir)
        jobs = []
        for markerId in binIdToModels[firstBin]:          # the reference builds one temporary model file per key of the first bin
            entries = []
            for binId, seqs in markerSeqs.get(markerId, {}).items():
                for seqId, seq in seqs.items():
                    st = markerStats[markerId][binId][seqId]
                    entries.append((binId + DefaultValues.SEQ_CONCAT_CHAR + seqId, '[e-value=%.4g,score=%.1f]' % (st[0], st[1]) if bReportHitStats else None, seq))
            if entries:
                jobs.append((os.path.join(alignOutputDir, markerId + '.masked.faa'), markerId, entries))
        self.logger.info("Aligning %d marker genes:" % len(binIdToModels[firstBin]))
        self._align_and_mask(hmmModelFile, jobs)
        return resultsParser

    def makeAlignmentsOfMultipleHits(self, outDir, markerFile, hmmTableFile, binIdToModels, binIdToBinMarkerSets,
                                     bIgnoreThresholds, evalueThreshold, lengthThreshold, alignOutputDir):
        """Align markers with multiple hits within a bin (hmmerAligner.py:125-173)."""
        makeSurePathExists(alignOutputDir)
        resultsParser = ResultsParser(binIdToModels)
        resultsParser.parseBinHits(outDir, hmmTableFile, False, bIgnoreThresholds, evalueThreshold, lengthThreshold)
        self.logger.info('Aligning marker genes with multiple hits in a single bin:')
        jobs = []                               # (output file, markerId, [(sequence id, header stats or None, residues)])
        for binId in binIdToModels:
            multi = self._extractMarkersWithMultipleHits(outDir, binId, resultsParser, binIdToBinMarkerSets[binId])
            for markerId, perBin in multi.items():
                entries = [(binId + DefaultValues.SEQ_CONCAT_CHAR + seqId, None, seq) for seqId, seq in perBin[binId].items()]
                jobs.append((os.path.join(alignOutputDir, binId, markerId + '.masked.faa'), markerId, entries))
        self._align_and_mask(markerFile, jobs)
        n = len(binIdToModels)
        if n and self.logger.getEffectiveLevel() <= logging.INFO:
            sys.stderr.write('    Finished processing %d of %d (%.2f%%) bins.\n' % (n, n, 100.0))

    def _align_and_mask(self, hmmModelFile, jobs):
        """All sequences of all jobs in ONE ckm_align call; one masked FASTA per job (what _alignMarker + _maskAlignment leave behind,
        hmmerAligner.py:275-352): '>' id [stats], then the residue of every match column or '-'."""
        if not jobs:
            return
        ctx = runtime.get_ctx()
        profiles = _lib.Profiles(ctx, hmmModelFile)
        try:
            slot = {}                           # hmmfetch finds a model by name or by accession
            for i, hd in enumerate(profiles.headers):
                slot.setdefault(hd["name"], i)
                if hd["acc"]:
                    slot.setdefault(hd["acc"], i)
            recs, model, owner = [], [], []
            for j, (_path, markerId, entries) in enumerate(jobs):
                if markerId not in slot:
                    self.logger.error('Model %s not found in %s.' % (markerId, hmmModelFile))
                    sys.exit(1)
                for seqId, _stats, residues in entries:
                    recs.append((seqId, '', residues))
                    model.append(slot[markerId]); owner.append(j)
            seqs = _lib.Seqs(ctx, [recs])
            try:
                paths = _lib.align(ctx, profiles, seqs, model, list(range(len(recs))))
            finally:
                seqs.close()
        finally:
            profiles.close()
        masked = defaultdict(list)
        for r, path in enumerate(paths):
            text = recs[r][2].upper()
            masked[owner[r]].append(''.join(text[i - 1] if i > 0 else '-' for i in path))
        for j, (path, _markerId, entries) in enumerate(jobs):
            makeSurePathExists(os.path.dirname(path))
            with open(path, 'w') as fout:
                for (seqId, stats, _res), seq in zip(entries, masked[j]):
                    fout.write('>%s %s\n' % (seqId, stats) if stats else '>' + seqId + '\n')
                    fout.write(seq + '\n')

    def _extractMarkerSeqsTopHits(self, outDir, resultsParser):
        """Sequence and (e-value, score) of the top hit of every marker in every bin (hmmerAligner.py:354-376).  As in the reference the
        hits are sorted from the HIGHEST e-value down and the first one is taken."""
        markerSeqs, markerStats = defaultdict(dict), defaultdict(dict)
        for binId in resultsParser.results:
            binORFs = {name: res for name, _d, res in read_fasta(os.path.join(outDir, 'bins', binId, DefaultValues.PRODIGAL_AA))}
            for markerId, hits in resultsParser.results[binId].markerHits.items():
                markerSeqs[markerId][binId] = {}
                markerStats[markerId][binId] = {}
                hits.sort(key=lambda x: x.full_e_value, reverse=True)
                top = hits[0]
                markerSeqs[markerId][binId][top.target_name] = self._extractSeq(top.target_name, binORFs)
                markerStats[markerId][binId][top.target_name] = [top.full_e_value, top.full_score]
        return markerSeqs, markerStats

    def _extractMarkerSeqsUnique(self, outDir, resultsParser):
        """Markers with exactly one hit in a bin (hmmerAligner.py:378-405)."""
        markerSeqs, markerStats = defaultdict(dict), defaultdict(dict)
        for binId in resultsParser.results:
            binORFs = {name: res for name, _d, res in read_fasta(os.path.join(outDir, 'bins', binId, DefaultValues.PRODIGAL_AA))}
            for markerId, hits in resultsParser.results[binId].markerHits.items():
                markerSeqs[markerId][binId] = {}
                markerStats[markerId][binId] = {}
                if len(hits) == 1:
                    hit = hits[0]
                    markerSeqs[markerId][binId][hit.target_name] = self._extractSeq(hit.target_name, binORFs)
                    markerStats[markerId][binId][hit.target_name] = [hit.full_e_value, hit.full_score]
        return markerSeqs, markerStats

    def _extractSeq(self, seqId, seqs):
        """Residues of an ORF, or of adjacent ORFs merged by the adjacency correction, without prodigal's final '*' (:407-426)."""
        out = ''
        for sid in seqId.split(DefaultValues.SEQ_CONCAT_CHAR):
            s = seqs[sid]
            if s[-1] == '*':
                s = s[0:-1]
            out += s
        return out

    def _extractMarkersWithMultipleHits(self, outDir, binId, resultsParser, binMarkerSet):
        """{markerId: {binId: {target_name: residues}}} for the markers of the selected set with >= 2 hits (:428-451)."""
        multi = defaultdict(dict)
        aaGeneFile = os.path.join(outDir, 'bins', binId, DefaultValues.PRODIGAL_AA)
        binORFs = {name: res for name, _desc, res in read_fasta(aaGeneFile)}
        markerGenes = binMarkerSet.selectedMarkerSet().getMarkerGenes()
        for markerId, hits in resultsParser.results[binId].markerHits.items():
            if markerId not in markerGenes or len(hits) < 2:
                continue
            hits.sort(key=lambda x: x.full_e_value, reverse=True)      # as the reference: highest e-value first
            multi[markerId][binId] = {}
            for hit in hits:
                multi[markerId][binId][hit.target_name] = self._extractSeq(hit.target_name, binORFs)
        return multi
