"""HmmerAligner (the multi-copy-marker part), API-compatible with checkm/hmmerAligner.py:36-44,125-205,275-352,428-451.

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

    def makeAlignmentsOfMultipleHits(self, outDir, markerFile, hmmTableFile, binIdToModels, binIdToBinMarkerSets,
                                     bIgnoreThresholds, evalueThreshold, lengthThreshold, alignOutputDir):
        """Align markers with multiple hits within a bin (hmmerAligner.py:125-173)."""
        makeSurePathExists(alignOutputDir)
        resultsParser = ResultsParser(binIdToModels)
        resultsParser.parseBinHits(outDir, hmmTableFile, False, bIgnoreThresholds, evalueThreshold, lengthThreshold)
        self.logger.info('Aligning marker genes with multiple hits in a single bin:')
        jobs = []                               # (binId, markerId, [(seqId, residues)])
        for binId in binIdToModels:
            multi = self._extractMarkersWithMultipleHits(outDir, binId, resultsParser, binIdToBinMarkerSets[binId])
            for markerId, perBin in multi.items():
                jobs.append((binId, markerId, list(perBin[binId].items())))
        self._align_and_mask(markerFile, jobs, alignOutputDir)
        n = len(binIdToModels)
        if n and self.logger.getEffectiveLevel() <= logging.INFO:
            sys.stderr.write('    Finished processing %d of %d (%.2f%%) bins.\n' % (n, n, 100.0))

    def _align_and_mask(self, hmmModelFile, jobs, alignOutputDir):
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
            for j, (binId, markerId, seqs) in enumerate(jobs):
                if markerId not in slot:
                    self.logger.error('Model %s not found in %s.' % (markerId, hmmModelFile))
                    sys.exit(1)
                for seqId, residues in seqs:
                    recs.append((binId + DefaultValues.SEQ_CONCAT_CHAR + seqId, '', residues))
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
            masked[owner[r]].append((recs[r][0], ''.join(text[i - 1] if i > 0 else '-' for i in path)))
        for j, (binId, markerId, _seqs) in enumerate(jobs):
            binDir = os.path.join(alignOutputDir, binId)
            makeSurePathExists(binDir)
            with open(os.path.join(binDir, markerId + '.masked.faa'), 'w') as fout:
                for seqId, seq in masked[j]:
                    fout.write('>' + seqId + '\n')
                    fout.write(seq + '\n')

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
