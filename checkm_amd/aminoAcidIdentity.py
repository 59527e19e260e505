"""AminoAcidIdentity, API-compatible with checkm/aminoAcidIdentity.py:30-161: amino-acid identity between the copies of a
multi-copy marker (aligned to its model and masked to the match columns by HmmerAligner) and the strain-heterogeneity summary
`checkm qa --aai_strain` prints.  Plain host arithmetic; the alignment itself is libcheckm_hip's (ckm_align)."""
from collections import defaultdict
import logging
import os
import sys

from checkm_amd.common import getBinIdsFromOutDir
from checkm_amd.defaultValues import DefaultValues


def _read_masked(path):
    """id -> sequence, ids cut at the first whitespace, blank lines skipped (checkm/util/seqUtils.py:180-211)."""
    seqs, cur = {}, None
    with open(path) as f:
        for line in f:
            if not line.strip():
                continue
            if line[0] == '>':
                cur = line[1:].split(None, 1)[0]
                seqs[cur] = []
            else:
                seqs[cur].append(line.rstrip('\n'))
    return {k: ''.join(v) for k, v in seqs.items()}


class AminoAcidIdentity(object):
    def __init__(self):
        self.logger = logging.getLogger('timestamp')
        self.aaiRawScores = defaultdict(dict)
        self.aaiHetero = defaultdict(dict)
        self.aaiMeanBinHetero = {}

    def run(self, aaiStrainThreshold, outDir, alignmentOutputFile):
        """AAI between all pairs of copies of every multi-copy marker of every bin (aminoAcidIdentity.py:39-98)."""
        self.logger.info('Calculating AAI between multi-copy marker genes.')
        fout = open(alignmentOutputFile, 'w') if alignmentOutputFile else None
        aaiOutputDir = os.path.join(outDir, 'storage', 'aai_qa')
        for binId in getBinIdsFromOutDir(outDir):
            binPath = os.path.join(aaiOutputDir, binId)
            if not os.path.exists(binPath):
                continue
            for f in os.listdir(binPath):
                if not f.endswith('.masked.faa'):
                    continue
                markerId = f[0:f.find('.')]          # cut at the FIRST dot, as the reference does (PF00318.15 -> PF00318)
                seqs = _read_masked(os.path.join(binPath, f))
                ids = list(seqs.keys())
                for i in range(len(ids)):
                    binIdI = ids[i][0:ids[i].find(DefaultValues.SEQ_CONCAT_CHAR)]
                    for j in range(i + 1, len(ids)):
                        binIdJ = ids[j][0:ids[j].find(DefaultValues.SEQ_CONCAT_CHAR)]
                        if binIdI != binIdJ:
                            self.logger.error('Bin ids do not match.')
                            sys.exit(1)
                        score = self.aai(seqs[ids[i]], seqs[ids[j]])
                        if fout:
                            fout.write(binId + ',' + markerId + '\n')
                            fout.write(ids[i] + '\t' + seqs[ids[i]] + '\n')
                            fout.write(ids[j] + '\t' + seqs[ids[j]] + '\n')
                            fout.write('AAI: %.3f\n' % score)
                            fout.write('\n')
                        if binIdI not in self.aaiRawScores:
                            self.aaiRawScores[binIdI] = defaultdict(list)
                        self.aaiRawScores[binIdI][markerId].append(score)
        if fout:
            fout.close()
        self.aaiHetero, self.aaiMeanBinHetero = self.strainHetero(self.aaiRawScores, aaiStrainThreshold)

    def strainHetero(self, aaiScores, aaiStrainThreshold):
        """Per marker: fraction of copy pairs above the threshold; per bin: percentage over all its pairs (:100-124)."""
        aaiHetero = defaultdict(dict)
        aaiMeanBinHetero = {}
        for binId, markers in aaiScores.items():
            strainCount = multiCopyPairs = 0
            aaiHetero[binId] = {}
            for markerId, scores in markers.items():
                local = 0
                for sc in scores:
                    multiCopyPairs += 1
                    if sc > aaiStrainThreshold:
                        strainCount += 1
                        local += 1
                aaiHetero[binId][markerId] = float(local) / len(scores)
            aaiMeanBinHetero[binId] = 100 * float(strainCount) / multiCopyPairs
        return aaiHetero, aaiMeanBinHetero

    def aai(self, seq1, seq2):
        """Identity over the columns between the leading and trailing gap runs (:126-161).  The trailing scan stops at index 1 and
        never looks at index 0, as the reference's does."""
        assert len(seq1) == len(seq2)
        n = len(seq1)
        start = 0
        for i in range(n):
            if seq1[i] == '-' or seq2[i] == '-':
                start = i + 1
            else:
                break
        end = n
        for i in range(n - 1, 0, -1):
            if seq1[i] == '-' or seq2[i] == '-':
                end = i
            else:
                break
        mismatches = seqLen = 0
        for i in range(start, end):
            if seq1[i] != seq2[i]:
                mismatches += 1
                seqLen += 1
            elif seq1[i] == '-' and seq2[i] == '-':
                pass
            else:
                seqLen += 1
        if seqLen == 0:
            return 0.0
        return 1.0 - (float(mismatches) / seqLen)
