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
        report = open(alignmentOutputFile, 'w') if alignmentOutputFile else None
        root = os.path.join(outDir, 'storage', 'aai_qa')
        sep = DefaultValues.SEQ_CONCAT_CHAR
        for binId in getBinIdsFromOutDir(outDir):
            folder = os.path.join(root, binId)
            if not os.path.isdir(folder):
                continue
            for name in os.listdir(folder):
                if not name.endswith('.masked.faa'):
                    continue
                marker = name[:name.find('.')]                 # cut at the FIRST dot, as the reference does (PF00318.15 -> PF00318)
                copies = list(_read_masked(os.path.join(folder, name)).items())
                owners = [cid[:cid.find(sep)] for cid, _ in copies]
                for a in range(len(copies)):
                    for b in range(a + 1, len(copies)):
                        if owners[a] != owners[b]:
                            self.logger.error('Bin ids do not match.')
                            sys.exit(1)
                        (ida, sa), (idb, sb) = copies[a], copies[b]
                        score = self.aai(sa, sb)
                        if report:
                            report.write('%s,%s\n%s\t%s\n%s\t%s\nAAI: %.3f\n\n' % (binId, marker, ida, sa, idb, sb, score))
                        if owners[a] not in self.aaiRawScores:       # (a plain defaultdict(dict) on first touch would do; kept as the reference builds it)
                            self.aaiRawScores[owners[a]] = defaultdict(list)
                        self.aaiRawScores[owners[a]][marker].append(score)
        if report:
            report.close()
        self.aaiHetero, self.aaiMeanBinHetero = self.strainHetero(self.aaiRawScores, aaiStrainThreshold)

    def strainHetero(self, aaiScores, aaiStrainThreshold):
        """Per marker: fraction of copy pairs above the threshold; per bin: percentage over all its pairs (:100-124)."""
        per_marker = defaultdict(dict)
        per_bin = {}
        for binId, markers in aaiScores.items():
            above_all = pairs_all = 0
            per_marker[binId] = {}
            for marker, scores in markers.items():
                above = sum(1 for sc in scores if sc > aaiStrainThreshold)
                per_marker[binId][marker] = float(above) / len(scores)
                above_all += above
                pairs_all += len(scores)
            per_bin[binId] = 100 * float(above_all) / pairs_all
        return per_marker, per_bin

    def aai(self, seq1, seq2):
        """Identity over the columns between the leading and trailing gap runs (:126-161): columns where both rows are gaps do not
        count, a gap against a residue is a mismatch.  The trailing scan never looks at column 0, as the reference's does not."""
        assert len(seq1) == len(seq2)
        gapped = [x == '-' or y == '-' for x, y in zip(seq1, seq2)]
        first = 0
        while first < len(gapped) and gapped[first]:
            first += 1
        last = len(gapped)
        while last > 1 and gapped[last - 1]:
            last -= 1
        compared = differing = 0
        for x, y in zip(seq1[first:last], seq2[first:last]):
            if x == '-' and y == '-':
                continue
            compared += 1
            differing += x != y
        return 1.0 - float(differing) / compared if compared else 0.0
