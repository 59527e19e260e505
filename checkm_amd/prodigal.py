"""Gene calling in front of the accelerated path: mirror of checkm/prodigal.py.

ProdigalRunner keeps the reference's contract -- `bins/<binId>/genes.faa`, `genes.gff` (+ `genes.fna`), translation table 4 chosen
only when its coding density beats table 11's by more than 0.05 AND exceeds 0.7 (checkm/prodigal.py:131-134), `-p meta` below
100 kb and as the retry when `-p single` fails (:74-78, :96-110) -- but runs the two prodigal processes of a bin side by side
(the reference runs them one after the other: half the wall-clock per bin) and skips table 4 altogether when table 11's density
already rules it out (d11 >= 0.95: d4 <= 1 cannot beat it by more than 0.05).  The choice needs BOTH densities in every other
case, so no further run can be skipped without changing results.

ProdigalGeneFeatureParser (checkm/prodigal.py:208-274) is restated with interval arithmetic instead of one numpy mask per
contig; pinned against the reference's class by tools/gen_prodigal_golden.py -> tests/golden/prodigal_cases.json.

The gene finder itself is the external `prodigal` binary when one is on PATH, and the library's own (checkm_amd/geneFinder.py: training,
scoring and the dynamic program on the device) when none is or when CKM_GENE_CALLER=device.
"""
import logging
import os
import shutil
import signal
import stat
import subprocess
import sys
import tempfile

from checkm_amd.defaultValues import DefaultValues


class ProdigalError(BaseException):
    pass


def _read_fasta_lengths(path):
    """Total bases and per-sequence ids of a (possibly gzipped) FASTA file; ids = header up to the first whitespace
    (checkm/util/seqUtils.py:180-211 with trimHeader=True)."""
    import gzip
    op = gzip.open if path.endswith('.gz') else open
    ids, total, seqs, cur = [], 0, {}, None
    with op(path, 'rt') as f:
        for line in f:
            if not line.strip():
                continue
            if line[0] == '>':
                cur = line[1:].split(None, 1)[0] if line[1:].strip() else ''
                ids.append(cur)
                seqs[cur] = []
            else:
                seqs[cur].append(line[0:-1] if line.endswith('\n') else line)
    out = {}
    for k, v in seqs.items():
        out[k] = ''.join(v)
        total += len(out[k])
    return out, total


class ProdigalRunner(object):
    """Wrapper for running prodigal (checkm/prodigal.py:41-182)."""

    def __init__(self, outDir):
        self.logger = logging.getLogger('timestamp')
        # the gene finder on the device (checkm_amd/geneFinder.py) when asked for (CKM_GENE_CALLER=device) or when no prodigal binary exists
        want = os.environ.get("CKM_GENE_CALLER", "")
        self.use_device = want == "device" or (want != "prodigal" and shutil.which("prodigal") is None)
        if not self.use_device:
            self.checkForProdigal()
        self.outDir = outDir
        self.aaGeneFile = os.path.join(outDir, DefaultValues.PRODIGAL_AA)
        self.ntGeneFile = os.path.join(outDir, DefaultValues.PRODIGAL_NT)
        self.gffFile = os.path.join(outDir, DefaultValues.PRODIGAL_GFF)

    def _cmd(self, procedure, table, prodigal_input, bNucORFs):
        aa, nt, gff = self.aaGeneFile + '.' + str(table), self.ntGeneFile + '.' + str(table), self.gffFile + '.' + str(table)
        if bNucORFs:
            return 'prodigal -p %s -q -m -f gff -g %d -a %s -d %s -i %s > %s 2> /dev/null' % (procedure, table, aa, nt, prodigal_input, gff)
        return 'prodigal -p %s -q -m -f gff -g %d -a %s -i %s > %s 2> /dev/null' % (procedure, table, aa, prodigal_input, gff)

    def _finish_table(self, table, rtn, cmd, procedure):
        """The retry / error path of checkm/prodigal.py:96-110."""
        aa = self.aaGeneFile + '.' + str(table)
        if rtn != 0 or not self._areORFsCalled(aa):
            msg = "Prodigal failed or returned no output (code: %s)." % rtn
            if procedure == 'single':
                self.logger.warning(msg + " Retrying with '-p meta' due to possible high N content in the genome.")
                rtn = os.system(cmd.replace('-p single', '-p meta'))
                if rtn != 0 or not self._areORFsCalled(aa):
                    self.logger.error("Prodigal failed again with '-p meta' (code: %s)." % rtn)
                    sys.exit(rtn if rtn else 1)
            else:
                self.logger.error(msg)
                sys.exit(rtn if rtn else 1)

    def _density(self, table, seqs, totalBases):
        parser = ProdigalGeneFeatureParser(self.gffFile + '.' + str(table))
        coding = 0
        for seqId in seqs:
            coding += parser.codingBases(seqId)
        return float(coding) / totalBases if totalBases != 0 else 0

    def run(self, query, bNucORFs=True):
        if self.use_device:
            from checkm_amd import geneFinder
            try:
                best, density = geneFinder.call_bin_files([(query, self.outDir)], bNucORFs, logger=self.logger)[query]
            except ValueError as e:
                self.logger.error(str(e))
                sys.exit(1)
            self.tableCodingDensity = density
            return best
        prodigal_input = query
        seqs, totalBases = _read_fasta_lengths(prodigal_input)
        tmp_dir = None
        if prodigal_input.endswith('.gz'):
            tmp_dir = tempfile.mkdtemp()
            prodigal_input = os.path.join(tmp_dir, os.path.basename(prodigal_input[0:-3]))
            with open(prodigal_input, 'w') as f:
                for k, v in seqs.items():
                    f.write('>' + k + '\n' + v + '\n')
        procedure = 'meta' if totalBases < 100000 else 'single'
        # table 11 and table 4 side by side; table 4 is only waited for (and only kept) when table 11's density leaves it a chance
        cmds = {t: self._cmd(procedure, t, prodigal_input, bNucORFs) for t in (4, 11)}
        # each in its own session: the command line is a shell pipeline (redirections, as the reference writes it), so stopping table 4
        # early has to reach the prodigal process behind the shell -- the whole process group is signalled
        procs = {t: subprocess.Popen(cmds[t], shell=True, start_new_session=True) for t in (4, 11)}

        def stop(proc):
            if proc.poll() is None:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except (ProcessLookupError, PermissionError):
                    pass
            proc.wait()
        density = {}
        try:
            self._finish_table(11, procs[11].wait() << 8, cmds[11], procedure)          # (os.system's encoding of the exit status, as the reference reports it)
            density[11] = self._density(11, seqs, totalBases)
            if density[11] >= 0.95:
                stop(procs[4])
                density[4] = None
                best = 11
            else:
                self._finish_table(4, procs[4].wait() << 8, cmds[4], procedure)
                density[4] = self._density(4, seqs, totalBases)
                best = 4 if (density[4] - density[11] > 0.05) and density[4] > 0.7 else 11
        finally:
            for t in (4, 11):            # (an error exit above must not leave a gene finder running)
                stop(procs[t])
        shutil.copyfile(self.aaGeneFile + '.' + str(best), self.aaGeneFile)
        shutil.copyfile(self.gffFile + '.' + str(best), self.gffFile)
        if bNucORFs:
            shutil.copyfile(self.ntGeneFile + '.' + str(best), self.ntGeneFile)
        for t in (4, 11):
            for f in (self.aaGeneFile, self.gffFile) + ((self.ntGeneFile,) if bNucORFs else ()):
                if os.path.exists(f + '.' + str(t)):
                    os.remove(f + '.' + str(t))
        if tmp_dir:
            shutil.rmtree(tmp_dir)
        self.tableCodingDensity = density
        return best

    def _areORFsCalled(self, aaGeneFile):
        return os.path.exists(aaGeneFile) and os.stat(aaGeneFile)[stat.ST_SIZE] != 0

    def areORFsCalled(self, bNucORFs):
        f = self.ntGeneFile if bNucORFs else self.aaGeneFile
        return os.path.exists(f) and os.stat(f)[stat.ST_SIZE] != 0

    def checkForProdigal(self):
        try:
            subprocess.call(['prodigal', '-h'], stdout=open(os.devnull, 'w'), stderr=subprocess.STDOUT)
        except Exception:
            self.logger.error("Make sure prodigal is on your system path.")
            sys.exit(1)


class ProdigalFastaParser(object):
    """Gene positions from prodigal's FASTA headers (checkm/prodigal.py:185-205)."""

    def genePositions(self, filename):
        if not os.path.exists(filename):
            logging.getLogger('timestamp').error('Input file does not exists: ' + filename + '\n')
            sys.exit(1)
        gp = {}
        with open(filename) as f:
            for line in f:
                if line[0] == '>':
                    t = line[1:].split()
                    gp[t[0]] = [int(t[2]), int(t[4])]
        return gp


class ProdigalGeneFeatureParser(object):
    """Genes of a prodigal GFF file and the number of coding bases (checkm/prodigal.py:208-274), overlapping genes counted once."""

    def __init__(self, filename):
        if not os.path.exists(filename):
            logging.getLogger('timestamp').error('Input file does not exists: ' + filename + '\n')
            sys.exit(1)
        self.genes = {}
        self.lastCodingBase = {}
        self.translationTable = None
        self._parseGFF(filename)
        self._merged = {s: self._merge(g.values()) for s, g in self.genes.items()}

    def _parseGFF(self, filename):
        geneCounter = 0
        with open(filename) as f:
            for line in f:
                if line.startswith('# Model Data') and not self.translationTable:
                    for token in line.split(';'):
                        if 'transl_table' in token:
                            self.translationTable = int(token[token.find('=') + 1:])
                if line[0] == '#' or line.strip() == '"':
                    continue
                t = line.split('\t')
                seqId = t[0]
                if seqId not in self.genes:
                    geneCounter = 0
                    self.genes[seqId] = {}
                    self.lastCodingBase[seqId] = 0
                geneId = seqId + '_' + str(geneCounter)
                geneCounter += 1
                start, end = int(t[3]), int(t[4])
                self.genes[seqId][geneId] = [start, end]
                self.lastCodingBase[seqId] = max(self.lastCodingBase[seqId], end)

    @staticmethod
    def _merge(intervals):
        """Union of 1-based inclusive [start, end] as sorted disjoint 0-based half-open pieces (the mask's ones)."""
        out = []
        for s, e in sorted((p[0] - 1, p[1]) for p in intervals):
            if e <= s:
                continue                                  # (a reversed interval marks nothing in the reference's mask either)
            if out and s <= out[-1][1]:
                out[-1][1] = max(out[-1][1], e)
            else:
                out.append([s, e])
        return out

    def codingBases(self, seqId, start=0, end=None):
        """Number of coding bases of seqId in [start, end) (0-based), as np.sum(mask[start:end]) gives it."""
        if seqId not in self.genes:
            return 0
        last = self.lastCodingBase[seqId]
        if end is None:
            end = last
        # Python slice semantics of the reference's mask[start:end]
        start, end, _ = slice(start, end).indices(last)
        n = 0
        for s, e in self._merged[seqId]:
            lo, hi = max(s, start), min(e, end)
            if hi > lo:
                n += hi - lo
        return n
