"""domtblout records and reader, API-compatible with checkm/hmmer.py:140-311 (HMMERParser,
HmmerHitDOM), plus HMMERRunner whose `search` runs on the MI355X instead of spawning hmmsearch
(checkm/hmmer.py:61-74)."""
import logging
import re
import sys


class FormatError(BaseException):
    pass


class HMMERError(BaseException):
    pass


class HMMMERModeError(BaseException):
    pass


_DOM_FIELDS = (("target_name", str), ("target_accession", str), ("target_length", int), ("query_name", str),
               ("query_accession", str), ("query_length", int), ("full_e_value", float), ("full_score", float),
               ("full_bias", float), ("dom", int), ("ndom", int), ("c_evalue", float), ("i_evalue", float),
               ("dom_score", float), ("dom_bias", float), ("hmm_from", int), ("hmm_to", int), ("ali_from", int),
               ("ali_to", int), ("env_from", int), ("env_to", int), ("acc", float), ("target_description", str))


class HmmerHitDOM(object):
    """One domtblout row (field contract: checkm/hmmer.py:255-285)."""

    def __init__(self, values):
        if len(values) == 23:
            for (name, conv), v in zip(_DOM_FIELDS, values):
                setattr(self, name, conv(v))
            if self.query_accession == '-':
                self.query_accession = self.query_name

    @classmethod
    def from_fields(cls, **kw):
        h = cls(())
        for k, v in kw.items():
            setattr(h, k, v)
        return h

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in _DOM_FIELDS}

    def __str__(self):
        return "\t".join(str(getattr(self, name)) for name, _ in _DOM_FIELDS)


class HMMERParser(object):
    """Iterates the hits of a domtblout handle (checkm/hmmer.py:140-200).  An empty line ends the
    table, as the reference's IndexError path does."""

    def __init__(self, fileHandle, mode='dom'):
        if mode != 'dom':
            raise HMMERError("Mode %s not understood, only 'dom' tables are produced on this path" % mode)
        self.handle = fileHandle
        self.mode = 'domtblout'

    def readHitsDOM(self):
        """Next hit of the table or None (checkm/hmmer.py:184-200)."""
        return self.next()

    def next(self):
        while True:
            line = self.handle.readline().rstrip()
            if len(line) == 0:
                return None
            if line[0] == '#':
                continue
            tok = re.split(r'\s+', line)
            if len(tok) < 23:
                raise FormatError("Error processing line:\n%s" % line)
            return HmmerHitDOM(tok[0:22] + [" ".join(tok[22:])])


def read_domtblout(path):
    with open(path) as fh:
        p = HMMERParser(fh)
        out = []
        while True:
            h = p.next()
            if h is None:
                return out
            out.append(h)


class HMMERRunner(object):
    """The reference probes for an hmmsearch binary (checkm/hmmer.py:131-137); here the probe is for a
    usable gfx950 device, and failure ends the run the same way (logger.error + sys.exit)."""

    def __init__(self, mode="dom"):
        self.logger = logging.getLogger('timestamp')
        if mode not in ("dom", "tbl", "align", "fetch"):
            raise HMMMERModeError("Mode %s not understood" % mode)
        self.mode = 'domtblout' if mode == 'dom' else mode
        self.checkForHMMER()

    def checkForHMMER(self):
        from checkm_amd import _lib
        try:
            n = _lib.device_count()
        except Exception as e:   # library missing
            self.logger.error("libcheckm_hip is not usable: %s" % e)
            sys.exit(1)
        if n < 1:
            self.logger.error("No MI355X (gfx950) device visible; the marker-gene scan has no CPU path.")
            sys.exit(1)

    def search(self, db, query, tableOut, hmmerOut, cmdlineOptions='', bKeepOutput=True):
        """Single-bin form of the scan: `db` HMM file against protein FASTA `query`, domtblout to tableOut."""
        if self.mode != 'domtblout':
            raise HMMMERModeError("Mode %s not compatible with search" % self.mode)
        from checkm_amd.markerGeneFinder import scan_files
        E, domE = 10.0, 10.0
        m = re.search(r'-E\s+(\S+)', cmdlineOptions)
        if m:
            E = float(m.group(1))
        m = re.search(r'--domE\s+(\S+)', cmdlineOptions)
        if m:
            domE = float(m.group(1))
        try:
            scan_files(db, [query], [tableOut], E, domE)
        except Exception as e:
            self.logger.error('marker-gene scan failed: %s' % e)
            sys.exit(1)
