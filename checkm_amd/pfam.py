"""Pfam clan/nesting tables (mirror of checkm/util/pfam.py:28-83,149-168).  Parsed ONCE per run
(the reference re-reads Pfam-A.hmm.dat for every bin, checkm/resultsParser.py:208)."""
from collections import defaultdict
import os


class PFAM(object):
    def __init__(self, pfamClanFile):
        self.pfamClanFile = pfamClanFile
        self.idToAcc = {}
        self.clan = {}
        self.nested = {}
        self._read = False

    def readClansAndNesting(self):
        """#=GF ID / AC (version stripped) / CL / NE records; nesting is made symmetric (pfam.py:34-56)."""
        if self._read:
            return
        if not os.path.exists(self.pfamClanFile):
            raise IOError("Input file does not exists: " + self.pfamClanFile)
        pairs = defaultdict(list)
        cur_id = cur_acc = None
        with open(self.pfamClanFile) as f:
            for line in f:
                if '#=GF ID' in line:
                    cur_id = line.split()[2].strip()
                elif '#=GF AC' in line:
                    cur_acc = line.split()[2].strip()
                    cur_acc = cur_acc[0:cur_acc.rfind('.')]
                    self.idToAcc[cur_id] = cur_acc
                elif '#=GF CL' in line:
                    self.clan[cur_acc] = line.split()[2].strip()
                elif '#=GF NE' in line:
                    other = line.split()[2].strip()
                    pairs[other].append(cur_id)
                    pairs[cur_id].append(other)
        for ident, others in pairs.items():
            self.nested[self.idToAcc[ident]] = set(self.idToAcc[x] for x in others)
        self._read = True

    def pfamIdToClanId(self):
        d = {}
        acc = None
        with open(self.pfamClanFile) as f:
            for line in f:
                if '#=GF AC' in line:
                    acc = line.split()[2].strip()
                elif '#=GF CL' in line:
                    d[acc] = line.split()[2].strip()
        return d

    def genesInClan(self):
        d = defaultdict(set)
        acc = None
        with open(self.pfamClanFile) as f:
            for line in f:
                if '#=GF AC' in line:
                    acc = line.split()[2].strip()
                elif '#=GF CL' in line:
                    d[line.split()[2].strip()].add(acc)
        return d

    def genesInSameClan(self, genes):
        """All other families of the clans the given genes span (pfam.py:149-168)."""
        to_clan = self.pfamIdToClanId()
        clans = set(to_clan[g] for g in genes if g in to_clan)
        members = self.genesInClan()
        out = set()
        for c in clans:
            out.update(members[c])
        return out - set(genes)
