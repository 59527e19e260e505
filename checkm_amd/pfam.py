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

    def filterHitsFromSameClan(self, markerHits):
        """Per gene, drop a Pfam hit that overlaps a more significant hit of the same clan unless the two families nest
        (checkm/util/pfam.py:86-147).  This is the row-at-a-time form; QA runs the same rule for all bins inside ckm_reduce.
        Quirks kept: families without a clan share the clan `None`; markers whose id does not start with 'PF' pass through
        first; the result is a defaultdict(list)."""
        self.readClansAndNesting()
        kept = defaultdict(list)
        by_gene = defaultdict(list)
        for marker, hits in markerHits.items():
            if marker.startswith('PF'):
                for h in hits:
                    by_gene[h.target_name].append(h)
            else:
                kept[marker] = hits

        def family(h):
            a = h.query_accession
            return a[0:a.rfind('.')]

        for hits in by_gene.values():
            hits.sort(key=lambda h: (h.full_e_value, h.i_evalue))       # stable: ties keep their order of arrival
            fam = [family(h) for h in hits]
            dropped = [False] * len(hits)
            for i, hi in enumerate(hits):
                if dropped[i]:
                    continue
                partners = self.nested.get(fam[i], ())
                for j in range(i + 1, len(hits)):
                    hj = hits[j]
                    if dropped[j] or self.clan.get(fam[i]) != self.clan.get(fam[j]):
                        continue
                    overlap = (hi.ali_from <= hj.ali_from < hi.ali_to) or (hj.ali_from <= hi.ali_from < hj.ali_to)
                    if overlap and fam[j] not in partners:
                        dropped[j] = True
            for h, gone in zip(hits, dropped):
                if not gone:
                    kept[h.query_accession].append(h)
        return kept

    def _clan_maps(self):
        """(family -> clan, clan -> families) of the clan file, read ONCE per PFAM object: the reference reads the file twice per call of
        genesInSameClan (pfam.py:149-168), i.e. per distinct marker set of a run -- 0.1 s of every analyze pass's start here."""
        maps = getattr(self, '_maps', None)
        if maps is None:
            to_clan, members = {}, defaultdict(set)
            acc = None
            with open(self.pfamClanFile) as f:
                for line in f:
                    if '#=GF AC' in line:
                        acc = line.split()[2].strip()
                    elif '#=GF CL' in line:
                        c = line.split()[2].strip()
                        to_clan[acc] = c
                        members[c].add(acc)
            maps = self._maps = (to_clan, members)
        return maps

    def pfamIdToClanId(self):
        return dict(self._clan_maps()[0])

    def genesInClan(self):
        d = defaultdict(set)
        for c, fams in self._clan_maps()[1].items():
            d[c] = set(fams)
        return d

    def genesInSameClan(self, genes):
        """All other families of the clans the given genes span (pfam.py:149-168)."""
        to_clan, members = self._clan_maps()
        clans = set(to_clan[g] for g in genes if g in to_clan)
        out = set()
        for c in clans:
            out.update(members[c])
        return out - set(genes)
