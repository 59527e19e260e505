"""Lineage-specific marker-set selection from the placed genome tree: the consumer of `countUniqueHits` that turns the tree pass's
hits into `lineage.ms` (mirror of checkm/treeParser.py:468-553 `TreeParser.getBinMarkerSets` and the helpers it uses, :151-221, :223-258,
:327-380, :430-466, :555-629).  Host logic only: the tree is pplacer's `concatenated.tre` (pplacer itself is a separate program and out
of scope), the node statistics are CheckM's `genome_tree.metadata.tsv` and `missing_duplicate_genes_50.tsv`.

The reference reads the tree with dendropy (rooting forced, underscores preserved); none is installed here, so the small reader below
restates what the reference relies on: a rooted tree whose internal nodes may carry a label `UID|taxonomy|bootstrap` (nodes inserted by
pplacer carry none) and whose leaves are taxa (`IMG_...` reference genomes and the bin ids).  Because the reference cannot run here
without dendropy, this module's parity is pinned by hand-derived cases (tests/test_tree_parser.py), not by the reference's output.
"""
import logging
import os
import re
import sys

from .common import getBinIdsFromOutDir
from .defaultValues import DefaultValues
from .markerSets import BinMarkerSets, MarkerSet

GENOME_TREE_METADATA = 'genome_tree.metadata.tsv'                 # checkm/defaultValues.py:69
GENOME_TREE_MISSING_DUPLICATE = 'missing_duplicate_genes_50.tsv'  # checkm/defaultValues.py:70
PPLACER_TREE_OUT = 'concatenated.tre'                             # checkm/defaultValues.py:89


class TreeNode(object):
    """One node of the rooted tree.  `label`: an internal node's label ('' for nodes pplacer inserted); `taxon`: a leaf's name."""
    __slots__ = ('parent', 'children', 'label', 'taxon', 'length')

    def __init__(self):
        self.parent = None
        self.children = []
        self.label = ''
        self.taxon = None
        self.length = None

    def is_leaf(self):
        return not self.children

    def leaves(self):
        stack = [self]
        while stack:
            n = stack.pop()
            if n.children:
                stack.extend(reversed(n.children))
            else:
                yield n


_NAME_END = set('(),:;[')


def read_newick(text):
    """A rooted tree from one Newick statement.  Quoted names ('...' with '' for a quote) are taken literally, unquoted names keep
    their underscores (the reference asks dendropy for preserve_underscores), [comments] are skipped (pplacer writes edge numbers as
    {n} after the length and may write [..] comments)."""
    root = cur = TreeNode()
    i, n = 0, len(text)
    seen_open = False

    def name_at(i):
        if text[i] == "'":
            out = []
            i += 1
            while i < n:
                c = text[i]
                if c == "'":
                    if i + 1 < n and text[i + 1] == "'":
                        out.append("'")
                        i += 2
                        continue
                    return ''.join(out), i + 1
                out.append(c)
                i += 1
            raise ValueError('unterminated quoted name in the tree')
        j = i
        while j < n and text[j] not in _NAME_END and not text[j].isspace():
            j += 1
        return text[i:j], j

    while i < n:
        c = text[i]
        if c.isspace():
            i += 1
        elif c == '[':
            j = text.find(']', i)
            if j < 0:
                raise ValueError('unterminated comment in the tree')
            i = j + 1
        elif c == '(':
            child = TreeNode()
            child.parent = cur
            cur.children.append(child)
            cur = child
            seen_open = True
            i += 1
        elif c == ',':
            if cur.parent is None:
                raise ValueError('a comma outside parentheses in the tree')
            sib = TreeNode()
            sib.parent = cur.parent
            cur.parent.children.append(sib)
            cur = sib
            i += 1
        elif c == ')':
            if cur.parent is None:
                raise ValueError('unbalanced parentheses in the tree')
            cur = cur.parent
            i += 1
        elif c == ':':
            j = i + 1
            while j < n and (text[j] in '+-.eE' or text[j].isdigit()):
                j += 1
            try:
                cur.length = float(text[i + 1:j])
            except ValueError:
                cur.length = None
            i = j
            while i < n and text[i] == '{':            # pplacer's edge numbers
                k = text.find('}', i)
                if k < 0:
                    raise ValueError('unterminated edge number in the tree')
                i = k + 1
        elif c == ';':
            break
        else:
            name, i = name_at(i)
            if cur.children:
                cur.label = name
            else:
                cur.taxon = name
    if cur is not root or not seen_open:
        raise ValueError('the tree is not one complete Newick statement')
    return Tree(root)


class Tree(object):
    def __init__(self, root):
        self.root = root
        self._leaf = {}
        for leaf in root.leaves():
            self._leaf.setdefault(leaf.taxon, leaf)

    @classmethod
    def from_path(cls, path):
        with open(path) as f:
            return read_newick(f.read())

    def find_leaf(self, taxon):
        """The leaf of that name or None (the reference: tree.find_node_with_taxon_label)."""
        return self._leaf.get(taxon)


_PY2_SET = re.compile(r'set\(\[(.*?)\]\)', re.S)


def parse_set_literal(text):
    """`[set(['a', 'b']), set(['c'])]` as the data files spell it (Python 2 repr) or `[{'a', 'b'}, {'c'}]` -> list of sets; a single
    `set([...])` -> one set.  The reference evals these strings (checkm/treeParser.py:378, :436-437)."""
    import ast
    t = _PY2_SET.sub(lambda m: '{%s}' % m.group(1) if m.group(1).strip() else '()', text.strip())
    t = t.replace('set()', '()')
    v = ast.literal_eval(t)
    if isinstance(v, (set, frozenset)):
        return set(v)
    if isinstance(v, tuple) and not v:
        return set()
    return [set(x) for x in v]


class TreeParser(object):
    """Marker sets, taxonomy and lineage statistics of bins placed in the genome tree."""

    def __init__(self):
        self.logger = logging.getLogger('timestamp')
        self.lineageSpecificGenesToRemove = None

    # ---- data files ---------------------------------------------------------------------------------------------------------------
    def _tree_dir(self):
        return os.path.join(DefaultValues.CHECKM_DATA_DIR, 'genome_tree')

    def readNodeMetadata(self):
        """uid -> statistics of that internal node (checkm/treeParser.py:555-584)."""
        stats = {}
        with open(os.path.join(self._tree_dir(), GENOME_TREE_METADATA)) as f:
            f.readline()
            for line in f:
                t = line.rstrip().split('\t')
                d = {'# genomes': int(t[1]), 'taxonomy': t[2]}
                try:
                    d['bootstrap'] = float(t[3])
                except ValueError:
                    d['bootstrap'] = 'NA'
                d['gc mean'] = float(t[4])
                d['gc std'] = float(t[5])
                d['genome size mean'] = float(t[6]) / 1e6
                d['genome size std'] = float(t[7]) / 1e6
                d['gene count mean'] = float(t[8])
                d['gene count std'] = float(t[9])
                d['marker set'] = t[10].rstrip()
                stats[t[0]] = d
        return stats

    def _readLineageSpecificGenesToRemove(self):
        """uid -> genes lost or duplicated in that lineage (checkm/treeParser.py:430-439)."""
        self.lineageSpecificGenesToRemove = {}
        with open(os.path.join(self._tree_dir(), GENOME_TREE_MISSING_DUPLICATE)) as f:
            for line in f:
                t = line.rstrip('\n').split('\t')
                self.lineageSpecificGenesToRemove[t[0]] = parse_set_literal(t[1]) | parse_set_literal(t[2])

    def _read_tree(self, outDir):
        return Tree.from_path(os.path.join(outDir, 'storage', 'tree', PPLACER_TREE_OUT))

    # ---- walks ----------------------------------------------------------------------------------------------------------------------
    @staticmethod
    def _first_labelled_ancestor(node):
        p = node.parent
        while p is not None and not p.label:
            p = p.parent
        return p

    def _findDomainNode(self, binNode):
        """The labelled node that defines the domain of a bin sitting on the bacterial or archaeal branch below the root
        (checkm/treeParser.py:223-258): up to the first ancestor holding a reference genome, then breadth first to a label."""
        cur = binNode.parent
        while cur is not None and not any((leaf.taxon or '').startswith('IMG_') for leaf in cur.leaves()):
            cur = cur.parent
        queue = [cur] if cur is not None else []
        while queue:
            cur = queue.pop(0)
            if cur.label:
                return cur
            queue.extend(c for c in cur.children if not c.is_leaf())
        self.logger.error('Failed to associate bin with a domain. Please report this bug.')
        sys.exit(1)

    def _getNextNamedNode(self, node, stats):
        """Taxonomy of the first ancestor that has one, else 'root' (checkm/treeParser.py:327-342)."""
        p = node.parent
        while p is not None:
            if p.label:
                tax = stats[p.label.split('|')[0]]['taxonomy']
                if tax != '':
                    return tax
            p = p.parent
        return 'root'

    def _getMarkerSet(self, parentNode, tree, stats, numGenomesMarkers, bootstrap, bForceDomain, bRequireTaxonomy):
        """(node, MarkerSet) of the first node from `parentNode` towards the root that meets the selection criteria; the root's
        when none does (checkm/treeParser.py:344-380)."""
        sel = parentNode
        taxonomyStr = 'root'
        uid = st = None
        while True:
            if sel.label:
                tokens = sel.label.split('|')
                uid = tokens[0]
                st = stats[uid]
                enough = st['# genomes'] == 'NA' or int(st['# genomes']) >= numGenomesMarkers
                supported = st['bootstrap'] == 'NA' or int(st['bootstrap']) >= bootstrap
                if enough and supported and (not bForceDomain or tokens[1] in ('k__Bacteria', 'k__Archaea')) \
                        and (not bRequireTaxonomy or st['taxonomy'] != ''):
                    taxonomyStr = st['taxonomy']
                    if not bRequireTaxonomy and taxonomyStr == '':
                        taxonomyStr = self._getNextNamedNode(sel, stats)
                    break
            if sel.parent is None:
                break
            sel = sel.parent
        if st is None:
            raise ValueError('no labelled node between the placement and the root of the genome tree')
        return sel, MarkerSet(uid, taxonomyStr.split(';')[-1], int(st['# genomes']), parse_set_literal(st['marker set']))

    def _removeInvalidLineageMarkerGenes(self, markerSet, toRemove):
        """Drop the genes subject to lineage-specific loss or duplication; Pfam accessions are compared as `pfamNNNNN`
        (checkm/treeParser.py:441-466)."""
        kept = []
        for ms in markerSet.markerSet:
            s = set()
            for gene in ms:
                probe = gene
                if probe.startswith('PF'):
                    probe = gene.replace('PF', 'pfam')
                    probe = probe[0:probe.rfind('.')]
                if probe not in toRemove:
                    s.add(gene)
            if s:
                kept.append(s)
        return MarkerSet(markerSet.UID, markerSet.lineageStr, markerSet.numGenomes, kept)

    # ---- the selection ----------------------------------------------------------------------------------------------------------------
    def getBinMarkerSets(self, outDir, markerFile, numGenomesMarkers, bootstrap, bNoLineageSpecificRefinement, bForceDomain,
                         bRequireTaxonomy, resultsParser, minUnique, maxMulti):
        """Write the Lineage marker file: for every bin the marker sets of the qualifying nodes between its placement and the root
        (checkm/treeParser.py:468-553).  `resultsParser.results[binId].countUniqueHits()` (the tree pass's reduction, on the device)
        decides whether a bin has enough phylogenetic markers to deserve more than its domain's set."""
        self.logger.info('Determining marker sets for each genome bin.')
        binIds = getBinIdsFromOutDir(outDir)
        stats = self.readNodeMetadata()
        tree = self._read_tree(outDir)
        root = tree.root
        with open(markerFile, 'w') as fout:
            fout.write(DefaultValues.LINEAGE_MARKER_FILE_HEADER + '\n')
            for binId in binIds:
                node = tree.find_leaf(binId)
                bms = BinMarkerSets(binId, BinMarkerSets.TREE_MARKER_SET)
                if node is None:                                        # not placed: the root's set
                    _sel, ms = self._getMarkerSet(root, tree, stats, numGenomesMarkers, bootstrap, bForceDomain, bRequireTaxonomy)
                    bms.addMarkerSet(ms)
                else:
                    first = self._first_labelled_ancestor(node)
                    if first is None:
                        raise ValueError('bin %s hangs below no labelled node of the genome tree' % binId)
                    if first.parent is None:                            # on the bacterial/archaeal branch right below the root:
                        cur = self._findDomainNode(node).children[0]    # start below the domain node so that its set is included
                    else:
                        cur = node
                    refinement = None
                    if not bNoLineageSpecificRefinement:
                        if self.lineageSpecificGenesToRemove is None:
                            self._readLineageSpecificGenesToRemove()
                        refinement = self.lineageSpecificGenesToRemove[first.label.split('|')[0]]
                    uniqueHits, multiCopyHits = resultsParser.results[binId].countUniqueHits()
                    force = bForceDomain or (uniqueHits < minUnique) or (multiCopyHits > maxMulti)
                    while cur.parent is not None:
                        cur, ms = self._getMarkerSet(cur.parent, tree, stats, numGenomesMarkers, bootstrap, force, bRequireTaxonomy)
                        if refinement is not None:
                            ms = self._removeInvalidLineageMarkerGenes(ms, refinement)
                        bms.addMarkerSet(ms)
                bms.write(fout)

    # ---- the reports' lookups -----------------------------------------------------------------------------------------------------
    def getInsertionBranchId(self, outDir, binIds):
        """bin -> UID of the first labelled node above its placement, 'NA' if not placed (checkm/treeParser.py:151-180)."""
        tree = self._read_tree(outDir)
        out = {}
        for binId in binIds:
            node = tree.find_leaf(binId)
            if node is None:
                out[binId] = 'NA'
                continue
            first = self._first_labelled_ancestor(node)
            out[binId] = first.label.split('|')[0]
        return out

    def getBinTaxonomy(self, outDir, binIds):
        """bin -> the taxon strings of its labelled ancestors, root first; the domain + ' (root)' when none carries one
        (checkm/treeParser.py:182-221)."""
        tree = self._read_tree(outDir)
        out = {}
        for binId in binIds:
            node = tree.find_leaf(binId)
            if node is None:
                out[binId] = 'NA'
                continue
            taxa = None
            p = node.parent
            while p is not None:
                if p.label:
                    tokens = p.label.split('|')
                    if tokens[1] != '':
                        taxa = tokens[1] + ';' + taxa if taxa else tokens[1]
                p = p.parent
            if not taxa:
                taxa = self._findDomainNode(node).label.split('|')[1] + ' (root)'
            out[binId] = taxa
        return out

    def readLineageMetadata(self, outDir, binIds):
        """bin -> statistics of the first labelled node above it (checkm/treeParser.py:586-629)."""
        stats = self.readNodeMetadata()
        tree = self._read_tree(outDir)
        out = {}
        for binId in binIds:
            node = tree.find_leaf(binId)
            if node is None:
                d = dict.fromkeys(('# genomes', 'gc mean', 'gc std', 'genome size mean', 'genome size std', 'gene count mean',
                                   'gene count std', 'marker set'), 'NA')
                d['taxonomy'] = 'unresolved'
                out[binId] = d
                continue
            first = self._first_labelled_ancestor(node)
            if first is None:
                self.logger.error('Failed to find lineage-specific statistics for inserted bin: ' + binId)
                sys.exit(1)
            out[binId] = stats[first.label.split('|')[0]]
        return out
