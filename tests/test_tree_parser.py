"""TreeParser.getBinMarkerSets and its lookups (the reference: checkm/treeParser.py:151-629): hand-derived cases on a hand-built genome
tree, and -- round 6 -- goldens written by the REFERENCE's own class on synthetic placed trees (tools/gen_tree_golden.py; the reference
needs DendroPy, which this image lacks: tests/shim/dendropy.py stands in for the calls it makes)."""
import os

import pytest

from checkm_amd.defaultValues import DefaultValues
from checkm_amd.markerSets import MarkerSetParser
from checkm_amd.treeParser import TreeParser, parse_set_literal, read_newick

#   UID1 (root, no taxonomy)
#   |-- (binC, UID2 k__Bacteria)               binC sits on the bacterial branch right below the root, in a node pplacer inserted
#   |        UID2 -- UID5 f__F -- UID10 g__A (IMG_1, IMG_2)
#   |                         \-- (binA, IMG_3)                      inserted node
#   |             \-- IMG_4
#   \-- UID3 k__Archaea -- (IMG_5, binB) , IMG_6
TREE = ("((binC:0.2{7},((('IMG_1':0.1,IMG_2:0.1)'UID10|g__A|100':0.1{3},(binA:0.1,IMG_3:0.1):0.05)'UID5|f__F|90':0.1,IMG_4:0.1)"
        "'UID2|k__Bacteria|100':0.2):0.1,((IMG_5:0.1,binB:1e-2):0.1,IMG_6:0.2)'UID3|k__Archaea|100':0.3)'UID1||';\n")

SETS = {
    'UID1': "[set(['PF00001.5', 'TIGR00001']), set(['PF00002.1'])]",
    'UID2': "[set(['PF00001.5', 'PF00003.2']), set(['TIGR00002'])]",
    'UID3': "[set(['PF00004.1'])]",
    'UID5': "[set(['PF00001.5', 'PF00005.9', 'TIGR00003']), set(['TIGR00004'])]",
    'UID10': "[set(['PF00006.3'])]",
}
GENOMES = {'UID1': 5000, 'UID2': 4000, 'UID3': 200, 'UID5': 60, 'UID10': 2}
TAXONOMY = {'UID1': '', 'UID2': 'k__Bacteria', 'UID3': 'k__Archaea', 'UID5': 'k__Bacteria;p__P;f__F', 'UID10': 'k__Bacteria;p__P;f__F;g__A'}
BOOT = {'UID1': 'NA', 'UID2': '100.0', 'UID3': '100.0', 'UID5': '90.0', 'UID10': '100.0'}


class _Hits(object):
    def __init__(self, unique, multi):
        self.v = (unique, multi)

    def countUniqueHits(self):
        return self.v


class _Results(object):
    def __init__(self, **bins):
        self.results = {b: _Hits(*v) for b, v in bins.items()}


@pytest.fixture
def world(tmp_path):
    data = tmp_path / 'data'
    (data / 'genome_tree').mkdir(parents=True)
    with open(data / 'genome_tree' / 'genome_tree.metadata.tsv', 'w') as f:
        f.write('UID\t# genomes\ttaxonomy\tbootstrap\tgc mean\tgc std\tsize mean\tsize std\tgenes mean\tgenes std\tmarker set\n')
        for uid in SETS:
            f.write('\t'.join([uid, str(GENOMES[uid]), TAXONOMY[uid], BOOT[uid], '50.0', '5.0', '3000000', '500000', '3000', '400', SETS[uid]]) + '\n')
    with open(data / 'genome_tree' / 'missing_duplicate_genes_50.tsv', 'w') as f:
        for uid in SETS:
            missing = "set(['pfam00005'])" if uid == 'UID5' else 'set([])'
            dup = "set(['TIGR00004', 'pfam00001'])" if uid == 'UID5' else 'set([])'
            f.write('%s\t%s\t%s\n' % (uid, missing, dup))
    with open(data / 'selected_marker_sets.tsv', 'w') as f:            # (internal node -> the node whose set is used for it)
        for uid, sel in (('UID1', 'UID1'), ('UID2', 'UID2'), ('UID3', 'UID3'), ('UID5', 'UID2'), ('UID10', 'UID5')):
            f.write('%s\t%s\n' % (uid, sel))
    out = tmp_path / 'out'
    for b in ('binA', 'binB', 'binC', 'binD'):
        (out / 'bins' / b).mkdir(parents=True)
    (out / 'storage' / 'tree').mkdir(parents=True)
    (out / 'storage' / 'tree' / 'concatenated.tre').write_text(TREE)
    old = DefaultValues.CHECKM_DATA_DIR
    DefaultValues.set_data_root(str(data))
    yield str(out)
    DefaultValues.set_data_root(old)


def _select(out, results, **kw):
    args = dict(numGenomesMarkers=30, bootstrap=0, bNoLineageSpecificRefinement=True, bForceDomain=False, bRequireTaxonomy=False,
                minUnique=10, maxMulti=10)
    args.update(kw)
    path = os.path.join(out, 'lineage.ms')
    TreeParser().getBinMarkerSets(out, path, args['numGenomesMarkers'], args['bootstrap'], args['bNoLineageSpecificRefinement'],
                                  args['bForceDomain'], args['bRequireTaxonomy'], results, args['minUnique'], args['maxMulti'])
    lines = open(path).read().splitlines()
    assert lines[0] == DefaultValues.LINEAGE_MARKER_FILE_HEADER
    got = {}
    for line in lines[1:]:
        t = line.split('\t')
        n = int(t[1])
        assert len(t) == 2 + 4 * n
        got[t[0]] = [(t[2 + 4 * i], t[3 + 4 * i], int(t[4 + 4 * i]), sorted(sorted(s) for s in parse_set_literal(t[5 + 4 * i]))) for i in range(n)]
    return got, path


def _sets(uid):
    return sorted(sorted(s) for s in parse_set_literal(SETS[uid]))


def test_newick_reader():
    t = read_newick("((a_b:1,'it''s c':2)'x|y z|3':0.5{12},[note]d)root_label;")
    assert [leaf.taxon for leaf in t.root.leaves()] == ['a_b', "it's c", 'd']
    inner = t.find_leaf('a_b').parent
    assert inner.label == 'x|y z|3' and inner.length == 0.5 and t.root.label == 'root_label'
    assert t.find_leaf('a_b').length == 1.0 and t.find_leaf('nope') is None
    for bad in ('((a,b);', 'a,b;', "('a,b);", ''):
        with pytest.raises(ValueError):
            read_newick(bad)


def test_set_literals_of_both_spellings():
    assert parse_set_literal("[set(['a', 'b']), set(['c'])]") == [{'a', 'b'}, {'c'}]
    assert parse_set_literal("[{'a', 'b'}, {'c'}]") == [{'a', 'b'}, {'c'}]
    assert parse_set_literal("set(['x'])") == {'x'} and parse_set_literal('set([])') == set() and parse_set_literal('set()') == set()


def test_marker_sets_between_placement_and_root(world):
    res = _Results(binA=(40, 0), binB=(40, 0), binC=(40, 0), binD=(40, 0))
    got, path = _select(world, res)
    # binA: UID10 is not an ancestor; UID5 (60 genomes >= 30), UID2, then the root, whose empty taxonomy reads 'root'
    assert got['binA'] == [('UID5', 'f__F', 60, _sets('UID5')), ('UID2', 'k__Bacteria', 4000, _sets('UID2')), ('UID1', 'root', 5000, _sets('UID1'))]
    assert got['binB'] == [('UID3', 'k__Archaea', 200, _sets('UID3')), ('UID1', 'root', 5000, _sets('UID1'))]
    # binC hangs right below the root: the walk starts under the domain node, so the bacterial set comes first
    assert got['binC'] == [('UID2', 'k__Bacteria', 4000, _sets('UID2')), ('UID1', 'root', 5000, _sets('UID1'))]
    # binD was not placed: the root's set alone
    assert got['binD'] == [('UID1', 'root', 5000, _sets('UID1'))]
    # and the file is a Lineage marker file the path's parser reads back
    bms = MarkerSetParser().parseLineageMarkerSetFile(path)
    assert [ms.UID for ms in bms['binA'].markerSetIter()] == ['UID5', 'UID2', 'UID1']
    assert bms['binA'].mostSpecificMarkerSet().getMarkerGenes() == {'PF00001.5', 'PF00005.9', 'TIGR00003', 'TIGR00004'}
    assert bms['binA'].selectedMarkerSet().UID == 'UID2'                     # UID5's selected set is UID2's


def test_thresholds_and_the_domain_rule(world):
    res = _Results(binA=(40, 0), binB=(3, 0), binC=(40, 11), binD=(0, 0))
    got, _ = _select(world, res, numGenomesMarkers=100)
    assert [g[0] for g in got['binA']] == ['UID2', 'UID1']                   # UID5 has 60 genomes < 100
    got, _ = _select(world, res, bootstrap=95)
    assert [g[0] for g in got['binA']] == ['UID2', 'UID1']                   # UID5's bootstrap is 90; the root's is NA and passes
    got, _ = _select(world, res)
    assert [g[0] for g in got['binB']] == ['UID3', 'UID1']                   # 3 unique hits < 10: domain sets only -- the same here
    got, _ = _select(world, _Results(binA=(3, 0), binB=(3, 0), binC=(40, 11), binD=(0, 0)))
    assert [g[0] for g in got['binA']] == ['UID2', 'UID1']                   # too few unique hits: f__F is skipped
    assert [g[0] for g in got['binC']] == ['UID2', 'UID1']                   # too many multi-copy hits
    got, _ = _select(world, res, bForceDomain=True)
    assert [g[0] for g in got['binA']] == ['UID2', 'UID1'] and [g[0] for g in got['binD']] == ['UID1']
    got, _ = _select(world, res, bRequireTaxonomy=True)
    assert [(g[0], g[1]) for g in got['binA']] == [('UID5', 'f__F'), ('UID2', 'k__Bacteria'), ('UID1', 'root')]   # the root never qualifies, its set closes the list


def test_lineage_specific_refinement(world):
    res = _Results(binA=(40, 0), binB=(40, 0), binC=(40, 0), binD=(40, 0))
    got, _ = _select(world, res, bNoLineageSpecificRefinement=False)
    # binA's first labelled ancestor is UID5: pfam00001, pfam00005 and TIGR00004 leave every set of the bin; emptied sets disappear
    assert got['binA'] == [('UID5', 'f__F', 60, [['TIGR00003']]), ('UID2', 'k__Bacteria', 4000, [['PF00003.2'], ['TIGR00002']]),
                           ('UID1', 'root', 5000, [['PF00002.1'], ['TIGR00001']])]
    assert got['binB'] == [('UID3', 'k__Archaea', 200, _sets('UID3')), ('UID1', 'root', 5000, _sets('UID1'))]


def test_report_lookups(world):
    tp = TreeParser()
    bins = ['binA', 'binB', 'binC', 'binD']
    assert tp.getInsertionBranchId(world, bins) == {'binA': 'UID5', 'binB': 'UID3', 'binC': 'UID1', 'binD': 'NA'}
    assert tp.getBinTaxonomy(world, bins) == {'binA': 'k__Bacteria;f__F', 'binB': 'k__Archaea', 'binC': 'k__Bacteria (root)', 'binD': 'NA'}
    meta = tp.readLineageMetadata(world, bins)
    assert meta['binA']['# genomes'] == 60 and meta['binA']['taxonomy'] == 'k__Bacteria;p__P;f__F' and meta['binA']['genome size mean'] == 3.0
    assert meta['binD']['taxonomy'] == 'unresolved' and meta['binD']['# genomes'] == 'NA'


def _run_mirror_on(case, tmp_path):
    data = tmp_path / 'data'
    (data / 'genome_tree').mkdir(parents=True)
    (data / 'genome_tree' / 'genome_tree.metadata.tsv').write_text(case['metadata'])
    (data / 'genome_tree' / 'missing_duplicate_genes_50.tsv').write_text(case['missing_duplicate'])
    out = tmp_path / 'out'
    for b in case['bins']:
        (out / 'bins' / b).mkdir(parents=True)
    (out / 'storage' / 'tree').mkdir(parents=True)
    (out / 'storage' / 'tree' / 'concatenated.tre').write_text(case['tree'])
    return str(data), str(out)


def _rows(p):
    out = {}
    lines = open(p).read().splitlines()
    assert lines[0] == DefaultValues.LINEAGE_MARKER_FILE_HEADER
    for line in lines[1:]:
        t = line.split('\t')
        out[t[0]] = [[t[2 + 4 * i], t[3 + 4 * i], t[4 + 4 * i], sorted(sorted(s) for s in parse_set_literal(t[5 + 4 * i]))] for i in range(int(t[1]))]
    return out, [ln.split('\t')[0] for ln in lines[1:]]


def test_reference_goldens(tmp_path):
    """checkm/treeParser.py itself, run by tools/gen_tree_golden.py on synthetic placed trees (bins on random edges, on the two branches
    below the root, beside each other, absent; sparse and dense taxonomy; every selection switch), wrote tests/golden/tree_cases.json:
    the mirror must write the same lineage.ms rows in the same order and return the same taxonomy / branch / lineage statistics."""
    import json
    gold = json.load(open(os.path.join(os.path.dirname(__file__), 'golden', 'tree_cases.json')))
    old = DefaultValues.CHECKM_DATA_DIR
    try:
        nsets = 0
        for k, case in enumerate(gold['cases']):
            data, out = _run_mirror_on(case, tmp_path / ('c%d' % k))
            DefaultValues.set_data_root(data)
            res = _Results(**{b: tuple(v) for b, v in case['hits'].items()})
            tp = TreeParser()
            for call in case['marker_sets']:
                a = call['args']
                path = os.path.join(out, 'lineage.ms')
                tp.getBinMarkerSets(out, path, a['numGenomesMarkers'], a['bootstrap'], a['bNoLineageSpecificRefinement'], a['bForceDomain'], a['bRequireTaxonomy'],
                                    res, a['minUnique'], a['maxMulti'])
                rows, order = _rows(path)
                assert order == call['order'], (k, a)
                assert rows == call['rows'], (k, a)
                nsets += sum(len(v) for v in rows.values())
            srt = sorted(case['bins'])
            assert tp.getBinTaxonomy(out, srt) == case['taxonomy'], k
            assert tp.getInsertionBranchId(out, srt) == case['branch'], k
            assert tp.readLineageMetadata(out, srt) == case['lineage_metadata'], k
        assert nsets > 500
    finally:
        DefaultValues.set_data_root(old)


def test_with_the_reference_when_it_can_run(world):
    """Where the reference package is on disk (this container, not the GPU box) its own TreeParser -- on tests/shim/dendropy.py, the
    stand-in for the DendroPy calls it makes -- must write the same file for the hand-built tree of this module."""
    if not os.path.isdir('/root/reference/checkm'):
        pytest.skip('no reference package')
    import subprocess
    import sys
    data = DefaultValues.CHECKM_DATA_DIR
    tre = os.path.join(world, 'storage', 'tree', 'concatenated.tre')
    import re
    text = open(tre).read()
    open(tre, 'w').write(re.sub(r'\{\d+\}', '', text))            # (pplacer's edge numbers: the mirror's reader skips them, DendroPy would not take them)
    code = ("import sys\nfrom checkm.treeParser import TreeParser\n"
            "class H:\n    def countUniqueHits(self): return (40, 0)\n"
            "class R:\n    results = {b: H() for b in ('binA', 'binB', 'binC', 'binD')}\n"
            "TreeParser().getBinMarkerSets(%r, %r, 30, 0, False, False, False, R(), 10, 10)\n" % (world, os.path.join(world, 'ref.ms')))
    shim = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'shim')
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(['/root/reference', shim]), CHECKM_DATA_PATH=data, PYTHONDONTWRITEBYTECODE='1')
    r = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    res = _Results(binA=(40, 0), binB=(40, 0), binC=(40, 0), binD=(40, 0))
    _got, path = _select(world, res, bNoLineageSpecificRefinement=False)
    assert _rows(path) == _rows(os.path.join(world, 'ref.ms'))
