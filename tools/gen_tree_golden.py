#!/usr/bin/env python
"""Goldens for checkm_amd/treeParser.py, produced by the REFERENCE's own TreeParser (imported read-only from /root/reference;
checkm/treeParser.py:151-221 getInsertionBranchId / getBinTaxonomy, :223-258 _findDomainNode, :327-380 _getMarkerSet, :430-466,
:468-553 getBinMarkerSets, :555-629 readNodeMetadata / readLineageMetadata) on synthetic PLACED genome trees.  The reference reads the
tree with DendroPy, which this image lacks: tests/shim/dendropy.py stands in for the dozen calls it makes (the shim's header lists
them) -- so the goldens pin the mirror to the reference's tree LOGIC; DendroPy's own Newick reader is not part of the pin.
Writes tests/golden/tree_cases.json (committed).  Run here only."""
import json
import os
import random
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = tempfile.mkdtemp(prefix="ckm_treedata_")
os.environ["CHECKM_DATA_PATH"] = DATA
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "tests", "shim"))
sys.path.insert(0, ROOT)

import logging  # noqa: E402
logging.getLogger("timestamp").setLevel(logging.ERROR)

from checkm.treeParser import TreeParser  # noqa: E402
from checkm.defaultValues import DefaultValues  # noqa: E402
from checkm_amd.treeParser import parse_set_literal  # noqa: E402   (only to normalise the written set literals)

RANKS = ["k__", "p__", "c__", "o__", "f__", "g__"]


class N(object):
    def __init__(self, label=None, taxon=None, length=0.1):
        self.label, self.taxon, self.length, self.kids, self.parent = label, taxon, length, [], None

    def add(self, k):
        k.parent = self
        self.kids.append(k)
        return k


def q(s):
    return "'%s'" % s if any(c in s for c in "|; ") else s


def newick(n):
    s = "(%s)" % ",".join(newick(k) for k in n.kids) if n.kids else ""
    s += q(n.label) if (n.kids and n.label) else (q(n.taxon) if not n.kids else "")
    if n.parent is not None:
        s += ":%g" % n.length
    return s


def make_world(rng, n_ref, n_bins, style):
    """A reference tree (every internal node labelled UID|taxon|bootstrap), bins inserted on random edges, metadata for every UID."""
    uid = [0]
    meta = {}
    img = [0]

    def new_uid():
        uid[0] += 1
        return "UID%d" % uid[0]

    def grow(depth, lineage, leaves):
        """subtree with `leaves` reference genomes under a node whose taxonomy string is `lineage`"""
        if leaves == 1:
            img[0] += 1
            return N(taxon="IMG_%d" % (2500000000 + img[0]), length=round(rng.uniform(0.01, 0.4), 4))
        u = new_uid()
        # the node's own taxon token: the ranks this node adds to its parent's lineage ('' when it adds none)
        add = []
        if depth < len(RANKS) and rng.random() < (0.75 if style != "sparse" else 0.35):
            add.append(RANKS[depth] + "T%d" % uid[0])
            if depth + 1 < len(RANKS) and rng.random() < 0.2:
                add.append(RANKS[depth + 1] + "T%dx" % uid[0])
        mine = lineage + add
        boot = rng.choice([100, 100, 97, 83, 61, 40]) if rng.random() < 0.9 else "NA"
        node = N(label="%s|%s|%s" % (u, ";".join(add), boot), length=round(rng.uniform(0.01, 0.3), 4))
        meta[u] = {"taxonomy": ";".join(mine), "bootstrap": boot, "leaves": leaves}
        left = rng.randint(1, leaves - 1)
        if rng.random() < 0.15 and leaves >= 3:          # a trifurcation now and then
            a = rng.randint(1, leaves - 2); b = rng.randint(1, leaves - a - 1)
            parts = [a, b, leaves - a - b]
        else:
            parts = [left, leaves - left]
        for p in parts:
            node.add(grow(depth + len(add), mine, p))
        return node

    root = N(label="UID1||")
    uid[0] = 1
    meta["UID1"] = {"taxonomy": "", "bootstrap": "NA", "leaves": n_ref}
    nb = max(2, int(n_ref * 0.7))
    for dom, cnt in (("k__Bacteria", nb), ("k__Archaea", max(2, n_ref - nb))):
        u = new_uid()
        d = N(label="%s|%s|100" % (u, dom), length=0.2)
        meta[u] = {"taxonomy": dom, "bootstrap": 100, "leaves": cnt}
        left = rng.randint(1, cnt - 1)
        d.add(grow(1, [dom], left)); d.add(grow(1, [dom], cnt - left))
        root.add(d)
    # bins: on random edges (a new unlabelled node takes the edge's place); some on the edges right below the root; some next to a bin
    # placed before (nested unlabelled nodes, or a second bin under the same inserted node); some stay out of the tree
    bins, placed = [], []

    def all_edges(n, out):
        for k in n.kids:
            out.append(k)
            all_edges(k, out)
        return out

    for b in range(n_bins):
        name = "bin_%02d" % b if b % 3 else "bin.%d_x" % b
        bins.append(name)
        r = rng.random()
        if r < 0.12:
            continue                                        # not in the tree
        if r < 0.30:
            child = rng.choice(root.kids)                   # the bacterial / archaeal branch below the root
        elif r < 0.42 and placed:
            child = rng.choice(placed)                      # beside an earlier bin
        else:
            child = rng.choice(all_edges(root, []))
        leaf = N(taxon=name, length=round(rng.uniform(0.0, 0.3), 4))
        if r >= 0.30 and r < 0.36 and placed and child.parent is not None and not child.parent.label:
            child.parent.add(leaf)                          # a second bin under the node pplacer inserted (a multifurcation)
        else:
            p = child.parent
            u = N(label=None, length=child.length / 2)
            child.length = child.length / 2
            idx = p.kids.index(child)
            p.kids[idx] = u; u.parent = p
            if rng.random() < 0.5:
                u.add(child); u.add(leaf)
            else:
                u.add(leaf); u.add(child)
        placed.append(leaf)
    # metadata
    pool = ["PF%05d.%d" % (rng.randint(1, 400), rng.randint(1, 20)) for _ in range(60)] + ["TIGR%05d" % rng.randint(1, 400) for _ in range(40)]
    rows, md = [], []
    for u in sorted(meta, key=lambda s: int(s[3:])):
        m = meta[u]
        ngen = m["leaves"] * rng.choice([1, 3, 10, 40])
        k = rng.randint(1, 5)
        sets = []
        genes = rng.sample(pool, rng.randint(k, min(len(pool), 12)))
        for g in genes:
            if len(sets) < k:
                sets.append([g])
            else:
                rng.choice(sets).append(g)
        lit = "[" + ", ".join("set([" + ", ".join("'%s'" % g for g in s) + "])" for s in sets) + "]"
        rows.append("\t".join([u, str(ngen), m["taxonomy"], str(m["bootstrap"]) if m["bootstrap"] != "NA" else "NA", "%.1f" % rng.uniform(30, 70), "%.1f" % rng.uniform(1, 8),
                               str(rng.randint(1500000, 7000000)), str(rng.randint(100000, 900000)), str(rng.randint(1200, 6500)), str(rng.randint(100, 900)), lit]))
        flat = [g for s in sets for g in s]
        def as_removed(g):
            return g.replace("PF", "pfam")[0:g.replace("PF", "pfam").rfind(".")] if g.startswith("PF") else g
        miss = set(as_removed(g) for g in rng.sample(flat, rng.randint(0, min(2, len(flat))))) | set(as_removed(g) for g in rng.sample(pool, rng.randint(0, 3)))
        dup = set(as_removed(g) for g in rng.sample(pool, rng.randint(0, 3)))
        def lit_set(s):
            return "set([" + ", ".join("'%s'" % g for g in sorted(s)) + "])"
        md.append("%s\t%s\t%s" % (u, lit_set(miss), lit_set(dup)))
    header = "UID\t# genomes\ttaxonomy\tbootstrap\tgc mean\tgc std\tsize mean\tsize std\tgenes mean\tgenes std\tmarker set"
    return newick(root) + ";\n", header + "\n" + "\n".join(rows) + "\n", "\n".join(md) + "\n", bins


class _Hits(object):
    def __init__(self, v):
        self.v = v

    def countUniqueHits(self):
        return self.v


class _Results(object):
    def __init__(self, d):
        self.results = {b: _Hits(tuple(v)) for b, v in d.items()}


def rows_of(path):
    out = {}
    for line in open(path).read().splitlines()[1:]:
        t = line.split("\t")
        out[t[0]] = [[t[2 + 4 * i], t[3 + 4 * i], t[4 + 4 * i], sorted(sorted(s) for s in parse_set_literal(t[5 + 4 * i]))] for i in range(int(t[1]))]
    return out


def main():
    rng = random.Random(20260930)
    cases = []
    specs = [(14, 8, "dense"), (30, 12, "dense"), (30, 12, "sparse"), (60, 16, "dense"), (8, 6, "sparse"), (45, 14, "dense"), (22, 10, "sparse"), (80, 18, "dense")]
    for n_ref, n_bins, style in specs:
        tree, metadata, missdup, bins = make_world(rng, n_ref, n_bins, style)
        gdir = os.path.join(DATA, "genome_tree")
        shutil.rmtree(gdir, ignore_errors=True)
        os.makedirs(gdir)
        open(os.path.join(gdir, DefaultValues.GENOME_TREE_METADATA), "w").write(metadata)
        open(os.path.join(gdir, DefaultValues.GENOME_TREE_MISSING_DUPLICATE), "w").write(missdup)
        out = tempfile.mkdtemp(prefix="ckm_treeout_")
        for b in bins:
            os.makedirs(os.path.join(out, "bins", b))
        os.makedirs(os.path.join(out, "storage", "tree"))
        open(os.path.join(out, "storage", "tree", DefaultValues.PPLACER_TREE_OUT), "w").write(tree)
        hits = {b: [rng.choice([0, 3, 9, 10, 11, 25, 40]), rng.choice([0, 0, 1, 9, 10, 11, 30])] for b in bins}
        tp = TreeParser()
        calls = []
        for args in ({"numGenomesMarkers": 30, "bootstrap": 0, "bNoLineageSpecificRefinement": True, "bForceDomain": False, "bRequireTaxonomy": False, "minUnique": 10, "maxMulti": 10},
                     {"numGenomesMarkers": 30, "bootstrap": 0, "bNoLineageSpecificRefinement": False, "bForceDomain": False, "bRequireTaxonomy": False, "minUnique": 10, "maxMulti": 10},
                     {"numGenomesMarkers": 2, "bootstrap": 70, "bNoLineageSpecificRefinement": False, "bForceDomain": False, "bRequireTaxonomy": True, "minUnique": 10, "maxMulti": 10},
                     {"numGenomesMarkers": 100, "bootstrap": 90, "bNoLineageSpecificRefinement": True, "bForceDomain": True, "bRequireTaxonomy": False, "minUnique": 0, "maxMulti": 1000},
                     {"numGenomesMarkers": 10, "bootstrap": 0, "bNoLineageSpecificRefinement": False, "bForceDomain": False, "bRequireTaxonomy": False, "minUnique": 26, "maxMulti": 0}):
            mf = os.path.join(out, "lineage.ms")
            tp.getBinMarkerSets(out, mf, args["numGenomesMarkers"], args["bootstrap"], args["bNoLineageSpecificRefinement"], args["bForceDomain"], args["bRequireTaxonomy"],
                                _Results(hits), args["minUnique"], args["maxMulti"])
            calls.append({"args": args, "rows": rows_of(mf), "order": [ln.split("\t")[0] for ln in open(mf).read().splitlines()[1:]]})
        srt = sorted(bins)
        lm = tp.readLineageMetadata(out, srt)
        cases.append({"tree": tree, "metadata": metadata, "missing_duplicate": missdup, "bins": bins, "hits": hits, "marker_sets": calls,
                      "taxonomy": tp.getBinTaxonomy(out, srt), "branch": tp.getInsertionBranchId(out, srt),
                      "lineage_metadata": {b: {k: v for k, v in d.items()} for b, d in lm.items()}})
        shutil.rmtree(out, ignore_errors=True)
    path = os.path.join(ROOT, "tests", "golden", "tree_cases.json")
    with open(path, "w") as f:
        json.dump({"generator": "tools/gen_tree_golden.py", "reference": "checkm/treeParser.py (v1.2.4) on tests/shim/dendropy.py", "cases": cases}, f, indent=0, sort_keys=True)
    print("wrote %s: %d cases, %d bytes" % (path, len(cases), os.path.getsize(path)))
    shutil.rmtree(DATA, ignore_errors=True)


if __name__ == "__main__":
    main()
