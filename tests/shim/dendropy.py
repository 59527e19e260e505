"""TEST INFRASTRUCTURE -- a stand-in for the part of DendroPy 4 that checkm/treeParser.py touches, so that the REFERENCE's own
TreeParser (imported read-only from /root/reference) can run here, where DendroPy is not installed: tools/gen_tree_golden.py puts this
directory on sys.path, runs the reference on synthetic placed trees and records what it writes (tests/golden/tree_cases.json);
tests/test_tree_parser.py feeds checkm_amd.treeParser the same inputs.  Never imported by checkm_amd.

What the reference relies on (checkm/treeParser.py:107-130, 160-176, 229-258, 268-300, 385-397, 485-540):
  Tree.get_from_path(path, schema='newick', rooting='force-rooted', preserve_underscores=True)
  tree.find_node_with_taxon_label(label) -> leaf or None;  tree.find_node(filter_fn) -> first node in pre-order or None
  tree.leaf_nodes(), tree.internal_nodes(), tree.seed_node, tree.as_string(schema='newick', suppress_rooting=True)
  node.label (internal nodes: the Newick label or None), node.taxon.label (leaves), node.parent_node, node.edge_length,
  node.child_nodes(), node.leaf_nodes(), node.sister_nodes(), node.postorder_iter(), node.is_leaf(), node.is_internal(),
  node.new_child(taxon=, edge_length=), node.remove_child(child);  Taxon(label=)
Newick as DendroPy reads it with these options: single-quoted labels verbatim ('' is a quote), unquoted labels keep their underscores,
[comments] dropped, a label behind ')' is the internal node's `label` (internal nodes get no taxon), a leaf's label is its taxon.
"""


class Taxon(object):
    def __init__(self, label=None):
        self.label = label

    def __repr__(self):
        return "<Taxon '%s'>" % self.label


class Node(object):
    def __init__(self, taxon=None, label=None, edge_length=None):
        self.taxon = taxon
        self.label = label
        self.edge_length = edge_length
        self.parent_node = None
        self._children = []

    def child_nodes(self):
        return list(self._children)

    def is_leaf(self):
        return not self._children

    def is_internal(self):
        return bool(self._children)

    def add_child(self, node):
        node.parent_node = self
        self._children.append(node)
        return node

    def new_child(self, **kwargs):
        return self.add_child(Node(**kwargs))

    def remove_child(self, node):
        self._children.remove(node)
        node.parent_node = None
        return node

    def sister_nodes(self):
        p = self.parent_node
        return [] if p is None else [c for c in p._children if c is not self]

    def preorder_iter(self):
        stack = [self]
        while stack:
            n = stack.pop()
            yield n
            stack.extend(reversed(n._children))

    def postorder_iter(self):
        stack = [(self, False)]
        while stack:
            n, done = stack.pop()
            if done or not n._children:
                yield n
            else:
                stack.append((n, True))
                stack.extend((c, False) for c in reversed(n._children))

    def leaf_nodes(self):
        return [n for n in self.preorder_iter() if not n._children]

    def leaf_iter(self):
        return iter(self.leaf_nodes())


_STOP = set("(),:;[")


def _label(text, i):
    n = len(text)
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
                return "".join(out), i + 1
            out.append(c)
            i += 1
        raise ValueError("unterminated quoted label")
    j = i
    while j < n and text[j] not in _STOP and not text[j].isspace():
        j += 1
    return text[i:j], j


class Tree(object):
    def __init__(self, seed_node=None):
        self.seed_node = seed_node if seed_node is not None else Node()

    @classmethod
    def get_from_path(cls, src, schema, **kwargs):
        with open(src) as f:
            return cls.get_from_string(f.read(), schema, **kwargs)

    @classmethod
    def get_from_string(cls, text, schema, rooting=None, preserve_underscores=False, **_kw):
        if schema != "newick":
            raise ValueError("only newick")
        root = cur = Node()
        i, n = 0, len(text)
        closed = False
        while i < n:
            c = text[i]
            if c.isspace():
                i += 1
            elif c == "[":
                j = text.find("]", i)
                if j < 0:
                    raise ValueError("unterminated comment")
                i = j + 1
            elif c == "(":
                cur = cur.add_child(Node())
                i += 1
            elif c == ",":
                if cur.parent_node is None:
                    raise ValueError("comma outside parentheses")
                cur = cur.parent_node.add_child(Node())
                i += 1
            elif c == ")":
                if cur.parent_node is None:
                    raise ValueError("unbalanced parentheses")
                cur = cur.parent_node
                i += 1
            elif c == ":":
                j = i + 1
                while j < n and text[j] not in _STOP and not text[j].isspace():
                    j += 1
                cur.edge_length = float(text[i + 1:j])          # (anything that is not a number is an error, as in DendroPy)
                i = j
            elif c == ";":
                closed = True
                break
            else:
                quoted = text[i] == "'"
                lab, i = _label(text, i)
                if not quoted and not preserve_underscores:
                    lab = lab.replace("_", " ")
                if cur._children:
                    cur.label = lab
                else:
                    cur.taxon = Taxon(lab)
        if cur is not root or not closed:
            raise ValueError("not one complete newick statement")
        return cls(root)

    def preorder_node_iter(self):
        return self.seed_node.preorder_iter()

    def postorder_node_iter(self):
        return self.seed_node.postorder_iter()

    def leaf_nodes(self):
        return self.seed_node.leaf_nodes()

    def internal_nodes(self):
        return [n for n in self.seed_node.preorder_iter() if n._children]

    def find_node(self, filter_fn):
        for n in self.seed_node.preorder_iter():
            if filter_fn(n):
                return n
        return None

    def find_node_with_taxon_label(self, label):
        for n in self.seed_node.preorder_iter():
            if n.taxon is not None and n.taxon.label == label:
                return n
        return None

    def as_string(self, schema="newick", **_kw):
        def q(s):
            return "'%s'" % s.replace("'", "''") if any(ch in s for ch in " ()[]':;,|") else s

        def w(n):
            s = "(%s)" % ",".join(w(c) for c in n._children) if n._children else ""
            if n._children:
                s += q(n.label) if n.label else ""
            elif n.taxon is not None and n.taxon.label is not None:
                s += q(n.taxon.label)
            if n.edge_length is not None:
                s += ":%s" % n.edge_length
            return s
        return w(self.seed_node) + ";\n"
