"""Newick / NEXUS-tree reading and writing without DendroPy.

The reference reads the tree with DendroPy (tapir/compute.py:61) and HyPhy re-reads the corrected tree
through scriptQueryTree.bf:16-72 (first tree of a NEXUS file, or a bare Newick string; rooted trees
accepted, models_and_rates.bf:913).  The engine needs four things from a tree: leaf names, topology,
branch lengths and the root-to-tip depth.
"""
import re


class Node:
    __slots__ = ("children", "name", "length", "parent")

    def __init__(self):
        self.children = []
        self.name = None
        self.length = None
        self.parent = None

    def is_leaf(self):
        return not self.children


class NewickError(ValueError):
    pass


_NUM = re.compile(r"\s*:\s*([+-]?(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)")


def parse(text):
    """Parse one Newick string -> root Node.  Comments in [...] are dropped, quoted labels honoured."""
    text = re.sub(r"\[[^\]]*\]", "", text).strip()
    if not text:
        raise NewickError("empty tree string")
    n = len(text)
    root = Node()
    cur = root
    # iterative descent (caterpillar trees can be thousands of levels deep)
    if text[0] != "(":
        raise NewickError("This doesn't seem to be a valid Newick string: can't find the opening parenthesis")
    stack = []
    i = 0
    while i < n:
        c = text[i]
        if c == "(":
            child = Node()
            child.parent = cur
            cur.children.append(child)
            stack.append(cur)
            cur = child
            i += 1
        elif c == ",":
            if not stack:
                raise NewickError("unexpected ',' outside parentheses")
            sib = Node()
            sib.parent = stack[-1]
            stack[-1].children.append(sib)
            cur = sib
            i += 1
        elif c == ")":
            if not stack:
                raise NewickError("can't match the parentheses")
            cur = stack.pop()
            i += 1
        elif c == ";":
            break
        elif c.isspace():
            i += 1
        elif c == ":":
            m = _NUM.match(text, i)
            if not m:
                raise NewickError("bad branch length at offset %d" % i)
            cur.length = float(m.group(1))
            i = m.end()
        elif c == "'":
            j = i + 1
            buf = []
            while j < n:
                if text[j] == "'":
                    if j + 1 < n and text[j + 1] == "'":
                        buf.append("'")
                        j += 2
                        continue
                    break
                buf.append(text[j])
                j += 1
            cur.name = "".join(buf)
            i = j + 1
        else:
            m = re.compile(r"[^:,;()\[\]\s']+").match(text, i)
            cur.name = m.group(0)
            i = m.end()
    if stack:
        raise NewickError("can't match the parentheses")
    return root


def read_tree(path, fmt="newick"):
    """Read the first tree of a Newick or NEXUS file (fmt as bin/tapir_compute.py --tree-format)."""
    with open(path) as fh:
        text = fh.read()
    if fmt == "nexus" or text.lstrip().upper().startswith("#NEXUS"):
        translate = {}
        tm = re.search(r"translate\s+(.*?);", text, flags=re.I | re.S)
        if tm:
            for item in tm.group(1).split(","):
                parts = item.split()
                if len(parts) >= 2:
                    translate[parts[0]] = parts[1].strip("'")
        m = re.search(r"^\s*tree\s+[^=]+=\s*(?:\[[^\]]*\]\s*)*(.*?;)", text, flags=re.I | re.S | re.M)
        if not m:
            raise NewickError("This NEXUS file doesn't contain a valid tree block")
        root = parse(m.group(1))
        if translate:
            for leaf in leaves(root):
                leaf.name = translate.get(leaf.name, leaf.name)
        return root
    return parse(text)


def postorder(root):
    out, stack = [], [(root, 0)]
    while stack:
        node, k = stack.pop()
        if k < len(node.children):
            stack.append((node, k + 1))
            stack.append((node.children[k], 0))
        else:
            out.append(node)
    return out


def leaves(root):
    return [n for n in postorder(root) if n.is_leaf()]


def tree_length(root):
    """DendroPy Tree.length(): sum of all edge lengths (missing lengths count 0)."""
    return sum((n.length or 0.0) for n in postorder(root) if n is not root)


def distance_from_tip(root):
    """DendroPy Node.distance_from_tip() of the seed node: root-to-tip distance along the longest path
    (equal for all tips of an ultrametric tree)."""
    depth = {}
    for n in postorder(root):
        depth[id(n)] = max(((c.length or 0.0) + depth[id(c)] for c in n.children), default=0.0)
    return depth[id(root)]


def _fmt_len(x):
    return repr(float(x))


def write(root):
    """Newick string with names and branch lengths."""
    parts = {}
    for n in postorder(root):
        label = n.name or ""
        if re.search(r"[\s:,;()\[\]']", label):
            label = "'" + label.replace("'", "''") + "'"
        s = "(" + ",".join(parts[id(c)] for c in n.children) + ")" + label if n.children else label
        if n is not root and n.length is not None:
            s += ":" + _fmt_len(n.length)
        parts[id(n)] = s
    return parts[id(root)] + ";"


def to_arrays(root, taxon_names):
    """Post-order arrays for tphip_plan_desc: (parent int32[], branch_len float64[], leaf_taxon int32[]).

    Leaves are matched to alignment rows by name; a leaf missing from the alignment is an error (HyPhy
    would refuse a tree/alignment mismatch as well)."""
    import numpy as np
    order = postorder(root)
    idx = {id(n): i for i, n in enumerate(order)}
    row = {name: i for i, name in enumerate(taxon_names)}
    parent = np.full(len(order), -1, dtype=np.int32)
    blen = np.zeros(len(order), dtype=np.float64)
    leaf = np.full(len(order), -1, dtype=np.int32)
    for n in order:
        if n is not root:
            parent[idx[id(n)]] = idx[id(n.parent)]
            blen[idx[id(n)]] = n.length or 0.0
        if n.is_leaf():
            if n.name not in row:
                raise NewickError("tree leaf %r is not in the alignment" % n.name)
            leaf[idx[id(n)]] = row[n.name]
    return parent, blen, leaf
