"""Randomised check of stage 1 (model-averaged exchangeabilities) against the independent CPU restatement on small
random cases: trees with polytomies, 3..7 taxa, 40..160 columns, gaps and ambiguity codes, skewed frequencies.
The restatement takes seconds per locus: the GPU tests run a five-case window of it.
usage: python tools/fuzz_stage1.py [NCASES] [SEED] [FIRST]   (cases before FIRST are generated -- the random stream is the same -- but not evaluated)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import stage1_oracle
from tapir_amd import engine, newick, nexus, pipeline

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
CODES = np.array([1, 2, 4, 8, 15, 5, 10], dtype=np.uint8)


def random_tree(n):
    nodes = ["t%d:%g" % (i, rng.gamma(2.0, 0.05)) for i in range(n)]
    while len(nodes) > 1:
        k = 2 if (len(nodes) < 3 or rng.random() < 0.8) else 3
        idx = rng.choice(len(nodes), size=k, replace=False)
        kids = [nodes[i] for i in idx]
        nodes = [x for j, x in enumerate(nodes) if j not in set(idx.tolist())]
        nodes.append("(%s):%g" % (",".join(kids), rng.gamma(2.0, 0.05)))
    return nodes[0].rsplit(":", 1)[0] + ";"


def evolve(root, names, ncol, pi, exch):
    """simulate columns down the tree under GTR (numpy, tiny sizes)"""
    R = np.zeros((4, 4)); k = 0
    for i in range(4):
        for j in range(i + 1, 4):
            R[i, j] = R[j, i] = exch[k]; k += 1
    Q = R * pi[None, :]; np.fill_diagonal(Q, 0); np.fill_diagonal(Q, -Q.sum(1))
    w, V = np.linalg.eig(Q)
    Vi = np.linalg.inv(V)
    out = {}
    def rec(node, state):
        if node.is_leaf():
            out[node.name] = state; return
        for c in node.children:
            P = np.real((V * np.exp(w * c.length)[None, :]) @ Vi)
            P = np.clip(P, 0, None); P /= P.sum(1, keepdims=True)
            u = rng.random(ncol)
            cs = (u[:, None] > np.cumsum(P[state], axis=1)).sum(1).clip(max=3)
            rec(c, cs)
    rec(root, rng.choice(4, size=ncol, p=pi))
    return np.stack([(1 << out[n]).astype(np.uint8) for n in names])


bad = 0
for case in range(ncases):
    nt = int(rng.integers(3, 8))
    root = newick.parse(random_tree(nt))
    names = [x.name for x in newick.leaves(root)]
    parent, blen, leaf = newick.to_arrays(root, names)
    ncol = int(rng.integers(40, 161))
    pi_true = rng.dirichlet(np.full(4, 8.0))
    ex_true = np.exp(rng.normal(0, 0.5, 6)); ex_true[1] = 1.0
    st = evolve(root, names, ncol, pi_true, ex_true)
    noise = rng.random(st.shape) < 0.05
    st = np.where(noise, rng.choice(CODES, size=st.shape), st).astype(np.uint8)
    off = np.array([0, ncol])
    if case < first:
        continue
    pi = nexus.base_frequencies_from_histogram(engine.state_histogram(st, off))
    t0 = time.time()
    got = pipeline.model_averaged_exchangeabilities(engine, st, off, pi, nt, parent, blen, leaf, 5, [1], [[0, 2]], 1.0)[0]
    t1 = time.time()
    ref = stage1_oracle.model_averaged(st, parent, np.asarray(blen), leaf, pi[0])
    t2 = time.time()
    rel = np.max(np.abs(got - ref["exch"]) / ref["exch"])
    line = "case %d: %d taxa x %d cols: gpu %.2f s, restatement %.1f s, max rel diff %.2e" % (case, nt, ncol, t1 - t0, t2 - t1, rel)
    if not (rel < 1e-3):
        bad += 1
        line += "  <-- gpu %s  ref %s" % (np.round(got, 4), np.round(ref["exch"], 4))
    print(line, flush=True)
print("%d cases, %d beyond 1e-3" % (ncases - first, bad))
