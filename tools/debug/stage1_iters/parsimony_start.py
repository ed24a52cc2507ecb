"""Start of the general model's branch lengths from Fitch parsimony (changes on the branch / columns) instead of the input
tree's shape times the best of a grid of scales (Stage1.initial_branch_lengths)."""
import os, sys, time, numpy as np
from harness import *

def fitch_branch_changes(states, parent, leaf):
    """changes[b] for every node with a branch; states [ntaxa, ncols] 4-bit masks (0 = gap = 15)."""
    nn = len(parent)
    kids = [[] for _ in range(nn)]
    for v, p in enumerate(parent):
        if p >= 0:
            kids[p].append(v)
    root = int(np.flatnonzero(np.asarray(parent) < 0)[0])
    order = []
    stack = [root]
    while stack:
        v = stack.pop(); order.append(v); stack.extend(kids[v])
    sets = [None] * nn
    for v in reversed(order):
        if not kids[v]:
            m = states[leaf[v]] & 15
            sets[v] = np.where(m == 0, 15, m).astype(np.uint8)
        else:
            a = sets[kids[v][0]]
            for c in kids[v][1:]:
                inter = a & sets[c]
                a = np.where(inter != 0, inter, a | sets[c])
            sets[v] = a
    low = lambda m: m & (~m + 1).astype(np.uint8) if False else (m & (-m.astype(np.int16)).astype(np.uint8))
    chosen = [None] * nn
    changes = np.zeros(nn)
    pairs = np.zeros((4, 4))
    chosen[root] = low(sets[root])
    bit = np.zeros(16, np.int64); bit[[1, 2, 4, 8]] = [0, 1, 2, 3]
    for v in order[1:]:
        p = parent[v]
        keep = (sets[v] & chosen[p]) != 0
        chosen[v] = np.where(keep, chosen[p], low(sets[v]))
        changes[v] = (~keep).sum()
        np.add.at(pairs, (bit[chosen[p][~keep]], bit[chosen[v][~keep]]), 1.0)
    fitch_branch_changes.pairs = pairs + pairs.T
    return changes

class S1(stage1.Stage1):
    def initial_branch_lengths(self):
        L = self.plan.nloci
        out = np.zeros((L, self.nn))
        tf = stage1.total_factor(self.pi, np.ones(6))
        for l in range(L):
            st = self.states[:, self.plan.off[l]:self.plan.off[l + 1]]
            ch = fitch_branch_changes(st, self.plan.parent, self.plan.leaf)
            b = np.maximum(ch, FLOOR) / max(st.shape[1], 1)
            out[l, self.branches] = b[self.branches] / tf[l]
            pr = fitch_branch_changes.pairs
            pi = self.pi[l]
            r = np.array([(pr[i, j] + 0.5) / (pi[i] * pi[j]) for i, j in ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))])
            self.rate_start[l] = np.log(np.clip(r / r[1], 0.05, 20.0))[[0, 2, 3, 4, 5]]
        return out

    def fit_grm(self, maxit=None):
        self.rate_start = np.zeros((self.plan.nloci, 5))
        if not os.environ.get("RATE_START"):
            return super().fit_grm(maxit)
        # same as the parent's, with the five log-rates started from the parsimony pair counts
        real = stage1._LBFGS
        outer = self
        class Wrapped(real):
            def __init__(self, value, vg, x0, **k):
                x0 = np.array(x0); x0[:, :5] = outer.rate_start
                # b = t * totalFactor: lengths were divided by totalFactor(ones); keep b as it is
                super().__init__(value, vg, x0, **k)
        stage1._LBFGS = Wrapped
        try:
            return super().fit_grm(maxit)
        finally:
            stage1._LBFGS = real

if __name__ == "__main__":
    nloci, ncols, ntaxa = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    FLOOR = float(sys.argv[4]) if len(sys.argv) > 4 else 0.3
    if os.environ.get("JUMP"):   # combine with jump_variant's iteration: JUMP="THRESHOLD FRACTION"
        argv = sys.argv
        sys.argv = argv[:4] + os.environ["JUMP"].split()
        import jump_variant
        sys.argv = argv
        stage1._LBFGS = jump_variant.NewLBFGS
    plan, st, pi, pin = make(nloci, ncols, ntaxa, 7)
    s1 = S1(plan, st, pi, pin["parent"], pin["blen"], analytic=True, device_fit=False)
    t = time.perf_counter()
    exch, tt, lnl = s1.fit_grm()
    print("parsimony start iters", s1.grm_iters.tolist(), "grads", s1.ngrads, "values", s1.nevals, "sec %.1f" % (time.perf_counter() - t))
    print("  lnl", np.round(lnl, 6).tolist(), flush=True)
