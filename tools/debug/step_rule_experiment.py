# Offline experiment (CPU oracle): accuracy of the first step from a parsimony-like start under
#   A: log(1 - f'/f'')                         (the product's rule: model m u - a e^u + c fitted to f', f'')
#   B: (1/b) log(m / (m - f')), b = -f''/(m - f')   (model m u - a e^{b u} + c with m = the column's parsimony length)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as orc
from tapir_amd import synth

ntaxa, ncols = int(sys.argv[1]), int(sys.argv[2])
d = synth.simulate(1, ncols, ntaxa, 11)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
parent, blen, leaf = np.asarray(pin["parent"]), np.asarray(pin["blen"]), np.asarray(pin["leaf"])
pi, exch = np.asarray(d["pi"][0]), np.asarray(d["exch"][0])
res = orc.site_rates(st, parent, blen, leaf, pi, exch)
lam, U, Ui, kappa = orc.gtr_eigen(pi, exch)
ok = (res["flag"] == 0) & (res["rate"] > 0)
ustar = np.log(np.where(ok, res["rate"], 1.0) / kappa)
# Fitch parsimony length per column
nn = len(parent)
sets = np.zeros((nn, ncols), dtype=np.uint8)
cnt = np.zeros(ncols, dtype=np.int64)
kids = [[] for _ in range(nn)]
for n in range(nn):
    if parent[n] >= 0:
        kids[parent[n]].append(n)
for n in range(nn):  # post-order arrays: children precede parents
    if leaf[n] >= 0:
        m = st[leaf[n]] & 15
        sets[n] = np.where(m == 0, 15, m)
    else:
        cur = sets[kids[n][0]].copy()
        for c in kids[n][1:]:
            inter = cur & sets[c]
            miss = inter == 0
            cnt += miss
            cur = np.where(miss, cur | sets[c], inter)
        sets[n] = cur
rng = np.random.default_rng(3)
idx = np.flatnonzero(ok & (cnt > 0))[:1500]
errA, errB, d0s = [], [], []
for c in idx:
    d0 = rng.uniform(-0.35, 0.25)
    u0 = ustar[c] + d0
    f, g, h = orc.column_curve(st, parent, blen, leaf, pi, exch, int(c), np.array([u0]))
    g, h = g[0], h[0]
    if not (h < 0):
        continue
    qa = 1.0 - g / h
    if qa <= 0:
        continue
    sA = np.log(qa)
    m = float(cnt[c])
    A = m - g
    if A <= 0:
        continue
    b = -h / A
    sB = np.log(m / A) / b
    errA.append(u0 + sA - ustar[c]); errB.append(u0 + sB - ustar[c]); d0s.append(d0)
errA, errB = np.abs(np.array(errA)), np.abs(np.array(errB))
print("columns %d; |u1 - u*| quantiles 50/90/99 %%:" % len(errA))
print("  rule A:", np.quantile(errA, [0.5, 0.9, 0.99]))
print("  rule B:", np.quantile(errB, [0.5, 0.9, 0.99]))
print("  fraction with |u1 - u*| < 0.03 (second step small enough for the two-point exit, roughly): A %.3f  B %.3f" % ((errA < 0.03).mean(), (errB < 0.03).mean()))
print("  fraction B better than A: %.3f" % (errB < errA).mean())
