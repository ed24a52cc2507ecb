# Offline experiment (CPU oracle): accuracy of the zero of f' predicted after TWO evaluations by
#   cubic  : Hermite interpolant of f' through (g0, h0), (g1, h1)                 (the product's exit (i))
#   quartic: the same plus the value condition  int f' = f1 - f0                 (one order higher)
# as a function of the predicted remaining step |t| and of the distance d between the points.
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as orc
from tapir_amd import synth

ntaxa, ncols, nsample = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rate_mean = float(sys.argv[4]) if len(sys.argv) > 4 else None
seed = int(sys.argv[5]) if len(sys.argv) > 5 else 11
d = synth.simulate(1, ncols, ntaxa, seed, **({} if rate_mean is None else {'rate_mean': rate_mean}))
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
parent, blen, leaf = np.asarray(pin["parent"]), np.asarray(pin["blen"]), np.asarray(pin["leaf"])
pi, exch = np.asarray(d["pi"][0]), np.asarray(d["exch"][0])
ref = orc.site_rates(st, parent, blen, leaf, pi, exch, start_mode=1)     # plain Newton to 1e-12
lam, U, Ui, kappa = orc.gtr_eigen(pi, exch)
ok = (ref["flag"] == 0) & (ref["rate"] > 0)
ustar = np.log(np.where(ok, ref["rate"], 1.0) / kappa)
# Fitch count + the product's start value
nn = len(parent)
sets = np.zeros((nn, ncols), dtype=np.uint8); cnt = np.zeros(ncols, dtype=np.int64)
kids = [[] for _ in range(nn)]
for n in range(nn):
    if parent[n] >= 0: kids[parent[n]].append(n)
for n in range(nn):
    if leaf[n] >= 0:
        m = st[leaf[n]] & 15; sets[n] = np.where(m == 0, 15, m)
    else:
        cur = sets[kids[n][0]].copy()
        for c in kids[n][1:]:
            inter = cur & sets[c]; miss = inter == 0; cnt += miss; cur = np.where(miss, cur | sets[c], inter)
        sets[n] = cur
resolved = ((st & 15) != 15).sum(0) - ((st & 15) == 0).sum(0)
chrono = blen[parent >= 0].sum()
B = np.maximum(2 * resolved - 3, 1)
mch = np.maximum(cnt, 1).astype(float)
mch = B * (-0.30 * np.log(1.0 - np.minimum(mch / B, 0.28) / 0.30))
u0s = np.clip(np.log(mch / (kappa * chrono * np.maximum(resolved, 1) / ntaxa)), -20, 8)
rows = []
extra = []
for c in np.flatnonzero(ok & (cnt > 0))[:nsample]:
    u0 = u0s[c]
    f0, g0, h0 = (x[0] for x in orc.column_curve(st, parent, blen, leaf, pi, exch, int(c), np.array([u0])))
    if not h0 < 0: continue
    m = float(cnt[c]); A = m - g0
    step = np.log(1 - g0 / h0) if 1 - g0 / h0 > 0 else -g0 / h0
    if ntaxa >= 32 and A > 0:
        sb = np.log(m / A) * (A / -h0)
        if abs(sb) <= 2: step = sb
    u1 = u0 + np.clip(step, -2, 2)
    f1, g1, h1 = (x[0] for x in orc.column_curve(st, parent, blen, leaf, pi, exch, int(c), np.array([u1])))
    if not h1 < 0: continue
    dd = u1 - u0
    # cubic (as in the kernel)
    c3 = 2.0 * (g0 - g1) / dd**3 + (h1 + h0) / dd**2
    c2 = (h1 - h0) / (2.0 * dd) + 1.5 * c3 * dd
    t = -g1 / h1
    for _ in range(3):
        p = g1 + t * (h1 + t * (c2 + t * c3)); dp = h1 + t * (2 * c2 + 3 * t * c3)
        if dp < 0: t -= p / dp
    # quartic: p(t) = g1 + h1 t + (a2/d^2) t^2 + (a3/d^3) t^3 + (a4/d^4) t^4, conditions at t = -d and the integral
    r1 = g0 - g1 + h1 * dd
    r2 = (h0 - h1) * dd
    r3 = (f1 - f0) / dd - g1 + 0.5 * h1 * dd
    Amat = np.array([[1.0, -1.0, 1.0], [-2.0, 3.0, -4.0], [1 / 3.0, -0.25, 0.2]])
    a2, a3, a4 = np.linalg.solve(Amat, np.array([r1, r2, r3]))
    q2, q3, q4 = a2 / dd**2, a3 / dd**3, a4 / dd**4
    tq = -g1 / h1
    for _ in range(4):
        p = g1 + tq * (h1 + tq * (q2 + tq * (q3 + tq * q4))); dp = h1 + tq * (2 * q2 + tq * (3 * q3 + 4 * tq * q4))
        if dp < 0: tq -= p / dp
    guard_c = abs(t * (c2 + t * c3)) < 0.02 * abs(h1)
    guard_q = abs(tq * (q2 + tq * (q3 + tq * q4))) < 0.02 * abs(h1)
    rows.append((abs(dd), abs(t), abs(u1 + t - ustar[c]), abs(tq), abs(u1 + tq - ustar[c]), abs(u1 - ustar[c]), float(guard_c), float(guard_q)))
    extra.append((dd, tq, q2, q3, q4, h1, h0, g1, u1, c2, c3))
R = np.array(rows)
print("columns", len(R), " |u1-u*| quantiles 50/90/99:", np.quantile(R[:, 5], [0.5, 0.9, 0.99]))
for lo, hi in ((0, 3e-4), (3e-4, 1e-3), (1e-3, 3e-3), (3e-3, 1e-2), (1e-2, 3e-2), (3e-2, 1.0)):
    sel = (R[:, 1] >= lo) & (R[:, 1] < hi)
    if sel.sum() == 0: continue
    print("predicted |t| in [%.0e, %.0e): %5d columns (%.1f %%)  error of the predicted zero: cubic max %.2e p99 %.2e | quartic max %.2e p99 %.2e"
          % (lo, hi, sel.sum(), 100 * sel.mean(), R[sel, 2].max(), np.quantile(R[sel, 2], 0.99), R[sel, 4].max(), np.quantile(R[sel, 4], 0.99)))
d_, tc, ec, tq_, eq, e1, gc, gq = R.T
def scen(name, acc, err):
    print("%-44s accepted %.1f %%  max err %.2e  p99.9 %.2e" % (name, 100 * acc.mean(), err[acc].max() if acc.any() else 0, np.quantile(err[acc], 0.999) if acc.any() else 0))
scen("cubic, current: t<1e-3, t*d<2e-4, guard", (tc < 1e-3) & (tc * d_ < 2e-4) & (tc < 0.5 * d_) & (gc > 0), ec)
for tmax, span in ((1e-2, 5e-4), (1e-2, 1e-3), (2e-2, 1e-3), (2e-2, 2e-3)):
    scen("quartic t<%.0e, t*d<%.0e, guard" % (tmax, span), (tq_ < tmax) & (tq_ * d_ < span) & (tq_ < 0.5 * d_) & (gq > 0), eq)
np.save("/tmp/two_point_%d_%s_%d.npy" % (ntaxa, str(rate_mean), seed), R)
np.save("/tmp/two_point_extra_%d_%s_%d.npy" % (ntaxa, str(rate_mean), seed), np.array(extra))
