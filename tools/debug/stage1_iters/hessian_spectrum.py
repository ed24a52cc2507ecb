import sys, time, numpy as np
from harness import *
nloci, ncols, ntaxa = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
plan, st, pi, pin = make(nloci, ncols, ntaxa, 7)
s1 = stage1.Stage1(plan, st, pi, pin["parent"], pin["blen"], analytic=True, device_fit=False)
exch, tt, lnl = s1.fit_grm()
tf = stage1.total_factor(s1.pi, exch)
for l in range(nloci):
    x = np.concatenate([np.log(exch[l, [0, 2, 3, 4, 5]]), np.log(np.maximum(tt[l, s1.branches] * tf[l], 1e-10))])
    D = len(x); h = 1e-3
    idx = np.array([l])
    f = lambda X: s1._grm_value(X, np.repeat(idx, len(X)))
    f0 = f(x[None])[0]
    E = np.eye(D) * h
    fp = f(x[None] + E); fm = f(x[None] - E)
    H = np.zeros((D, D))
    for i in range(D):
        H[i, i] = (fp[i] + fm[i] - 2 * f0) / h ** 2
    ii, jj = np.triu_indices(D, 1)
    fpp = f(x[None] + E[ii] + E[jj]); fmm = f(x[None] - E[ii] - E[jj])
    H[ii, jj] = (fpp + fmm - fp[ii] - fp[jj] - fm[ii] - fm[jj] + 2 * f0) / (2 * h * h)
    H[jj, ii] = H[ii, jj]
    np.save("/tmp/H_%d_%d_%d.npy" % (ntaxa, ncols, l), H)
    w = np.linalg.eigvalsh(H)
    def cond(P):
        Li = np.linalg.inv(np.linalg.cholesky(P))
        w = np.linalg.eigvalsh(Li @ H @ Li.T)
        return w
    d = np.maximum(np.diag(H), 1e-8)
    wd = cond(np.diag(d))
    P = np.diag(d); P[:5, :5] = H[:5, :5]
    wb = cond(P)
    print("locus", l, "x rates", np.round(x[:5], 2), "min logb %.1f" % x[5:].min())
    print("  H eig min/max %.3g %.3g" % (w[0], w[-1]), " diag-precond eig: %.3g .. %.3g  (n<0.3: %d, n>3: %d)" % (wd[0], wd[-1], (wd < 0.3).sum(), (wd > 3).sum()),
          " block-precond eig: %.3g .. %.3g (n<0.3: %d, n>3: %d)" % (wb[0], wb[-1], (wb < 0.3).sum(), (wb > 3).sum()))
    print("  diag-precond spectrum", np.round(wd, 2))
    print("  block-precond spectrum", np.round(wb, 2))
