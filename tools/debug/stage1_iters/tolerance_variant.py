"""Stopping rule of the general model's fit: iterations and what is lost (lnL, rates) with looser ftol / ptol."""
import sys, time, numpy as np
from harness import *
nloci, ncols, ntaxa = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
real = stage1._LBFGS
res = {}
for ftol, ptol in ((1e-10, 1e-9), (1e-8, 1e-8), (1e-7, 1e-7)):
    class Loose(real):
        def __init__(self, *a, **k):
            k["ftol"], k["ptol"] = ftol, ptol
            super().__init__(*a, **k)
    stage1._LBFGS = Loose
    plan, st, pi, pin = make(nloci, ncols, ntaxa, 7)
    s1 = stage1.Stage1(plan, st, pi, pin["parent"], pin["blen"], analytic=True, device_fit=False)
    exch, tt, lnl = s1.fit_grm()
    res[(ftol, ptol)] = (s1.grm_iters.copy(), lnl, exch)
    b = res[(1e-10, 1e-9)]
    print("ftol %.0e ptol %.0e: iters mean %.1f max %d  grads %d values %d | lnL loss max %.2e  rates max rel diff %.2e" % (
        ftol, ptol, s1.grm_iters.mean(), s1.grm_iters.max(), s1.ngrads, s1.nevals, (b[1] - lnl).max(), np.abs(exch / b[2] - 1).max()), flush=True)
