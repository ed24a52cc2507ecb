"""The trap of the log parametrisation between 1e-6 and the escape length: Stage1._grm_escape looks only at branches below
1e-6, a branch that crawled to ~5e-6 while its gradient was positive and then wants to grow has g_u = b g_b ~ 1e-5 and passes the
gradient test.  Variant: the same first-order test for every branch below a quarter of the escape length."""
import sys, time, numpy as np
from harness import *

class S1(stage1.Stage1):
    def _grm_escape(self, idx, X, G):
        Xn = X.copy()
        logb = X[:, 5:]
        b = np.exp(logb)
        slope = G[:, 5:] / b
        fscale = 1.0 + np.abs(self._last_f[idx]) if hasattr(self, "_last_f") else 1.0
        kick = (b < 0.25 * stage1.ESCAPE_LENGTH) & (slope * (stage1.ESCAPE_LENGTH - b) < -1e-7 * np.reshape(fscale, (-1, 1)))
        kick &= self._kicks[idx][:, None] < 3
        Xn[:, 5:] = np.where(kick, np.log(stage1.ESCAPE_LENGTH), logb)
        rslope = G[:, :5] / np.exp(X[:, :5])
        rkick = (X[:, :5] < stage1.LOG_RATE_MIN + 1.0) & (rslope * stage1.ESCAPE_RATE < -1e-7 * np.reshape(fscale, (-1, 1)))
        rkick &= self._kicks[idx][:, None] < 3
        Xn[:, :5] = np.where(rkick, np.log(stage1.ESCAPE_RATE), X[:, :5])
        moved = kick.any(axis=1) | rkick.any(axis=1)
        self._kicks[idx] += moved
        self.nkicks = getattr(self, "nkicks", 0) + int(moved.sum())
        return Xn, moved

if __name__ == "__main__":
    nloci, ncols, ntaxa = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    plan, st, pi, pin = make(nloci, ncols, ntaxa, 7)
    s1 = S1(plan, st, pi, pin["parent"], pin["blen"], analytic=True, device_fit=False)
    t = time.perf_counter()
    exch, tt, lnl = s1.fit_grm()
    print("wide escape iters", s1.grm_iters.tolist(), "grads", s1.ngrads, "values", s1.nevals, "kicks", getattr(s1, "nkicks", 0), "sec %.1f" % (time.perf_counter() - t))
    print("  lnl", np.round(lnl, 6).tolist(), flush=True)
