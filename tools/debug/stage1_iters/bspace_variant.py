"""b-space variant: branch coordinates are the lengths themselves with a lower bound; coordinates sitting on the bound with the
gradient pointing outwards are frozen for the iteration (active set); step limits relative to b."""
import sys, time, inspect, numpy as np
from harness import *
BMIN = np.exp(stage1.LOG_BLEN_MIN)
src = inspect.getsource(stage1._LBFGS)
def rep(old, new):
    global src
    assert old in src, old
    src = src.replace(old, new)
# freeze bound coordinates: gradient entries zeroed before the two-loop, direction zero there
rep('''            q = gl.copy()
''', '''            frozen = (xl <= self.lo * (1 + 1e-9) + 0 * xl) & (gl > 0)
            frozen[:, :5] = False
            self.nfrozen = int(frozen.sum())
            gl = np.where(frozen, 0.0, gl)
            q = gl.copy()
''')
rep('''            d = np.clip(-r, -MAX_LOG_STEP, MAX_LOG_STEP)   # per coordinate: one runaway parameter (a branch collapsing
            gd = np.einsum("pd,pd->p", gl, d)              # to zero has almost no curvature) must not shrink the others' step
''', '''            d = -r
            d[:, :5] = np.clip(d[:, :5], -MAX_LOG_STEP, MAX_LOG_STEP)
            bb = xl[:, 5:]
            d[:, 5:] = np.clip(d[:, 5:], -bb, bb * (np.exp(MAX_LOG_STEP) - 1.0))    # down to zero, up by e^2
            d = np.where(frozen, 0.0, d)
            gd = np.einsum("pd,pd->p", gl, d)
''')
rep('''            dmax = np.abs(d).max(1)
            t = np.minimum(step0, MAX_LOG_STEP / np.maximum(dmax, 1e-300))
''', '''            dmax = np.abs(d[:, :5]).max(1)
            t = np.minimum(step0, MAX_LOG_STEP / np.maximum(dmax, 1e-300))
''')
rep('''            gmax = np.abs(gl).max(1)
''', '''            gmax = np.abs(gl * np.where(np.arange(gl.shape[1]) >= 5, xl, 1.0)).max(1)   # in log units, as before
''')
rep('''            y_ = gx - gl
''', '''            y_ = np.where(frozen, 0.0, gx - gl)
            s_ = np.where(frozen, 0.0, s_)
''')
rep("                gamma = gamma[keep]\n", "                gamma = gamma[keep]\n                frozen = frozen[keep]\n")
ns = dict(vars(stage1)); exec(src, ns); BL = ns["_LBFGS"]

class S1(stage1.Stage1):
    def _grm_point(self, X, idx):
        exch = self._exch_from_free(X[:, :5])
        scale = 1.0 / stage1.total_factor(self.pi[idx], exch)
        vecs = np.zeros((len(idx), self.nn))
        vecs[:, self.branches] = X[:, 5:]
        return exch, scale, vecs
    def _grm_value_and_grad(self, X, idx):
        Xl = X.copy(); Xl[:, 5:] = np.log(X[:, 5:])
        orig = stage1.Stage1._grm_point
        f, g, h = stage1.Stage1._grm_value_and_grad(self, Xl, idx) if False else self._vg_log(Xl, idx)
        b = X[:, 5:]
        gb = g[:, 5:] / b
        hb = (h[:, 5:] - g[:, 5:]) / (b * b)
        g2 = g.copy(); h2 = h.copy()
        g2[:, 5:] = gb; h2[:, 5:] = np.where(hb > 0, hb, np.nan)
        return f, g2, h2
    def _vg_log(self, Xl, idx):
        # the parent's routine works on log b: give it a log point through a temporary parent-style _grm_point
        saved = self._grm_point
        self._grm_point = lambda X, i: stage1.Stage1._grm_point(self, X, i)
        try:
            return stage1.Stage1._grm_value_and_grad(self, Xl, idx)
        finally:
            self._grm_point = saved
    def fit_grm(self, maxit=None):
        L = self.plan.nloci
        maxit = max(300, 4 * (5 + len(self.branches)))
        x0 = np.zeros((L, 5 + len(self.branches)))
        x0[:, 5:] = self.initial_branch_lengths()[:, self.branches] * stage1.total_factor(self.pi, np.ones(6))[:, None]
        lo = np.concatenate([np.full(5, stage1.LOG_RATE_MIN), np.full(len(self.branches), BMIN)])
        hi = np.concatenate([np.full(5, stage1.LOG_RATE_MAX), np.full(len(self.branches), np.exp(stage1.LOG_BLEN_MAX))])
        self._kicks = np.zeros(L, dtype=np.int64)
        opt = BL(self._grm_value, self._grm_value_and_grad, x0, maxit=maxit, lo=lo, hi=hi, escape=None)
        x, f = opt.run()
        self.grm_iters = opt.iters
        exch = self._exch_from_free(x[:, :5])
        t = np.zeros((L, self.nn))
        t[:, self.branches] = x[:, 5:] / stage1.total_factor(self.pi, exch)[:, None]
        return exch, t, -f

if __name__ == "__main__":
    nloci, ncols, ntaxa = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    plan, st, pi, pin = make(nloci, ncols, ntaxa, 7)
    s1 = S1(plan, st, pi, pin["parent"], pin["blen"], analytic=True, device_fit=False)
    t = time.perf_counter()
    exch, tt, lnl = s1.fit_grm()
    print("bspace iters", s1.grm_iters.tolist(), "grads", s1.ngrads, "values", s1.nevals, "sec %.1f" % (time.perf_counter() - t))
    print("  lnl", np.round(lnl, 6).tolist(), flush=True)
