import os, sys, time, inspect, numpy as np
from harness import *
src = inspect.getsource(stage1._LBFGS)
# 1) coordinates (>= 5: log branch lengths) whose Newton step in b itself reaches zero jump to the lower bound
old = '''            d = np.clip(-r, -MAX_LOG_STEP, MAX_LOG_STEP)   # per coordinate: one runaway parameter (a branch collapsing
            gd = np.einsum("pd,pd->p", gl, d)              # to zero has almost no curvature) must not shrink the others' step
'''
new = '''            d = np.clip(-r, -MAX_LOG_STEP, MAX_LOG_STEP)
            # collapse test in the branch length b = exp(u) itself: g_b = g_u / b, h_b = (h_u - g_u) / b^2; Newton in b: b - g_b / h_b <= 0
            gu, hu = gl, hl
            jump = np.zeros_like(d, bool)
            if JUMP:
                hb_b2 = hu - gu                      # h_b * b^2
                jump = np.isfinite(hu) & (gu > 0) & (hb_b2 <= JUMP_FRAC * gu) & (xl < JUMP_BELOW) & (xl > self.lo + 1e-9)
                jump[:, :5] = False
                d = np.where(jump, np.maximum(-gu / np.maximum(hu, 1e-300), -1.0) * 0 - 1.0, d)   # for the model: one log-unit
            gd = np.einsum("pd,pd->p", gl, d)
            self.njump = getattr(self, "njump", 0) + int(jump.sum())
'''
assert old in src; src = src.replace(old, new)
old = '''                xt = np.clip(xl[pending] + t[pending, None] * d[pending], self.lo, self.hi)
'''
new = '''                xt = np.clip(xl[pending] + t[pending, None] * d[pending], self.lo, self.hi)
                xt = np.where(jump[pending], np.broadcast_to(self.lo, xt.shape), xt)
'''
assert old in src; src = src.replace(old, new)
# the jumped coordinates leave the correction pair
old = '''            s_ = xnew - xl
            y_ = gx - gl
'''
new = '''            s_ = np.where(jump, 0.0, xnew - xl)
            y_ = np.where(jump, 0.0, gx - gl)
'''
assert old in src; src = src.replace(old, new)
old = "                gamma = gamma[keep]\n"
assert old in src; src = src.replace(old, old + "                jump = jump[keep]\n")
if os.environ.get("ANYTIME_ESCAPE"):   # the escape test on every iteration (not only at stopping points), history kept
    o = "            if stop.any() and self.escape is not None:\n"
    assert o in src
    src = src.replace(o, "            if self.escape is not None:\n                stop_real = stop\n                stop = np.ones_like(stop)\n")
    o = """                    rho[:, mi] = 0.0
                    nhist[mi] = 0
                    last_df[mi] = np.inf
                    continue          # recompute every direction from the new points"""
    assert o in src
    src = src.replace(o, """                    last_df[mi] = np.inf
                    continue          # recompute every direction from the new points
                stop = stop_real""")
ns = dict(vars(stage1)); ns["JUMP"] = True; ns["JUMP_FRAC"] = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0; ns["JUMP_BELOW"] = float(sys.argv[4]) if len(sys.argv) > 4 else -5.0
exec(src, ns)
NewLBFGS = ns["_LBFGS"]
if __name__ == "__main__":
    nloci, ncols, ntaxa = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    for name, cls in (("jump", NewLBFGS),):
        stage1._LBFGS = cls
        plan, st, pi, pin = make(nloci, ncols, ntaxa, 7)
        s1 = stage1.Stage1(plan, st, pi, pin["parent"], pin["blen"], analytic=True, device_fit=False)
        t = time.perf_counter()
        exch, tt, lnl = s1.fit_grm()
        print(name, "iters", s1.grm_iters.tolist(), "grads", s1.ngrads, "values", s1.nevals, "sec %.1f" % (time.perf_counter() - t))
        print("  lnl", np.round(lnl, 6).tolist())
        print("  exch0", np.round(exch[0], 5).tolist())
