"""HyPhy stage 1 (model-averaged GTR exchangeabilities per locus, models_and_rates.bf:405-897): entry point and a
second-opinion host optimiser.

What stage 1 is (tapir/data/models_and_rates.bf, "next" row #1 of SURVEY.md section 8f):

  bf:487-520   general reversible model (012345): `Optimize(lf_MLES, lf)` over the 5 exchangeabilities
               AC, AT, CG, CT, GT (AG = 1) AND every branch length;
  bf:522-540   stash the fitted branch lengths in expected substitutions (t_b * totalFactor,
               totalFactor = 2 sum_{i<j} pi_i pi_j r_ij);
  bf:542-661   the other 202 set partitions of the six rates (restricted growth strings, in the script's loop
               order): rates of one class are equal, the class holding AG is 1, branch lengths are the stashed
               ones divided by the model's own totalFactor (bf:613-619); optimise the <= 4 free class rates;
  bf:806-831   Akaike weights from c_m = 2 (np_m - lnL_m + log sites): w_m proportional to exp(lnL_m - k_m),
               k_m = number of free rates (everything else in np_m is the same for all models);
  bf:838-847   modelAveragedRates[v] = sum_m w_m rate_m[v]  ->  the AC, AT, CG, CT, GT handed to stage 2.

The PRODUCT path is one engine call, `Plan.stage1_fit` -> tphip_stage1_fit (csrc/stage1_driver.hip +
csrc/stage1_opt_kernels.hpp): likelihoods, gradients AND the optimisers are device kernels, the host sequences
launches; `model_averaged_exchangeabilities` below goes there.

The `Stage1` class is the round-1/2 implementation -- the same likelihood kernels driven by a batched L-BFGS written in
numpy -- kept as an independent second opinion for the tests (tests/test_gpu_stage1.py) and for A/B timing
(tools/stage1_timing.py compare).  It is not on the product path.

Parity: HyPhy's own optimiser and its results for this stage are pinned by no fixture of the reference
("parity unpinned"); the stage is checked against an independent CPU restatement
(tests/test_gpu_stage1.py: oracle likelihood + scipy L-BFGS-B) at 1e-3 relative on the averaged rates.
"""
import os
import time

import numpy as np

RATE_ORDER = ("AC", "AG", "AT", "CG", "CT", "GT")
# Bounds (HyPhy: rates in [0, 10000], lengths >= 0).  The lower ones keep every transition probability
# q_ij * t >= ~1e-15 above the rounding noise (~2e-16) of its eigen-sum; below them a rate or a length is zero for
# every purpose of this stage.
LOG_RATE_MIN, LOG_RATE_MAX = -7.0, 9.2      # exchangeabilities within [9e-4, 1e4]
LOG_BLEN_MIN, LOG_BLEN_MAX = -23.0, 4.0     # branch lengths within [1e-10, 55]
MAX_LOG_STEP = 2.0                          # largest move of a log-parameter in one iteration
ESCAPE_RATE = 0.05                          # where a wrongly collapsed exchangeability is put back (Stage1._sub_escape)
ESCAPE_LENGTH = 1e-3                        # where a wrongly collapsed branch is put back (Stage1._grm_escape)
PRUNE_NATS = 21.0                           # see Stage1.fit_submodels: a model this far behind weighs < e^-21 = 8e-10
_PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))  # AC AG AT CG CT GT as (i, j) over A C G T


def model_strings():
    """The 203 rate-class models as restricted growth strings over (AC, AG, AT, CG, CT, GT): '012345' (the general
    reversible model, fitted first, bf:487) followed by the other 202 in the order of the loops at bf:544-566."""
    out = ["012345"]
    for v2 in range(0, 2):
        for v3 in range(0, v2 + 2):
            ub4 = v3 if v3 > v2 else v2
            for v4 in range(0, ub4 + 2):
                ub5 = v4 if v4 >= ub4 else ub4
                for v5 in range(0, ub5 + 2):
                    ub6 = v5 if v5 > ub5 else ub5
                    for v6 in range(0, ub6 + 2):
                        if v6 == 5:
                            break
                        out.append("0%d%d%d%d%d" % (v2, v3, v4, v5, v6))
    assert len(out) == 203 and len(set(out)) == 203
    return out


def model_design(strings=None):
    """For each model: class index of every rate with the AG class mapped to -1 (fixed at 1) and the free classes
    renumbered 0..k-1.  Returns (cls int[203, 6], k int[203])."""
    strings = strings or model_strings()
    cls = np.zeros((len(strings), 6), dtype=np.int64)
    k = np.zeros(len(strings), dtype=np.int64)
    for m, s in enumerate(strings):
        ag = s[1]
        free = []
        for c in s:
            if c != ag and c not in free:
                free.append(c)
        k[m] = len(free)
        cls[m] = [-1 if c == ag else free.index(c) for c in s]
    return cls, k


def total_factor(pi, exch):
    """totalFactor of bf:531-534 = expected substitutions per unit t: 2 sum_{i<j} pi_i pi_j r_ij."""
    pi = np.asarray(pi, dtype=np.float64)
    exch = np.asarray(exch, dtype=np.float64)
    w = np.stack([2.0 * pi[..., i] * pi[..., j] for i, j in _PAIRS], axis=-1)
    return (w * exch).sum(axis=-1)


def _backtrack(t, ft, f0, gd):
    """Next trial step after `t` failed the Armijo test: the minimiser of the parabola through f(0) = f0, f'(0) = gd < 0 and
    f(t) = ft, kept within [0.1 t, 0.5 t] (0.25 t where the trial value is not finite).  Plain halving needed 6-8 rounds --
    6-8 likelihood calls one after the other -- whenever one locus of a batch wanted a step of 1/100."""
    den = 2.0 * (ft - f0 - gd * t)
    good = np.isfinite(ft) & (den > 0)
    tq = np.where(good, -gd * t * t / np.where(good, den, 1.0), 0.25 * t)
    return np.clip(tq, 0.1 * t, 0.5 * t)


class _LBFGS:
    """Batched L-BFGS (maximisation written as minimisation of -lnL) over P independent problems of dimension D.
    `value(X, idx)` evaluates points X[len(idx), D] of problems idx; `value_and_grad(X, idx)` adds gradients."""

    def __init__(self, value, value_and_grad, x0, active=None, history=8, maxit=200, ftol=1e-10, gtol=2e-6, ptol=1e-9,
                 lo=-np.inf, hi=np.inf, prune=None, escape=None):
        self.escape = escape  # optional: escape(problem indices, x, g) -> (x', moved mask): called on problems about to stop
        self.ptol = ptol      # stop when the quasi-Newton model predicts a further decrease below ptol * (1 + |f|)
        self.prune = prune   # optional: prune(problem indices, f, last decrease) -> mask of problems to abandon
        self.value, self.vg = value, value_and_grad
        self.x = np.array(x0, dtype=np.float64)
        P, D = self.x.shape
        self.lo, self.hi = np.broadcast_to(lo, (D,)), np.broadcast_to(hi, (D,))
        self.active = np.ones((P, D), bool) if active is None else active
        self.m, self.maxit, self.ftol, self.gtol = min(history, max(D, 2)), maxit, ftol, gtol   # D pairs span R^D

    def _vg(self, x, idx):
        out = self.vg(x, idx)
        return out if len(out) == 3 else (out[0], out[1], None)

    def run(self):
        P, D = self.x.shape
        live = np.arange(P)
        f, g, hd = self._vg(self.x, live)
        g = np.where(self.active, g, 0.0)
        hdiag = np.full((P, D), np.nan) if hd is None else hd    # per-parameter curvature (NaN: unknown)
        # correction pairs in a ring shared by all problems: slot it % m holds the pair of iteration it, rho = 0 marks
        # "no pair" (problem not live, or curvature condition failed), which makes the two-loop skip it
        m = self.m
        S = np.zeros((m, P, D)); Y = np.zeros((m, P, D)); rho = np.zeros((m, P))
        gamma_all = np.ones(P)
        nhist = np.zeros(P, dtype=np.int64)
        converged = np.zeros(P, bool)
        last_df = np.full(P, np.inf)
        fresh = np.zeros(P, bool)        # history was just reset after a failed line search
        self.iters = np.zeros(P, dtype=np.int64)
        for it in range(self.maxit):
            live = np.flatnonzero(~converged)
            if live.size == 0:
                break
            all_live = live.size == P
            sl = slice(None) if all_live else live
            xl, fl, gl = self.x[sl], f[sl], g[sl]
            # two-loop recursion on the live problems, newest pair first
            slots = [(it - 1 - j) % m for j in range(min(it, m))]
            q = gl.copy()
            alpha = {}
            for i in slots:
                a = rho[i, sl] * np.einsum("pd,pd->p", S[i, sl], q)
                alpha[i] = a
                q -= a[:, None] * Y[i, sl]
            gamma = gamma_all[sl]
            # initial inverse Hessian: 1 / curvature where the objective supplies a positive one (the gradient kernel's
            # Hessian diagonal for the branch lengths), the usual sy/yy scalar elsewhere
            hl = hdiag[sl]
            have_h = np.isfinite(hl).any(axis=1)
            if have_h.any():
                r = np.where(np.isfinite(hl) & (hl > 1e-12), 1.0 / np.where(hl > 1e-12, hl, 1.0), gamma[:, None]) * q
            else:
                r = gamma[:, None] * q
            for i in reversed(slots):
                bcoef = rho[i, sl] * np.einsum("pd,pd->p", Y[i, sl], r)
                r += (alpha[i] - bcoef)[:, None] * S[i, sl]
            d = np.clip(-r, -MAX_LOG_STEP, MAX_LOG_STEP)   # per coordinate: one runaway parameter (a branch collapsing
            gd = np.einsum("pd,pd->p", gl, d)              # to zero has almost no curvature) must not shrink the others' step
            bad = ~(gd < 0)
            if bad.any():
                d[bad] = -gl[bad]
                gd[bad] = -(gl[bad] ** 2).sum(1)
            # Convergence: the last step gained (almost) nothing AND the quasi-Newton model, whose metric knows about
            # flat directions (branches collapsing to zero have tiny gradients and tiny curvatures alike), predicts that
            # the next one will not either.  A test on the raw gradient norm stops early exactly on those directions.
            scale_f = 1.0 + np.abs(fl)
            gmax = np.abs(gl).max(1)
            stop = (last_df[sl] <= self.ftol * scale_f) & (-gd <= self.ptol * scale_f) & (gmax <= self.gtol * scale_f)
            stop |= gmax <= 1e-9
            if stop.any() and self.escape is not None:
                # a stopping point may be a trap of the parametrisation rather than an optimum: let the objective move it
                xs, moved = self.escape(live[stop], xl[stop], gl[stop])
                if moved.any():
                    mi = live[stop][moved]
                    self.x[mi] = xs[moved]
                    fm, gm, hm = self._vg(self.x[mi], mi)
                    f[mi] = fm
                    g[mi] = np.where(self.active[mi], gm, 0.0)
                    if hm is not None:
                        hdiag[mi] = hm
                    rho[:, mi] = 0.0
                    nhist[mi] = 0
                    last_df[mi] = np.inf
                    continue          # recompute every direction from the new points
            if stop.any():
                if os.environ.get("TPHIP_STAGE1_TRACE") and P <= 16:
                    print("it %d stop %s last_df %s -gd %s" % (it, live[stop].tolist(), last_df[live[stop]].tolist(), (-gd[stop]).tolist()), flush=True)
                converged[live[stop]] = True
                if stop.all():
                    continue
                keep = ~stop
                live = live[keep]
                sl = live
                xl, fl, gl, d, gd, hl, have_h = xl[keep], fl[keep], gl[keep], d[keep], gd[keep], hl[keep], have_h[keep]
                gamma = gamma[keep]
            # first iteration: a cautious step length
            step0 = np.where((nhist[sl] == 0) & ~have_h, np.minimum(1.0, 1.0 / np.maximum(np.abs(gl).max(1), 1e-300)), 1.0)
            # cap the step in log-parameter space
            dmax = np.abs(d).max(1)
            t = np.minimum(step0, MAX_LOG_STEP / np.maximum(dmax, 1e-300))
            xnew = xl.copy(); fnew = fl.copy()
            pending = np.arange(live.size)
            for _ in range(30):
                if pending.size == 0:
                    break
                xt = np.clip(xl[pending] + t[pending, None] * d[pending], self.lo, self.hi)
                ft = self.value(xt, live[pending])
                ok = ft <= fl[pending] + 1e-4 * t[pending] * gd[pending]
                ok &= np.isfinite(ft)
                acc = pending[ok]
                xnew[acc] = xt[ok]; fnew[acc] = ft[ok]
                t[pending[~ok]] = _backtrack(t[pending[~ok]], ft[~ok], fl[pending[~ok]], gd[pending[~ok]])
                pending = pending[~ok]
            failed = np.zeros(live.size, bool)
            failed[pending] = True  # no decrease found: treat as converged at the current point
            fx, gx, hx = self._vg(xnew, live)
            gx = np.where(self.active[sl], gx, 0.0)
            if hx is not None:
                hdiag[sl] = hx
            s_ = xnew - xl
            y_ = gx - gl
            sy = np.einsum("pd,pd->p", s_, y_)
            yy = np.einsum("pd,pd->p", y_, y_)
            upd = (sy > 1e-12 * np.sqrt(np.einsum("pd,pd->p", s_, s_) * yy + 1e-300)) & ~failed
            slot = it % m
            rho[slot] = 0.0
            S[slot, sl] = s_; Y[slot, sl] = y_
            rho[slot, sl] = np.where(upd, 1.0 / np.where(upd, sy, 1.0), 0.0)
            # (measured: scaling the curvature diagonal D by the secant factor s.y / (y.D y), the textbook choice, makes the steps
            #  far too short here -- 250 iterations instead of 40 on 16 loci x 20 000 x 64 taxa; the bare 1 / curvature overshoots
            #  the unit step for about half the loci of an iteration, which costs one more likelihood call, not an iteration)
            gamma_all[sl] = np.where(upd, sy / np.maximum(yy, 1e-300), gamma)
            nhist[sl] += upd
            df = fl - fx
            last_df[sl] = np.where(failed, 0.0, df)
            # a failed line search with correction pairs in play: forget them and try once more along the
            # (preconditioned) gradient before giving the point up as converged
            retry = failed & (nhist[sl] > 0)
            if retry.any():
                rho[:, live[retry]] = 0.0
                nhist[live[retry]] = 0
                last_df[live[retry]] = np.inf
            done = failed & ~retry & fresh[sl]
            done |= failed & ~retry & (nhist[sl] == 0)
            fresh[sl] = retry
            if self.prune is not None:
                done |= self.prune(live, fx, df)
            self.x[sl] = xnew; f[sl] = fx; g[sl] = gx
            self.iters[sl] += 1
            converged[live[done]] = True
            if os.environ.get("TPHIP_STAGE1_TRACE") and P <= 16:
                print("it %d live %s f %s df %s -gd %s failed %s retry %s done %s" % (
                    it, live.tolist(), np.round(fx, 6).tolist(), ["%.2e" % v for v in df], ["%.2e" % v for v in -gd],
                    failed.astype(int).tolist(), retry.astype(int).tolist(), done.astype(int).tolist()), flush=True)
        self.f, self.g = f, g
        return self.x, f


class Stage1:
    """Model-averaged exchangeabilities for every locus of a plan (see module docstring)."""

    def __init__(self, plan, states, pi, parent, blen, fd_step=1e-4, analytic=None, sub_analytic=None, prune_models=True,
                 precondition=True, verbose=False, screen_models=True):
        self.screen_models = screen_models   # quadratic screen of the 202 constrained models (see _screen_submodels)
        self.plan, self.states = plan, np.ascontiguousarray(states, dtype=np.uint8)
        self.pi = np.asarray(pi, dtype=np.float64).reshape(plan.nloci, 4)
        self.pi = self.pi / self.pi.sum(1, keepdims=True)
        parent = np.asarray(parent)
        self.branches = np.flatnonzero(parent >= 0)          # nodes that carry a branch
        self.nn = len(parent)
        self.blen0 = np.asarray(blen, dtype=np.float64)
        self.h = fd_step
        self.cache = plan.device_cache()
        # gradients: the reverse-mode kernel (tphip_locus_gradient) when the engine has it, else central differences
        self.analytic = hasattr(plan, "locus_gradient") if analytic is None else analytic
        # the rate-class models have <= 4 free parameters: a central-difference stencil is <= 8 value evaluations,
        # which is what one reverse-mode gradient costs at present (measured, tools/stage1_timing.py), so they default
        # to the stencil; the general model (5 + 2N-3 parameters) always uses the gradient kernel
        self.sub_analytic = (analytic is True and sub_analytic is not False) if sub_analytic is None else sub_analytic
        self.lik_seconds = 0.0       # wall time spent inside likelihood calls (kernels + copies)
        self.prune_models = prune_models
        self.precondition = precondition
        self.ngrads = 0
        self.verbose = verbose
        self.nevals = 0

    def close(self):
        self.cache.release()

    # ---- likelihood calls ---------------------------------------------------------------------------------
    def _lik(self, vecs, locus, exch, vec=None, scale=None, pidx=None, pfac=None):
        self.nevals += len(locus)
        t0 = time.perf_counter()
        out = self.plan.locus_loglik(self.states, vecs, locus, exch, vec, scale, pidx, pfac, cache=self.cache)
        self.lik_seconds += time.perf_counter() - t0
        return out

    @staticmethod
    def _exch_from_free(logr5):
        """[.., 5] log(AC, AT, CG, CT, GT) -> [.., 6] AC, AG=1, AT, CG, CT, GT."""
        r = np.exp(logr5)
        return np.stack([r[..., 0], np.ones_like(r[..., 0]), r[..., 1], r[..., 2], r[..., 3], r[..., 4]], axis=-1)

    # ---- general reversible model: 5 rates + all branch lengths -------------------------------------------
    # The optimiser's coordinates are (log rates, log b) with b_b = t_b * totalFactor(r) the branch length in expected
    # substitutions: at fixed b a change of the rates does not change how much evolution the tree carries, which
    # removes the strong rates-vs-lengths correlation of the raw (r, t) coordinates (what Q normalisation does in
    # most phylogenetics codes; HyPhy's script leaves Q unnormalised, bf:405-463, so t is what it reports).
    def _grm_point(self, X, idx):
        exch = self._exch_from_free(X[:, :5])
        scale = 1.0 / total_factor(self.pi[idx], exch)
        vecs = np.zeros((len(idx), self.nn))
        vecs[:, self.branches] = np.exp(X[:, 5:])
        return exch, scale, vecs

    def _grm_value(self, X, idx):
        exch, scale, vecs = self._grm_point(X, idx)
        return -self._lik(vecs, idx, exch, None, scale)

    def _grm_value_and_grad(self, X, idx):
        n, D = X.shape
        if self.analytic:
            exch, scale, vecs = self._grm_point(X, idx)
            self.ngrads += n
            out = self.plan.locus_gradient(self.states, vecs, idx, exch, None, scale, cache=self.cache,
                                           curvature=self.precondition)
            lnl, dex, dlt, sdl = out[:4]
            if not hasattr(self, "_last_f"):
                self._last_f = np.zeros(self.plan.nloci)
            self._last_f[idx] = -lnl
            # t_b = b_b / totalFactor(r): d log t_b / d r_q = -(2 pi_i pi_j) / totalFactor for every branch
            pi = self.pi[idx]
            dk = np.stack([2.0 * pi[:, i] * pi[:, j] for i, j in _PAIRS], axis=1)
            dr = (dex - sdl[:, None] * dk * scale[:, None]) * exch        # d lnL / d log r_q at fixed b
            g = np.empty((n, D))
            g[:, :5] = dr[:, [0, 2, 3, 4, 5]]
            g[:, 5:] = dlt[:, self.branches]                              # d/d log b_b = d/d log t_b
            if not self.precondition:
                return -lnl, -g
            h = np.full((n, D), np.nan)
            h[:, 5:] = -out[4][:, self.branches]   # curvature of -lnL in log b_b; the five rates get the scalar scaling
            return -lnl, -g, h
        nb = len(self.branches)
        h = self.h
        vecs = np.zeros((n, self.nn))
        vecs[:, self.branches] = np.exp(X[:, 5:])
        # candidates per problem: base, +-h on each log-rate (exchangeabilities and the scale 1/totalFactor change),
        # +-h on each log-branch (the kernel multiplies one branch of the shared vector by exp(+-h))
        per = 1 + 2 * D
        loc = np.repeat(idx, per)
        vec = np.repeat(np.arange(n), per)
        logr = np.repeat(X[:, None, :5], per, axis=1)
        pidx = np.full((n, per), -1, dtype=np.int64)
        pfac = np.ones((n, per))
        for j in range(5):
            logr[:, 1 + 2 * j, j] += h
            logr[:, 2 + 2 * j, j] -= h
        for b in range(nb):
            pidx[:, 1 + 2 * (5 + b)] = self.branches[b]; pfac[:, 1 + 2 * (5 + b)] = np.exp(h)
            pidx[:, 2 + 2 * (5 + b)] = self.branches[b]; pfac[:, 2 + 2 * (5 + b)] = np.exp(-h)
        exch = self._exch_from_free(logr.reshape(-1, 5))
        scale = 1.0 / total_factor(self.pi[loc], exch)
        f = -self._lik(vecs, loc, exch, vec, scale, pidx.reshape(-1), pfac.reshape(-1))
        f = f.reshape(n, per)
        g = (f[:, 1::2] - f[:, 2::2]) / (2 * h)
        return f[:, 0], g

    def _grm_escape(self, idx, X, G):
        """Log-parameters have a trap at zero: d f / d log b = b * d f / d b vanishes with b whatever d f / d b is, so a
        branch that overshot towards zero early on can sit there although the likelihood wants it longer (seen on 64
        taxa x 200 columns: 0.08 lnL left on the table).  At a would-be stopping point every collapsed branch is
        checked in the ORIGINAL parametrisation (optimality at the bound b = 0 needs d f / d b >= 0); offenders are
        put back at a small positive length and the search continues from there."""
        Xn = X.copy()
        logb = X[:, 5:]
        b = np.exp(logb)
        slope = G[:, 5:] / b                                 # d(-lnL) / d b
        fscale = 1.0 + np.abs(self._last_f[idx]) if hasattr(self, "_last_f") else 1.0
        kick = (b < 1e-6) & (slope * ESCAPE_LENGTH < -1e-7 * np.reshape(fscale, (-1, 1)))
        kick &= self._kicks[idx][:, None] < 3                # a branch that keeps coming back really is zero
        Xn[:, 5:] = np.where(kick, np.log(ESCAPE_LENGTH), logb)
        # the five rates: same test at their lower bound (see _sub_escape)
        rslope = G[:, :5] / np.exp(X[:, :5])
        rkick = (X[:, :5] < LOG_RATE_MIN + 1.0) & (rslope * ESCAPE_RATE < -1e-7 * np.reshape(fscale, (-1, 1)))
        rkick &= self._kicks[idx][:, None] < 3
        Xn[:, :5] = np.where(rkick, np.log(ESCAPE_RATE), X[:, :5])
        moved = kick.any(axis=1) | rkick.any(axis=1)
        self._kicks[idx] += moved
        return Xn, moved

    def _sub_escape(self, idx, X, G):
        """The same trap for a rate class that sits at its lower bound (typically because the general model put that
        rate there and the class starts from it): if the likelihood rises with the rate itself, d f / d r < 0, the
        point is an artefact of the log scale, not a boundary optimum; put the rate at ESCAPE_RATE and carry on."""
        r = np.exp(X)
        slope = G / r                                        # d(-lnL) / d r
        fscale = 1.0 + np.abs(self._sub_last_f[idx])
        kick = (X < LOG_RATE_MIN + 1.0) & self._sub_active[idx] & (slope * ESCAPE_RATE < -1e-7 * fscale[:, None])
        kick &= self._sub_kicks[idx][:, None] < 2
        Xn = np.where(kick, np.log(ESCAPE_RATE), X)
        moved = kick.any(axis=1)
        self._sub_kicks[idx] += moved
        return Xn, moved

    def initial_branch_lengths(self):
        """Start of the branch-length search: the input tree's shape (its lengths are in time units, not in
        substitutions), rescaled per locus to the best of a coarse grid of mean branch lengths 1e-4 .. 1 under the
        all-ones model.  (HyPhy starts from its own defaults; the optimum does not depend on the start.)"""
        L = self.plan.nloci
        shape = np.zeros(self.nn)
        b = np.maximum(self.blen0[self.branches], 0.0)
        shape[self.branches] = np.maximum(b / max(b.mean(), 1e-300), 1e-3)
        grid = 10.0 ** np.linspace(-4.0, 0.0, 17)
        loc = np.repeat(np.arange(L), len(grid))
        lnl = self._lik(shape[None, :], loc, np.ones((len(loc), 6)), np.zeros(len(loc), dtype=np.int64),
                        np.tile(grid, L)).reshape(L, len(grid))
        best = grid[np.argmax(lnl, axis=1)]
        return best[:, None] * shape[None, :]

    def fit_grm(self, maxit=None):
        L = self.plan.nloci
        if maxit is None:
            maxit = max(300, 4 * (5 + len(self.branches)))   # L-BFGS needs O(dimension) iterations on big trees
        x0 = np.zeros((L, 5 + len(self.branches)))
        x0[:, 5:] = np.log(self.initial_branch_lengths()[:, self.branches] * total_factor(self.pi, np.ones(6))[:, None])
        lo = np.concatenate([np.full(5, LOG_RATE_MIN), np.full(len(self.branches), LOG_BLEN_MIN)])
        hi = np.concatenate([np.full(5, LOG_RATE_MAX), np.full(len(self.branches), LOG_BLEN_MAX)])
        self._kicks = np.zeros(L, dtype=np.int64)
        opt = _LBFGS(self._grm_value, self._grm_value_and_grad, x0, maxit=maxit, lo=lo, hi=hi,
                     escape=self._grm_escape if self.analytic else None)
        x, f = opt.run()
        self.grm_iters = opt.iters
        exch = self._exch_from_free(x[:, :5])
        t = np.zeros((L, self.nn))
        t[:, self.branches] = np.exp(x[:, 5:]) / total_factor(self.pi, exch)[:, None]
        return exch, t, -f

    # ---- the 202 constrained models ---------------------------------------------------------------------
    def _sub_exch(self, X, cls):
        """free log class rates X [n, 4] + class map cls [n, 6] -> exchangeabilities [n, 6]."""
        r = np.exp(X)
        pick = r[np.arange(len(X))[:, None], np.maximum(cls, 0)]
        return np.where(cls < 0, 1.0, pick)

    def _sub_eval(self, X, prob, cls):
        exch = self._sub_exch(X, cls)
        loc = self._sub_locus[prob]
        scale = 1.0 / (exch * self._w6[loc]).sum(axis=1)      # branch lengths = stash / totalFactor(model), bf:613-619
        return exch, loc, scale

    def _sub_value(self, X, idx):
        exch, loc, scale = self._sub_eval(X, idx, self._sub_cls[idx])
        return -self._lik(self._stash, loc, exch, loc, scale)

    def _sub_value_and_grad(self, X, idx):
        n, D = X.shape
        if self.sub_analytic:
            cls = self._sub_cls[idx]
            exch, loc, scale = self._sub_eval(X, idx, cls)
            self.ngrads += n
            lnl, dex, _, sdl = self.plan.locus_gradient(self.states, self._stash, loc, exch, loc, scale, cache=self.cache,
                                                        per_branch=False)
            # t_b = stash_b / totalFactor(r): d log t_b / d r_q = -(2 pi_i pi_j) / totalFactor for every branch
            dr = (dex - sdl[:, None] * self._w6[loc] * scale[:, None]) * exch    # d lnL / d log r_q, q over the six rates
            g = np.zeros((n, D))
            for c in range(D):
                g[:, c] = (dr * (cls == c)).sum(1)
            self._sub_last_f[idx] = -lnl
            if self._sub_hdiag is not None:
                return -lnl, -np.where(self._sub_active[idx], g, 0.0), self._sub_hdiag[idx]
            return -lnl, -np.where(self._sub_active[idx], g, 0.0)
        # central differences: the stencil's exchangeabilities are the base point's times precomputed factors
        # (member rates of class j times e^{+-h}), so a whole iteration's candidates are two broadcasts
        per = 1 + 2 * D
        base = self._sub_exch(X, self._sub_cls[idx])                             # [n, 6]
        exch = (base[:, None, :] * self._sub_stencil[idx]).reshape(n * per, 6)   # [n * per, 6]
        loc = np.repeat(self._sub_locus[idx], per)
        scale = 1.0 / (exch * self._w6[loc]).sum(axis=1)
        # skip the stencil points of inactive dimensions (their gradient is defined as zero)
        need = self._sub_need[idx].reshape(-1)
        f = np.zeros(n * per)
        f[need] = -self._lik(self._stash, loc[need], exch[need], loc[need], scale[need])
        f = f.reshape(n, per)
        g = (f[:, 1::2] - f[:, 2::2]) / (2 * self.h)
        self._sub_last_f[idx] = f[:, 0]
        if self._sub_hdiag is not None:
            return f[:, 0], np.where(self._sub_active[idx], g, 0.0), self._sub_hdiag[idx]
        return f[:, 0], np.where(self._sub_active[idx], g, 0.0)

    # ---- quadratic screen of the 202 constrained models ------------------------------------------------------
    # Every constrained model maximises THE SAME function as the general model -- f(rho) = lnL at the stashed branch
    # lengths b* (in expected substitutions, bf:522-540, 613-619) as a function of the five log-rates rho -- over a linear
    # subspace: rates of one class are equal, the class of AG is 0 (bf:577-600), i.e. rho = A_m theta.  So one local
    # model of f serves all 202: its Hessian at the general model's optimum (31 likelihood evaluations per locus) gives
    # every model's constrained optimum in closed form, theta_m = (A'HA)^-1 A'(H rho* - g).  One TRUE likelihood
    # evaluation there (202 per locus) then tells which models can carry weight at all: on long loci a handful -- the
    # others trail by hundreds of log-units and are never fitted (before: one 9-point stencil iteration for each of
    # them, 90 % of the stage's kernel time at 50 000 columns) -- and the fitted ones start next to their optimum with
    # the right metric.  Nothing is approximated in what is reported: a fitted model is polished on the true likelihood
    # as before, an abandoned one carries its true likelihood at theta_m (a lower bound of its maximum).
    def _screen_submodels(self, grm_exch, cls, kk, h=2e-2):
        L = self.plan.nloci
        M = cls.shape[0]
        free6 = np.array([0, 2, 3, 4, 5])                   # positions of AC, AT, CG, CT, GT among the six rates
        rho = np.log(grm_exch[:, free6])                     # [L, 5]
        pairs = [(i, j) for i in range(5) for j in range(i + 1, 5)]
        pts = [np.zeros(5)]
        for i in range(5):
            for sgn in (1.0, -1.0):
                e = np.zeros(5); e[i] = sgn * h; pts.append(e)
        for i, j in pairs:
            for sgn in (1.0, -1.0):
                e = np.zeros(5); e[i] = sgn * h; e[j] = sgn * h; pts.append(e)
        pts = np.array(pts)                                   # [31, 5]
        npts = len(pts)
        loc = np.repeat(np.arange(L), npts)
        logr = (rho[:, None, :] + pts[None, :, :]).reshape(-1, 5)
        exch = self._exch_from_free(logr)
        scale = 1.0 / (exch * self._w6[loc]).sum(axis=1)
        f = self._lik(self._stash, loc, exch, loc, scale).reshape(L, npts)
        f0 = f[:, 0]
        g = np.zeros((L, 5)); H = np.zeros((L, 5, 5))
        for i in range(5):
            fp, fm = f[:, 1 + 2 * i], f[:, 2 + 2 * i]
            g[:, i] = (fp - fm) / (2 * h)
            H[:, i, i] = (fp - 2 * f0 + fm) / (h * h)
        for q, (i, j) in enumerate(pairs):
            fpp, fmm = f[:, 11 + 2 * q], f[:, 12 + 2 * q]
            fi = f[:, 1 + 2 * i] + f[:, 2 + 2 * i]
            fj = f[:, 1 + 2 * j] + f[:, 2 + 2 * j]
            H[:, i, j] = H[:, j, i] = (fpp + fmm - fi - fj + 2 * f0) / (2 * h * h)
        # curvature of -f, made safely positive definite (a rate on the edge of its box, or a flat direction)
        w, V = np.linalg.eigh(-H)
        # (absolute floor 1e-3: a direction along which lnL changes by less than 5e-4 per unit of log-rate is flat for
        #  every purpose; on a locus without data the stencil's gradient is rounding noise, which a smaller floor amplifies)
        floor = np.maximum(1e-6 * np.abs(w).max(axis=1, keepdims=True), 1e-3)
        K = np.einsum("lik,lk,ljk->lij", V, np.maximum(w, floor), V)      # [L, 5, 5]
        self._screen_K, self._screen_g = K, g
        # constrained optimum of the quadratic model per (locus, model)
        theta = np.zeros((L, M, 4))
        hdiag = np.full((L, M, 4), np.nan)
        for m in range(M):
            k = int(kk[m])
            A = np.zeros((5, max(k, 1)))
            for q5, q6 in enumerate(free6):
                if cls[m, q6] >= 0:
                    A[q5, cls[m, q6]] = 1.0
            if k == 0:
                continue
            A = A[:, :k]
            AK = np.einsum("qa,lqr->lar", A, K)                # [L, k, 5]
            AKA = np.einsum("lar,rb->lab", AK, A)              # [L, k, k]
            rhs = np.einsum("lar,lr->la", AK, rho) + g @ A     # maximise f: K (A theta - rho*) = g  (g is d f / d rho)
            theta[:, m, :k] = np.linalg.solve(AKA, rhs[..., None])[..., 0]
            hdiag[:, m, :k] = np.einsum("laa->la", AKA)
        theta = np.clip(theta, LOG_RATE_MIN, LOG_RATE_MAX)
        return theta.reshape(L * M, 4), hdiag.reshape(L * M, 4)

    def fit_submodels(self, grm_exch, grm_t, maxit=100, grm_lnl=None):
        L = self.plan.nloci
        strings = model_strings()
        cls, k = model_design(strings)
        M = len(strings) - 1
        # stashed branch lengths in expected substitutions (bf:522-540)
        self._stash = grm_t * total_factor(self.pi, grm_exch)[:, None]
        self._sub_locus = np.repeat(np.arange(L), M)
        self._sub_cls = np.tile(cls[1:], (L, 1))
        kk = np.tile(k[1:], L)
        self._sub_active = np.arange(4)[None, :] < kk[:, None]
        self._w6 = np.stack([2.0 * self.pi[:, i] * self.pi[:, j] for i, j in _PAIRS], axis=1)   # totalFactor = exch . w6
        # per model: factors of the 9-point stencil [1 + 2*4, 6] and which of its points are needed
        up, dn = np.exp(self.h), np.exp(-self.h)
        sten = np.ones((M, 9, 6))
        need = np.zeros((M, 9), bool)
        need[:, 0] = True
        for j in range(4):
            member = cls[1:] == j                                   # [M, 6]
            sten[:, 1 + 2 * j, :] = np.where(member, up, 1.0)
            sten[:, 2 + 2 * j, :] = np.where(member, dn, 1.0)
            need[:, 1 + 2 * j] = need[:, 2 + 2 * j] = k[1:] > j
        self._sub_stencil = np.tile(sten, (L, 1, 1))
        self._sub_need = np.tile(need, (L, 1))
        self._sub_hdiag = None
        if self.screen_models:
            # start: every model's optimum under the shared quadratic model of the likelihood (_screen_submodels)
            x0, self._sub_hdiag = self._screen_submodels(grm_exch, cls[1:], k[1:])
            x0 = np.where(self._sub_active, x0, 0.0)
        else:
            # start: geometric mean of the general model's rates over each class
            lg = np.log(grm_exch)[self._sub_locus]                        # [P, 6]
            x0 = np.zeros((L * M, 4))
            for c in range(4):
                inc = self._sub_cls == c
                cnt = inc.sum(1)
                x0[:, c] = np.where(cnt > 0, (lg * inc).sum(1) / np.maximum(cnt, 1), 0.0)
        # Models that cannot matter are abandoned early: a model whose Akaike score lnL - k trails the best of its
        # locus by more than PRUNE_NATS even after crediting three times its last improvement carries a weight below
        # e^-21 ~ 8e-10 -- invisible in the averaged rates -- so polishing its optimum is wasted likelihood evaluations
        # (with thousands of columns all but a handful of the 203 models are in that state after one iteration).
        best = (np.asarray(grm_lnl) - 5.0).copy() if grm_lnl is not None else np.full(L, -np.inf)
        self.pruned = 0

        self._sub_last_f = np.zeros(L * M)
        self._sub_kicks = np.zeros(L * M, dtype=np.int64)
        sel = None
        if self.screen_models and self.prune_models and grm_lnl is not None:
            # one true likelihood per model at its screened optimum; a model that trails the best of its locus by more than
            # PRUNE_NATS even after crediting three times what the quadratic model missed there is never fitted
            f_at = self._sub_value(x0, np.arange(L * M))                          # -lnL at the screened optima
            score = -f_at - kk
            np.maximum.at(best, self._sub_locus, score)
            rho5 = np.log(grm_exch[:, [0, 2, 3, 4, 5]])[self._sub_locus]
            A_theta = np.where(self._sub_cls[:, [0, 2, 3, 4, 5]] >= 0,
                               x0[np.arange(L * M)[:, None], np.maximum(self._sub_cls[:, [0, 2, 3, 4, 5]], 0)], 0.0)
            # what the quadratic model (metric K of the locus) predicted for this point vs what the likelihood says
            keep = score + 3.0 * np.abs(self._screen_miss(A_theta - rho5, f_at, grm_lnl)) >= best[self._sub_locus] - PRUNE_NATS
            self.pruned = int((~keep).sum())
            sel = np.flatnonzero(keep)
            self._sub_last_f[:] = f_at
        if sel is None:
            sel = np.arange(L * M)
        x, fall = x0.copy(), self._sub_last_f.copy()
        self.sub_iters = np.zeros(L * M, dtype=np.int64)
        if sel.size:
            full = dict(cls=self._sub_cls, locus=self._sub_locus, active=self._sub_active, stencil=self._sub_stencil,
                        need=self._sub_need, hdiag=self._sub_hdiag, kk=kk)
            # the optimiser sees the surviving problems only
            self._sub_cls, self._sub_locus, self._sub_active = full["cls"][sel], full["locus"][sel], full["active"][sel]
            self._sub_stencil, self._sub_need = full["stencil"][sel], full["need"][sel]
            self._sub_hdiag = None if full["hdiag"] is None else full["hdiag"][sel]
            self._sub_last_f = np.zeros(sel.size)
            self._sub_kicks = np.zeros(sel.size, dtype=np.int64)
            kk_sel = kk[sel]

            def prune_sel(idx, f, df):
                score = -f - kk_sel[idx]
                loc = self._sub_locus[idx]
                np.maximum.at(best, loc, score)
                drop = score + 3.0 * np.maximum(df, 0.0) < best[loc] - PRUNE_NATS
                self.pruned += int(drop.sum())
                return drop

            opt = _LBFGS(self._sub_value, self._sub_value_and_grad, x0[sel], active=self._sub_active, maxit=maxit,
                         lo=LOG_RATE_MIN, hi=LOG_RATE_MAX, prune=prune_sel if self.prune_models else None,
                         escape=self._sub_escape)
            xs, fs = opt.run()
            its = opt.iters
            x[sel], fall[sel] = xs, fs
            self.sub_iters[sel] = its
            self._sub_cls, self._sub_locus, self._sub_active = full["cls"], full["locus"], full["active"]
            self._sub_stencil, self._sub_need, self._sub_hdiag = full["stencil"], full["need"], full["hdiag"]
        exch = self._sub_exch(x, self._sub_cls).reshape(L, M, 6)
        return exch, (-fall).reshape(L, M), k

    def _screen_miss(self, delta, f_at, grm_lnl):
        """How far the true likelihood at a screened optimum is from what the quadratic model predicted there (the
        model's error is the scale of what a fit can still gain): predicted -lnL = -lnL* + 1/2 delta' K delta."""
        K = self._screen_K[self._sub_locus]
        pred = -np.asarray(grm_lnl)[self._sub_locus] - np.einsum("pq,pq->p", self._screen_g[self._sub_locus], delta) \
            + 0.5 * np.einsum("pi,pij,pj->p", delta, K, delta)
        return f_at - pred

    # ---- model averaging (bf:806-847) --------------------------------------------------------------------
    def run(self):
        grm_exch, grm_t, grm_lnl = self.fit_grm()
        sub_exch, sub_lnl, k = self.fit_submodels(grm_exch, grm_t, grm_lnl=grm_lnl)
        lnl = np.concatenate([grm_lnl[:, None], sub_lnl], axis=1)          # [L, 203]
        exch = np.concatenate([grm_exch[:, None, :], sub_exch], axis=1)    # [L, 203, 6]
        score = lnl - k[None, :]                                           # = -AIC/2 + const
        w = np.exp(score - score.max(axis=1, keepdims=True))
        w /= w.sum(axis=1, keepdims=True)
        avg = (w[:, :, None] * exch).sum(axis=1)
        avg[:, 1] = 1.0
        return dict(exch=avg, weights=w, lnl=lnl, model_exch=exch, grm_exch=grm_exch, grm_blen=grm_t,
                    models=model_strings(), nevals=self.nevals, ngrads=self.ngrads)


def model_averaged_exchangeabilities(plan, states, pi, parent, blen, engine_fit=None, **kw):
    """Returns dict with `exch` [L, 6] (AC, AG=1, AT, CG, CT, GT) and diagnostics.

    The product path is ONE engine call, tphip_stage1_fit (csrc/stage1_driver.hip: the optimisers are device kernels,
    the host sequences launches).  engine_fit=False -- or any option of the class below -- runs this module's host
    optimiser over the same likelihood kernels instead: the round-1/2 implementation, kept as a second opinion for
    tests and A/B timing (tools/stage1_timing.py host)."""
    if engine_fit is None:
        engine_fit = hasattr(plan, "stage1_fit") and not kw and os.environ.get("TPHIP_STAGE1_HOST") is None
    if engine_fit:
        out = plan.stage1_fit(states)
        st = out.pop("stats")
        out.update(models=model_strings(), nevals=st["nevals"], ngrads=st["ngrads"], stats=st, grm_exch=out["model_exch"][:, 0])
        return out
    s1 = Stage1(plan, states, pi, parent, blen, **kw)
    try:
        return s1.run()
    finally:
        s1.close()
