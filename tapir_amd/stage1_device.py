"""Device-resident optimiser for the 202 constrained models of HyPhy's stage 1 (models_and_rates.bf:542-661).

`tapir_amd/stage1.py` fits them with a batched L-BFGS written in numpy; on batches of many short loci (C2 / C4 shapes:
10^5 small problems per block of loci) the numpy bookkeeping, not the likelihood kernel, was 70 % of the stage.  This
module keeps the optimiser's whole state on the GPU -- torch tensors as device memory, `tphip_locus_loglik_dev` for the
likelihoods, no host round trip per likelihood call -- and replaces the limited-memory update by what problems of at
most four parameters allow: a dense BFGS matrix per problem, started from the metric of the quadratic screen
(`Stage1._screen_submodels`), i.e. Newton-like steps from the first iteration.

Same objective, same stencil gradients (central differences of the value kernel), same stopping, pruning and
boundary-escape rules as `stage1._LBFGS` + `Stage1.fit_submodels`; the results agree with the numpy path within the
stage's tolerance (tests/test_gpu_stage1.py::test_stage1_device_fitter_matches_host_fitter).
"""
import numpy as np

from . import stage1 as _s1


def available(plan):
    """The device path needs the real engine (device-pointer likelihood entry) and torch with a GPU."""
    if not hasattr(plan, "locus_loglik_dev"):
        return False
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def _backtrack(torch, t, ft, f0, gd):
    """stage1._backtrack on tensors: next trial step from the parabola through f(0), f'(0) and the failed trial."""
    den = 2.0 * (ft - f0 - gd * t)
    good = torch.isfinite(ft) & (den > 0)
    tq = torch.where(good, -gd * t * t / torch.where(good, den, torch.ones_like(den)), 0.25 * t)
    return torch.maximum(torch.minimum(tq, 0.5 * t), 0.1 * t)


class DeviceSubmodelFitter:
    def __init__(self, plan, d_states_ptr, stash, w6, locus, cls, active, stencil, need, kk, h, device=0):
        import torch
        self.torch = torch
        self.plan = plan
        self.dev = torch.device("cuda", device)
        self.d_states_ptr = int(d_states_ptr)
        t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a), dtype=dt, device=self.dev)  # noqa: E731
        self.stash = t(stash, torch.float64)                  # [L, nnodes]
        self.w6 = t(w6, torch.float64)                        # [L, 6]
        self.locus = t(locus, torch.int64)                    # [P]
        self.cls = t(cls, torch.int64)                        # [P, 6], -1 = the class of AG
        self.active = t(active, torch.bool)                   # [P, 4]
        self.stencil = t(stencil, torch.float64)              # [P, 9, 6]
        self.need = t(need, torch.bool)                       # [P, 9]
        self.kk = t(kk, torch.float64)                        # [P]
        self.h = float(h)
        self.nevals = 0
        self.stream = torch.cuda.current_stream(self.dev).cuda_stream

    # ---- likelihood -------------------------------------------------------------------------------------------
    def _exch(self, x, idx):
        r = self.torch.exp(x)
        cls = self.cls[idx]
        pick = self.torch.gather(r, 1, cls.clamp(min=0))
        return self.torch.where(cls < 0, self.torch.ones_like(pick), pick)

    def _lik(self, loc, exch):
        """-lnL of candidates (locus loc[c], exchangeabilities exch[c]) at the stashed lengths / totalFactor(model)."""
        torch = self.torch
        n = loc.numel()
        if n == 0:
            return torch.zeros(0, dtype=torch.float64, device=self.dev)
        scale = 1.0 / (exch * self.w6[loc]).sum(dim=1)
        loc32 = loc.to(torch.int32).contiguous()
        out = torch.empty(n, dtype=torch.float64, device=self.dev)
        pidx = torch.full((n,), -1, dtype=torch.int32, device=self.dev)
        pfac = torch.ones(n, dtype=torch.float64, device=self.dev)
        self.plan.locus_loglik_dev(self.d_states_ptr, n, loc32, exch.contiguous(), self.stash, loc32, scale.contiguous(), pidx,
                                   pfac, out, self.stream)
        self.nevals += n
        return -out

    def value(self, x, idx):
        return self._lik(self.locus[idx], self._exch(x, idx))

    def value_and_grad(self, x, idx):
        """Central differences on the 9-point stencil, skipping the points of inactive dimensions."""
        torch = self.torch
        n = idx.numel()
        base = self._exch(x, idx)                                         # [n, 6]
        need = self.need[idx]                                             # [n, 9]
        row, pt = torch.nonzero(need, as_tuple=True)
        exch = base[row] * self.stencil[idx[row], pt]                     # [m, 6]
        fc = self._lik(self.locus[idx[row]], exch)
        F = torch.zeros((n, 9), dtype=torch.float64, device=self.dev)
        F[row, pt] = fc
        g = (F[:, 1::2] - F[:, 2::2]) / (2.0 * self.h)
        return F[:, 0], torch.where(self.active[idx], g, torch.zeros_like(g))

    # ---- the optimiser ----------------------------------------------------------------------------------------
    def run(self, x0, hdiag, best, maxit=100, ftol=1e-10, gtol=2e-6, ptol=1e-9, prune=True):
        """x0 [P, 4] start (the screened optima), hdiag [P, 4] curvature of -lnL per parameter from the screen (NaN where
        unknown), best [L] best Akaike score lnL - k seen per locus so far.  Returns (x [P, 4], f [P] = -lnL, iterations)."""
        torch = self.torch
        dev = self.dev
        P = x0.shape[0]
        x = torch.as_tensor(x0, dtype=torch.float64, device=dev).clone()
        hd = torch.as_tensor(hdiag, dtype=torch.float64, device=dev)
        best = torch.as_tensor(best, dtype=torch.float64, device=dev).clone()
        lo, hi = _s1.LOG_RATE_MIN, _s1.LOG_RATE_MAX
        all_idx = torch.arange(P, device=dev)
        f, g = self.value_and_grad(x, all_idx)
        # inverse metric: 1 / curvature on the diagonal where the screen supplied one, identity-scaled otherwise
        inv = torch.where(torch.isfinite(hd) & (hd > 1e-12), 1.0 / hd.clamp(min=1e-12), torch.ones_like(hd))
        Hinv = torch.diag_embed(inv)
        converged = torch.zeros(P, dtype=torch.bool, device=dev)
        last_df = torch.full((P,), float("inf"), dtype=torch.float64, device=dev)
        kicks = torch.zeros(P, dtype=torch.int64, device=dev)
        iters = torch.zeros(P, dtype=torch.int64, device=dev)
        pruned = 0
        for it in range(maxit):
            live = torch.nonzero(~converged, as_tuple=True)[0]
            if live.numel() == 0:
                break
            xl, fl, gl, Hl = x[live], f[live], g[live], Hinv[live]
            d = -torch.bmm(Hl, gl.unsqueeze(2)).squeeze(2)
            d = torch.where(self.active[live], d, torch.zeros_like(d)).clamp(-_s1.MAX_LOG_STEP, _s1.MAX_LOG_STEP)
            gd = (gl * d).sum(dim=1)
            bad = ~(gd < 0)
            d = torch.where(bad.unsqueeze(1), -gl, d)
            gd = torch.where(bad, -(gl * gl).sum(dim=1), gd)
            scale_f = 1.0 + fl.abs()
            gmax = gl.abs().max(dim=1).values
            stop = (last_df[live] <= ftol * scale_f) & (-gd <= ptol * scale_f) & (gmax <= gtol * scale_f)
            stop |= gmax <= 1e-9
            if bool(stop.any()):
                # boundary trap of the log scale (Stage1._sub_escape): a class rate sitting at its lower bound although the
                # likelihood rises with the rate itself is put back at ESCAPE_RATE and the search continues from there
                si = live[stop]
                xs, gs = x[si], g[si]
                slope = gs / torch.exp(xs)
                fscale = (1.0 + f[si].abs()).unsqueeze(1)
                kick = (xs < lo + 1.0) & self.active[si] & (slope * _s1.ESCAPE_RATE < -1e-7 * fscale) & (kicks[si] < 2).unsqueeze(1)
                moved = kick.any(dim=1)
                if bool(moved.any()):
                    mi = si[moved]
                    x[mi] = torch.where(kick[moved], torch.full_like(xs[moved], float(np.log(_s1.ESCAPE_RATE))), xs[moved])
                    kicks[mi] += 1
                    fm, gm = self.value_and_grad(x[mi], mi)
                    f[mi], g[mi] = fm, gm
                    Hinv[mi] = torch.diag_embed(inv[mi])
                    last_df[mi] = float("inf")
                    stop_idx = si[~moved]
                    converged[stop_idx] = True
                    continue
                converged[si] = True
                keep = ~stop
                if not bool(keep.any()):
                    continue
                live, xl, fl, gl, Hl, d, gd = live[keep], xl[keep], fl[keep], gl[keep], Hl[keep], d[keep], gd[keep]
            # line search: Armijo backtracking from the full quasi-Newton step
            n = live.numel()
            t = torch.minimum(torch.ones(n, dtype=torch.float64, device=dev),
                              _s1.MAX_LOG_STEP / d.abs().max(dim=1).values.clamp(min=1e-300))
            xnew, fnew = xl.clone(), fl.clone()
            pending = torch.arange(n, device=dev)
            for _ in range(30):
                if pending.numel() == 0:
                    break
                xt = (xl[pending] + t[pending].unsqueeze(1) * d[pending]).clamp(lo, hi)
                ft = self.value(xt, live[pending])
                ok = (ft <= fl[pending] + 1e-4 * t[pending] * gd[pending]) & torch.isfinite(ft)
                acc = pending[ok]
                xnew[acc], fnew[acc] = xt[ok], ft[ok]
                bad = ~ok
                t[pending[bad]] = _backtrack(torch, t[pending[bad]], ft[bad], fl[pending[bad]], gd[pending[bad]])
                pending = pending[bad]
            failed = torch.zeros(n, dtype=torch.bool, device=dev)
            failed[pending] = True
            fx, gx = self.value_and_grad(xnew, live)
            # dense BFGS update of the inverse metric (problems of <= 4 parameters)
            s_ = xnew - xl
            y_ = gx - gl
            sy = (s_ * y_).sum(dim=1)
            upd = (sy > 1e-12 * torch.sqrt((s_ * s_).sum(dim=1) * (y_ * y_).sum(dim=1) + 1e-300)) & ~failed
            rho = torch.where(upd, 1.0 / torch.where(upd, sy, torch.ones_like(sy)), torch.zeros_like(sy))
            Hy = torch.bmm(Hl, y_.unsqueeze(2)).squeeze(2)
            yHy = (y_ * Hy).sum(dim=1)
            r1 = rho.unsqueeze(1).unsqueeze(2)
            term = (1.0 + rho * yHy).unsqueeze(1).unsqueeze(2) * r1 * (s_.unsqueeze(2) * s_.unsqueeze(1)) \
                - r1 * (Hy.unsqueeze(2) * s_.unsqueeze(1) + s_.unsqueeze(2) * Hy.unsqueeze(1))
            Hinv[live] = torch.where(upd.unsqueeze(1).unsqueeze(2), Hl + term, Hl)
            df = fl - fx
            last_df[live] = torch.where(failed, torch.zeros_like(df), df)
            # a failed line search with a learned metric: fall back to the screen's metric once before giving up
            learned = (Hl - torch.diag_embed(inv[live])).abs().amax(dim=(1, 2)) > 0
            retry = failed & learned
            Hinv[live[retry]] = torch.diag_embed(inv[live[retry]])
            last_df[live[retry]] = float("inf")
            done = failed & ~retry
            if prune:
                score = -fx - self.kk[live]
                loc = self.locus[live]
                best = best.scatter_reduce(0, loc, score, reduce="amax", include_self=True)
                drop = score + 3.0 * df.clamp(min=0.0) < best[loc] - _s1.PRUNE_NATS
                pruned += int(drop.sum())
                done = done | drop
            x[live], f[live], g[live] = xnew, fx, gx
            iters[live] += 1
            converged[live[done]] = True
        self.pruned = pruned
        return x.cpu().numpy(), f.cpu().numpy(), iters.cpu().numpy()


# ---- the general reversible model (bf:487-520): 5 rates + every branch length ------------------------------------------
class DeviceLBFGS:
    """`stage1._LBFGS` with its state in HBM: the same batched L-BFGS (two-loop recursion over a ring of correction pairs
    shared by all problems, diagonal initial metric from the objective's curvature, Armijo backtracking, the same stopping,
    retry and escape rules), torch tensors instead of numpy arrays.  On 4000 loci x 131 parameters the numpy two-loop
    recursion alone was 16 ms per iteration -- more than the kernels."""

    def __init__(self, torch, value, value_and_grad, x0, history=8, maxit=200, ftol=1e-10, gtol=2e-6, ptol=1e-9, lo=None, hi=None,
                 escape=None):
        self.torch = torch
        self.value, self.vg, self.escape = value, value_and_grad, escape
        self.x = x0.clone()
        self.m, self.maxit, self.ftol, self.gtol, self.ptol = history, maxit, ftol, gtol, ptol
        self.lo, self.hi = lo, hi

    def run(self):
        """Fixed-shape formulation: the optimiser's state covers a WORKING SET of problems (all of them at first) and every
        update is a masked whole-array operation -- no gather / scatter per correction pair, which made an iteration ~800
        small launches (7-11 ms) whatever the block held.  Only the likelihood calls take index lists (converged problems
        cost no kernel work), and the working set is compacted whenever fewer than 70 % of its rows are still live."""
        torch = self.torch
        P, D = self.x.shape
        dev = self.x.device
        f64 = torch.float64
        inf = float("inf")
        ids = torch.arange(P, device=dev)                  # original index of every working-set row
        x = self.x.clone()
        f, g, hd = self.vg(x, ids)
        hdiag = hd.clone() if hd is not None else torch.full((P, D), float("nan"), dtype=f64, device=dev)
        m = self.m
        S = torch.zeros((m, P, D), dtype=f64, device=dev)
        Y = torch.zeros((m, P, D), dtype=f64, device=dev)
        rho = torch.zeros((m, P), dtype=f64, device=dev)
        gamma = torch.ones(P, dtype=f64, device=dev)
        nhist = torch.zeros(P, dtype=torch.int64, device=dev)
        conv = torch.zeros(P, dtype=torch.bool, device=dev)
        last_df = torch.full((P,), inf, dtype=f64, device=dev)
        fresh = torch.zeros(P, dtype=torch.bool, device=dev)
        iters = torch.zeros(P, dtype=torch.int64, device=dev)
        out_x, out_f = self.x.clone(), torch.zeros(P, dtype=f64, device=dev)
        out_iters = torch.zeros(P, dtype=torch.int64, device=dev)
        MAXS = _s1.MAX_LOG_STEP

        def nz(mask):
            return torch.nonzero(mask, as_tuple=True)[0]

        for it in range(self.maxit):
            W = x.shape[0]
            nlive = W - int(conv.sum())
            if nlive == 0:
                break
            if nlive < 0.7 * W:
                # compact the working set: results of the rows that leave go to the output arrays
                gone, keep = nz(conv), nz(~conv)
                out_x[ids[gone]], out_f[ids[gone]], out_iters[ids[gone]] = x[gone], f[gone], iters[gone]
                ids, x, f, g, hdiag = ids[keep], x[keep], f[keep], g[keep], hdiag[keep]
                S, Y, rho = S[:, keep].contiguous(), Y[:, keep].contiguous(), rho[:, keep].contiguous()
                gamma, nhist, last_df, fresh, iters = gamma[keep], nhist[keep], last_df[keep], fresh[keep], iters[keep]
                conv = torch.zeros(keep.numel(), dtype=torch.bool, device=dev)
                W = keep.numel()
            live = ~conv
            # two-loop recursion, newest pair first, on the whole working set (rho = 0 marks "no pair")
            slots = [(it - 1 - j) % m for j in range(min(it, m))]
            q = g.clone()
            alpha = {}
            for i in slots:
                a = rho[i] * (S[i] * q).sum(dim=1)
                alpha[i] = a
                q = torch.addcmul(q, Y[i], a.unsqueeze(1), value=-1.0)
            okh = torch.isfinite(hdiag) & (hdiag > 1e-12)
            have_h = torch.isfinite(hdiag).any(dim=1)
            r = torch.where(okh, 1.0 / torch.where(okh, hdiag, torch.ones_like(hdiag)), gamma.unsqueeze(1).expand_as(hdiag)) * q
            for i in reversed(slots):
                bcoef = rho[i] * (Y[i] * r).sum(dim=1)
                r = torch.addcmul(r, S[i], (alpha[i] - bcoef).unsqueeze(1))
            d = (-r).clamp(-MAXS, MAXS)
            gd = (g * d).sum(dim=1)
            bad = ~(gd < 0)
            d = torch.where(bad.unsqueeze(1), -g, d)
            gd = torch.where(bad, -(g * g).sum(dim=1), gd)
            scale_f = 1.0 + f.abs()
            gmax = g.abs().max(dim=1).values
            stop = (last_df <= self.ftol * scale_f) & (-gd <= self.ptol * scale_f) & (gmax <= self.gtol * scale_f)
            stop = (stop | (gmax <= 1e-9)) & live
            if bool(stop.any()):
                if self.escape is not None:
                    si = nz(stop)
                    xs, moved = self.escape(ids[si], x[si], g[si])
                    if bool(moved.any()):
                        mi = si[moved]
                        x[mi] = xs[moved]
                        fm, gm, hm = self.vg(x[mi], ids[mi])
                        f[mi], g[mi] = fm, gm
                        if hm is not None:
                            hdiag[mi] = hm
                        rho[:, mi] = 0.0
                        nhist[mi] = 0
                        last_df[mi] = inf
                        continue          # every direction is recomputed from the new points
                conv = conv | stop
                live = ~conv
                if not bool(live.any()):
                    continue
            # line search (Armijo, parabolic backtracking) on the live problems
            step0 = torch.where((nhist == 0) & ~have_h, (1.0 / gmax.clamp(min=1e-300)).clamp(max=1.0), torch.ones_like(gmax))
            dmax = d.abs().max(dim=1).values
            t = torch.minimum(step0, MAXS / dmax.clamp(min=1e-300))
            xnew, fnew = x.clone(), f.clone()
            pend = live.clone()
            all_rows = bool(live.all())
            for rnd in range(30):
                pidx = None if (rnd == 0 and all_rows) else nz(pend)
                if pidx is not None and pidx.numel() == 0:
                    break
                if pidx is None:
                    xt = torch.maximum(torch.minimum(torch.addcmul(x, d, t.unsqueeze(1)), self.hi), self.lo)
                    ft = self.value(xt, ids)
                    ok = (ft <= f + 1e-4 * t * gd) & torch.isfinite(ft)
                    xnew = torch.where(ok.unsqueeze(1), xt, xnew)
                    fnew = torch.where(ok, ft, fnew)
                    t = torch.where(ok, t, _backtrack(torch, t, ft, f, gd))
                    pend = ~ok
                    if bool(ok.all()):
                        break
                else:
                    tp, fp, gp = t[pidx], f[pidx], gd[pidx]
                    xt = torch.maximum(torch.minimum(torch.addcmul(x[pidx], d[pidx], tp.unsqueeze(1)), self.hi), self.lo)
                    ft = self.value(xt, ids[pidx])
                    ok = (ft <= fp + 1e-4 * tp * gp) & torch.isfinite(ft)
                    acc = pidx[ok]
                    xnew[acc], fnew[acc] = xt[ok], ft[ok]
                    rej = ~ok
                    t[pidx[rej]] = _backtrack(torch, tp[rej], ft[rej], fp[rej], gp[rej])
                    pend[acc] = False
            failed = pend & live
            if all_rows:
                fx, gx, hx = self.vg(xnew, ids)
            else:
                li = nz(live)
                fl_, gl_, hl_ = self.vg(xnew[li], ids[li])
                fx, gx = f.clone(), g.clone()
                fx[li], gx[li] = fl_, gl_
                hx = None
                if hl_ is not None:
                    hx = hdiag.clone()
                    hx[li] = hl_
            if hx is not None:
                hdiag = hx
            lv = live.unsqueeze(1)
            s_ = torch.where(lv, xnew - x, torch.zeros_like(x))
            y_ = torch.where(lv, gx - g, torch.zeros_like(g))
            sy = (s_ * y_).sum(dim=1)
            yy = (y_ * y_).sum(dim=1)
            upd = (sy > 1e-12 * torch.sqrt((s_ * s_).sum(dim=1) * yy + 1e-300)) & ~failed & live
            slot = it % m
            S[slot], Y[slot] = s_, y_
            rho[slot] = torch.where(upd, 1.0 / torch.where(upd, sy, torch.ones_like(sy)), torch.zeros_like(sy))
            gamma = torch.where(upd, sy / yy.clamp(min=1e-300), gamma)
            nhist = nhist + upd.to(torch.int64)
            df = f - fx
            last_df = torch.where(live, torch.where(failed, torch.zeros_like(df), df), last_df)
            # a failed line search with correction pairs in play: forget them and try once more along the (preconditioned)
            # gradient before giving the point up as converged
            retry = failed & (nhist > 0)
            rho = torch.where(retry.unsqueeze(0), torch.zeros_like(rho), rho)
            nhist = torch.where(retry, torch.zeros_like(nhist), nhist)
            last_df = torch.where(retry, torch.full_like(last_df, inf), last_df)
            done = failed & ~retry & (fresh | (nhist == 0))
            fresh = torch.where(live, retry, fresh)
            x = torch.where(lv, xnew, x)
            f = torch.where(live, fx, f)
            g = torch.where(lv, gx, g)
            iters = iters + live.to(torch.int64)
            conv = conv | done
        out_x[ids], out_f[ids], out_iters[ids] = x, f, iters
        self.iters = out_iters
        return out_x, out_f


class DeviceGrmFitter:
    """Objective of the general model on the device: coordinates (log rates [5], log b [branches]) with b = t * totalFactor(r)
    the branch lengths in expected substitutions (stage1.Stage1._grm_point), value from `tphip_locus_loglik_dev`, value +
    gradient + curvature of the branch lengths from `tphip_locus_gradient_dev`, chain rule and boundary escape as in
    stage1.py (`_grm_value_and_grad`, `_grm_escape`)."""

    def __init__(self, plan, d_states_ptr, pi, branches, nn, device=0):
        import torch
        self.torch = torch
        self.plan = plan
        self.dev = torch.device("cuda", device)
        self.d_states_ptr = int(d_states_ptr)
        self.pi = torch.as_tensor(np.ascontiguousarray(pi), dtype=torch.float64, device=self.dev)       # [L, 4]
        self.branches = torch.as_tensor(np.ascontiguousarray(branches), dtype=torch.int64, device=self.dev)
        self.nn = int(nn)
        self.dk = torch.stack([2.0 * self.pi[:, i] * self.pi[:, j] for i, j in _s1._PAIRS], dim=1)       # [L, 6]
        self.free = torch.tensor([0, 2, 3, 4, 5], dtype=torch.int64, device=self.dev)
        L = self.pi.shape[0]
        self.kicks = torch.zeros(L, dtype=torch.int64, device=self.dev)
        self.last_f = torch.zeros(L, dtype=torch.float64, device=self.dev)
        self.nevals = 0
        self.ngrads = 0
        self.stream = torch.cuda.current_stream(self.dev).cuda_stream

    def _point(self, X, idx):
        torch = self.torch
        n = X.shape[0]
        r = torch.exp(X[:, :5])
        exch = torch.ones((n, 6), dtype=torch.float64, device=self.dev)
        exch[:, self.free] = r
        scale = 1.0 / (exch * self.dk[idx]).sum(dim=1)
        vecs = torch.zeros((n, self.nn), dtype=torch.float64, device=self.dev)
        vecs[:, self.branches] = torch.exp(X[:, 5:])
        return exch, scale, vecs

    def _cand(self, idx, n):
        torch = self.torch
        return (idx.to(torch.int32).contiguous(), torch.arange(n, dtype=torch.int32, device=self.dev),
                torch.full((n,), -1, dtype=torch.int32, device=self.dev), torch.ones(n, dtype=torch.float64, device=self.dev))

    def value(self, X, idx):
        torch = self.torch
        n = X.shape[0]
        if n == 0:
            return torch.zeros(0, dtype=torch.float64, device=self.dev)
        exch, scale, vecs = self._point(X, idx)
        loc, vec, pidx, pfac = self._cand(idx, n)
        out = torch.empty(n, dtype=torch.float64, device=self.dev)
        self.plan.locus_loglik_dev(self.d_states_ptr, n, loc, exch, vecs, vec, scale.contiguous(), pidx, pfac, out, self.stream)
        self.nevals += n
        return -out

    def value_and_grad(self, X, idx):
        torch = self.torch
        n, D = X.shape
        exch, scale, vecs = self._point(X, idx)
        loc, vec, pidx, pfac = self._cand(idx, n)
        lnl = torch.empty(n, dtype=torch.float64, device=self.dev)
        dex = torch.empty((n, 6), dtype=torch.float64, device=self.dev)
        dlt = torch.empty((n, self.nn), dtype=torch.float64, device=self.dev)
        sdl = torch.empty(n, dtype=torch.float64, device=self.dev)
        d2 = torch.empty((n, self.nn), dtype=torch.float64, device=self.dev)
        self.plan.locus_gradient_dev(self.d_states_ptr, n, loc, exch, vecs, vec, scale.contiguous(), pidx, pfac, lnl, dex, dlt, sdl,
                                     d2, self.stream)
        self.nevals += n
        self.ngrads += n
        self.last_f[idx] = -lnl
        # t_b = b_b / totalFactor(r): d log t_b / d r_q = -(2 pi_i pi_j) / totalFactor for every branch
        dr = (dex - sdl.unsqueeze(1) * self.dk[idx] * scale.unsqueeze(1)) * exch        # d lnL / d log r_q at fixed b
        g = torch.empty((n, D), dtype=torch.float64, device=self.dev)
        g[:, :5] = dr[:, self.free]
        g[:, 5:] = dlt[:, self.branches]
        h = torch.full((n, D), float("nan"), dtype=torch.float64, device=self.dev)
        h[:, 5:] = -d2[:, self.branches]
        return -lnl, -g, h

    def escape(self, idx, X, G):
        """stage1.Stage1._grm_escape: collapsed branches (and rates at their lower bound) are tested in the original
        parametrisation at a would-be stopping point and put back if the likelihood wants them longer."""
        torch = self.torch
        Xn = X.clone()
        logb = X[:, 5:]
        b = torch.exp(logb)
        slope = G[:, 5:] / b
        fscale = (1.0 + self.last_f[idx].abs()).unsqueeze(1)
        few = (self.kicks[idx] < 3).unsqueeze(1)
        kick = (b < 1e-6) & (slope * _s1.ESCAPE_LENGTH < -1e-7 * fscale) & few
        Xn[:, 5:] = torch.where(kick, torch.full_like(logb, float(np.log(_s1.ESCAPE_LENGTH))), logb)
        rslope = G[:, :5] / torch.exp(X[:, :5])
        rkick = (X[:, :5] < _s1.LOG_RATE_MIN + 1.0) & (rslope * _s1.ESCAPE_RATE < -1e-7 * fscale) & few
        Xn[:, :5] = torch.where(rkick, torch.full_like(X[:, :5], float(np.log(_s1.ESCAPE_RATE))), X[:, :5])
        moved = kick.any(dim=1) | rkick.any(dim=1)
        self.kicks[idx] += moved.to(torch.int64)
        return Xn, moved

    def fit(self, x0, lo, hi, maxit):
        torch = self.torch
        t = lambda a: torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64, device=self.dev)  # noqa: E731
        opt = DeviceLBFGS(torch, self.value, self.value_and_grad, t(x0), maxit=maxit, lo=t(lo), hi=t(hi), escape=self.escape)
        x, f = opt.run()
        return x.cpu().numpy(), f.cpu().numpy(), opt.iters.cpu().numpy()
