"""CPU restatement of HyPhy stage 1 (model-averaged exchangeabilities) -- TEST INFRASTRUCTURE ONLY.

Follows tapir/data/models_and_rates.bf:
  bf:487-520  general reversible model: maximise lnL over AC, AT, CG, CT, GT (AG = 1) and all branch lengths;
  bf:522-540  stash branch lengths as expected substitutions (t * totalFactor);
  bf:542-661  every other partition of the six rates into classes (class of AG fixed at 1), branch lengths :=
              stashed / totalFactor(model), maximise over the free class rates;
  bf:806-847  Akaike weights and the weighted mean of every rate.

Deliberately independent of tapir_amd/stage1.py: the likelihood is the oracle's C pruning (orc_locus_loglik), the
optimiser is scipy's L-BFGS-B with its own forward-difference gradient, and the 203 partitions are enumerated
recursively rather than with the script's nested loops.

Pinned loosely: the only stage-1 output the reference holds is the 2-decimal header of
tapir/tests/test-hyphy/chr1_918.subsmodel.phydesign.rates (PhyDesign's own HyPhy run on the bundled locus); this
restatement reproduces its five model-averaged rates within 5 % (tests/test_oracle_golden.py).  Beyond that parity is
unpinned (no per-model output anywhere; HyPhy is absent and never run here): the GPU stage is checked against this
file, not against HyPhy.
"""
import numpy as np
from scipy.optimize import minimize

from . import oracle as orc

_PAIRS = ((0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3))  # AC AG AT CG CT GT


def partitions6():
    """All 203 set partitions of 6 items as restricted growth strings."""
    out = []

    def rec(prefix, mx):
        if len(prefix) == 6:
            out.append("".join(map(str, prefix)))
            return
        for c in range(mx + 2):
            rec(prefix + [c], max(mx, c))

    rec([0], 0)
    return out


def total_factor(pi, exch):
    return sum(2.0 * pi[i] * pi[j] * exch[k] for k, (i, j) in enumerate(_PAIRS))


def _fit(fun, x0, bounds):
    best = minimize(fun, x0, method="L-BFGS-B", bounds=bounds,
                    options=dict(maxiter=500, ftol=1e-13, gtol=1e-7, eps=1e-6, maxfun=100000, maxls=40))
    # restart once from the optimum: L-BFGS-B sometimes stops early on the flat forward-difference gradient
    again = minimize(fun, best.x, method="L-BFGS-B", bounds=bounds,
                     options=dict(maxiter=500, ftol=1e-14, gtol=1e-8, eps=1e-7, maxfun=100000, maxls=40))
    return again if again.fun <= best.fun else best


def model_averaged(states, parent, blen, leaf_taxon, pi):
    """One locus.  Returns dict(exch[6], weights{model: w}, lnl{model: lnL}, grm_exch, grm_blen)."""
    parent = np.asarray(parent)
    br = np.flatnonzero(parent >= 0)
    pi = np.asarray(pi, dtype=np.float64) / np.sum(pi)

    def exch_of(logr5):
        r = np.exp(logr5)
        return np.array([r[0], 1.0, r[1], r[2], r[3], r[4]])

    def grm_obj(x):
        t = np.zeros(len(parent)); t[br] = np.exp(x[5:])
        return -orc.locus_loglik(states, parent, t, leaf_taxon, pi, exch_of(x[:5]))

    x0 = np.concatenate([np.zeros(5), np.log(np.maximum(np.asarray(blen, dtype=np.float64)[br], 1e-6))])
    bounds = [(-7.0, 9.2)] * 5 + [(-23.0, 4.0)] * len(br)   # same box as the product (a modelling choice, not arithmetic)
    r = _fit(grm_obj, x0, bounds)
    grm_exch = exch_of(r.x[:5])
    grm_t = np.zeros(len(parent)); grm_t[br] = np.exp(r.x[5:])
    stash = grm_t * total_factor(pi, grm_exch)
    lnl = {"012345": -r.fun}
    rates = {"012345": grm_exch}
    nfree = {"012345": 5}
    for s in partitions6():
        if s == "012345":
            continue
        free = sorted(set(s) - {s[1]})
        k = len(free)

        def exch_m(x, s=s, free=free):
            return np.array([1.0 if c == s[1] else np.exp(x[free.index(c)]) for c in s])

        def obj(x):
            e = exch_m(x)
            return -orc.locus_loglik(states, parent, stash / total_factor(pi, e), leaf_taxon, pi, e)

        if k == 0:
            lnl[s], rates[s], nfree[s] = -obj(np.zeros(0)), np.ones(6), 0
            continue
        lg = np.log(grm_exch)
        x0 = np.array([np.mean([lg[i] for i in range(6) if s[i] == c]) for c in free])
        rr = _fit(obj, x0, [(-7.0, 9.2)] * k)
        lnl[s], rates[s], nfree[s] = -rr.fun, exch_m(rr.x), k
    keys = list(lnl)
    score = np.array([lnl[m] - nfree[m] for m in keys])
    w = np.exp(score - score.max()); w /= w.sum()
    avg = sum(wi * rates[m] for wi, m in zip(w, keys))
    return dict(exch=avg, weights=dict(zip(keys, w)), lnl=lnl, rates=rates, grm_exch=grm_exch, grm_blen=grm_t)
