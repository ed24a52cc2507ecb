"""CPU experiment: iteration counts of the general-model fit (Stage1.fit_grm) with the oracle likelihood standing in for
the kernels; gradient and Hessian diagonal by central differences (what the gradient kernel returns analytically)."""
import sys, os, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import oracle_engine
from tapir_amd import stage1, synth

class Plan(oracle_engine.Plan):
    device = 0
    def locus_gradient(self, states, blen_vecs, cand_locus, cand_exch, cand_vec=None, cand_scale=None, cand_pidx=None,
                       cand_pfac=None, cache=None, per_branch=True, curvature=False):
        bv = np.asarray(blen_vecs, np.float64); bv = bv.reshape(-1, bv.shape[-1])
        n = len(cand_locus); nn = bv.shape[1]
        ce = np.asarray(cand_exch, np.float64).reshape(n, 6)
        cs = np.ones(n) if cand_scale is None else np.asarray(cand_scale, np.float64)
        br = np.flatnonzero(self.parent >= 0)
        h = 1e-4
        lnl = np.empty(n); dex = np.zeros((n, 6)); dlt = np.zeros((n, nn)); d2 = np.zeros((n, nn))
        for c in range(n):
            b0 = bv[c if cand_vec is None else cand_vec[c]] * cs[c]
            l = cand_locus[c]
            st = np.asarray(states, np.uint8)[:, self.off[l]:self.off[l + 1]]
            f = lambda b, e: oracle_engine.orc.locus_loglik(st, self.parent, b, self.leaf, self.pi[l], e)
            f0 = f(b0, ce[c]); lnl[c] = f0
            for q in range(6):
                e1 = ce[c].copy(); e2 = ce[c].copy(); d = h * max(ce[c][q], 1e-3)
                e1[q] += d; e2[q] -= d
                dex[c, q] = (f(b0, e1) - f(b0, e2)) / (2 * d)
            for b in br:
                b1 = b0.copy(); b2 = b0.copy(); b1[b] *= np.exp(h); b2[b] *= np.exp(-h)
                fp, fm = f(b1, ce[c]), f(b2, ce[c])
                dlt[c, b] = (fp - fm) / (2 * h); d2[c, b] = (fp + fm - 2 * f0) / (h * h)
        sdl = dlt.sum(1)
        return (lnl, dex, dlt, sdl, d2) if curvature else (lnl, dex, dlt, sdl)

def make(nloci, ncols, ntaxa, seed):
    d = synth.simulate(nloci, ncols, ntaxa, seed)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    off = d["locus_offsets"]
    hist = oracle_engine.state_histogram(st, off)
    from tapir_amd import nexus
    pi = nexus.base_frequencies_from_histogram(hist)
    plan = Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off, pi, np.ones((nloci, 6)), 10, [1], [[0, 1]])
    return plan, st, pi, pin

if __name__ == "__main__":
    nloci, ncols, ntaxa = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    plan, st, pi, pin = make(nloci, ncols, ntaxa, 7)
    s1 = stage1.Stage1(plan, st, pi, pin["parent"], pin["blen"], analytic=True, device_fit=False)
    t = time.perf_counter()
    exch, tt, lnl = s1.fit_grm()
    print("iters", s1.grm_iters.tolist(), "grads", s1.ngrads, "values", s1.nevals, "sec %.1f" % (time.perf_counter() - t))
    print("lnl", np.round(lnl, 6).tolist())
    print("exch", np.round(exch, 4).tolist())
