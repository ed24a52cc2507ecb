# CPU oracle: |f'/f''| at the optimiser's answer for every optimised column of a synthetic locus (the size of the Newton
# correction that is still on the table = the error of the reported log-rate), for the product's optimiser (start_mode 0).
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as orc
from tapir_amd import synth
ntaxa, ncols, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
kw = {}
if len(sys.argv) > 4: kw["rate_mean"] = float(sys.argv[4])
if len(sys.argv) > 5: kw["gap_frac"] = float(sys.argv[5])
d = synth.simulate(1, ncols, ntaxa, seed, **kw)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
args = (st, np.asarray(pin["parent"]), np.asarray(pin["blen"]), np.asarray(pin["leaf"]), np.asarray(d["pi"][0]), np.asarray(d["exch"][0]))
res = orc.site_rates(*args)
kappa = orc.gtr_eigen(args[4], args[5])[3]
idx = np.flatnonzero((res["flag"] == 0) & (res["rate"] > 0))
worst = []
for c in idx:
    u = np.log(res["rate"][c] / kappa)
    f, g, h = orc.column_curve(*args, int(c), np.array([u]))
    if h[0] < 0:
        worst.append((abs(g[0] / h[0]), abs(h[0]), abs(g[0]), int(c)))
W = np.array(worst)
r = W[:, 0]
print("taxa %d cols %d %s: optimised %d, evaluations/optimised column %.3f; |f'/f''|: max %.2e, > 1e-6: %d, > 1e-7: %d, > 1e-8: %d"
      % (ntaxa, ncols, kw, len(r), res["nevals"] / max(len(idx), 1), r.max(), (r > 1e-6).sum(), (r > 1e-7).sum(), (r > 1e-8).sum()))
for row in W[np.argsort(-r)[:5]]:
    print("   residual %.2e  |h| %.2e  |g| %.2e  column %d" % tuple(row))
