"""Stress the stage-1 optimiser on awkward shapes (many taxa, few columns; very low / very high rates) and report
iterations, the gradient-kernel run against the finite-difference run, and finiteness.
usage: python tools/stage1_stress.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tapir_amd import engine, nexus, stage1, synth

for (L, n, nt, mean) in ((4, 200, 64, 0.004), (4, 150, 32, 0.0002), (4, 300, 24, 0.05), (2, 1000, 128, 0.004)):
    d = synth.simulate(L, n, nt, 41, rate_mean=mean)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    pi = nexus.base_frequencies_from_histogram(engine.state_histogram(st, d["locus_offsets"]))
    plan = engine.Plan(nt, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], pi, np.ones((L, 6)), pin["T"], [1], [[0, 1]],
                       correction=pin["correction"])
    res = {}
    for name, kw in (("analytic", dict()), ("fd", dict(analytic=False))):
        s1 = stage1.Stage1(plan, st, pi, pin["parent"], np.asarray(pin["blen"]) / pin["correction"], prune_models=False, **kw)
        t0 = time.time()
        r = s1.run()
        res[name] = (r, time.time() - t0, s1.grm_iters.max())
        s1.close()
    a, b = res["analytic"][0], res["fd"][0]
    print("%d loci x %d cols x %d taxa, rate %.4g: finite %s; grm iters %d / %d; %.2f s / %.2f s; max rel diff exch %.2e; lnL(analytic) - lnL(fd) general model in [%.2e, %.2e]; all models in [%.2e, %.2e]"
          % (L, n, nt, mean, bool(np.all(np.isfinite(a["exch"])) and np.all(np.isfinite(a["lnl"]))), res["analytic"][2], res["fd"][2],
             res["analytic"][1], res["fd"][1], np.max(np.abs(a["exch"] - b["exch"]) / b["exch"]),
             np.min(a["lnl"][:, 0] - b["lnl"][:, 0]), np.max(a["lnl"][:, 0] - b["lnl"][:, 0]),
             np.min(a["lnl"] - b["lnl"]), np.max(a["lnl"] - b["lnl"])))
    plan.close()
