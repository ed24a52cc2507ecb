"""Stress stage 1 on awkward shapes (many taxa, few columns; very low / very high rates): the engine call (tphip_stage1_fit,
no pruning) against the host second-opinion optimiser (tapir_amd/stage1.Stage1, gradient kernel) -- iterations, time,
finiteness, and who ends higher.   usage: python tools/stage1_stress.py"""
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
    t0 = time.time()
    a = plan.stage1_fit(st, prune_models=False)
    ta = time.time() - t0
    s1 = stage1.Stage1(plan, st, pi, pin["parent"], np.asarray(pin["blen"]) / pin["correction"], prune_models=False)
    t0 = time.time()
    b = s1.run()
    tb = time.time() - t0
    hi = s1.grm_iters.max()
    s1.close()
    d0 = a["lnl"][:, 0] - b["lnl"][:, 0]
    dall = a["lnl"] - b["lnl"]
    print("%d loci x %d cols x %d taxa, rate %.4g: finite %s; general-model iterations engine %d / host %d; %.2f s / %.2f s; averaged "
          "rates max rel diff %.2e; lnL(engine) - lnL(host): general model in [%.2e, %.2e]; all models in [%.2e, %.2e]"
          % (L, n, nt, mean, bool(np.all(np.isfinite(a["exch"])) and np.all(np.isfinite(a["lnl"]))), a["grm_iters"].max(), hi, ta, tb,
             np.max(np.abs(a["exch"] - b["exch"]) / b["exch"]), d0.min(), d0.max(), dall.min(), dall.max()))
    plan.close()
