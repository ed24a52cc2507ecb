"""General-model start: the input tree's shape times a grid of scales against the better of that and per-branch parsimony counts,
when the input tree's branch lengths are off (each multiplied by exp(N(0, SIGMA))) from the tree the data were simulated on.
usage: s1_start_perturbed.py LOCI COLS TAXA SIGMA [pars|shrunk|both ...]   (values of TPHIP_S1_START compared with grid)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tapir_amd import engine, nexus, synth
L, n, nt = (int(x) for x in sys.argv[1:4])
sigma = float(sys.argv[4])
d = synth.simulate(L, n, nt, 5)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
pi = nexus.base_frequencies_from_histogram(engine.state_histogram(st, d["locus_offsets"]))
rng = np.random.default_rng(3)
blen = np.asarray(pin["blen"], dtype=np.float64) * np.exp(sigma * rng.standard_normal(len(pin["blen"])))
plan = engine.Plan(nt, pin["parent"], blen, pin["leaf"], d["locus_offsets"], pi, np.ones((L, 6)), pin["T"], [1], [[0, 1]], correction=pin["correction"])
cache = plan.device_cache()
res = {}
for name in ["grid"] + sys.argv[5:]:
    os.environ["TPHIP_S1_START"] = name
    plan.stage1_fit(st, cache=cache)
    t0 = time.time(); out = plan.stage1_fit(st, cache=cache); dt = time.time() - t0
    res[name] = out
    print("%-5s %.3f s  general model: iterations mean %.1f max %d, gradients %d, values %d" % (name, dt, out["grm_iters"].mean(), out["grm_iters"].max(),
          out["stats"]["grm_grads"], out["stats"]["grm_evals"]), flush=True)
for name in sys.argv[5:]:
    dl = res[name]["lnl"][:, 0] - res["grid"]["lnl"][:, 0]
    rel = np.abs(res[name]["exch"] - res["grid"]["exch"]) / res["grid"]["exch"]
    print("%s vs grid: general-model lnL better by > 1e-3 on %d loci (max %.3e), worse on %d (max %.3e); averaged rates max rel diff %.2e" % (
        name, (dl > 1e-3).sum(), dl.max(), (dl < -1e-3).sum(), -dl.min(), rel.max()))
cache.release(); plan.close()
