"""Wall time of stage 1 (model-averaged exchangeabilities) on the GPU for a synthetic batch.
usage: python tools/stage1_timing.py LOCI COLS TAXA [analytic|fd]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tapir_amd import engine, nexus, stage1, synth

try:   # the device-resident fitter uses torch for its device memory: pay torch's one-time CUDA start-up outside the timing
    import torch
    if torch.cuda.is_available():
        torch.zeros(1, device="cuda").exp_().sum().item()
except ImportError:
    pass
L, n, nt = (int(x) for x in sys.argv[1:4])
d = synth.simulate(L, n, nt, 5)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
pi = nexus.base_frequencies_from_histogram(engine.state_histogram(st, d["locus_offsets"]))
plan = engine.Plan(nt, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], pi, np.ones((L, 6)), pin["T"], [1], [[0, 1]],
                   correction=pin["correction"])
analytic = None if len(sys.argv) < 5 or sys.argv[4] not in ("analytic", "fd") else (sys.argv[4] == "analytic")
host = "host" in sys.argv[4:]            # fit the constrained models with the numpy L-BFGS instead of the device-resident optimiser
for rep in range(2):                     # the second pass is the steady state (torch loads its kernels lazily on first use)
    s1 = stage1.Stage1(plan, st, pi, pin["parent"], np.asarray(pin["blen"]) / pin["correction"], analytic=analytic,
                       precondition=os.environ.get("S1_NO_PRECOND") is None, device_fit=not host)
    t0 = time.time(); ge, gt, gl = s1.fit_grm(); t1 = time.time()
    e1, g1, l1 = s1.nevals, s1.ngrads, s1.lik_seconds
    sub = s1.fit_submodels(ge, gt, grm_lnl=gl); t2 = time.time()
    s1.close()
print("loci %d cols %d taxa %d (second of two passes; constrained models fitted %s)" % (L, n, nt, "on the device" if s1.sub_device else "on the host"))
print("gradients: general model %d, rate-class models %d" % (g1, s1.ngrads - g1))
print("general model: %.2f s, %d likelihood evaluations, iterations max %d mean %.1f" % (t1 - t0, e1, s1.grm_iters.max(), s1.grm_iters.mean()))
print("202 models   : %.2f s, %d likelihood evaluations, iterations max %d mean %.1f" % (t2 - t1, s1.nevals - e1, s1.sub_iters.max(), s1.sub_iters.mean()))
print("inside likelihood calls (value kernel + copies): general model %.2f s, 202 models %.2f s" % (l1, s1.lik_seconds - l1))
print("models abandoned early: %d of %d" % (getattr(s1, "pruned", 0), L * 202))
print("column-evaluations/s: %.3e" % (s1.nevals * n / (t2 - t0)))
true = np.asarray(d["exch"]); print("max rel err of general-model rates vs generating:", np.max(np.abs(ge - true / true[:, 1:2]) / true))
