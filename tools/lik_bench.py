"""Micro-benchmark of the stage-1 kernels (locus_loglik_kernel / locus_grad_kernel) on a fixed synthetic batch.
usage: python tools/lik_bench.py LOCI COLS TAXA NCAND_PER_LOCUS [REPS]   (wall time includes the small host copies)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tapir_amd import engine, synth

L, n, nt, per = (int(x) for x in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
d = synth.simulate(L, n, nt, 5)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
plan = engine.Plan(nt, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], np.ones((L, 6)), pin["T"], [1], [[0, 1]],
                   correction=pin["correction"])
rng = np.random.default_rng(0)
ncand = L * per
cl = np.repeat(np.arange(L), per)
ce = np.exp(rng.normal(0, 0.3, (ncand, 6)))
vec = (np.asarray(pin["blen"]) / pin["correction"])[None, :] * np.ones((L, 1)) * 0.01
cache = plan.device_cache()
plan.locus_loglik(st, vec, cl, ce, cl, cache=cache)
plan.locus_gradient(st, vec, cl, ce, cl, cache=cache, per_branch=False)
t0 = time.time()
for _ in range(reps):
    plan.locus_loglik(st, vec, cl, ce, cl, cache=cache)
t1 = time.time()
for _ in range(reps):
    plan.locus_gradient(st, vec, cl, ce, cl, cache=cache, per_branch=False)
t2 = time.time()
cols = ncand * n
print("%d candidates x %d cols x %d taxa: value %.3f ms (%.3e col-evals/s), gradient %.3f ms (%.3e col-grads/s), ratio %.2f"
      % (ncand, n, nt, 1e3 * (t1 - t0) / reps, cols * reps / (t1 - t0), 1e3 * (t2 - t1) / reps, cols * reps / (t2 - t1), (t2 - t1) / (t1 - t0)))
cache.release(); plan.close()
