"""New transition-matrix gradient kernel (taken when the alignment is the library's cached copy with packed codes) against
locus_grad_kernel (no cache) on random candidates: every output.  usage: grad2_check.py [TAXA] [COLS] [LOCI]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tapir_amd import engine, synth
nt = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 700
L = int(sys.argv[3]) if len(sys.argv) > 3 else 5
d = synth.simulate(L, n, nt, 7)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
st[1, 3:40] = 15; st[2, 50:60] = 5; st[0, 100:110] = 10
plan = engine.Plan(nt, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], np.ones((L, 6)), pin["T"], [1], [[0, 1]],
                   correction=pin["correction"])
rng = np.random.default_rng(1)
nc = 3 * L
loc = np.repeat(np.arange(L), 3)
exch = np.exp(rng.normal(0, 0.5, (nc, 6))); exch[:, 1] = 1.0
nn = len(pin["parent"])
vecs = np.asarray(pin["blen"])[None, :] * np.exp(rng.normal(-4.5, 1.0, (nc, nn)))
vecs[:, -1] = 0.0
vecs[0, 2] = 1e-9; vecs[1, 5] = 3.0
w = rng.integers(1, 4, st.shape[1]).astype(np.float64)
for weights in (None, w):
    plan.set_column_weights(weights)
    old = plan.locus_gradient(st, vecs, loc, exch, curvature=True)
    cache = plan.device_cache()
    t0 = time.time()
    new = plan.locus_gradient(st, vecs, loc, exch, curvature=True, cache=cache)
    cache.release()
    for name, a, b in zip(("lnl", "dexch", "dlogt", "sum_dlogt", "d2logt"), old, new):
        scale = np.maximum(np.abs(a).max(), 1e-300)
        print("%-10s max |old| %.3e  max |new - old| %.3e  (relative to the largest entry %.2e)" % (name, np.abs(a).max(), np.abs(a - b).max(), np.abs(a - b).max() / scale))
plan.close()
