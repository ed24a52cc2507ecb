"""Dump engine vs host-optimiser stage-1 results for offline comparison.  usage: s1_ab_dump.py LOCI COLS TAXA OUT.npz"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tapir_amd import engine, nexus, stage1, synth
L, n, nt = (int(x) for x in sys.argv[1:4])
d = synth.simulate(L, n, nt, 5)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
pi = nexus.base_frequencies_from_histogram(engine.state_histogram(st, d["locus_offsets"]))
plan = engine.Plan(nt, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], pi, np.ones((L, 6)), pin["T"], [1], [[0, 1]],
                   correction=pin["correction"])
cache = plan.device_cache()
out = plan.stage1_fit(st, cache=cache)
s1 = stage1.Stage1(plan, st, pi, pin["parent"], np.asarray(pin["blen"]))
ge, gt, gl = s1.fit_grm()
np.savez(sys.argv[4], e_lnl=out["lnl"], e_exch=out["exch"], e_mexch=out["model_exch"], e_blen=out["grm_blen"], e_iters=out["grm_iters"],
         h_lnl0=gl, h_exch0=ge, h_blen=gt, h_iters=s1.grm_iters)
dl = out["lnl"][:, 0] - gl
print("GRM lnL engine - host: min %.3e max %.3e; |diff| > 1e-3: %d of %d; > 1e-2: %d" % (dl.min(), dl.max(), (np.abs(dl) > 1e-3).sum(), L, (np.abs(dl) > 1e-2).sum()))
print("iters engine mean %.1f max %d; host mean %.1f max %d" % (out["grm_iters"].mean(), out["grm_iters"].max(), s1.grm_iters.mean(), s1.grm_iters.max()))
w = np.argsort(-np.abs(dl))[:10]
for l in w:
    print("locus %d: dlnL %.3e engine iters %d host iters %d  min blen engine %.2e host %.2e" % (l, dl[l], out["grm_iters"][l], s1.grm_iters[l],
          out["grm_blen"][l][out["grm_blen"][l] > 0].min(), gt[l][gt[l] > 0].min()))
s1.close(); cache.release(); plan.close()
