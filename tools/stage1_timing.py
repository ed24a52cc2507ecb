"""Wall time of stage 1 through the ENGINE call (tphip_stage1_fit: optimisers as device kernels) on a synthetic batch,
optionally beside the round-2 host optimiser (tapir_amd/stage1.py) on the same bytes.
usage: python tools/stage1_timing.py LOCI COLS TAXA [compare] [reps=N] [cuda]   (cuda: simulate the batch on the GPU -- another
random stream than the default CPU simulation, much faster to set up)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tapir_amd import engine, nexus, stage1, synth

L, n, nt = (int(x) for x in sys.argv[1:4])
compare = "compare" in sys.argv[4:]
reps = max([int(a.split("=")[1]) for a in sys.argv[4:] if a.startswith("reps=")] + [2])
d = synth.simulate(L, n, nt, 5, device="cuda") if "cuda" in sys.argv[4:] else synth.simulate(L, n, nt, 5)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].cpu().numpy()
pi = nexus.base_frequencies_from_histogram(engine.state_histogram(st, d["locus_offsets"]))
plan = engine.Plan(nt, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], pi, np.ones((L, 6)), pin["T"], [1], [[0, 1]],
                   correction=pin["correction"])
cache = plan.device_cache()
for rep in range(reps):
    t0 = time.time()
    out = plan.stage1_fit(st, cache=cache)
    dt = time.time() - t0
    print("engine pass %d: %.3f s  %s" % (rep, dt, out["stats"]), flush=True)
print("loci %d cols %d taxa %d: engine stage 1 %.3f s = %.3e columns/s; general model iterations max %d mean %.1f; "
      "rate-class iterations max %d mean %.2f" % (L, n, nt, dt, L * n / dt, out["grm_iters"].max(), out["grm_iters"].mean(),
                                                   out["sub_iters"].max(), out["sub_iters"].mean()))
if compare:
    t0 = time.time()
    ref = stage1.model_averaged_exchangeabilities(plan, st, pi, pin["parent"], np.asarray(pin["blen"]), engine_fit=False)
    print("host optimiser (round 2): %.3f s" % (time.time() - t0))
    rel = np.abs(out["exch"] - ref["exch"]) / ref["exch"]
    print("averaged rates: max rel diff %.3e (median %.3e)" % (rel.max(), np.median(rel)))
    print("general-model lnL: max |diff| %.3e; engine better by at most %.3e, worse by at most %.3e" % (
        np.abs(out["lnl"][:, 0] - ref["lnl"][:, 0]).max(), (out["lnl"][:, 0] - ref["lnl"][:, 0]).max(),
        (ref["lnl"][:, 0] - out["lnl"][:, 0]).max()))
    keep = ref["weights"] > 1e-9
    print("lnL of models with weight > 1e-9: max |diff| %.3e; weights max |diff| %.3e" % (
        np.abs(out["lnl"] - ref["lnl"])[keep].max(), np.abs(out["weights"] - ref["weights"]).max()))
cache.release()
plan.close()
