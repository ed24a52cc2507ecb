"""Engine stage 1 under environment-variable variants on the same batch: general-model lnL per locus, iterations, time.
usage: s1_variants.py LOCI COLS TAXA VAR1=val,VAR2=val [more variant specs ...]   ('' = defaults)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tapir_amd import engine, nexus, synth
L, n, nt = (int(x) for x in sys.argv[1:4])
d = synth.simulate(L, n, nt, 5)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
pi = nexus.base_frequencies_from_histogram(engine.state_histogram(st, d["locus_offsets"]))
plan = engine.Plan(nt, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], pi, np.ones((L, 6)), pin["T"], [1], [[0, 1]],
                   correction=pin["correction"])
cache = plan.device_cache()
res = []
for spec in sys.argv[4:]:
    kv = [x.split("=") for x in spec.split(",") if x]
    opts = {}
    for k, v in kv:
        if k.startswith("opt:"):
            opts[k[4:]] = int(v)
        else:
            os.environ[k] = v
    plan.stage1_fit(st, cache=cache, **opts)
    t0 = time.time(); out = plan.stage1_fit(st, cache=cache, **opts); dt = time.time() - t0
    for k, v in kv:
        os.environ.pop(k, None)
    res.append((spec or "default", out, dt))
    print("%-40s %.3f s  grm iters mean %.1f max %d  grads %d evals %d  sub iters mean %.2f" % (spec or "default", dt, out["grm_iters"].mean(),
          out["grm_iters"].max(), out["stats"]["grm_grads"], out["stats"]["nevals"], out["sub_iters"].mean()), flush=True)
base = res[0][1]
for name, out, dt in res[1:]:
    dl = out["lnl"][:, 0] - base["lnl"][:, 0]
    rel = np.abs(out["exch"] - base["exch"]) / base["exch"]
    print("%s vs %s: GRM lnL better by >1e-3 on %d loci (max %.3e), worse on %d (max %.3e); sum %.4f; averaged rates max rel diff %.2e" % (
        name, res[0][0], (dl > 1e-3).sum(), dl.max(), (dl < -1e-3).sum(), -dl.min(), dl.sum(), rel.max()))
cache.release(); plan.close()
