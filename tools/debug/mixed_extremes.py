import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tapir_amd import engine, synth
rng = np.random.default_rng(5)
def run(ntaxa, lens, seed, env):
    for k in ("TPHIP_SITE_MIXED", "TPHIP_SITE_WAVES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    L = len(lens)
    d = synth.simulate(1, int(sum(lens)) + 1, ntaxa, seed)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = np.ascontiguousarray(d["states"].numpy()[:, : int(sum(lens))])
    off = np.zeros(L + 1, np.int64); off[1:] = np.cumsum(lens)
    pi = np.tile(d["pi"][0], (L, 1)) * rng.uniform(0.8, 1.2, (L, 4)); pi /= pi.sum(1, keepdims=True)
    ex = np.tile(d["exch"][0], (L, 1)) * rng.uniform(0.5, 2.0, (L, 6)); ex[:, 1] = 1.0
    plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off, pi, ex, pin["T"], [10], [[5, 15]], correction=pin["correction"])
    out = plan.run_fused(st)
    plan.close()
    return out
cases = [("5000 loci of 0-3 columns, 16 taxa", 16, rng.integers(0, 4, 5000)),
         ("one locus of 100000 columns, 16 taxa", 16, np.array([100000])),
         ("3 loci 1 / 70000 / 2 columns, 40 taxa", 40, np.array([1, 70000, 2])),
         ("20000 loci of 8-40 columns, 8 taxa", 8, rng.integers(8, 41, 20000)),
         ("700 loci of 500-1500 columns, 64 taxa", 64, rng.integers(500, 1501, 700))]
for name, nt, lens in cases:
    rng = np.random.default_rng(11)
    a = run(nt, lens, 3, dict(TPHIP_SITE_MIXED="0"))
    rng = np.random.default_rng(11)
    b = run(nt, lens, 3, {})
    rng = np.random.default_rng(11)
    c = run(nt, lens, 3, dict(TPHIP_SITE_MIXED="1", TPHIP_SITE_WAVES="37"))
    same = all(np.array_equal(a[k], b[k], equal_nan=True) and np.array_equal(a[k], c[k], equal_nan=True) for k in ("rate", "subst", "lnl", "flag", "nres", "tables"))
    print("%-45s columns %8d optimised %8d: mixed (default grid, 37 waves) == slices: %s" % (name, int(sum(lens)), int((a["flag"] == 0).sum()), same), flush=True)
    assert same
