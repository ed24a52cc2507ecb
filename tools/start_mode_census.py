"""How many columns end on a different optimum when the optimiser starts where HyPhy does?

The reference starts every column at siteRate = 1 (models_and_rates.bf:1050) and reports the local optimum uphill of it
(SURVEY F4); the product starts at the column's parsimony rate and stops through accelerated exits.  This census runs
the CPU oracle in both modes (oracle.site_rates(start_mode=0 / 1); the GPU agrees with mode 0 to 1e-6 by the parity
tests) on samples of the BASELINE shapes and on noisy 5-taxon columns, and counts columns whose flags differ or whose
rates differ by more than 1e-6 relative.  CPU only; a few minutes.

    python tools/start_mode_census.py [columns per shape, default 100000]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from tapir_amd import synth  # noqa: E402


def census(name, nloci, ncols, ntaxa, seed, **kw):
    d = synth.simulate(nloci, ncols, ntaxa, seed, **kw)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    n = flagdiff = ratediff = opt = 0
    worst = 0.0
    ev0 = ev1 = 0
    t0 = time.time()
    examples = []
    for l in range(nloci):
        sl = slice(l * ncols, (l + 1) * ncols)
        a = orc.site_rates(st[:, sl], pin["parent"], pin["blen"], pin["leaf"], d["pi"][l], d["exch"][l], start_mode=0)
        b = orc.site_rates(st[:, sl], pin["parent"], pin["blen"], pin["leaf"], d["pi"][l], d["exch"][l], start_mode=1)
        ev0 += a["nevals"]
        ev1 += b["nevals"]
        n += ncols
        fd = a["flag"] != b["flag"]
        flagdiff += int(fd.sum())
        both = (a["flag"] == 0) & (b["flag"] == 0)
        opt += int(both.sum())
        rel = np.zeros(ncols)
        rel[both] = np.abs(a["rate"][both] - b["rate"][both]) / np.maximum(b["rate"][both], 1e-300)
        bad = rel > 1e-6
        ratediff += int(bad.sum())
        worst = max(worst, float(rel[~bad].max()) if (~bad).any() else 0.0)
        for c in np.flatnonzero(bad | fd)[:3]:
            examples.append((l, int(c), int(a["flag"][c]), int(b["flag"][c]), float(a["rate"][c]), float(b["rate"][c]),
                             float(a["lnl"][c]), float(b["lnl"][c])))
    print("%-28s columns %7d  optimised in both %7d  flag differs %4d  rate differs >1e-6 %4d  worst agreeing rel %.2e  "
          "evals/col %.2f vs %.2f  (%.0f s)" % (name, n, opt, flagdiff, ratediff, worst, ev0 / n, ev1 / n, time.time() - t0))
    for e in examples[:8]:
        print("    locus %d col %d: flags %d/%d rate %.6g / %.6g  lnL %.9f / %.9f" % e)
    return n, flagdiff, ratediff


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    census("C2 shape (16 taxa)", N // 500, 500, 16, synth.WORKLOAD_SEED["C2"])
    census("C3 shape (64 taxa)", max(1, N // 50000), min(N, 50000), 64, synth.WORKLOAD_SEED["C3"])
    census("5 taxa, fast noisy columns", N // 1000, 1000, 5, 4242, rate_mean=0.02, gap_frac=0.15)
    census("C5 shape (256 taxa)", max(1, N // 40000), min(N, 2000), 256, synth.WORKLOAD_SEED["C5"])
