#!/usr/bin/env python3
"""Capture golden vectors from the reference's own code (run in the build container ONLY).

The reference (/root/reference) never travels to the GPU box, so this script is run once here and
its *outputs* (data: inputs + expected outputs) are committed under tests/golden/.

What is captured
----------------
* The data files the reference's own tests hold (tapir/tests/test-data/*.nex, *.tree, *.npy, *.json and
  tapir/tests/test-hyphy/chr1_918.subsmodel.phydesign.rates parsed to a table).  These are BSD-3
  licensed data (see tests/golden/LICENSE.reference-data).
* Outputs of /root/reference/tapir/compute.py (imported in-process with two shims, nothing fetched:
  an empty `dendropy` module and `scipy.vectorize = numpy.vectorize`) on
  the bundled 100-rate JSON and on seeded random rate vectors:
    get_townsend_pi, net PI (nansum), get_net_pi_for_periods, get_net_integral_for_epochs
    (scipy.integrate.quad integral and abserr, per site and summed), cull_uninformative_rates,
    parse_site_rates(test=True).

Usage:  python tests/golden/make_golden.py      (needs /root/reference; writes tests/golden/*)
"""
import importlib.util
import json
import os
import re
import shutil
import sys
import types

import numpy as np
import scipy

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def load_reference_compute():
    # shim 1: DendroPy is not installed; compute.py only uses it inside two functions we do not call.
    sys.modules.setdefault("dendropy", types.ModuleType("dendropy"))
    # shim 2: scipy.vectorize was an alias of numpy.vectorize and has been removed from modern scipy.
    if not hasattr(scipy, "vectorize"):
        scipy.vectorize = np.vectorize
    spec = importlib.util.spec_from_file_location("ref_compute", os.path.join(REF, "tapir", "compute.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def copy_data_files():
    src = os.path.join(REF, "tapir", "tests", "test-data")
    for name in ("chr1_918.nex", "Euteleost.tree", "informativeness_cutoff.nex",
                 "test-uniform-draw-weights.rates.json", "chr1_918-test-cutoff-values.npy",
                 "test-30-50-integral.npy", "test-R-townsend-output.npy", "test-culled-rates.npy",
                 "test-parsed-rates.npy"):
        dst = os.path.join(HERE, name)
        shutil.copyfile(os.path.join(src, name), dst)
        os.chmod(dst, 0o644)
    shutil.copyfile(os.path.join(REF, "LICENSE.txt"), os.path.join(HERE, "LICENSE.reference-data"))
    os.chmod(os.path.join(HERE, "LICENSE.reference-data"), 0o644)


def parse_phydesign_rates():
    """tapir/tests/test-hyphy/chr1_918.subsmodel.phydesign.rates -> JSON table (data only)."""
    path = os.path.join(REF, "tapir", "tests", "test-hyphy", "chr1_918.subsmodel.phydesign.rates")
    txt = open(path).read()
    out = {}
    out["chronogram_length"] = float(re.search(r"Chronogram length \(time units\):\s*([0-9.]+)", txt).group(1))
    freqs = re.search(r"Base freqs: \{\s*\{\s*([0-9.]+)\}\s*\{\s*([0-9.]+)\}\s*\{\s*([0-9.]+)\}\s*\{\s*([0-9.]+)\}", txt)
    out["freqs_ACGT"] = [float(freqs.group(i)) for i in range(1, 5)]
    for k in ("AC", "AG", "AT", "CG", "CT", "GT"):
        out[k] = float(re.search(r"\b%s:\s*([0-9.]+)" % k, txt).group(1))
    rows = re.findall(r"Site\s+(\d+) Total subst =\s*([0-9.\-eE+]+) subst, Rate =\s*([0-9.\-eE+]+) subst/time, "
                      r"Log\(L\)\s*([0-9.\-eE+]+)", txt)
    out["site"] = [int(r[0]) for r in rows]
    out["subst"] = [float(r[1]) for r in rows]
    out["rate"] = [float(r[2]) for r in rows]
    out["ll"] = [float(r[3]) for r in rows]
    assert out["site"] == list(range(1, 227)), len(rows)
    out["source"] = "tapir/tests/test-hyphy/chr1_918.subsmodel.phydesign.rates (PhyDesign web site, hyphy1)"
    with open(os.path.join(HERE, "chr1_918_phydesign_rates.json"), "w") as fh:
        json.dump(out, fh, indent=1)


def capture(ref):
    loc = os.path.join(REF, "tapir", "tests", "test-data")
    cases = {}
    # ---- case A: the bundled 100-rate JSON, time 0..173 (test_compute.py:41-43), README/test intervals
    rates = ref.parse_site_rates(os.path.join(loc, "test-uniform-draw-weights.rates.json"), test=True)
    rates10 = ref.parse_site_rates(os.path.join(loc, "test-uniform-draw-weights.rates.json"), 10.0, test=True)
    cases["A"] = dict(rates=rates, T=174, times=[10, 20, 50],
                      intervals=[[0, 10], [10, 15], [15, 20], [20, 30], [20, 70], [20, 100], [30, 50]])
    # ---- case B: seeded gamma rates with NaNs (culled) mixed in, T=100, the synthetic-bench times/intervals
    rng = np.random.default_rng(0)
    rb = rng.gamma(0.5, 0.02, 4096)
    rb[rng.random(4096) < 0.1] = np.nan
    cases["B"] = dict(rates=rb, T=100, times=[10, 30, 50, 90],
                      intervals=[[5, 15], [25, 35], [45, 55], [85, 95]])
    # ---- case C: wide dynamic range incl. very small and very large rates (adaptive quad paths)
    rng = np.random.default_rng(1)
    rc = np.concatenate([10.0 ** rng.uniform(-7, 1.2, 1500), [0.0, 2.5e-6, 1e-12, 4.942, 0.662, 4.783, 25.0, 300.0]])
    cases["C"] = dict(rates=rc, T=64, times=[0, 1, 63],
                      intervals=[[0, 10], [0, 1], [1, 2], [3, 7], [20, 100], [50, 51], [0, 500]])
    out = {"parsed_rates_A": rates, "parsed_rates_A_div10": rates10}
    for name, c in cases.items():
        r = np.asarray(c["rates"], dtype=np.float64)
        tv = ref.get_time(0, c["T"])
        pi = ref.get_townsend_pi(tv, r)
        net = np.nansum(pi, axis=1)
        disc = ref.get_net_pi_for_periods(pi, c["times"])
        fin = r[np.isfinite(r)]
        vec = np.vectorize(ref.get_integral_over_times)
        integ = np.zeros((len(c["intervals"]), fin.size))
        err = np.zeros_like(integ)
        for k, (a, b) in enumerate(c["intervals"]):
            integ[k], err[k] = vec(a, b, fin)
        ep = ref.get_net_integral_for_epochs(fin, c["intervals"])
        out[name + "_rates"] = r
        out[name + "_T"] = np.int64(c["T"])
        out[name + "_times"] = np.asarray(c["times"], dtype=np.int64)
        out[name + "_intervals"] = np.asarray(c["intervals"], dtype=np.int64)
        out[name + "_net"] = net
        out[name + "_disc"] = np.asarray([disc[t] for t in c["times"]])
        out[name + "_site_integral"] = integ
        out[name + "_site_abserr"] = err
        out[name + "_sum_integral"] = np.asarray([ep["%d-%d" % (a, b)]["sum(integral)"] for a, b in c["intervals"]])
        out[name + "_sum_error"] = np.asarray([ep["%d-%d" % (a, b)]["sum(error)"] for a, b in c["intervals"]])
        if name == "A":
            out["A_pi_dense"] = pi
    # culling: reference cull_uninformative_rates on the bundled mask golden (first 100) x rates/10
    mask = np.load(os.path.join(loc, "chr1_918-test-cutoff-values.npy"))[:100]
    out["A_culled_div10"] = ref.cull_uninformative_rates(rates10, mask)
    np.savez_compressed(os.path.join(HERE, "reference_compute_outputs.npz"), **out)
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("make_golden.py must run in the build container (needs %s)" % REF)
    copy_data_files()
    parse_phydesign_rates()
    ref = load_reference_compute()
    out = capture(ref)
    print("captured", sorted(out.keys()))
    print("A net[1:6]      ", out["A_net"][1:6])
    print("A disc          ", out["A_disc"])
    print("A sum_integral  ", out["A_sum_integral"])
    print("A sum_error     ", out["A_sum_error"])


if __name__ == "__main__":
    main()
