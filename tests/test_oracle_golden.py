"""Pins the CPU oracle (oracle/) against every golden vector the reference holds for this path, and
against outputs of the reference's own compute.py captured by tests/golden/make_golden.py.  CPU only."""
import json
import os

import numpy as np
import pytest


def _rel(a, b, floor=1e-300):
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def test_phydesign_net_pi_and_integrals(oracle, golden_dir):
    """test_compute.py:44-71: PhyDesign web-site values for the bundled 100-rate JSON."""
    data = json.load(open(os.path.join(golden_dir, "test-uniform-draw-weights.rates.json")))
    rates = np.array([r["rate"] for r in data["sites"]["rates"]])
    assert np.array_equal(rates, np.load(os.path.join(golden_dir, "test-parsed-rates.npy")).ravel())
    net = oracle.net_pi(rates, 174)
    np.testing.assert_allclose(net[:6], [0.00000, 0.10778, 0.14484, 0.14616, 0.13132, 0.11089], atol=5e-6)
    for t, e in zip([10, 20, 50], [0.03448, 0.01111, 0.02293]):
        assert abs(net[t] - e) < 5e-6
    iv = [[0, 10], [10, 15], [15, 20], [20, 30], [20, 70], [20, 100]]
    si, se = oracle.net_integrals(rates, iv, 0)
    np.testing.assert_allclose(si, [0.93453, 0.10628, 0.05855, 0.12638, 1.03698, 2.08840], atol=1.01e-5, rtol=0)  # PhyDesign prints 5 truncated decimals
    sc, _ = oracle.net_integrals(rates, iv, 1)
    assert _rel(sc, si).max() < 1e-9


def test_unreferenced_npy_goldens(oracle, golden_dir):
    """test-R-townsend-output.npy and test-30-50-integral.npy (shipped by the reference, loaded by no test)."""
    rates = np.load(os.path.join(golden_dir, "test-parsed-rates.npy")).ravel()
    r_out = np.load(os.path.join(golden_dir, "test-R-townsend-output.npy"))  # (100 sites, 101 times)
    mine = oracle.get_townsend_pi(oracle.get_time(0, 101), rates)
    assert _rel(mine[1:], r_out.T[1:]).max() < 1e-14
    g = np.load(os.path.join(golden_dir, "test-30-50-integral.npy")).ravel()
    got = np.array([oracle.quad_townsend(30, 50, r)[0] for r in rates])
    assert _rel(got, g).max() < 1e-14


@pytest.mark.parametrize("case", ["A", "B", "C"])
def test_c_restatement_vs_reference_compute_outputs(oracle, golden, case):
    """C restatement (PI sums, dqagse) against outputs of /root/reference/tapir/compute.py."""
    rates = golden[case + "_rates"]
    T = int(golden[case + "_T"])
    net = oracle.net_pi(rates, T)
    assert _rel(net[1:], golden[case + "_net"][1:]).max() < 1e-13
    fin = rates[np.isfinite(rates)]
    iv = golden[case + "_intervals"]
    si, se = oracle.net_integrals(rates, iv, 0)
    assert _rel(si, golden[case + "_sum_integral"]).max() < 1e-13
    assert _rel(se, golden[case + "_sum_error"]).max() < 1e-6
    # per-site integral and abserr (a sample of sites per interval keeps this test in seconds)
    idx = np.unique(np.linspace(0, fin.size - 1, 300).astype(int))
    for k, (a, b) in enumerate(iv):
        for i in idx:
            res, err, neval, ier = oracle.quad_townsend(a, b, fin[i])
            ref = golden[case + "_site_integral"][k, i]
            assert abs(res - ref) <= 1e-14 * abs(ref) + 1e-300
            referr = golden[case + "_site_abserr"][k, i]
            assert abs(err - referr) <= 0.1 * referr + 1e-300


def test_numpy_restatement_matches_reference_outputs(oracle, golden):
    """The numpy/scipy restatement of worker()'s PI half (bin/tapir_compute.py:114-122)."""
    for case in "AB":
        rates = golden[case + "_rates"]
        T = int(golden[case + "_T"])
        times = [int(t) for t in golden[case + "_times"]]
        iv = [[int(a), int(b)] for a, b in golden[case + "_intervals"]]
        if case == "B":
            iv = iv[:2]
        pi_net, pi_times, pi_epochs = oracle.worker_tables(rates, T, times, iv)
        assert np.array_equal(pi_net, golden[case + "_net"])
        assert np.array_equal(np.array([pi_times[t] for t in times]), golden[case + "_disc"])
        for k, (a, b) in enumerate(iv):
            assert pi_epochs["%d-%d" % (a, b)]["sum(integral)"] == golden[case + "_sum_integral"][k]
            assert pi_epochs["%d-%d" % (a, b)]["sum(error)"] == golden[case + "_sum_error"][k]


def test_closed_form_vs_quad(oracle, golden):
    r = golden["C_rates"]
    r = r[np.isfinite(r) & (r < 1.0)]
    for a, b in ([0, 10], [3, 7], [20, 100]):
        for rate in r[::25]:
            q = oracle.quad_townsend(a, b, rate)[0]
            c = oracle.lib().orc_integral_closed(float(a), float(b), float(rate))
            assert abs(q - c) <= 3e-8 * abs(q) + 1e-15  # quad's own tolerance is 1.49e-8


def test_site_rates_known_answer_file(oracle, golden_dir):
    """tapir/tests/test-hyphy/chr1_918.subsmodel.phydesign.rates: the only stage-2 pin (4 decimals)."""
    kat = json.load(open(os.path.join(golden_dir, "chr1_918_phydesign_rates.json")))
    names, rows = oracle.read_nexus_matrix(os.path.join(golden_dir, "chr1_918.nex"))
    st = oracle.encode_rows(rows)
    depth, factor, root = oracle.correct_branch_lengths_values(open(os.path.join(golden_dir, "Euteleost.tree")).read())
    assert depth == 174.0 and factor == 100
    parent, blen, leaf = oracle.tree_arrays(root, names)
    assert abs(blen.sum() - kat["chronogram_length"]) < 1e-12
    pi = np.array(kat["freqs_ACGT"])
    exch = np.array([kat[k] for k in ("AC", "AG", "AT", "CG", "CT", "GT")])
    r = oracle.site_rates(st, parent, blen, leaf, pi, exch)
    inf = r["nres"] >= 3
    assert inf.sum() == 180
    assert np.abs(r["lnl"] - np.array(kat["ll"]))[inf].max() < 5.1e-5
    ok = inf & ((r["flag"] == 0) | (r["flag"] == 3))
    assert ok.sum() == 177
    assert np.abs(r["rate"] - np.array(kat["rate"]))[ok].max() < 5.1e-5
    assert np.abs(r["subst"] - np.array(kat["subst"]))[ok].max() < 5.1e-5
    # the three columns HyPhy reports with junk rates (494.2, 66.2, 478.3) are the saturating ones
    sat = np.flatnonzero(inf & (r["flag"] == 2)) + 1
    assert sat.tolist() == [25, 174, 179]
    # single-taxon columns: lnL = ln(pi_x) exactly, rate = kappa * start value
    assert abs(r["lnl"][0] - np.log(0.19)) < 1e-14 and r["flag"][0] == 1
    # 131 constant columns -> exactly zero
    assert (r["rate"][ok] == 0).sum() == 131


@pytest.mark.parametrize("start_mode", [1, 2])
def test_reference_start_modes_on_the_known_answer_file(oracle, golden_dir, start_mode):
    """The oracle's reference-faithful mode (start_mode 1: every column starts at siteRate = 1 as in
    models_and_rates.bf:1050, plain safeguarded Newton to 1e-12, none of the product's accelerations) and the product's
    optimiser started at 1 (start_mode 2) against the PhyDesign known-answer file: the same 180/180 log-likelihoods and
    177/177 rates as the default mode, and NO column of the bundled locus on which the three modes disagree."""
    kat = json.load(open(os.path.join(golden_dir, "chr1_918_phydesign_rates.json")))
    names, rows = oracle.read_nexus_matrix(os.path.join(golden_dir, "chr1_918.nex"))
    st = oracle.encode_rows(rows)
    depth, factor, root = oracle.correct_branch_lengths_values(open(os.path.join(golden_dir, "Euteleost.tree")).read())
    parent, blen, leaf = oracle.tree_arrays(root, names)
    pi = np.array(kat["freqs_ACGT"])
    exch = np.array([kat[k] for k in ("AC", "AG", "AT", "CG", "CT", "GT")])
    r = oracle.site_rates(st, parent, blen, leaf, pi, exch, start_mode=start_mode)
    inf = r["nres"] >= 3
    assert inf.sum() == 180
    assert np.abs(r["lnl"] - np.array(kat["ll"]))[inf].max() < 5.1e-5
    ok = inf & ((r["flag"] == 0) | (r["flag"] == 3))
    assert ok.sum() == 177
    assert np.abs(r["rate"] - np.array(kat["rate"]))[ok].max() < 5.1e-5
    d = oracle.site_rates(st, parent, blen, leaf, pi, exch)          # default: parsimony start, accelerated exits
    assert np.array_equal(d["flag"], r["flag"])
    both = (d["flag"] == 0)
    assert (np.abs(d["rate"][both] - r["rate"][both]) > 1e-6 * r["rate"][both]).sum() == 0
    assert np.abs(d["lnl"] - r["lnl"]).max() < 1e-9


def test_start_modes_agree_on_unimodal_columns_and_differ_rarely(oracle):
    """Parsimony start + accelerated exits (mode 0) against the reference-faithful mode (1) on seeded synthetic columns:
    wherever both end on the same local optimum they agree far below 1e-6; the columns on which they do not are
    multimodal ones (two local maxima more than a log-unit apart), a few per 10^5 at 16 taxa (tools/start_mode_census.py
    has the large-sample figures quoted in DESIGN.md)."""
    from tapir_amd import synth
    d = synth.simulate(40, 500, 16, 20261005)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    n = differ = 0
    for l in range(40):
        sl = slice(l * 500, (l + 1) * 500)
        a = oracle.site_rates(st[:, sl], pin["parent"], pin["blen"], pin["leaf"], d["pi"][l], d["exch"][l])
        b = oracle.site_rates(st[:, sl], pin["parent"], pin["blen"], pin["leaf"], d["pi"][l], d["exch"][l], start_mode=1)
        both = (a["flag"] == 0) & (b["flag"] == 0)
        rel = np.abs(a["rate"][both] - b["rate"][both]) / b["rate"][both]
        same = rel < 1e-2                      # the same local optimum ...
        assert rel[same].max() < 1e-6          # ... located to the parity tolerance by both
        differ += int((~same).sum()) + int((a["flag"] != b["flag"]).sum())
        n += 500
        assert b["nevals"] > a["nevals"]       # the faithful mode pays for it
    assert differ <= 5, differ                 # 20 000 columns: expect ~1


def test_optimiser_tail_on_small_trees(oracle):
    """Evaluations per column from HyPhy's start value on 5-taxon columns (the shape of the reference's bundled locus), column
    by column: a small batch lasts as long as its slowest column, so the TAIL of this distribution is a performance contract
    of the optimiser the kernel shares with the oracle.  With the Halley step at weakly curved points the slowest of these
    1525 optimised columns takes 15 evaluations and 3 take 14 or more (without it: 27 and 24; mean 4.36 instead of 4.18)."""
    from tapir_amd import synth
    d = synth.simulate(8, 500, 5, 3)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    off = d["locus_offsets"]
    ev = []
    for l in range(8):
        sub = st[:, off[l]:off[l + 1]]
        whole = oracle.site_rates(np.ascontiguousarray(sub), pin["parent"], pin["blen"], pin["leaf"], d["pi"][l], d["exch"][l])
        total = 0
        for i in range(sub.shape[1]):
            r = oracle.site_rates(np.ascontiguousarray(sub[:, i:i + 1]), pin["parent"], pin["blen"], pin["leaf"], d["pi"][l], d["exch"][l])
            total += r["nevals"]
            if r["flag"][0] in (0, 2):
                ev.append(r["nevals"])
            assert r["flag"][0] == whole["flag"][i] and r["rate"][0] == whole["rate"][i]   # a column does not depend on its neighbours
        assert total == whole["nevals"]
    ev = np.array(ev)
    assert len(ev) > 1400
    assert ev.max() <= 20 and (ev >= 14).sum() <= 10 and ev.mean() < 4.25, (ev.max(), (ev >= 14).sum(), ev.mean())


def test_informative_mask_goldens(oracle, golden_dir):
    """chr1_918-test-cutoff-values.npy (threshold 3) and the 4-column toy alignment of
    test_compute.py:99-103 (expected [nan, nan, 1, 1])."""
    names, rows = oracle.read_nexus_matrix(os.path.join(golden_dir, "chr1_918.nex"))
    mask = oracle.informative_mask_from_chars(rows, 3)
    exp = np.load(os.path.join(golden_dir, "chr1_918-test-cutoff-values.npy"))
    assert np.array_equal(np.isnan(mask), np.isnan(exp)) and np.array_equal(mask[~np.isnan(mask)], exp[~np.isnan(exp)])
    assert np.isnan(exp).sum() == 46
    counts = oracle.informative_counts(oracle.encode_rows(rows))
    assert np.array_equal(counts >= 3, ~np.isnan(exp))
    names, rows = oracle.read_nexus_matrix(os.path.join(golden_dir, "informativeness_cutoff.nex"))
    small = oracle.informative_mask_from_chars(rows, 3)
    assert np.isnan(small[:2]).all() and np.array_equal(small[2:], [1.0, 1.0])


def test_culled_rates_golden(oracle, golden, golden_dir):
    """test-culled-rates.npy = cull(mask[:100], rates / 10) (test_compute.py:111-121)."""
    exp = np.load(os.path.join(golden_dir, "test-culled-rates.npy"))
    mask = np.load(os.path.join(golden_dir, "chr1_918-test-cutoff-values.npy"))[:100]
    got = oracle.cull_uninformative_rates(golden["parsed_rates_A_div10"], mask)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    assert np.array_equal(got[~np.isnan(got)], exp[~np.isnan(exp)])
    assert np.array_equal(np.isnan(golden["A_culled_div10"]), np.isnan(exp))


def test_oracle_derivatives_by_finite_differences(oracle):
    from tapir_amd import synth
    d = synth.simulate(2, 64, 12, 3)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()[:, :64]
    u = np.array([-2.0, -0.5, 0.0, 0.7, 2.0])
    eps = 1e-5
    for c in range(0, 64, 5):
        args = (st, pin["parent"], pin["blen"], pin["leaf"], d["pi"][0], d["exch"][0], c)
        f, g, h = oracle.column_curve(*args, u)
        fp, gp, _ = oracle.column_curve(*args, u + eps)
        fm, gm, _ = oracle.column_curve(*args, u - eps)
        assert np.abs(g - (fp - fm) / (2 * eps)).max() < 1e-8 * max(1, np.abs(g).max())
        assert np.abs(h - (gp - gm) / (2 * eps)).max() < 1e-7 * max(1, np.abs(h).max())


def test_oracle_rescaling_matches_log_space_sum(oracle):
    """1000-taxon caterpillar with long branches: partials underflow fp64 without the rescale branch."""
    n = 600
    rng = np.random.default_rng(5)
    # caterpillar: leaf, leaf, internal, leaf, internal, ...
    parent, blen, leaf = [], [], []
    # nodes: 0 = leaf0, 1 = leaf1, 2 = int(0,1), 3 = leaf2, 4 = int(2,3), ...
    parent = [2, 2]
    leaf = [0, 1]
    for k in range(2, n):
        cur = len(parent)          # internal node joining previous internal/leaf pair
        parent.append(cur + 2)     # placeholder, fixed below
        leaf.append(-1)
        parent.append(cur + 2)
        leaf.append(k)
    parent.append(-1)
    leaf.append(-1)
    # fix parent pointers: internal at index i (i even >= 2) and leaf at i+1 join at i+2
    parent = np.array(parent, dtype=np.int32)
    nn = len(parent)
    parent[0] = parent[1] = 2
    for i in range(2, nn - 1, 2):
        parent[i] = i + 2
        parent[i + 1] = i + 2
    parent[nn - 1] = -1
    blen = rng.uniform(0.5, 1.5, nn)
    blen[nn - 1] = 0
    st = (1 << rng.integers(0, 4, size=(n, 3))).astype(np.uint8)
    pi = np.array([0.1, 0.2, 0.3, 0.4]); exch = np.array([1.0, 2.0, 0.5, 0.7, 3.0, 1.0])
    f, g, h = oracle.column_curve(st, parent, blen, np.array(leaf, dtype=np.int32), pi, exch, 0, np.array([1.0]))
    assert np.isfinite(f[0]) and f[0] < -700  # below log(DBL_MIN) ~ -708 only reachable with rescaling
    assert np.isfinite(g[0]) and np.isfinite(h[0])


def test_stage1_restatement_against_phydesign_header(golden_dir):
    """The only published stage-1 output the reference holds: the header of
    tapir/tests/test-hyphy/chr1_918.subsmodel.phydesign.rates (2 decimals) gives the model-averaged
    exchangeabilities PhyDesign's HyPhy run estimated for chr1_918.nex on Euteleost.tree: AC .96, AT .58, CG .36,
    CT 1.87, GT .51.  The independent restatement (oracle/stage1_oracle.py: 203 models, Akaike weights with the
    script's parameter counts) lands within 5 % of every one of them -- as close as a 226-column, 5-taxon likelihood
    surface and HyPhy's 0.001 lnL optimisation precision allow; the general model alone (AC .88, AT .51, CG .20,
    CT 1.86, GT .45) does not, so the file pins the AVERAGING, loosely."""
    import json
    from oracle import stage1_oracle
    from tapir_amd import compute, newick, nexus
    names, states = nexus.read_states(os.path.join(golden_dir, "chr1_918.nex"))
    root = newick.read_tree(os.path.join(golden_dir, "Euteleost.tree"))
    depth, factor = compute.correct_tree(root)
    parent, blen, leaf = newick.to_arrays(root, names)
    kat = json.load(open(os.path.join(golden_dir, "chr1_918_phydesign_rates.json")))
    hist = np.bincount(np.where(states == 0, 15, states & 15).ravel(), minlength=16)[None, :16]
    pi = nexus.base_frequencies_from_histogram(hist)[0]
    assert np.max(np.abs(pi - np.array(kat["freqs_ACGT"]))) < 0.007          # the header rounds to 2 decimals
    ref = stage1_oracle.model_averaged(states, parent, np.asarray(blen) / factor, leaf, pi)
    want = np.array([kat[k] for k in ("AC", "AG", "AT", "CG", "CT", "GT")])
    assert np.max(np.abs(ref["exch"] - want) / want) < 0.06, ref["exch"]
    assert np.max(np.abs(ref["grm_exch"] - want) / want) > 0.3              # it is the averaging that matches
    assert abs(sum(ref["weights"].values()) - 1.0) < 1e-12 and max(ref["weights"].values()) < 0.2
