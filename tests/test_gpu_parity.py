"""GPU parity tests (run on the MI355X box with -m gpu): HIP path through the C ABI vs the CPU oracle,
vs the reference's golden vectors, and size-independent properties at larger shapes.

Tolerances (north_star: 1e-6 relative on per-site rates and PI integrals):
  rate / lnL  GPU vs oracle   rel 1e-6 on columns whose flag is OK or ZERO (flat and saturated columns have
                              no unique maximiser: their policy values are compared exactly by flag).
                              A column whose log-likelihood is almost flat around its maximum (|d2f/du2| below
                              ~1e-8: nearly saturated, seen with <= 5 taxa) cannot be located to 1e-6 in fp64 by
                              any implementation; for those the test requires instead that the GPU's maximiser is
                              a stationary point of the ORACLE's likelihood to fp64 noise (|df/du| <= 5e-14) and
                              that the two maxima agree to 1e-12 in log L.
  PI tables   GPU vs golden   rel 1e-12 (both are fp64 evaluations of the same closed-form expressions)
  sum(error)  GPU vs golden   rel 1e-6 (QUADPACK's abserr is partly rounding noise)
  KAT         GPU vs PhyDesign file  5e-5 absolute (the file holds 4 decimals)
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL_RATE = 1e-6
RTOL_PI = 1e-12


def _engine():
    from tapir_amd import engine
    if engine.device_count() < 1:
        pytest.fail("no GPU visible: the -m gpu tests must run on the MI355X box")
    return engine


def _rel(a, b, floor=1e-300):
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def _assert_rates_match(oracle, got, ref, sl, states, pin, pi, exch, kappa):
    ok = (ref["flag"] == 0) | (ref["flag"] == 3)
    rr = np.zeros(ok.size)
    rr[ok] = _rel(got["rate"][sl][ok], ref["rate"][ok], 1e-12)
    for c in np.flatnonzero(rr >= RTOL_RATE):
        u_gpu = np.log(got["rate"][sl][c] / kappa)
        f, g, h = oracle.column_curve(states, pin["parent"], pin["blen"], pin["leaf"], pi, exch, int(c), np.array([u_gpu]))
        assert abs(g[0]) <= 5e-14 and abs(h[0]) < 1e-7 and abs(got["lnl"][sl][c] - ref["lnl"][c]) <= 1e-12, \
            (int(c), rr[c], g[0], h[0])


def _plan_for(engine, c, T, times, intervals, correction=1.0, threshold=3, round_decimals=4, integ_mode=0):
    ncols = c["states"].shape[1]
    return engine.Plan(c["states"].shape[0], c["parent"], c["blen"], c["leaf"], [0, ncols], [c["pi"]], [c["exch"]],
                       T, times, intervals, correction=correction, threshold=threshold,
                       round_decimals=round_decimals, integ_mode=integ_mode)


def test_chr1_918_site_rates_vs_phydesign_and_oracle(chr1_918, oracle):
    """The reference's only stage-2 known-answer file, through the HIP kernel."""
    engine = _engine()
    c = chr1_918
    plan = _plan_for(engine, c, 174, [10, 20, 50], [[0, 10]], correction=c["factor"])
    got = plan.site_rates(c["states"])
    ref = oracle.site_rates(c["states"], c["parent"], c["blen"], c["leaf"], c["pi"], c["exch"])
    assert np.array_equal(got["flag"], ref["flag"])
    assert np.array_equal(got["nres"], ref["nres"])
    ok = (ref["flag"] == 0) | (ref["flag"] == 3)
    assert _rel(got["rate"][ok], ref["rate"][ok], 1e-12).max() < RTOL_RATE
    assert np.abs(got["lnl"] - ref["lnl"]).max() < 1e-9
    assert _rel(got["subst"][ok], ref["subst"][ok], 1e-12).max() < RTOL_RATE
    # flat columns keep the start value: rate == kappa exactly
    assert np.array_equal(got["rate"][ref["flag"] == 1], ref["rate"][ref["flag"] == 1])
    kat = c["kat"]
    inf = ref["nres"] >= 3
    assert inf.sum() == 180
    assert np.abs(got["lnl"] - np.array(kat["ll"]))[inf].max() < 5.1e-5
    okk = inf & ok
    assert okk.sum() == 177
    assert np.abs(got["rate"] - np.array(kat["rate"]))[okk].max() < 5.1e-5
    assert np.abs(got["subst"] - np.array(kat["subst"]))[okk].max() < 5.1e-5
    assert abs(plan.chrono_length - kat["chronogram_length"]) < 1e-12
    plan.close()


def test_gtr_eigen_systems(oracle):
    engine = _engine()
    from tapir_amd import synth
    pi, exch = synth.locus_parameters(50, 7)
    root, names = synth.yule_tree(8, 3)
    pin = synth.plan_inputs(root, names)
    off = np.arange(51) * 10
    plan = engine.Plan(8, pin["parent"], pin["blen"], pin["leaf"], off, pi, exch, 10, [1], [[0, 1]])
    lam, U, Ui, kappa = plan.models()
    for l in range(50):
        Q = U[l] @ np.diag(lam[l]) @ Ui[l]
        olam, oU, oUi, okappa = oracle.gtr_eigen(pi[l], exch[l])
        Qo = oU @ np.diag(olam) @ oUi
        assert np.abs(Q - Qo).max() < 1e-13
        assert np.abs(U[l] @ Ui[l] - np.eye(4)).max() < 1e-13
        assert abs(kappa[l] - okappa) < 1e-14
        assert lam[l][0] == 0.0 and np.all(U[l][:, 0] == 1.0) and np.allclose(Ui[l][0], pi[l], rtol=0, atol=1e-15)
    plan.close()


@pytest.mark.parametrize("ntaxa,nloci,ncols,seed", [(16, 24, 500, 11), (64, 6, 700, 12), (5, 10, 333, 13), (256, 2, 200, 14)])
def test_synthetic_site_rates_vs_oracle(oracle, ntaxa, nloci, ncols, seed):
    """Same seeded bytes to the HIP kernel and to the C restatement; ragged last chunks on purpose."""
    engine = _engine()
    from tapir_amd import synth
    d = synth.simulate(nloci, ncols, ntaxa, seed)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"],
                       pin["T"], [10, 30, 50, 90], [[5, 15], [25, 35]], correction=pin["correction"])
    got = plan.site_rates(st)
    kappa = plan.models()[3]
    for l in range(nloci):
        sl = slice(l * ncols, (l + 1) * ncols)
        ref = oracle.site_rates(st[:, sl], pin["parent"], pin["blen"], pin["leaf"], d["pi"][l], d["exch"][l])
        assert np.array_equal(got["nres"][sl], ref["nres"])
        assert np.array_equal(got["flag"][sl], ref["flag"]), (l, np.flatnonzero(got["flag"][sl] != ref["flag"]))
        _assert_rates_match(oracle, got, ref, sl, st[:, sl], pin, d["pi"][l], d["exch"][l], kappa[l])
        assert np.abs(got["lnl"][sl] - ref["lnl"]).max() < 1e-10 * max(1.0, np.abs(ref["lnl"]).max())
    assert plan.last_eval_count() > 0
    plan.close()


@pytest.mark.parametrize("case", ["A", "B", "C"])
def test_pi_tables_vs_reference_outputs(golden, case):
    """tphip_pi_tables against outputs of the reference's own compute.py (tests/golden/make_golden.py)."""
    engine = _engine()
    rates = golden[case + "_rates"]
    T = int(golden[case + "_T"])
    times, iv = golden[case + "_times"], golden[case + "_intervals"]
    # any 3-taxon tree will do: the PI stage never looks at it
    parent, blen, leaf = [2, 2, 4, 4, -1], [1.0, 1.0, 1.0, 1.0, 0.0], [0, 1, -1, 2, -1]
    plan = engine.Plan(3, parent, blen, leaf, [0, len(rates)], [[.25] * 4], [[1.0] * 6], T, times, iv,
                       correction=1.0, threshold=0, round_decimals=-1)
    tab = plan.pi_tables(rates)[0]
    n_t, n_i = len(times), len(iv)
    net, disc = tab[:T], tab[T:T + n_t]
    integ, err = tab[T + n_t:T + n_t + n_i], tab[T + n_t + n_i:]
    assert _rel(net[1:], golden[case + "_net"][1:]).max() < RTOL_PI
    assert net[0] == 0.0
    assert _rel(disc, golden[case + "_disc"], 1e-300).max() < RTOL_PI
    assert _rel(integ, golden[case + "_sum_integral"]).max() < RTOL_PI
    assert _rel(err, golden[case + "_sum_error"]).max() < 1e-6
    plan.close()


def test_phydesign_known_answers(golden_dir):
    """test_compute.py:44-71 (PhyDesign web-site values) through the mirrored host functions."""
    import os
    from tapir_amd import compute
    _engine()
    rates = compute.parse_site_rates(os.path.join(golden_dir, "test-uniform-draw-weights.rates.json"), test=True)
    townsend = compute.get_townsend_pi(compute.get_time(0, 174), np.array(rates))
    assert townsend.shape == (174, 100)
    net = np.sum(townsend, axis=1)
    np.testing.assert_allclose(net[:6], [0.00000, 0.10778, 0.14484, 0.14616, 0.13132, 0.11089], atol=5e-6)
    for v, e in zip([10, 20, 50], [0.03448, 0.01111, 0.02293]):
        assert abs(net[v] - e) < 5e-6
    expected = [0.93453, 0.10628, 0.05855, 0.12638, 1.03698, 2.08840]
    for pair, e in zip(([0, 10], [10, 15], [15, 20], [20, 30], [20, 70], [20, 100]), expected):
        integral, error = compute.get_integral_over_times(pair[0], pair[1], rates)
        assert abs(sum(integral) - e) < 1e-5
    # the two goldens the reference ships but never loads
    r_out = np.load(os.path.join(golden_dir, "test-R-townsend-output.npy"))          # (100 sites, 101 times)
    mine = compute.get_townsend_pi(compute.get_time(0, 101), np.array(rates))
    assert _rel(mine[1:], r_out.T[1:]).max() < 1e-14
    g3050 = np.load(os.path.join(golden_dir, "test-30-50-integral.npy")).ravel()
    integral, _ = compute.get_integral_over_times(30, 50, rates)
    assert _rel(integral, g3050).max() < 1e-14


def test_per_site_quad_vs_reference(golden):
    engine = _engine()
    for case in "ABC":
        r = golden[case + "_rates"]
        fin = r[np.isfinite(r)]
        for k, (a, b) in enumerate(golden[case + "_intervals"]):
            integral, abserr = engine.quad_townsend(a, b, fin)
            ref_int = golden[case + "_site_integral"][k]
            # 1e-13 relative; integrals below 1e-15 (far under quad's own epsabs = 1.49e-8) only absolutely
            assert np.all(np.abs(integral - ref_int) <= 1e-13 * np.abs(ref_int) + 1e-28)
            # abserr: the deterministic floor 50*eps*resabs and the real adaptive errors agree closely; a few
            # values are rounding noise of (resk - resg) and may differ by several percent
            ref_err = golden[case + "_site_abserr"][k]
            rel = _rel(abserr, ref_err, 1e-300)
            big = ref_err > 1e-12  # real (adaptive) error estimates, far above the 50*eps*resabs floor
            assert np.all(rel[big] < 1e-3) and np.median(rel) < 1e-9 and rel.max() < 0.5


def test_run_fused_matches_staged_and_oracle(chr1_918, oracle):
    """worker() end to end for the bundled locus: rates rounded to 4 dp (Format(x,0,4)), / correction,
    threshold-3 cull, PI tables; versus the numpy/scipy restatement fed with the oracle's rates."""
    engine = _engine()
    c = chr1_918
    times, iv = [10, 20, 50], [[0, 10], [10, 15], [20, 100]]
    plan = _plan_for(engine, c, int(c["depth"]), times, iv, correction=c["factor"], threshold=3, round_decimals=4)
    out = plan.run_fused(c["states"])
    ref = oracle.site_rates(c["states"], c["parent"], c["blen"], c["leaf"], c["pi"], c["exch"])
    rates = oracle.round_dp(ref["rate"], 4) / c["factor"]
    rates[ref["nres"] < 3] = np.nan
    pi_net, pi_times, pi_epochs = oracle.worker_tables(rates, int(c["depth"]), times, iv)
    T = int(c["depth"])
    tab = out["tables"][0]
    assert _rel(tab[1:T], pi_net[1:], 1e-300).max() < 1e-9
    assert _rel(tab[T:T + 3], np.array([pi_times[t] for t in times])).max() < 1e-9
    assert _rel(tab[T + 3:T + 6], np.array([pi_epochs["%d-%d" % (a, b)]["sum(integral)"] for a, b in iv])).max() < 1e-9
    staged = plan.pi_tables(out["rate"], out["nres"])
    assert np.array_equal(staged, out["tables"])
    plan.close()


def test_cli_end_to_end_on_gpu(golden_dir, tmp_path, oracle):
    """BASELINE config C1 through the real engine: tapir_compute.py on the bundled locus + tree; checks the
    output directory contents, the JSON schema and values, and the byte-identical sqlite schema."""
    _engine()
    from test_host_logic import _run_cli, check_cli_outputs
    outdir, _ = _run_cli(golden_dir, tmp_path)
    check_cli_outputs(outdir, oracle, golden_dir)


def test_state_histogram_and_dense_pi_edge_cases():
    engine = _engine()
    from tapir_amd import synth
    d = synth.simulate(5, 300, 7, 21)
    st = d["states"].numpy()
    off = np.array([0, 300, 300, 900, 1200, 1500])  # an empty locus in the middle
    hist = engine.state_histogram(st, off)
    for l in range(5):
        blk = st[:, off[l]:off[l + 1]].ravel()
        assert np.array_equal(hist[l], np.bincount(blk, minlength=16)[:16])
    # NaN and zero rates, scalar time, time 0
    out = engine.townsend_pi_dense([0.0, 1.0, 2.5], [0.0, np.nan, 0.3])
    assert out[0][0] == 0.0 and np.isnan(out[1][1]) and np.isnan(out[0][1])
    assert abs(out[2][2] - 16 * 0.09 * 2.5 * np.exp(-4 * 0.3 * 2.5)) < 1e-15
    # empty inputs
    assert engine.townsend_pi_dense([], [0.1]).shape == (0, 1)
    i, e = engine.quad_townsend(0, 10, [])
    assert i.size == 0 and e.size == 0


def test_plan_rejects_bad_arguments(chr1_918):
    engine = _engine()
    c = chr1_918
    args = (5, c["parent"], c["blen"], c["leaf"], [0, 226], [c["pi"]], [c["exch"]], 174)
    with pytest.raises(engine.TphipError, match="outside 0..T-1"):
        engine.Plan(*args, [174], [[0, 10]])
    with pytest.raises(engine.TphipError, match="Start time"):
        engine.Plan(*args, [10], [[10, 10]])
    with pytest.raises(engine.TphipError, match="post-order"):
        engine.Plan(5, c["parent"][::-1].copy(), c["blen"], c["leaf"], [0, 226], [c["pi"]], [c["exch"]], 174, [1], [[0, 1]])
    with pytest.raises(engine.TphipError, match="fewer than two bases"):
        engine.Plan(5, c["parent"], c["blen"], c["leaf"], [0, 226], [[1.0, 0.0, 0.0, 0.0]], [c["exch"]], 174, [1], [[0, 1]])
    with pytest.raises(engine.TphipError, match="frequencies of locus 0"):
        engine.Plan(5, c["parent"], c["blen"], c["leaf"], [0, 226], [[0.5, 0.6, -0.1, 0.0]], [c["exch"]], 174, [1], [[0, 1]])


def test_locus_with_an_absent_base(oracle):
    """A short gap-free locus in which one base never occurs has empirical frequency 0 for it (HarvestFrequencies,
    bf:968); HyPhy fits such a locus, so the plan must accept it (it used to abort the whole batch).  The engine floors
    the zero at 1e-12 of the total (tphip.hip: kPiFloor); checked against the oracle on the same floored frequencies,
    and the floor itself is shown not to matter: the oracle's log L moves by < 5e-8 between floors of 1e-12 and 1e-9."""
    engine = _engine()
    from tapir_amd import nexus, synth
    d = synth.simulate(2, 400, 12, 77, gap_frac=0.0)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy().copy()
    a = st[:, :400]
    a[a == 8] = 1                                        # locus 0: every T becomes an A -> pi_T = 0
    hist = engine.state_histogram(st, d["locus_offsets"])
    pi = nexus.base_frequencies_from_histogram(hist)
    assert pi[0, 3] == 0.0 and (pi[1] > 0).all()
    plan = engine.Plan(12, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], pi, d["exch"], pin["T"], [10], [[5, 15]],
                       correction=pin["correction"])
    got = plan.run_fused(st)
    assert np.isfinite(got["tables"]).all() and np.isfinite(got["lnl"]).all()
    kappa = plan.models()[3]

    def floored(p, eps):
        q = np.maximum(p, eps * p.sum())
        return q / q.sum()

    for l in range(2):
        sl = slice(l * 400, (l + 1) * 400)
        ref = oracle.site_rates(st[:, sl], pin["parent"], pin["blen"], pin["leaf"], floored(pi[l], 1e-12), d["exch"][l])
        assert np.array_equal(got["flag"][sl], ref["flag"])
        assert np.abs(got["lnl"][sl] - ref["lnl"]).max() < 1e-9
        _assert_rates_match(oracle, got, ref, sl, st[:, sl], pin, floored(pi[l], 1e-12), d["exch"][l], kappa[l])
    coarse = oracle.site_rates(st[:, :400], pin["parent"], pin["blen"], pin["leaf"], floored(pi[0], 1e-9), d["exch"][0])
    fine = oracle.site_rates(st[:, :400], pin["parent"], pin["blen"], pin["leaf"], floored(pi[0], 1e-12), d["exch"][0])
    # (linear in the floor: ~1e-9 x the number of changes on the column, i.e. ~1e-11 at the floor the engine uses)
    assert np.abs(coarse["lnl"] - fine["lnl"]).max() < 5e-8
    plan.close()


def test_four_decimal_round_trip_matches_printf_on_ties(chr1_918):
    """The rate the PI stage uses is what tapir reads back from HyPhy's Format(x,0,4) text (bf:1093-1095): printf-style
    rounding of the EXACT binary value.  Constructed near-ties: doubles next to (k + 0.5) * 1e-4, whose product with
    1e4 rounds onto the half-integer although the true product is above or below it -- rint(r * 1e4) / 1e4 (round 1)
    gets about half of them wrong by 1e-4.  Device (tphip_corrected_rates) and host helper against "%.4f"."""
    engine = _engine()
    from tapir_amd import compute
    c = chr1_918
    rng = np.random.default_rng(5)
    k = rng.integers(0, 10 ** 7, 20000)
    v = (k + 0.5) / 1e4
    v = np.concatenate([v, np.nextafter(v, np.inf), np.nextafter(v, -np.inf), rng.gamma(0.5, 0.02, 20000),
                        [0.03125, 0.09375, 0.00005, 1.00015, 0.0, 2.5e-5, 9999.99995, 1e4]])
    want = np.array([float("%.4f" % x) for x in v])
    naive = np.round(v * 1e4) / 1e4
    assert (naive != want).sum() > 1000                  # the cases are real
    assert np.array_equal(compute.round_like_hyphy(v, 4), want)
    n = v.size
    plan = engine.Plan(5, c["parent"], c["blen"], c["leaf"], [0, n], [c["pi"]], [c["exch"]], 174, [10], [[0, 10]],
                       correction=100.0, threshold=3, round_decimals=4)
    got = plan.corrected_rates(v)
    assert np.array_equal(got, want / 100.0)
    nres = np.where(np.arange(n) % 3 == 0, 2, 3).astype(np.int32)
    got = plan.corrected_rates(v, nres)
    assert np.isnan(got[nres < 3]).all() and np.array_equal(got[nres >= 3], (want / 100.0)[nres >= 3])
    plan.close()
    raw = engine.Plan(5, c["parent"], c["blen"], c["leaf"], [0, n], [c["pi"]], [c["exch"]], 174, [10], [[0, 10]],
                      correction=100.0, round_decimals=-1)
    assert np.array_equal(raw.corrected_rates(v), v / 100.0)
    raw.close()


def test_reference_start_rule(chr1_918, oracle):
    """tphip_plan_desc.start_rule = TPHIP_START_REFERENCE: every column starts at siteRate = 1 (models_and_rates.bf:1050).
      * against the oracle running the same optimiser from the same start (start_mode 2): flags exact, rates 1e-6;
      * the bundled locus: default start, reference start and the oracle's independent plain-Newton-from-1 restatement
        (start_mode 1) agree on EVERY column -- the count of differing columns is asserted to be 0;
      * synthetic samples: the product's DEFAULT (parsimony start) against the reference-faithful restatement: columns
        that end on a different local optimum are counted (they are multimodal columns; DESIGN.md section 5 quotes the
        large-sample frequencies) and bounded."""
    engine = _engine()
    from tapir_amd import synth
    c = chr1_918
    n = c["states"].shape[1]
    mk = lambda rule: engine.Plan(5, c["parent"], c["blen"], c["leaf"], [0, n], [c["pi"]], [c["exch"]], 174, [10], [[0, 10]],  # noqa: E731
                                  correction=c["factor"], start_rule=rule)
    p0, p1 = mk(engine.START_PARSIMONY), mk(engine.START_REFERENCE)
    g0, g1 = p0.site_rates(c["states"]), p1.site_rates(c["states"])
    p0.close()
    p1.close()
    plain = oracle.site_rates(c["states"], c["parent"], c["blen"], c["leaf"], c["pi"], c["exch"], start_mode=1)
    same = oracle.site_rates(c["states"], c["parent"], c["blen"], c["leaf"], c["pi"], c["exch"], start_mode=2)
    assert np.array_equal(g1["flag"], same["flag"]) and np.array_equal(g0["flag"], plain["flag"]) and np.array_equal(g1["flag"], plain["flag"])
    ok = plain["flag"] == 0
    for g in (g0, g1):
        assert (np.abs(g["rate"][ok] - plain["rate"][ok]) > 1e-6 * plain["rate"][ok]).sum() == 0
        assert np.abs(g["lnl"] - plain["lnl"]).max() < 1e-9
    report = []
    for name, nloci, ncols, ntaxa, seed, kw, bound in [("C2 shape", 100, 500, 16, 20261005, {}, 3e-4),
                                                       ("C3 shape", 2, 10000, 64, 20261006, {}, 1e-4),
                                                       ("5 noisy taxa", 40, 1000, 5, 4242, dict(rate_mean=0.02, gap_frac=0.15), 4e-3)]:
        d = synth.simulate(nloci, ncols, ntaxa, seed, **kw)
        pin = synth.plan_inputs(d["root"], d["names"])
        st = d["states"].numpy()
        args = (ntaxa, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"], pin["T"], [10], [[5, 15]])
        pa, pb = engine.Plan(*args, correction=pin["correction"]), engine.Plan(*args, correction=pin["correction"], start_rule=1)
        ga, gb = pa.site_rates(st), pb.site_rates(st)
        kappa = pb.models()[3]
        pa.close()
        pb.close()
        differ = 0
        for l in range(nloci):
            sl = slice(l * ncols, (l + 1) * ncols)
            o2 = oracle.site_rates(st[:, sl], pin["parent"], pin["blen"], pin["leaf"], d["pi"][l], d["exch"][l], start_mode=2)
            assert np.array_equal(gb["flag"][sl], o2["flag"])
            _assert_rates_match(oracle, gb, o2, sl, st[:, sl], pin, d["pi"][l], d["exch"][l], kappa[l])
            o1 = oracle.site_rates(st[:, sl], pin["parent"], pin["blen"], pin["leaf"], d["pi"][l], d["exch"][l], start_mode=1)
            both = (ga["flag"][sl] == 0) & (o1["flag"] == 0)
            rel = np.abs(ga["rate"][sl][both] - o1["rate"][both]) / o1["rate"][both]
            differ += int((rel > 1e-2).sum()) + int((ga["flag"][sl] != o1["flag"]).sum())
            # the same local optimum is located to the parity tolerance (nearly flat columns excepted: |f''| < 1e-7,
            # they are compared through log L)
            near = rel <= 1e-2
            loose = rel[near] >= 1e-6
            assert np.abs(ga["lnl"][sl][both][near][loose] - o1["lnl"][both][near][loose]).max(initial=0.0) < 1e-11
        report.append((name, nloci * ncols, differ))
        assert differ <= bound * nloci * ncols, (name, differ)
    print("columns on another local optimum than the start-at-1 restatement (default start):", report)


def test_stage2_pattern_dedup(chr1_918, monkeypatch):
    """HyPhy fits one rate per unique column pattern of a locus (models_and_rates.bf:1033-1044).  The engine does the
    same when a locus repeats columns (tphip_plan_desc.pattern_dedup; pattern_kernels.hpp): outputs must be bit-identical
    with de-duplication off, forced on and automatic, on
      * bootstrap resamples of the reference's bundled locus (2000 columns drawn from its 226: at most 57 patterns),
      * resamples of 64- and 130-taxon synthetic loci mixed with loci that repeat nothing (ragged lengths, an empty locus);
    the optimiser must run on unique patterns only (evaluation counts), the automatic mode must switch on for the
    resampled loci and stay off for the synthetic batch with random gaps, and the dominant kernel must get faster."""
    engine = _engine()
    import torch
    from tapir_amd import synth
    monkeypatch.delenv("TPHIP_DEDUP", raising=False)
    rng = np.random.default_rng(8)
    c = chr1_918
    L, S = 12, 2000
    st = np.concatenate([c["states"][:, rng.integers(0, 226, S)] for _ in range(L)], axis=1)
    off = np.arange(L + 1) * S
    runs = {}
    for mode in (engine.DEDUP_OFF, engine.DEDUP_ON, engine.DEDUP_AUTO):
        plan = engine.Plan(5, c["parent"], c["blen"], c["leaf"], off, np.tile(c["pi"], (L, 1)), np.tile(c["exch"], (L, 1)), 174,
                           [10, 20], [[0, 10], [20, 100]], correction=c["factor"], pattern_dedup=mode)
        runs[mode] = (plan.run_fused(st), plan.last_eval_count())
        plan.close()
    base, ev_off = runs[engine.DEDUP_OFF]
    for mode in (engine.DEDUP_ON, engine.DEDUP_AUTO):
        got, ev = runs[mode]
        for k in base:
            assert np.array_equal(got[k], base[k], equal_nan=True), (mode, k)
    assert runs[engine.DEDUP_ON][1] < 0.2 * ev_off            # <= 57 patterns (fewer need the optimiser) of 2000 columns
    assert runs[engine.DEDUP_AUTO][1] == ev_off               # a batch this small is left alone in automatic mode
    # duplicates of one pattern carry one answer
    first = {}
    for col in range(S):
        key = st[:, col].tobytes()
        if key in first:
            assert base["rate"][col] == base["rate"][first[key]] and base["lnl"][col] == base["lnl"][first[key]]
        else:
            first[key] = col
    # larger trees: resampled loci next to loci that repeat nothing, ragged offsets, an empty locus
    for ntaxa, seed in ((64, 41), (130, 42)):
        d = synth.simulate(6, 1200, ntaxa, seed)
        pin = synth.plan_inputs(d["root"], d["names"])
        s0 = d["states"].numpy().copy()
        for l in (1, 4):   # loci 1 and 4 become bootstrap resamples of their own first 150 columns
            s0[:, l * 1200:(l + 1) * 1200] = s0[:, l * 1200 + rng.integers(0, 150, 1200)]
        off2 = np.array([0, 1200, 2400, 2400, 4800, 6000, 7200])   # locus 2 is empty, locus 3 twice as long
        pi6 = d["pi"]
        outs = {}
        for mode in (engine.DEDUP_OFF, engine.DEDUP_ON, engine.DEDUP_AUTO):
            plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off2, pi6, d["exch"], pin["T"], [10], [[5, 15]],
                               correction=pin["correction"], pattern_dedup=mode)
            outs[mode] = (plan.run_fused(s0), plan.last_eval_count())
            plan.close()
        for mode in (engine.DEDUP_ON, engine.DEDUP_AUTO):
            for k in outs[engine.DEDUP_OFF][0]:
                assert np.array_equal(outs[mode][0][k], outs[engine.DEDUP_OFF][0][k], equal_nan=True), (ntaxa, mode, k)
        assert outs[engine.DEDUP_ON][1] < 0.8 * outs[engine.DEDUP_OFF][1]     # 2400 of 7200 columns collapse to <= 300
        assert outs[engine.DEDUP_AUTO][1] == outs[engine.DEDUP_OFF][1]   # small batch: automatic mode stays off
    # per-locus decision of the automatic mode (batches of >= 2^20 columns): 24 loci x 50 000 columns, every other locus a
    # resample of its own first 500 columns -- those are de-duplicated, the loci with random gaps are not
    d = synth.simulate(24, 500, 64, 45)
    pin = synth.plan_inputs(d["root"], d["names"])
    src = d["states"].numpy()
    fresh = synth.simulate(24, 50000, 64, 46, device="cuda", tree=(d["root"], d["names"]))["states"].cpu().numpy()
    mix = np.concatenate([src[:, l * 500 + rng.integers(0, 500, 50000)] if l % 2 else fresh[:, l * 50000:(l + 1) * 50000]
                          for l in range(24)], axis=1)
    mix = np.ascontiguousarray(mix)
    res = {}
    for mode in (engine.DEDUP_OFF, engine.DEDUP_ON, engine.DEDUP_AUTO):
        plan = engine.Plan(64, pin["parent"], pin["blen"], pin["leaf"], np.arange(25) * 50000, d["pi"], d["exch"], pin["T"], [10],
                           [[5, 15]], correction=pin["correction"], pattern_dedup=mode)
        res[mode] = (plan.run_fused(mix), plan.last_eval_count())
        plan.close()
    for mode in (engine.DEDUP_ON, engine.DEDUP_AUTO):
        for k in res[engine.DEDUP_OFF][0]:
            assert np.array_equal(res[mode][0][k], res[engine.DEDUP_OFF][0][k], equal_nan=True), (mode, k)
    assert res[engine.DEDUP_AUTO][1] < 0.6 * res[engine.DEDUP_OFF][1]          # the resampled half collapses
    assert res[engine.DEDUP_ON][1] <= res[engine.DEDUP_AUTO][1] <= 1.05 * res[engine.DEDUP_ON][1]
    # synthetic batch without repeats: automatic mode leaves it alone (same evaluations as off)
    d = synth.simulate(4, 3000, 64, 43)
    pin = synth.plan_inputs(d["root"], d["names"])
    ev = {}
    for mode in (engine.DEDUP_OFF, engine.DEDUP_AUTO):
        plan = engine.Plan(64, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"], pin["T"], [10],
                           [[5, 15]], correction=pin["correction"], pattern_dedup=mode)
        plan.run_fused(d["states"].numpy())
        ev[mode] = plan.last_eval_count()
        plan.close()
    assert ev[engine.DEDUP_AUTO] > 0.97 * ev[engine.DEDUP_OFF]
    # kernel time: 40 loci x 50 000 columns resampled from 1000 columns each, 64 taxa
    d = synth.simulate(40, 1000, 64, 44)
    pin = synth.plan_inputs(d["root"], d["names"])
    src = d["states"].numpy()
    big = np.concatenate([src[:, l * 1000 + rng.integers(0, 1000, 50000)] for l in range(40)], axis=1)
    t_big = torch.from_numpy(np.ascontiguousarray(big)).cuda()   # (fancy indexing + concatenate give a Fortran-ordered array)
    ms = {}
    for mode in (engine.DEDUP_OFF, engine.DEDUP_AUTO):
        plan = engine.Plan(64, pin["parent"], pin["blen"], pin["leaf"], np.arange(41) * 50000, d["pi"], d["exch"], pin["T"], [10],
                           [[5, 15]], correction=pin["correction"], pattern_dedup=mode)
        n = plan.ncols
        o = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(3)]
        fl, nr = torch.empty(n, dtype=torch.uint8, device="cuda"), torch.empty(n, dtype=torch.int32, device="cuda")
        tb = torch.empty((40, plan.width), dtype=torch.float64, device="cuda")
        ws = torch.empty(plan.workspace_bytes, dtype=torch.uint8, device="cuda")
        plan.profile_enable(True)
        for _ in range(3):
            plan.run_dev(t_big, o[0], o[1], o[2], fl, nr, tb, ws, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        site_ms, _, launches = plan.profile_read()
        ms[mode] = (site_ms / launches, o[0].clone(), tb.clone(), plan.last_eval_count(),
                    torch.bincount(fl.to(torch.int64), minlength=5).cpu().numpy().tolist())
        plan.close()
    assert torch.equal(ms[engine.DEDUP_OFF][1], ms[engine.DEDUP_AUTO][1]) and torch.equal(ms[engine.DEDUP_OFF][2], ms[engine.DEDUP_AUTO][2])
    print("site_rate_kernel, 2 000 000 resampled columns: off %.3f ms (%d evaluations), auto %.3f ms (%d evaluations); flags %s"
          % (ms[engine.DEDUP_OFF][0], ms[engine.DEDUP_OFF][3], ms[engine.DEDUP_AUTO][0], ms[engine.DEDUP_AUTO][3], ms[engine.DEDUP_OFF][4]))
    assert ms[engine.DEDUP_AUTO][3] < 0.05 * ms[engine.DEDUP_OFF][3]
    assert ms[engine.DEDUP_AUTO][0] < 0.5 * ms[engine.DEDUP_OFF][0]


def test_engine_before_torch_in_one_process():
    """Round 1 saw torch report "No HIP GPUs are available" when it initialised after libtphip in the same process: the
    library had bound /opt/rocm's HIP runtime, torch then mapped its bundled copy, and the second ROCr in a process
    cannot acquire the GPU.  engine.load() now loads torch's copy first (engine._preload_torch_hip_runtime); this runs
    the failing order in a fresh process."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tapir_amd import engine\n"
            "assert engine.device_count() >= 1\n"
            "import numpy as np\n"
            "pi = engine.townsend_pi_dense([1.0, 2.0], [0.1, 0.2])\n"
            "assert np.isfinite(pi).all()\n"
            "import torch\n"
            "torch.cuda.init()\n"
            "x = torch.ones(4, device='cuda') * 2\n"
            "assert float(x.sum()) == 8.0\n"
            "maps = open('/proc/self/maps').read()\n"
            "libs = {l.split()[-1] for l in maps.splitlines() if 'libamdhip64' in l}\n"
            "assert len(libs) == 1, libs\n"
            "print('ok')\n") % __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout, r.stderr[-3000:])


def test_all_gap_and_ragged_loci(oracle):
    """Edge cases the domain has: empty locus, all-gap columns, single-column locus, a 2-taxon tree."""
    engine = _engine()
    parent, blen, leaf = [2, 2, -1], [0.3, 0.7, 0.0], [0, 1, -1]
    st = np.array([[1, 15, 2, 4, 8, 1, 15], [1, 15, 15, 4, 1, 2, 8]], dtype=np.uint8)
    off = [0, 0, 1, 7]
    pi = np.array([[.1, .2, .3, .4]] * 3)
    ex = np.array([[1, 2, .5, .7, 3, 1.]] * 3)
    plan = engine.Plan(2, parent, blen, leaf, off, pi, ex, 5, [1], [[0, 2]], threshold=2, round_decimals=-1)
    got = plan.run_fused(st)
    ref = oracle.site_rates(st, np.array(parent, np.int32), np.array(blen), np.array(leaf, np.int32), pi[0], ex[0])
    assert np.array_equal(got["flag"], ref["flag"]) and np.array_equal(got["nres"], ref["nres"])
    ok = (ref["flag"] == 0) | (ref["flag"] == 3)
    assert np.allclose(got["rate"][ok], ref["rate"][ok], rtol=1e-6, atol=0)
    assert np.allclose(got["lnl"], ref["lnl"], rtol=0, atol=1e-12)
    assert got["lnl"][1] == 0.0 and got["flag"][1] == 1          # all-gap column: L = 1
    assert np.all(got["tables"][0] == 0.0)                       # empty locus -> zero row
    plan.close()


@pytest.mark.parametrize("persistent,byte_path,mixed", [("0", "0", "0"), ("0", "0", "1"), ("1", "0", "0"), ("0", "1", "0"), ("1", "1", "0")])
def test_site_rate_kernel_variants(oracle, monkeypatch, persistent, byte_path, mixed):
    """Every scheduling mode (persistent equal shares / locus-aligned slices / waves carrying several loci) and both
    tip-state paths (register-resident packed words / one-op-ahead byte loads) against the oracle on the same bytes."""
    engine = _engine()
    from tapir_amd import synth
    monkeypatch.setenv("TPHIP_SITE_PERSISTENT", persistent)
    monkeypatch.setenv("TPHIP_FORCE_BYTE_PATH", byte_path)
    monkeypatch.setenv("TPHIP_SITE_MIXED", mixed)
    for ntaxa, nloci, ncols, seed in [(20, 9, 777, 31), (130, 3, 300, 32)]:
        d = synth.simulate(nloci, ncols, ntaxa, seed)
        pin = synth.plan_inputs(d["root"], d["names"])
        st = d["states"].numpy()
        off = d["locus_offsets"].copy()
        off[1] = off[0]  # an empty first locus, and loci of different lengths
        off[2] = off[3] - 5
        plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off, d["pi"], d["exch"], pin["T"], [10], [[5, 15]],
                           correction=pin["correction"])
        got = plan.site_rates(st)
        kappa = plan.models()[3]
        for l in range(nloci):
            sl = slice(int(off[l]), int(off[l + 1]))
            if sl.stop == sl.start:
                continue
            ref = oracle.site_rates(st[:, sl], pin["parent"], pin["blen"], pin["leaf"], d["pi"][l], d["exch"][l])
            assert np.array_equal(got["flag"][sl], ref["flag"])
            _assert_rates_match(oracle, got, ref, sl, st[:, sl], pin, d["pi"][l], d["exch"][l], kappa[l])
            assert np.abs(got["lnl"][sl] - ref["lnl"]).max() < 1e-10 * max(1.0, np.abs(ref["lnl"]).max())
        plan.close()


@pytest.mark.parametrize("ntaxa", [12, 40])
def test_mixed_loci_mode_is_bit_identical(monkeypatch, ntaxa):
    """Small batches of short loci run with waves that carry columns of several loci at once, each lane holding its own
    locus' model.  A column's arithmetic does not depend on the lane or wave that carries it: every output
    equals the slice mode's and the persistent mode's bit for bit -- ragged loci from 0 to 200 columns, shares that span
    more loci than one group holds (few waves), one- and four-word trees; default mode = mixed on this shape."""
    engine = _engine()
    from tapir_amd import synth
    rng = np.random.default_rng(77 + ntaxa)
    nloci = 300
    d = synth.simulate(nloci, 200, ntaxa, 900 + ntaxa)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    lens = rng.integers(0, 201, size=nloci)
    tiny = rng.random(nloci) < 0.3
    lens[tiny] = rng.integers(0, 4, size=int(tiny.sum()))        # runs of loci with 0-3 columns
    off = np.zeros(nloci + 1, dtype=np.int64)
    off[1:] = np.cumsum(lens)
    st = np.ascontiguousarray(st[:, : int(off[-1])])
    runs = {}
    for name, env in [("default", {}), ("mixed", dict(TPHIP_SITE_MIXED="1")), ("mixed_few_waves", dict(TPHIP_SITE_MIXED="1", TPHIP_SITE_WAVES="5")),
                      ("slices", dict(TPHIP_SITE_MIXED="0", TPHIP_SITE_PERSISTENT="0")), ("persistent", dict(TPHIP_SITE_MIXED="0", TPHIP_SITE_PERSISTENT="1"))]:
        for k in ("TPHIP_SITE_MIXED", "TPHIP_SITE_WAVES", "TPHIP_SITE_PERSISTENT"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off, d["pi"], d["exch"], pin["T"], [10], [[5, 15]],
                           correction=pin["correction"])
        runs[name] = plan.run_fused(st)
        plan.close()
    base = runs["slices"]
    assert (base["flag"] == 0).sum() > off[-1] // 4
    for name, got in runs.items():
        for key in ("rate", "subst", "lnl", "flag", "nres", "tables"):
            assert np.array_equal(got[key], base[key]), (name, key)


@pytest.mark.parametrize("ntaxa,ncols,kw", [(9, 30000, dict(rate_mean=0.05, gap_frac=0.3)), (64, 20000, dict(rate_mean=0.3)),
                                            (16, 30000, {}), (5, 30000, dict(rate_mean=0.02, gap_frac=0.15))])
def test_optimiser_exits_leave_no_residual_on_hard_shapes(ntaxa, ncols, kw):
    """The two exits that save the confirming evaluation (two-point quartic, third-order corrected Newton step) are sized by the
    residual tails of the BASELINE configs (tests/test_gpu_fullsize.py); this is the same bound, |f'/f''| < 1e-6 at every
    reported interior maximiser, on the shapes where low-order models of f' are least at home: few taxa with many gaps (nearly
    flat columns), fast sites near saturation, 16 taxa, 5 noisy taxa."""
    engine = _engine()
    from tapir_amd import synth
    d = synth.simulate(1, ncols, ntaxa, 20261007 + ntaxa, **kw)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"], pin["T"], [10], [[5, 15]],
                       correction=pin["correction"])
    got = plan.site_rates(st)
    ok = got["flag"] == 0
    kappa = plan.models()[3][0]
    u = np.zeros(ncols)
    u[ok] = np.log(got["rate"][ok] / kappa)
    f, g, h = plan.eval_columns(st, u)
    plan.close()
    assert ok.sum() > ncols // 4 and np.all(h[ok] < 0)
    resid = np.abs(g[ok] / h[ok])
    assert resid.max() < 1e-6, (resid.max(), np.abs(h[ok])[np.argmax(resid)])
    assert np.abs(f[ok] - got["lnl"][ok]).max() < 1e-9 * np.abs(got["lnl"][ok]).max()


def test_fused_cherries_follow_the_branch_lengths(oracle):
    """The packed path folds TIP_SET + TIP_MUL on EQUALLY long branches into one CHERRY op (shared exponentials).
    A chronogram's cherries all qualify; lengthening one tip of a cherry must take exactly that pair out of the
    fused stream and leave the results equal to the oracle's either way."""
    engine = _engine()
    from tapir_amd import synth
    d = synth.simulate(2, 600, 64, synth.WORKLOAD_SEED["C3"], tree=synth.yule_tree(64, synth.WORKLOAD_SEED["C3"]))
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    parent, blen, leaf = np.asarray(pin["parent"]), np.asarray(pin["blen"], dtype=np.float64).copy(), np.asarray(pin["leaf"])
    # cherries of the tree: internal nodes whose children are two leaves
    kids = {}
    for n, p in enumerate(parent):
        if p >= 0:
            kids.setdefault(int(p), []).append(n)
    cherries = [k for k in kids.values() if len(k) == 2 and all(leaf[c] >= 0 for c in k)]
    assert len(cherries) == 21
    counts = []
    for variant in range(2):
        if variant == 1:
            blen[cherries[0][0]] *= 1.25   # no longer equal to its sibling
        plan = engine.Plan(64, parent, blen, leaf, d["locus_offsets"], d["pi"], d["exch"], pin["T"], [10], [[5, 15]],
                           correction=pin["correction"])
        counts.append(plan.op_counts["cherry"])
        got = plan.site_rates(st)
        kappa = plan.models()[3]
        off = d["locus_offsets"]
        for l in range(2):
            sl = slice(int(off[l]), int(off[l + 1]))
            ref = oracle.site_rates(st[:, sl], parent, blen, leaf, d["pi"][l], d["exch"][l])
            assert np.array_equal(got["flag"][sl], ref["flag"])
            _assert_rates_match(oracle, got, ref, sl, st[:, sl], dict(pin, blen=blen), d["pi"][l], d["exch"][l], kappa[l])
        plan.close()
    assert counts == [21, 20]


def test_results_do_not_depend_on_how_the_work_list_is_shared_out(monkeypatch):
    """A column's result depends on the column alone: whatever the persistent grid's share boundaries (resident-wave
    count, grid multiplier, first-round fraction) or the small-batch slice length, every output must be bit-identical --
    which also proves that the shares cover each optimiser column exactly once."""
    engine = _engine()
    from tapir_amd import synth
    d = synth.simulate(7, 3001, 24, 77)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    off = d["locus_offsets"].copy()
    off[3] = off[2]          # an empty locus in the middle
    settings = [dict(),
                dict(TPHIP_SITE_PERSISTENT="1", TPHIP_SITE_WAVES="37"),
                dict(TPHIP_SITE_PERSISTENT="1", TPHIP_SITE_WAVES="16", TPHIP_SITE_GRID_MULT="3", TPHIP_SITE_FIRST_FRACTION="0.8"),
                dict(TPHIP_SITE_PERSISTENT="1", TPHIP_SITE_WAVES="5", TPHIP_SITE_GRID_MULT="7", TPHIP_SITE_FIRST_FRACTION="0.33"),
                dict(TPHIP_SITE_PERSISTENT="1", TPHIP_SITE_WAVES="2048", TPHIP_SITE_GRID_MULT="2", TPHIP_SITE_FIRST_FRACTION="0.999"),
                dict(TPHIP_SITE_PERSISTENT="0", TPHIP_SITE_CHUNK="64"),
                dict(TPHIP_SITE_PERSISTENT="0", TPHIP_SITE_CHUNK="1000")]
    ref = None
    for env in settings:
        for k in ("TPHIP_SITE_PERSISTENT", "TPHIP_SITE_WAVES", "TPHIP_SITE_GRID_MULT", "TPHIP_SITE_FIRST_FRACTION", "TPHIP_SITE_CHUNK"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        plan = engine.Plan(24, pin["parent"], pin["blen"], pin["leaf"], off, d["pi"], d["exch"], pin["T"], [10], [[5, 15]],
                           correction=pin["correction"])
        got = plan.site_rates(st)
        plan.close()
        if ref is None:
            ref = got
            assert (got["flag"] == 0).sum() > 5000
        else:
            for key in ("rate", "subst", "lnl", "flag", "nres"):
                assert np.array_equal(got[key], ref[key], equal_nan=(got[key].dtype.kind == "f")), (env, key)


def test_hyphy_protocol_shim(golden_dir, tmp_path, oracle):
    """Boundary #1: `--hyphy /path/to/tphip_hyphy` -- argv = [exe, template], three stdin lines, JSON
    file out, stdout must not start with "Error"; the file must parse the way tapir/compute.py:24-44 parses it."""
    import json
    import os
    import subprocess
    import sys
    _engine()
    from tapir_amd import compute, newick
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    depth, factor, tree = compute.correct_branch_lengths(os.path.join(golden_dir, "Euteleost.tree"), "newick", d=str(tmp_path))
    out = str(tmp_path / "chr1_918.nex.rates")
    towrite = "\n".join([os.path.join(golden_dir, "chr1_918.nex"), tree, out])
    env = dict(os.environ, TPHIP_EXCHANGEABILITIES="0.96,1,0.58,0.36,1.87,0.51")
    p = subprocess.run([sys.executable, os.path.join(root, "bin", "tphip_hyphy"), "models_and_rates.bf"], input=towrite,
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0 and not p.stdout.startswith("Error"), p.stdout[:500] + p.stderr[-500:]
    rates = compute.parse_site_rates(out, correction=factor)      # rewrites the file with corrected_rates
    doc = json.load(open(out))
    assert len(doc["sites"]["rates"]) == 226 and len(doc["sites"]["corrected_rates"]) == 226
    got = np.array([r["rate"] for r in doc["sites"]["rates"]])
    # same numbers as the oracle fed with the file's own (empirical) base frequencies
    from tapir_amd import nexus
    names, st = nexus.read_states(os.path.join(golden_dir, "chr1_918.nex"))
    troot = newick.read_tree(tree)
    leaf_names = [n.name for n in newick.leaves(troot)]
    parent, blen, leaf = newick.to_arrays(troot, leaf_names)
    st = st[[names.index(n) for n in leaf_names]]
    # the file prints pi with 6 significant digits (HyPhy's default print precision); the engine used full precision
    pi = nexus.base_frequencies_from_histogram(np.bincount(st.ravel(), minlength=16)[None, :16])[0]
    ref = oracle.site_rates(st, parent, blen, leaf, pi, [0.96, 1, 0.58, 0.36, 1.87, 0.51])
    ok = (ref["flag"] == 0) | (ref["flag"] == 3)
    assert np.abs(got - ref["rate"])[ok].max() < 5.1e-5
    assert np.abs(np.array([r["ll"] for r in doc["sites"]["rates"]]) - ref["lnl"]).max() < 5.1e-5
    assert np.allclose(rates, got / factor)
    # without the override the shim estimates the exchangeabilities itself, as HyPhy's script does (stage 1), and
    # reports them in the JSON header (bf:1018-1031): within 6 % of PhyDesign's published values for this locus
    env2 = {k: v for k, v in os.environ.items() if k != "TPHIP_EXCHANGEABILITIES"}
    out2 = str(tmp_path / "chr1_918.stage1.rates")
    p = subprocess.run([sys.executable, os.path.join(root, "bin", "tphip_hyphy"), "models_and_rates.bf"],
                       input="\n".join([os.path.join(golden_dir, "chr1_918.nex"), tree, out2]), capture_output=True, text=True,
                       env=env2, timeout=300)
    assert p.returncode == 0 and not p.stdout.startswith("Error"), p.stdout[:500] + p.stderr[-500:]
    sm = json.load(open(out2))["sites"]["subs_matrix"]
    for key, want in (("AC", 0.96), ("AG", 1.0), ("AT", 0.58), ("CG", 0.36), ("CT", 1.87), ("GT", 0.51)):
        assert abs(sm[key] - want) < 0.06 * want, sm
    # a missing alignment must surface as an "Error" on stdout, which tapir turns into "hyphy error: ..."
    p = subprocess.run([sys.executable, os.path.join(root, "bin", "tphip_hyphy"), "x.bf"], input="nope.nex\n%s\n%s" % (tree, out),
                       capture_output=True, text=True, env=env, timeout=300)
    assert p.stdout.startswith("Error")


def _random_states(rng, ntaxa, ncols, ambig=0.15, gaps=0.1):
    base = (1 << rng.integers(0, 4, size=(ntaxa, ncols))).astype(np.uint8)
    # low-rate columns: copy a single base down most of the column
    const = rng.random(ncols) < 0.4
    base[:, const] = base[0, const]
    amb = rng.random((ntaxa, ncols)) < ambig
    base[amb] = rng.integers(1, 16, size=int(amb.sum())).astype(np.uint8)  # any IUPAC union, incl. N = 15
    base[rng.random((ntaxa, ncols)) < gaps] = 15
    return base


@pytest.mark.parametrize("newick_text", [
    "((a:0.3,b:0.1,c:0.2,d:0.05):0.2,(e:0.4,f:0.0):0.1,g:0.3);",                       # polytomies, trifurcating root, zero branch
    "(((((((a:.1,b:.2):.1,c:.3):.1,d:.2):.1,e:.1):.2,f:.3):.1,g:.2):.1,h:.4);",          # caterpillar: stack depth 0
    "(((a:.1,b:.2):.3,(c:.1,d:.2):.2):.1,((e:.3,f:.1):.2,(g:.2,h:.1):.3):.2);",          # balanced: deepest stack
    "((a:0.2,b:0.3):0.0,(c:0.1,(d:0.2,e:0.3):0.4)x:0.2)root;",                           # internal labels, zero internal branch
])
def test_tree_shapes_and_ambiguity_codes(oracle, newick_text):
    """Multifurcations, zero-length branches, unrooted (trifurcating) roots and IUPAC ambiguity masks: the
    tree compiler + kernel against the oracle's plain post-order recursion."""
    engine = _engine()
    from tapir_amd import newick
    root = newick.parse(newick_text)
    names = [n.name for n in newick.leaves(root)]
    parent, blen, leaf = newick.to_arrays(root, names)
    rng = np.random.default_rng(len(newick_text))
    ncols = 700
    st = _random_states(rng, len(names), ncols)
    pi = np.array([0.31, 0.19, 0.23, 0.27])
    ex = np.array([1.3, 1.0, 0.6, 0.8, 2.1, 0.9])
    plan = engine.Plan(len(names), parent, blen, leaf, [0, ncols], [pi], [ex], 3, [1], [[0, 2]], threshold=2)
    got = plan.site_rates(st)
    ref = oracle.site_rates(st, parent, blen, leaf, pi, ex)
    assert np.array_equal(got["flag"], ref["flag"])
    assert np.array_equal(got["nres"], ref["nres"])
    kappa = plan.models()[3]
    pin = dict(parent=parent, blen=blen, leaf=leaf)
    _assert_rates_match(oracle, got, ref, slice(0, ncols), st, pin, pi, ex, kappa[0])
    assert np.abs(got["lnl"] - ref["lnl"]).max() < 1e-10 * max(1.0, np.abs(ref["lnl"]).max())
    # the diagnostic (byte path) agrees with the oracle's curve everywhere, also on flat / ambiguous columns
    u = rng.uniform(-2, 2, ncols)
    f, g, h = plan.eval_columns(st, u)
    for c in range(0, ncols, 37):
        fo, go, ho = oracle.column_curve(st, parent, blen, leaf, pi, ex, c, np.array([u[c]]))
        assert abs(f[c] - fo[0]) < 1e-11 * max(1, abs(fo[0])) and abs(g[c] - go[0]) < 1e-10 and abs(h[c] - ho[0]) < 1e-10
    plan.close()


def test_plan_rejects_tree_alignment_mismatch(chr1_918):
    engine = _engine()
    c = chr1_918
    with pytest.raises(engine.TphipError, match="number of leaves"):
        engine.Plan(6, c["parent"], c["blen"], c["leaf"], [0, 10], [c["pi"]], [c["exch"]], 10, [1], [[0, 1]])


def test_locus_loglik_vs_oracle(oracle):
    """Stage-1 objective (whole-locus log-likelihood at site rate 1) for batches of candidate exchangeabilities
    and branch lengths, against the oracle's plain recursion; includes ambiguity codes, a ragged batch and the
    device cache of the alignment."""
    engine = _engine()
    from tapir_amd import synth
    rng = np.random.default_rng(7)
    for ntaxa, nloci, ncols in ((9, 5, 333), (70, 2, 600)):
        d = synth.simulate(nloci, ncols, ntaxa, 40 + ntaxa, rate_mean=0.01)
        pin = synth.plan_inputs(d["root"], d["names"])
        st = d["states"].numpy().copy()
        st[rng.random(st.shape) < 0.03] = 5  # R = A|G
        off = d["locus_offsets"].copy()
        off[1] -= 7
        plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off, d["pi"], d["exch"], 3, [1], [[0, 1]])
        nn = len(pin["parent"])
        ncand = 23
        cl = rng.integers(0, nloci, ncand)
        ce = np.exp(rng.normal(0, 0.5, (ncand, 6)))
        cb = pin["blen"][None, :] * np.exp(rng.normal(0, 0.7, (ncand, nn)))
        cache = plan.device_cache()
        got = plan.locus_loglik(st, cb, cl, ce, cache=cache)
        got2 = plan.locus_loglik(st, cb, cl, ce, cache=cache)  # second call reuses the device copy
        assert np.array_equal(got, got2)
        # shared-vector form: vector 3 scaled and with one branch perturbed == the explicit vector
        exp_vec = cb[3] * 0.7
        exp_vec[5] *= 1.25
        a = plan.locus_loglik(st, exp_vec[None, :], [cl[3]], [ce[3]], cache=cache)
        b = plan.locus_loglik(st, cb, [cl[3]], [ce[3]], cand_vec=[3], cand_scale=[0.7], cand_pidx=[5], cand_pfac=[1.25], cache=cache)
        assert abs(a[0] - b[0]) <= 1e-12 * abs(a[0])
        cache.release()
        for c in range(ncand):
            l = int(cl[c])
            ref = oracle.locus_loglik(st[:, off[l]:off[l + 1]], pin["parent"], cb[c], pin["leaf"], d["pi"][l], ce[c])
            assert abs(got[c] - ref) < 1e-9 * abs(ref), (c, got[c], ref)
        plan.close()


def test_host_pointer_pipeline_of_locus_groups_is_bit_identical(monkeypatch):
    """tphip_run_fused from pinned host memory cuts a big batch into locus groups (the upload of group k + 1 under the
    kernels of group k); whatever the number of groups, ragged loci and an empty one included, every output must equal
    the unsplit run's bit for bit -- and pageable buffers must take the plain path and agree too."""
    engine = _engine()
    from tapir_amd import synth
    nloci, ncols, ntaxa = 40, 56000, 12
    d = synth.simulate(nloci, ncols, ntaxa, 97)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    off = d["locus_offsets"].copy()
    off[7] -= 1234                # ragged
    off[20] = off[19]             # an empty locus
    stp = engine.pinned_empty(st.shape, np.uint8)
    stp[...] = st
    outs = {}
    for k in ("1", "3", "4", "9"):
        monkeypatch.setenv("TPHIP_HOST_SPLIT", k)
        plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off, d["pi"], d["exch"], pin["T"], [3, 9], [[1, 8], [2, 20]],
                           correction=pin["correction"])
        outs[k] = plan.run_fused(stp, pinned=True)
        if k == "4":
            outs["pageable"] = plan.run_fused(st)
        plan.close()
    monkeypatch.delenv("TPHIP_HOST_SPLIT")
    for k in ("3", "4", "9", "pageable"):
        for key in outs["1"]:
            assert np.array_equal(outs["1"][key], outs[k][key], equal_nan=True), (k, key)
    # a batch that does not cut into groups with columns (one locus holds everything, the others are empty): run whole
    off1 = np.array([0, st.shape[1], st.shape[1], st.shape[1], st.shape[1]])
    pi1, ex1 = np.repeat(d["pi"][:1], 4, axis=0), np.repeat(d["exch"][:1], 4, axis=0)
    plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off1, pi1, ex1, pin["T"], [3, 9], [[1, 8], [2, 20]],
                       correction=pin["correction"])
    a, b = plan.run_fused(stp, pinned=True), plan.run_fused(st)
    plan.close()
    for key in a:
        assert np.array_equal(a[key], b[key], equal_nan=True), key


def _balanced_tree(levels, rng):
    """Post-order arrays of a perfectly balanced binary tree with 2**levels leaves: levels - 1 partials are parked at the
    deepest point.  The value kernel's register stack holds 5, so 6 levels = 64 leaves need all of them and 7 levels fall
    back to the eigenbasis kernel with its LDS stack."""
    parent, blen, leaf = [], [], []
    counter = [0]

    def build(level):
        if level == 0:
            parent.append(-1); blen.append(float(rng.uniform(0.01, 0.3))); leaf.append(counter[0]); counter[0] += 1
            return len(parent) - 1
        a = build(level - 1)
        b = build(level - 1)
        parent.append(-1); blen.append(float(rng.uniform(0.01, 0.3))); leaf.append(-1)
        me = len(parent) - 1
        parent[a] = me; parent[b] = me
        return me

    build(levels)
    blen[-1] = 0.0
    return np.array(parent, np.int32), np.array(blen), np.array(leaf, np.int32)


def test_locus_value_kernel_codes_trees_and_widths(oracle, monkeypatch):
    """The transition-matrix value kernel (locus_value_kernel.hpp) against the oracle and against the eigenbasis kernel it
    replaces: every IUPAC state set, gaps and zero bytes; one and two columns per thread; a caterpillar (stack depth 1), a
    balanced 64-leaf tree (all 5 register slots) and a balanced 128-leaf tree (deeper than the register stack: falls
    back); column weights; a near-zero and a long branch."""
    engine = _engine()
    rng = np.random.default_rng(11)
    ncols, nloci = 700, 3
    for levels in (3, 6, 7):
        parent, blen, leaf = _balanced_tree(levels, rng)
        ntaxa = 1 << levels
        blen[1] = 1e-9
        blen[3] = 4.0
        st = (1 << rng.integers(0, 4, (ntaxa, ncols * nloci))).astype(np.uint8)
        st[:, ::3] = st[0, ::3]                                    # a third of the columns constant
        amb = rng.random(st.shape) < 0.08
        st[amb] = rng.integers(0, 16, int(amb.sum())).astype(np.uint8)   # every mask 0..15 (0 is read as a gap)
        off = np.array([0, ncols - 13, 2 * ncols, 3 * ncols])
        pi = rng.dirichlet([5, 5, 5, 5], nloci)
        w = rng.integers(1, 5, ncols * nloci).astype(np.float64)
        ncand = 17
        cl = rng.integers(0, nloci, ncand)
        ce = np.exp(rng.normal(0, 0.5, (ncand, 6)))
        cb = blen[None, :] * np.exp(rng.normal(0, 0.3, (ncand, len(parent))))
        results = {}
        for variant in ("1", "2", "eigenbasis"):
            monkeypatch.delenv("TPHIP_VALUE_COLS", raising=False)
            monkeypatch.delenv("TPHIP_VALUE_EIGENBASIS", raising=False)
            if variant == "eigenbasis":
                monkeypatch.setenv("TPHIP_VALUE_EIGENBASIS", "1")
            else:
                monkeypatch.setenv("TPHIP_VALUE_COLS", variant)
            plan = engine.Plan(ntaxa, parent, blen, leaf, off, pi, np.ones((nloci, 6)), 3, [1], [[0, 1]])
            plain = plan.locus_loglik(st, cb, cl, ce)
            cache = plan.device_cache()      # the library's own device copy: state codes packed once at upload
            assert np.array_equal(plain, plan.locus_loglik(st, cb, cl, ce, cache=cache)), (levels, variant)
            cache.release()
            plan.set_column_weights(w)
            results[variant] = (plain, plan.locus_loglik(st, cb, cl, ce))
            plan.close()
        for variant in ("1", "2"):
            for k in (0, 1):
                assert np.max(np.abs(results[variant][k] - results["eigenbasis"][k]) / np.abs(results["eigenbasis"][k])) < 1e-11, (levels, variant)
        for c in range(ncand):
            l = int(cl[c])
            ref = oracle.locus_loglik(st[:, off[l]:off[l + 1]], parent, cb[c], leaf, pi[l], ce[c])
            assert abs(results["2"][0][c] - ref) < 1e-9 * abs(ref), (levels, c)
    # a caterpillar: one parked partial at most
    nt = 12
    parent, blen, leaf = [], [], []
    parent.append(-1); blen.append(0.1); leaf.append(0)
    prev = 0
    for t in range(1, nt):
        parent.append(-1); blen.append(0.05 + 0.01 * t); leaf.append(t)
        tip = len(parent) - 1
        parent.append(-1); blen.append(0.02 * t); leaf.append(-1)
        me = len(parent) - 1
        parent[prev] = me; parent[tip] = me
        prev = me
    blen[-1] = 0.0
    parent, blen, leaf = np.array(parent, np.int32), np.array(blen), np.array(leaf, np.int32)
    st = (1 << rng.integers(0, 4, (nt, 300))).astype(np.uint8)
    st[:, 100:] = st[0, 100:]
    st[3, 100:200] = 15
    off = np.array([0, 300])
    pi = np.array([[0.3, 0.2, 0.2, 0.3]])
    monkeypatch.delenv("TPHIP_VALUE_COLS", raising=False)
    monkeypatch.delenv("TPHIP_VALUE_EIGENBASIS", raising=False)
    plan = engine.Plan(nt, parent, blen, leaf, off, pi, np.ones((1, 6)), 3, [1], [[0, 1]])
    ce = np.array([[1.0, 2.0, 0.5, 0.7, 3.0, 1.0]])
    got = plan.locus_loglik(st, blen[None, :], [0], ce)
    plan.close()
    ref = oracle.locus_loglik(st, parent, blen, leaf, pi[0], ce[0])
    assert abs(got[0] - ref) < 1e-10 * abs(ref)


def _fd_gradient(oracle, st, parent, leaf, pi, exch, blen, h=1e-5):
    """Central differences of the oracle's log-likelihood w.r.t. the six exchangeabilities and every log branch
    length (the independent check of the reverse-mode kernel)."""
    parent = np.asarray(parent)
    f = lambda e, b: oracle.locus_loglik(st, parent, b, leaf, pi, e)
    dex = np.zeros(6)
    for q in range(6):
        ep, em = exch.copy(), exch.copy()
        ep[q] *= 1 + h
        em[q] *= 1 - h
        dex[q] = (f(ep, blen) - f(em, blen)) / (2 * h * exch[q])
    dlt = np.zeros(len(parent))
    for b in np.flatnonzero(parent >= 0):
        bp, bm = blen.copy(), blen.copy()
        bp[b] *= np.exp(h)
        bm[b] *= np.exp(-h)
        dlt[b] = (f(exch, bp) - f(exch, bm)) / (2 * h)
    return dex, dlt


def test_locus_gradient_vs_oracle_finite_differences(oracle):
    """Reverse-mode gradient kernel: lnL bit-compatible with the value kernel, derivatives w.r.t. all six
    exchangeabilities and every branch length against central differences of the ORACLE's likelihood.
    Covers ambiguity codes, ragged loci, more candidates than resident workgroups' first wave, scaled and
    perturbed shared vectors, equal eigenvalues (all rates 1: the F_kl limit) and a polytomy."""
    engine = _engine()
    from tapir_amd import synth
    rng = np.random.default_rng(17)
    for ntaxa, nloci, ncols in ((9, 4, 211), (40, 2, 300)):
        d = synth.simulate(nloci, ncols, ntaxa, 60 + ntaxa, rate_mean=0.01)
        pin = synth.plan_inputs(d["root"], d["names"])
        st = d["states"].numpy().copy()
        st[rng.random(st.shape) < 0.03] = 5
        off = d["locus_offsets"].copy()
        off[1] -= 5
        plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off, d["pi"], d["exch"], 3, [1], [[0, 1]])
        nn = len(pin["parent"])
        ncand = 11
        cl = rng.integers(0, nloci, ncand)
        ce = np.exp(rng.normal(0, 0.5, (ncand, 6)))
        ce[0] = 1.0  # Jukes-Cantor-like exchangeabilities: three equal eigenvalues
        cb = pin["blen"][None, :] * np.exp(rng.normal(0, 0.7, (ncand, nn)))
        cs = np.exp(rng.normal(0, 0.3, ncand))
        ci = rng.integers(-1, nn - 1, ncand)
        cf = np.exp(rng.normal(0, 0.2, ncand))
        cache = plan.device_cache()
        val = plan.locus_loglik(st, cb, cl, ce, None, cs, ci, cf, cache=cache)
        lnl, dex, dlt, sdl = plan.locus_gradient(st, cb, cl, ce, None, cs, ci, cf, cache=cache)
        lnl2, dex2, none, sdl2 = plan.locus_gradient(st, cb, cl, ce, None, cs, ci, cf, cache=cache, per_branch=False)
        cache.release()
        assert none is None and np.array_equal(lnl, lnl2) and np.array_equal(dex, dex2) and np.array_equal(sdl, sdl2)
        assert np.max(np.abs(lnl - val) / np.abs(val)) < 1e-13
        assert np.max(np.abs(dlt.sum(1) - sdl)) < 1e-9 * np.abs(dlt).sum(1).max()
        for c in range(ncand):
            l = int(cl[c])
            b = cb[c] * cs[c]
            if ci[c] >= 0:
                b[ci[c]] *= cf[c]
            rdex, rdlt = _fd_gradient(oracle, st[:, off[l]:off[l + 1]], pin["parent"], pin["leaf"], d["pi"][l], ce[c], b)
            tol = 2e-6 * max(1.0, np.abs(rdex).max(), np.abs(rdlt).max())
            assert np.max(np.abs(dex[c] - rdex)) < tol, (c, dex[c], rdex)
            assert np.max(np.abs(dlt[c] - rdlt)) < tol, (c, np.abs(dlt[c] - rdlt).max())
        plan.close()
    # a polytomy (three internal children under the root) and a 2-taxon tree
    from tapir_amd import newick
    for text in ("((a:1,b:2):1,(c:1,d:1.5):2,(e:0.5,f:1):1,g:3);", "(a:1,b:2);"):
        root = newick.parse(text)
        names = [n.name for n in newick.leaves(root)]
        parent, blen, leaf = newick.to_arrays(root, names)
        st = rng.choice(np.array([1, 2, 4, 8, 15, 5], dtype=np.uint8), size=(len(names), 97))
        pi = np.array([0.1, 0.2, 0.3, 0.4])
        exch = np.array([0.7, 1.0, 1.9, 0.4, 2.5, 1.1])
        plan = engine.Plan(len(names), parent, blen, leaf, [0, 97], [pi], [exch], 3, [1], [[0, 1]])
        b = np.asarray(blen) * 0.1
        lnl, dex, dlt, sdl = plan.locus_gradient(st, b[None, :], [0], [exch])
        ref = oracle.locus_loglik(st, parent, b, leaf, pi, exch)
        rdex, rdlt = _fd_gradient(oracle, st, parent, leaf, pi, exch, b)
        assert abs(lnl[0] - ref) < 1e-10 * abs(ref)
        tol = 2e-6 * max(1.0, np.abs(rdex).max(), np.abs(rdlt).max())
        assert np.max(np.abs(dex[0] - rdex)) < tol and np.max(np.abs(dlt[0] - rdlt)) < tol
        plan.close()


def test_compress_columns_exact_and_weighted_likelihood(oracle):
    """Site-pattern compression (HyPhy's dupInfo, bf:960-963): the patterns of every locus are exactly its set of
    distinct columns (after 0 -> 15 normalisation), counts add up, the map points every column at an identical
    pattern, loci never mix (identical columns in different loci stay separate), empty loci survive; and the
    weighted likelihood / gradient on the patterns equals the plain one on all columns."""
    engine = _engine()
    from tapir_amd import synth
    rng = np.random.default_rng(3)
    ntaxa = 11
    d = synth.simulate(4, 500, ntaxa, 77, rate_mean=0.002)   # slow: plenty of duplicate columns
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy().copy()
    st[3, 7] = 0                             # 0 is read as 15 (all states)
    st[:, 1000:1500] = st[:, 0:500]          # locus 3 repeats locus 0 column by column
    off = np.array([0, 500, 500, 1000, 1500, 2000])   # an empty locus in the middle
    pst, poff, w, cmap = engine.compress_columns(st, off)
    assert pst.shape[0] == ntaxa and poff[0] == 0 and poff[-1] == pst.shape[1] == len(w)
    assert w.sum() == 2000 and np.all(w >= 1)
    norm = np.where(st == 0, 15, st & 15)
    for l in range(5):
        cols = norm[:, off[l]:off[l + 1]]
        pats = pst[:, poff[l]:poff[l + 1]]
        uniq, counts = np.unique(cols, axis=1, return_counts=True)
        assert pats.shape[1] == uniq.shape[1]
        order = np.lexsort(pats[::-1])                      # np.unique sorts columns lexicographically
        assert np.array_equal(pats[:, order], uniq)
        assert np.array_equal(w[poff[l]:poff[l + 1]][order], counts)
        assert np.all((cmap[off[l]:off[l + 1]] >= poff[l]) & (cmap[off[l]:off[l + 1]] < poff[l + 1]))
    assert np.array_equal(pst[:, cmap], norm)
    assert poff[1] == poff[2] and (poff[1] - poff[0]) == (poff[4] - poff[3])
    # deterministic
    pst2, poff2, w2, cmap2 = engine.compress_columns(st, off)
    assert np.array_equal(pst, pst2) and np.array_equal(w, w2) and np.array_equal(cmap, cmap2)
    assert pst.shape[1] < 1200
    # weighted likelihood and gradient on patterns == unweighted on columns
    pi = np.vstack([d["pi"][0], d["pi"][1], d["pi"][1], d["pi"][2], d["pi"][3]])
    ex = np.ones((5, 6))
    full = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off, pi, ex, 3, [1], [[0, 1]])
    comp = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], poff, pi, ex, 3, [1], [[0, 1]])
    comp.set_column_weights(w)
    cl = np.array([0, 1, 2, 3, 4, 2])
    ce = np.exp(rng.normal(0, 0.4, (6, 6)))
    cb = np.asarray(pin["blen"])[None, :] * np.exp(rng.normal(0, 0.5, (6, len(pin["parent"])))) * 0.01
    a = full.locus_gradient(st, cb, cl, ce)
    b = comp.locus_gradient(pst, cb, cl, ce)
    assert a[0][1] == 0.0 and b[0][1] == 0.0   # the empty locus
    for x, y in zip(a, b):
        assert np.max(np.abs(x - y)) <= 1e-10 * max(1.0, np.abs(x).max())
    va, vb = full.locus_loglik(st, cb, cl, ce), comp.locus_loglik(pst, cb, cl, ce)
    assert np.max(np.abs(va - vb)) <= 1e-11 * np.abs(va).max()
    comp.set_column_weights(None)
    assert abs(comp.locus_loglik(pst, cb, cl, ce)[0] - vb[0]) > 1.0   # weights really were in use
    full.close()
    comp.close()
    # all-gap alignment, single column, a locus of identical columns
    one = engine.compress_columns(np.full((3, 1), 15, np.uint8), [0, 1])
    assert one[0].shape == (3, 1) and one[2][0] == 1
    same = engine.compress_columns(np.tile(np.array([[1], [2], [4]], np.uint8), (1, 1000)), [0, 400, 1000])
    assert same[0].shape == (3, 2) and list(same[2]) == [400, 600] and list(same[1]) == [0, 1, 2]


def test_locus_gradient_kernels_agree(oracle):
    """The two gradient kernels: the transition-matrix one (locus_grad2_kernel: taken when the alignment is the library's
    cached copy, whose packed state codes it reads, and the tree is binary) and the eigenbasis one (locus_grad_kernel: any
    caller-owned array, any tree) compute the same numbers by different routes -- value, six rate derivatives, every branch
    derivative and curvature within 1e-10 of the largest entry -- on 5 ... 300 taxa (rescaled partials at 300), with column
    weights, gaps and ambiguity codes, zero-length and long branches, equal eigenvalues, and a column count that leaves
    padding lanes.  Finite differences of the oracle pin one of them (test above); a polytomy takes the eigenbasis kernel
    with or without the cache."""
    engine = _engine()
    from tapir_amd import newick, synth
    rng = np.random.default_rng(23)
    for ntaxa, nloci, ncols in ((5, 3, 77), (16, 4, 333), (64, 2, 1000), (300, 1, 140)):
        d = synth.simulate(nloci, ncols, ntaxa, 200 + ntaxa, rate_mean=0.01)
        pin = synth.plan_inputs(d["root"], d["names"])
        st = d["states"].numpy().copy()
        st[rng.random(st.shape) < 0.02] = rng.choice(np.array([3, 5, 6, 7, 9, 10, 11, 12, 13, 14, 15], np.uint8))
        plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"], 3, [1], [[0, 1]])
        nn = len(pin["parent"])
        ncand = 7
        cl = rng.integers(0, nloci, ncand)
        ce = np.exp(rng.normal(0, 0.5, (ncand, 6)))
        ce[0] = 1.0
        cb = np.asarray(pin["blen"])[None, :] * np.exp(rng.normal(-3.5, 1.0, (ncand, nn)))
        cb[1, rng.integers(0, nn - 1, 3)] = 0.0
        cb[2, rng.integers(0, nn - 1, 2)] = 4.0
        w = rng.integers(1, 5, st.shape[1]).astype(float)
        for weights in (None, w):
            plan.set_column_weights(weights)
            old = plan.locus_gradient(st, cb, cl, ce, curvature=True)
            cache = plan.device_cache()
            new = plan.locus_gradient(st, cb, cl, ce, curvature=True, cache=cache)
            val = plan.locus_loglik(st, cb, cl, ce, cache=cache)
            cache.release()
            assert np.max(np.abs(new[0] - val) / np.abs(val)) < 1e-13
            for name, a, b in zip(("lnl", "dexch", "dlogt", "sum_dlogt", "d2logt"), old, new):
                assert np.all(np.isfinite(b)), name
                assert np.max(np.abs(a - b)) <= 1e-10 * max(np.abs(a).max(), 1e-12), (ntaxa, name, np.max(np.abs(a - b)), np.abs(a).max())
        plan.close()
    root = newick.parse("((a:1,b:2):1,(c:1,d:1.5):2,(e:0.5,f:1):1,g:3);")
    names = [n.name for n in newick.leaves(root)]
    parent, blen, leaf = newick.to_arrays(root, names)
    st = rng.choice(np.array([1, 2, 4, 8, 15, 5], dtype=np.uint8), size=(len(names), 97))
    plan = engine.Plan(len(names), parent, blen, leaf, [0, 97], [[0.1, 0.2, 0.3, 0.4]], [np.ones(6)], 3, [1], [[0, 1]])
    b = np.asarray(blen)[None, :] * 0.1
    old = plan.locus_gradient(st, b, [0], [np.ones(6)], curvature=True)
    cache = plan.device_cache()
    new = plan.locus_gradient(st, b, [0], [np.ones(6)], curvature=True, cache=cache)
    cache.release()
    plan.close()
    for a, c in zip(old, new):
        assert np.array_equal(a, c)


def test_locus_gradient_hessian_diagonal(oracle):
    """d2 lnL / d (log t_b)^2 from the gradient kernel against second central differences of the oracle likelihood."""
    engine = _engine()
    from tapir_amd import synth
    rng = np.random.default_rng(5)
    ntaxa = 10
    d = synth.simulate(2, 400, ntaxa, 91, rate_mean=0.01)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    off = d["locus_offsets"]
    plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off, d["pi"], d["exch"], 3, [1], [[0, 1]])
    parent = np.asarray(pin["parent"])
    ce = np.exp(rng.normal(0, 0.4, (2, 6)))
    cb = np.asarray(pin["blen"])[None, :] * np.exp(rng.normal(0, 0.5, (2, len(parent)))) * 0.02
    w = rng.integers(1, 4, st.shape[1]).astype(float)
    for weights in (None, w):
        plan.set_column_weights(weights)
        lnl, dex, dlt, sdl, d2 = plan.locus_gradient(st, cb, [0, 1], ce, curvature=True)
        for c in range(2):
            cols = slice(off[c], off[c + 1])
            reps = np.ones(400, dtype=int) if weights is None else w[cols].astype(int)
            stc = np.repeat(st[:, cols], reps, axis=1)       # the oracle has no weights: repeat the columns
            f = lambda b: oracle.locus_loglik(stc, parent, b, pin["leaf"], d["pi"][c], ce[c])
            h = 1e-3
            for b in np.flatnonzero(parent >= 0)[::3]:
                bp, bm = cb[c].copy(), cb[c].copy()
                bp[b] *= np.exp(h)
                bm[b] *= np.exp(-h)
                ref = (f(bp) - 2 * f(cb[c]) + f(bm)) / (h * h)
                assert abs(d2[c, b] - ref) < 1e-4 * max(1.0, abs(ref)), (c, b, d2[c, b], ref)
    plan.close()


def test_rate_mixture_vs_oracle(oracle):
    """Opt-in GTR+G extension (tphip_plan_desc.ncat): L(s) = sum_k w_k L(s rho_k) inside the per-site optimiser, HIP vs
    the oracle's restatement of the same mixture (1e-6 on rates, exact flags); categories that all equal 1 give the
    plain model's answers; the diagnostic curve kernel (f, g, h at chosen u) agrees with the oracle's mixture too.
    The reference's script has no mixture, so K = 1 is the only setting with reference parity."""
    engine = _engine()
    from tapir_amd import compute, synth
    for ntaxa, ncols, seed in ((8, 700, 5), (40, 500, 6)):
        d = synth.simulate(2, ncols, ntaxa, seed)
        pin = synth.plan_inputs(d["root"], d["names"])
        st = d["states"].numpy()
        r, w = compute.discrete_gamma(0.5, 4)
        args = (ntaxa, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"], pin["T"], [10], [[5, 15]])
        plain = engine.Plan(*args, correction=pin["correction"])
        ones = engine.Plan(*args, correction=pin["correction"], cat_rates=[1.0, 1.0, 1.0], cat_weights=[0.5, 0.25, 0.25])
        mix = engine.Plan(*args, correction=pin["correction"], cat_rates=r, cat_weights=w)
        a, b, c = plain.site_rates(st), ones.site_rates(st), mix.site_rates(st)
        ok = a["flag"] == 0
        assert np.array_equal(a["flag"], b["flag"]) and _rel(b["rate"][ok], a["rate"][ok], 1e-12).max() < 1e-9
        assert np.abs(a["lnl"] - b["lnl"]).max() < 1e-9
        lam, U, Ui, kappa = mix.models()
        for l in range(2):
            sl = slice(l * ncols, (l + 1) * ncols)
            ref = oracle.site_rates(st[:, sl], pin["parent"], pin["blen"], pin["leaf"], d["pi"][l], d["exch"][l], r, w)
            assert np.array_equal(c["flag"][sl], ref["flag"])
            okm = (ref["flag"] == 0) | (ref["flag"] == 3)
            rel = _rel(c["rate"][sl][okm], ref["rate"][okm], 1e-12)
            lnl_ok = np.abs(c["lnl"][sl] - ref["lnl"]) < 1e-9
            # almost flat maxima cannot be located to 1e-6 by anyone: there the log-likelihoods must agree instead
            assert np.all((rel < RTOL_RATE) | lnl_ok[okm]), rel.max()
            assert lnl_ok.mean() > 0.99
        assert np.abs(c["rate"] - a["rate"])[ok & (c["flag"] == 0)].max() > 1e-3     # the mixture does change the optimum
        u = np.linspace(-2.0, 2.0, st.shape[1])
        f, g, h = mix.eval_columns(st, u)
        for col in (3, 77, ncols + 5):
            l = col // ncols
            fo, go, ho = _mixture_curve(oracle, st[:, l * ncols:(l + 1) * ncols], pin, d["pi"][l], d["exch"][l], col - l * ncols, u[col], r, w)
            assert abs(f[col] - fo) < 1e-9 * max(1, abs(fo)) and abs(g[col] - go) < 1e-8 * max(1, abs(go))
            assert abs(h[col] - ho) < 1e-7 * max(1, abs(ho))
        for p in (plain, ones, mix):
            p.close()
    with pytest.raises(engine.TphipError, match="positive"):
        engine.Plan(*args, cat_rates=[1.0, -1.0], cat_weights=[0.5, 0.5])


def _mixture_curve(oracle, st, pin, pi, exch, col, u, rates, weights):
    """f, g, h of the mixture at u from the oracle's single-category curve (independent of its own mixture code)."""
    fk, gk, hk = [], [], []
    for r in rates:
        f, g, h = oracle.column_curve(st, pin["parent"], pin["blen"], pin["leaf"], pi, exch, col, np.array([u + np.log(r)]))
        fk.append(f[0]); gk.append(g[0]); hk.append(h[0])
    fk, gk, hk = np.array(fk) + np.log(weights), np.array(gk), np.array(hk)
    top = fk.max()
    p = np.exp(fk - top)
    z = p.sum()
    g = (p * gk).sum() / z
    return top + np.log(z), g, (p * (hk + gk * gk)).sum() / z - g * g


def test_locus_kernels_on_a_400_taxon_tree(oracle):
    """Trees far beyond the bench shapes: the stage-1 kernels shrink their per-branch LDS accumulators instead of
    refusing the tree (400 taxa: 799 nodes), rescaling of tiny partials included."""
    engine = _engine()
    from tapir_amd import synth
    rng = np.random.default_rng(9)
    ntaxa, ncols = 400, 130
    d = synth.simulate(1, ncols, ntaxa, 123, rate_mean=0.02)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], [0, ncols], d["pi"], d["exch"], 3, [1], [[0, 1]])
    parent = np.asarray(pin["parent"])
    b = np.asarray(pin["blen"]) * 0.02
    ce = np.exp(rng.normal(0, 0.3, (1, 6)))
    val = plan.locus_loglik(st, b[None, :], [0], ce)
    lnl, dex, dlt, sdl, d2 = plan.locus_gradient(st, b[None, :], [0], ce, curvature=True)
    ref = oracle.locus_loglik(st, parent, b, pin["leaf"], d["pi"][0], ce[0])
    assert abs(val[0] - ref) < 1e-9 * abs(ref) and abs(lnl[0] - ref) < 1e-9 * abs(ref)
    h = 1e-5
    for node in np.flatnonzero(parent >= 0)[::97]:
        up, dn = b.copy(), b.copy()
        up[node] *= np.exp(h)
        dn[node] *= np.exp(-h)
        fd = (oracle.locus_loglik(st, parent, up, pin["leaf"], d["pi"][0], ce[0]) -
              oracle.locus_loglik(st, parent, dn, pin["leaf"], d["pi"][0], ce[0])) / (2 * h)
        assert abs(dlt[0, node] - fd) < 2e-6 * max(1.0, np.abs(dlt).max()), (node, dlt[0, node], fd)
    assert abs(dlt.sum() - sdl[0]) < 1e-9 * np.abs(dlt).sum() and np.all(np.isfinite(d2))
    plan.close()


def test_site_rates_on_300_and_500_taxon_trees(oracle):
    """Beyond 256 tips the site-rate kernel leaves its register-resident packed states for the byte path (NW = 0) and
    partials need rescaling more than once per column; both against the oracle."""
    engine = _engine()
    from tapir_amd import synth
    for ntaxa, ncols, seed in ((300, 96, 7), (500, 70, 8)):
        d = synth.simulate(1, ncols, ntaxa, seed, rate_mean=0.01)
        pin = synth.plan_inputs(d["root"], d["names"])
        st = d["states"].numpy()
        plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], [0, ncols], d["pi"], d["exch"], pin["T"], [10], [[5, 15]],
                           correction=pin["correction"])
        got = plan.site_rates(st)
        ref = oracle.site_rates(st, pin["parent"], pin["blen"], pin["leaf"], d["pi"][0], d["exch"][0])
        assert np.array_equal(got["flag"], ref["flag"]) and np.array_equal(got["nres"], ref["nres"])
        ok = (ref["flag"] == 0) | (ref["flag"] == 3)
        assert ok.sum() > ncols // 2
        assert _rel(got["rate"][ok], ref["rate"][ok], 1e-12).max() < RTOL_RATE
        assert np.abs(got["lnl"] - ref["lnl"]).max() < 1e-9 * np.abs(ref["lnl"]).max()
        plan.close()


def test_randomised_parity_sweep():
    """tools/fuzz_parity.py on a fixed seed: random trees with polytomies (2..90 taxa), ragged loci, gaps / IUPAC codes,
    noisy and saturating columns, skewed frequencies and rates -- site rates (exact flags, 1e-6 on rates, lnL) and the
    stage-1 value / gradient kernels against the oracle.  This sweep is what exposed, and now guards, the
    noise-dependent saturation flags (fixed by confirming far-out optima by value) and the child order of the
    parsimony start at polytomies."""
    _engine()
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "30", "3"], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "30 cases, 0 with discrepancies" in p.stdout, p.stdout[-3000:]


def test_cli_site_rates_and_subset_paths_on_gpu(golden_dir, tmp_path, oracle):
    """The two re-analysis paths of the command line through the real engine: --site-rates (PI kernels alone on the
    rates of existing .rates files: divided by the correction again and NOT culled, as bin/tapir_compute.py:103-104,
    153-158 does) and --subset-pi-map-file (rates sliced before PI, tapir/base.py:116-123)."""
    _engine()
    import json
    import os
    import shutil
    import sqlite3
    from test_host_logic import _run_cli
    from tapir_amd import cli, nexus
    outdir, _ = _run_cli(golden_dir, tmp_path)
    rates_dir = tmp_path / "rates"
    rates_dir.mkdir()
    shutil.copy(os.path.join(outdir, "chr1_918.nex.rates"), rates_dir)
    out2 = tmp_path / "out2"
    out2.mkdir()
    cli.main([str(rates_dir), os.path.join(golden_dir, "Euteleost.tree"), "--output", str(out2), "--times", "10,50",
              "--intervals", "0-10,20-70", "--site-rates"])
    conn = sqlite3.connect(os.path.join(str(out2), "phylogenetic-informativeness.sqlite"))
    assert conn.execute("select locus from loci").fetchall() == [("chr1_918.nex",)]
    doc = json.load(open(rates_dir / "chr1_918.nex.rates"))
    r = np.array([x["rate"] for x in doc["sites"]["rates"]]) / 100
    for t in (10, 50):
        got = conn.execute("select pi from net where time=?", (t,)).fetchone()[0]
        assert abs(got - np.nansum(oracle.get_townsend_pi(t, r))) <= 1e-9 * got
    si, _ = oracle.net_integrals(r, [[0, 10], [20, 70]], 0)
    rows = dict(conn.execute("select interval, pi from interval").fetchall())
    assert abs(rows["0-10"] - si[0]) <= 1e-9 * si[0] and abs(rows["20-70"] - si[1]) <= 1e-9 * si[1]
    conn.close()
    # subset map
    m = tmp_path / "map.tsv"
    m.write_text("chr1_918.nex\t50\t150\n")
    sub = tmp_path / "sub"
    sub.mkdir()
    outdir3, _ = _run_cli(golden_dir, sub, extra=["--subset-pi-map-file", str(m)])
    doc = json.load(open(os.path.join(outdir3, "chr1_918.nex.rates")))
    _, st = nexus.read_states(os.path.join(golden_dir, "chr1_918.nex"))
    inf = ((st == 1) | (st == 2) | (st == 4) | (st == 8)).sum(axis=0) >= 3
    rr = np.array([x["rate"] for x in doc["sites"]["corrected_rates"]])
    rr[~inf] = np.nan
    rr = rr[50:150]
    conn = sqlite3.connect(os.path.join(outdir3, "phylogenetic-informativeness.sqlite"))
    net20 = conn.execute("select pi from net where time=20").fetchone()[0]
    assert abs(net20 - np.nansum(oracle.get_townsend_pi(20, rr))) <= 1e-9 * net20
    conn.close()
