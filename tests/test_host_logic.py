"""Host logic above the C ABI, on CPU (`-m "not gpu"`): argparse helpers (the reference's test_base.py),
tree / NEXUS readers, tree correction (test_compute.py:76-91), JSON and sqlite writers, the CLI end to end
with the oracle-backed test engine, the C ABI's symbol table, and the gloo all-gather of PI tables."""
import argparse
import ctypes
import json
import os
import re
import shutil
import sqlite3
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- tapir/tests/test_base.py, restated -----------------------------------------------------------
def test_is_dir(tmp_path):
    from tapir_amd.base import is_dir
    assert is_dir(str(tmp_path))
    with pytest.raises(argparse.ArgumentTypeError):
        is_dir(str(tmp_path / "nope"))


def test_get_output_type():
    from tapir_amd.base import get_output_type
    assert get_output_type('test.jpg') == 'jpg'
    with pytest.raises(AssertionError):
        get_output_type('test.bob')
    with pytest.raises(AssertionError):
        get_output_type('test')


def test_list_parsers():
    from tapir_amd.base import get_list_from_ints, get_list_from_ranges, get_strings_from_items
    assert get_list_from_ints('1,2,3') == [1, 2, 3]
    assert get_strings_from_items('1,2,3') == ['1', '2', '3']
    assert get_list_from_ranges('1-2,2-3,3-4') == [[1, 2], [2, 3], [3, 4]]
    with pytest.raises(argparse.ArgumentTypeError):
        get_list_from_ints('1,b')
    with pytest.raises(argparse.ArgumentTypeError):
        get_list_from_ranges('1-x')


def test_get_files(golden_dir, tmp_path):
    from tapir_amd.base import get_files
    for f in ("chr1_918.nex", "informativeness_cutoff.nex"):
        shutil.copy(os.path.join(golden_dir, f), tmp_path)
    (tmp_path / "test-extension.nexus").write_text("")
    observed = [os.path.basename(i) for i in get_files(str(tmp_path), '*.nex,*.nexus')]
    assert set(observed) == {'chr1_918.nex', 'informativeness_cutoff.nex', 'test-extension.nexus'}
    with pytest.raises(IOError):
        get_files('test-data', '*.rrwrr')


def test_create_unique_dir(tmp_path):
    from tapir_amd.base import create_unique_dir
    d = tmp_path / "out"
    d.mkdir()
    assert create_unique_dir(str(d)) == str(d)          # empty: used as is
    (d / "x").write_text("1")
    assert create_unique_dir(str(d)) == str(d) + ".1"   # non-empty: sibling .1
    assert create_unique_dir(str(d)) == str(d) + ".2"


def test_parse_subset_map_file(tmp_path):
    from tapir_amd.base import parse_subset_map_file
    p = tmp_path / "m.tsv"
    p.write_text("a.nex\t0\t10\n\nb.nex\t5\t7\n")
    assert dict(parse_subset_map_file(str(p))) == {"a.nex": [0, 10], "b.nex": [5, 7]}


# ---- trees and alignments ---------------------------------------------------------------------------
def test_tree_adjustment_matches_reference_test(golden_dir, tmp_path):
    """TestTreeAdjustment (test_compute.py:76-91): Euteleost.tree / 100."""
    from tapir_amd import compute, newick
    depth, factor, pth = compute.correct_branch_lengths(os.path.join(golden_dir, 'Euteleost.tree'), 'newick', d=str(tmp_path))
    assert depth == 174.0 and factor == 100
    assert os.path.basename(pth) == "Tree_100_174.0.newick"
    got = newick.read_tree(pth)
    exp = newick.parse('(danRer6:1.74,(oryLat2:1,(gasAcu1:0.93,(fr2:0.37,tetNig2:0.37):0.56):0.07):0.74);')
    ga, ea = newick.postorder(got), newick.postorder(exp)
    assert [n.name for n in ga] == [n.name for n in ea]
    assert np.allclose([n.length or 0 for n in ga], [n.length or 0 for n in ea], rtol=0, atol=1e-15)


def test_newick_roundtrip_and_errors():
    from tapir_amd import newick
    t = "((a:1.5,'b c':2):0.25,(d:1e-3,e:3,f:4)g:1)root;"
    root = newick.parse(t)
    assert [n.name for n in newick.leaves(root)] == ["a", "b c", "d", "e", "f"]
    assert abs(newick.tree_length(root) - (1.5 + 2 + .25 + 1e-3 + 3 + 4 + 1)) < 1e-12
    again = newick.parse(newick.write(root))
    assert newick.write(again) == newick.write(root)
    for bad in ("", "a,b;", "((a,b);", "(a,b));"):
        with pytest.raises(newick.NewickError):
            newick.parse(bad)


def test_nexus_tree_file(tmp_path):
    from tapir_amd import newick
    p = tmp_path / "t.nex"
    p.write_text("#NEXUS\nbegin trees;\n translate 1 alpha, 2 beta, 3 gamma;\n tree one = [&R] ((1:1,2:1):1,3:2);\nend;\n")
    root = newick.read_tree(str(p), "nexus")
    assert [n.name for n in newick.leaves(root)] == ["alpha", "beta", "gamma"]
    assert newick.distance_from_tip(root) == 2.0


def test_nexus_matrix_and_masks(golden_dir, tmp_path):
    from tapir_amd import compute, nexus
    names, st = nexus.read_states(os.path.join(golden_dir, "chr1_918.nex"))
    assert names == ["danRer6", "fr2", "oryLat2", "gasAcu1", "tetNig2"] and st.shape == (5, 226)
    assert set(np.unique(st)) <= {1, 2, 4, 8, 15}
    exp = np.load(os.path.join(golden_dir, "chr1_918-test-cutoff-values.npy"))
    got = compute.get_informative_sites(os.path.join(golden_dir, "chr1_918.nex"), 3)
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    small = compute.get_informative_sites(os.path.join(golden_dir, "informativeness_cutoff.nex"), 3)
    assert np.isnan(small[:2]).all() and np.array_equal(small[2:], [1.0, 1.0])
    # interleaved + ambiguity codes + lowercase
    p = tmp_path / "i.nex"
    p.write_text("#NEXUS\nbegin data;\n dimensions ntax=2 nchar=6;\n format datatype=dna missing=? gap=- interleave;\n"
                 "matrix\n a ACg\n b R?-\n\n a TNY\n b ttt\n;\nend;\n")
    names, st = nexus.read_states(str(p))
    assert st.tolist() == [[1, 2, 4, 8, 15, 10], [5, 15, 15, 8, 8, 8]]
    with pytest.raises(nexus.NexusError):
        nexus.read_matrix(os.path.join(str(tmp_path), "i.nex").replace("i.nex", "missing.nex")) if False else nexus.encode(["AZ"])
    empty = tmp_path / "e.nexus"
    empty.write_text("")
    with pytest.raises(nexus.NexusError):
        nexus.read_matrix(str(empty))


def test_base_frequencies_match_fixture_header(golden_dir):
    """HarvestFrequencies counts a gap as 1/4 of each base: gives the PhyDesign header 0.14/0.19/0.33/0.34."""
    from tapir_amd import nexus
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_engine
    names, st = nexus.read_states(os.path.join(golden_dir, "chr1_918.nex"))
    pi = nexus.base_frequencies_from_histogram(oracle_engine.state_histogram(st, [0, 226]))[0]
    # 0.137/0.187/0.329/0.346 (gap-free counting would give 0.112/0.173/0.347/0.368)
    assert np.allclose(pi, [0.13738938, 0.1869469, 0.32942478, 0.34623894], atol=1e-8)
    assert np.abs(pi - np.array([0.14, 0.19, 0.33, 0.34])).max() < 0.0065


def test_parse_site_rates_and_cull(golden_dir, tmp_path):
    """TestTransform (test_compute.py:19-36) with real comparisons, and the rewrite with corrected_rates."""
    from tapir_amd import compute
    src = os.path.join(golden_dir, "test-uniform-draw-weights.rates.json")
    exp = np.load(os.path.join(golden_dir, "test-parsed-rates.npy")).ravel()
    assert np.array_equal(compute.parse_site_rates(src, test=True), exp)
    assert np.array_equal(compute.parse_site_rates(src, 10., test=True), exp / 10.)
    cp = tmp_path / "x.rates"
    shutil.copy(src, cp)
    got = compute.parse_site_rates(str(cp), correction=100)
    doc = json.load(open(cp))
    assert [d["site"] for d in doc["sites"]["corrected_rates"]] == list(range(1, 101))
    assert np.array_equal([d["rate"] for d in doc["sites"]["corrected_rates"]], got)
    mask = np.load(os.path.join(golden_dir, "chr1_918-test-cutoff-values.npy"))[:100]
    culled = compute.cull_uninformative_rates(compute.parse_site_rates(src, 10., test=True), mask)
    expc = np.load(os.path.join(golden_dir, "test-culled-rates.npy"))
    assert np.array_equal(np.isnan(culled), np.isnan(expc)) and np.array_equal(culled[~np.isnan(culled)], expc[~np.isnan(expc)])
    t = compute.get_time(0, 5)
    assert t.shape == (5, 1) and t.ravel().tolist() == [0, 1, 2, 3, 4]


# ---- tree program (host half of the kernel) ----------------------------------------------------------
def test_tree_program_shapes():
    """The compiled traversal: op counts and Sethi-Ullman stack depth for known shapes, via the C ABI is not
    possible without a GPU, so the same invariants are checked on the oracle-side arrays."""
    from tapir_amd import newick, synth
    root, names = synth.yule_tree(64, 1)
    parent, blen, leaf = newick.to_arrays(root, names)
    assert (parent[:-1] > np.arange(len(parent) - 1)).all() and parent[-1] == -1
    assert sorted(leaf[leaf >= 0].tolist()) == list(range(64))
    assert abs(newick.distance_from_tip(root) - 100.0) < 1e-9
    pin = synth.plan_inputs(root, names)
    assert pin["T"] == 100 and pin["correction"] == 100


# ---- sqlite and the CLI end to end (oracle-backed engine) ----------------------------------------------
REF_SCHEMA = [
    "CREATE TABLE loci (id INTEGER PRIMARY KEY AUTOINCREMENT, locus TEXT)",
    "CREATE TABLE net (id INT, time INT, pi FLOAT,\n            FOREIGN KEY(id) REFERENCES loci(id) DEFERRABLE INITIALLY\n            DEFERRED)",
    "CREATE TABLE discrete (id INT, time INT, pi FLOAT, \n            FOREIGN KEY(id) REFERENCES loci(id) DEFERRABLE INITIALLY\n            DEFERRED)",
    "CREATE TABLE interval (id INT, interval TEXT, pi FLOAT,\n            error FLOAT, FOREIGN KEY(id) REFERENCES loci(id) DEFERRABLE\n            INITIALLY DEFERRED)",
]


def _run_cli(golden_dir, tmp_path, extra=(), engine_mod=None):
    from tapir_amd import cli
    aln = tmp_path / "aln"
    aln.mkdir()
    shutil.copy(os.path.join(golden_dir, "chr1_918.nex"), aln)
    out = tmp_path / "out"
    out.mkdir()
    argv = [str(aln), os.path.join(golden_dir, "Euteleost.tree"), "--output", str(out), "--times", "10,20,50",
            "--intervals", "0-10,10-15,20-100", "--exchangeabilities", "0.96,1,0.58,0.36,1.87,0.51"] + list(extra)
    return cli.main(argv, engine_mod=engine_mod), aln


def check_cli_outputs(outdir, oracle, golden_dir):
    files = sorted(os.listdir(outdir))
    assert files == ["Tree_100_174.0.newick", "chr1_918.nex.rates", "phylogenetic-informativeness.sqlite"]
    doc = json.load(open(os.path.join(outdir, "chr1_918.nex.rates")))
    assert list(doc.keys()) == ["sites"]
    assert set(doc["sites"].keys()) == {"freqs", "subs_matrix", "rates", "corrected_rates"}
    assert set(doc["sites"]["freqs"]) == set("ACGT") and set(doc["sites"]["subs_matrix"]) == {"AC", "AG", "AT", "CG", "CT", "GT"}
    rows = doc["sites"]["rates"]
    assert [r["site"] for r in rows] == list(range(1, 227)) and set(rows[0]) == {"site", "subst", "rate", "ll"}
    for r in rows[:40]:
        for k in ("subst", "rate", "ll"):
            assert abs(r[k] * 1e4 - round(r[k] * 1e4)) < 1e-6  # 4 decimals, Format(x,0,4)
    # values: empirical pi (unrounded) + the fixture's exchangeabilities -> same numbers as the oracle
    from tapir_amd import newick, nexus
    names, st = nexus.read_states(os.path.join(golden_dir, "chr1_918.nex"))
    root = newick.read_tree(os.path.join(outdir, "Tree_100_174.0.newick"))
    leaf_names = [n.name for n in newick.leaves(root)]
    parent, blen, leaf = newick.to_arrays(root, leaf_names)
    st = st[[names.index(n) for n in leaf_names]]
    pi = np.array([doc["sites"]["freqs"][b] for b in "ACGT"])
    ref = oracle.site_rates(st, parent, blen, leaf, pi, [0.96, 1, 0.58, 0.36, 1.87, 0.51])
    ok = (ref["flag"] == 0) | (ref["flag"] == 3)
    assert np.abs(np.array([r["rate"] for r in rows]) - ref["rate"])[ok].max() < 5.1e-5
    assert np.abs(np.array([r["ll"] for r in rows]) - ref["lnl"]).max() < 5.1e-5
    corr = np.array([r["rate"] for r in doc["sites"]["corrected_rates"]])
    assert np.array_equal(corr, np.array([r["rate"] for r in rows]) / 100)
    # sqlite: byte-identical DDL, row counts, values vs the numpy/scipy restatement of worker()
    conn = sqlite3.connect(os.path.join(outdir, "phylogenetic-informativeness.sqlite"))
    schema = [r[0] for r in conn.execute("select sql from sqlite_master where type='table' and name!='sqlite_sequence' order by rowid")]
    assert schema == REF_SCHEMA
    assert conn.execute("select locus from loci").fetchall() == [("chr1_918",)]
    net = conn.execute("select time, pi from net order by time").fetchall()
    assert [t for t, _ in net] == list(range(174))
    rates = corr.copy()
    rates[ref["nres"] < 3] = np.nan
    pi_net, pi_times, pi_epochs = oracle.worker_tables(rates, 174, [10, 20, 50], [[0, 10], [10, 15], [20, 100]])
    assert np.allclose([p for _, p in net], pi_net, rtol=1e-9, atol=1e-300)
    disc = dict(conn.execute("select time, pi from discrete").fetchall())
    assert set(disc) == {10, 20, 50} and all(abs(disc[t] - pi_times[t]) <= 1e-9 * abs(pi_times[t]) for t in disc)
    iv = {k: (p, e) for k, p, e in conn.execute("select interval, pi, error from interval")}
    assert set(iv) == {"0-10", "10-15", "20-100"}
    for k in iv:
        assert abs(iv[k][0] - pi_epochs[k]["sum(integral)"]) <= 1e-9 * pi_epochs[k]["sum(integral)"]
        assert abs(iv[k][1] - pi_epochs[k]["sum(error)"]) <= 2e-2 * pi_epochs[k]["sum(error)"]  # abserr ~1e-12 here is mostly rounding noise of (resk - resg)
    conn.close()


def test_cli_end_to_end_with_oracle_engine(golden_dir, tmp_path, oracle, capsys):
    """BASELINE config C1 plumbing: bundled locus + tree through the CLI; engine = CPU oracle stand-in."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_engine
    outdir, aln = _run_cli(golden_dir, tmp_path, engine_mod=oracle_engine)
    check_cli_outputs(outdir, oracle, golden_dir)
    # --site-rates re-analysis of the files just written (bin/tapir_compute.py:153-158): no cull, divides again
    from tapir_amd import cli
    rates_dir = tmp_path / "rates"
    rates_dir.mkdir()
    shutil.copy(os.path.join(outdir, "chr1_918.nex.rates"), rates_dir)
    out2 = tmp_path / "out2"
    out2.mkdir()
    cli.main([str(rates_dir), os.path.join(golden_dir, "Euteleost.tree"), "--output", str(out2), "--times", "10",
              "--intervals", "0-10", "--site-rates"], engine_mod=oracle_engine)
    conn = sqlite3.connect(os.path.join(str(out2), "phylogenetic-informativeness.sqlite"))
    assert conn.execute("select locus from loci").fetchall() == [("chr1_918.nex",)]  # one extension stripped
    doc = json.load(open(rates_dir / "chr1_918.nex.rates"))
    r = np.array([x["rate"] for x in doc["sites"]["rates"]]) / 100
    net10 = conn.execute("select pi from net where time=10").fetchone()[0]
    assert abs(net10 - np.nansum(oracle.get_townsend_pi(10, r))) <= 1e-9 * net10
    conn.close()


def test_cli_subset_map(golden_dir, tmp_path, oracle):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_engine
    m = tmp_path / "map.tsv"
    m.write_text("chr1_918.nex\t50\t150\n")
    outdir, _ = _run_cli(golden_dir, tmp_path, extra=["--subset-pi-map-file", str(m)], engine_mod=oracle_engine)
    doc = json.load(open(os.path.join(outdir, "chr1_918.nex.rates")))
    from tapir_amd import nexus
    _, st = nexus.read_states(os.path.join(golden_dir, "chr1_918.nex"))
    inf = ((st == 1) | (st == 2) | (st == 4) | (st == 8)).sum(axis=0) >= 3
    r = np.array([x["rate"] for x in doc["sites"]["corrected_rates"]])
    r[~inf] = np.nan
    r = r[50:150]
    conn = sqlite3.connect(os.path.join(outdir, "phylogenetic-informativeness.sqlite"))
    net20 = conn.execute("select pi from net where time=20").fetchone()[0]
    assert abs(net20 - np.nansum(oracle.get_townsend_pi(20, r))) <= 1e-9 * net20


def test_cli_rejects_bad_input(golden_dir, tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_engine
    with pytest.raises(IndexError):  # --times beyond the tree depth: numpy IndexError in the reference too
        _run_cli(golden_dir, tmp_path, extra=["--times", "500"], engine_mod=oracle_engine)


# ---- C ABI ------------------------------------------------------------------------------------------
def test_cabi_exports_every_declared_symbol():
    """libtphip.so loads without a GPU and exports exactly what include/tphip.h declares."""
    import __graft_entry__ as ge
    ge.build()
    from tapir_amd import engine
    header = open(os.path.join(ROOT, "include", "tphip.h")).read()
    declared = set(re.findall(r"\b(tphip_[a-z_0-9]+)\s*\(", header))
    lib = ctypes.CDLL(engine.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    bound = {s[0] for s in engine.SYMBOLS}
    assert bound == declared, (bound ^ declared)
    nm = subprocess.run(["nm", "-D", "--defined-only", engine.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (tphip_[a-z_0-9]+)$", nm, flags=re.M))
    assert exported == declared, (exported ^ declared)
    assert lib.tphip_version() == 110


def test_structurizer_switch_touches_only_the_kernels_it_is_meant_for():
    """`-mllvm -structurizecfg-skip-uniform-regions=true` (a default-off LLVM switch that once mis-merged two stores in a
    helper kernel) is confined to translation units of their own.  Every flagged unit is compiled to
    gfx950 assembly with and without it: the functions whose code changes must be the interpreter-style kernels the switch is
    there for, and every other function of those units must come out byte-identical -- so a helper kernel that moves into
    such a unit cannot be affected silently (tools/flag_containment.py)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import flag_containment as fc
    units = fc.flagged_units()
    assert sorted(units) == ["locus_grad2_launch.hip", "locus_value_launch.hip", "site_rate_launch.hip"]
    meant = {"site_rate_launch.hip": ("tphip::site_rate_kernel<", "tphip::eval_columns_kernel("),   # (the diagnostic twin of the
             "locus_value_launch.hip": ("tphip::locus_value_kernel<",),                                #  site-rate evaluation)
             "locus_grad2_launch.hip": ("tphip::locus_grad2_kernel<",)}
    for u in units:
        changed, every = fc.changed_kernels(u)
        assert changed, u                                  # the switch does something for the unit, or it should go
        for k in changed:
            assert any(m in k for m in meant[u]), (u, k)
        helpers = [k for k in every if not any(m in k for m in meant[u])]
        assert helpers and all(k not in changed for k in helpers), (u, helpers)


def test_one_hip_runtime_whatever_the_import_order():
    """engine.load() before `import torch` must not leave two HIP runtimes in the process (the cause of round 1's
    "No HIP GPUs are available": see engine._preload_torch_hip_runtime).  Needs no GPU: it reads the process' map."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tapir_amd import engine\n"
            "engine.load()\n"
            "import torch\n"
            "maps = open('/proc/self/maps').read()\n"
            "for stem in ('libamdhip64', 'libhsa-runtime64'):\n"
            "    libs = {l.split()[-1] for l in maps.splitlines() if stem in l}\n"
            "    assert len(libs) == 1, libs\n"
            "print('ok')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout, r.stderr[-3000:])


def test_round_like_hyphy_is_the_printf_round_trip():
    """Host twin of the device rounding (csrc/pi_kernels.hpp round_like_printf): float("%.4f" % x) on constructed
    near-ties, where numpy.round(x * 1e4) / 1e4 is wrong about half the time."""
    from tapir_amd import compute
    rng = np.random.default_rng(2)
    k = rng.integers(0, 10 ** 8, 50000)
    v = (k + 0.5) / 1e4
    v = np.concatenate([v, np.nextafter(v, np.inf), np.nextafter(v, -np.inf), rng.gamma(0.5, 0.02, 20000),
                        [0.03125, 0.09375, 0.00005, 1.00015, 0.0, 2.5e-5]])
    want = np.array([float("%.4f" % x) for x in v])
    assert np.array_equal(compute.round_like_hyphy(v, 4), want)
    assert (np.round(v * 1e4) / 1e4 != want).sum() > 10000
    assert np.array_equal(compute.round_like_hyphy(v.reshape(2, -1), 4), want.reshape(2, -1))
    assert np.array_equal(compute.round_like_hyphy([0.125, 0.375], 2), [0.12, 0.38])   # exact ties: to even, as printf


def test_absent_base_has_zero_frequency():
    """A gap-free locus over three bases gives pi = 0 for the fourth (HarvestFrequencies); the plan accepts it
    (GPU test test_locus_with_an_absent_base)."""
    from tapir_amd import nexus
    hist = np.zeros((1, 16), np.int64)
    hist[0, 1], hist[0, 2], hist[0, 4] = 10, 20, 30
    pi = nexus.base_frequencies_from_histogram(hist)
    assert pi[0, 3] == 0.0 and abs(pi[0].sum() - 1) < 1e-15


def test_engine_fails_loudly_without_gpu(chr1_918):
    """No silent CPU fallback: on a GPU-less machine plan creation raises with TPHIP_ERR_NO_DEVICE."""
    from tapir_amd import engine
    if engine.device_count() > 0:
        pytest.skip("GPU present")
    c = chr1_918
    with pytest.raises(engine.TphipError, match="no HIP device"):
        engine.Plan(5, c["parent"], c["blen"], c["leaf"], [0, 226], [c["pi"]], [c["exch"]], 174, [10], [[0, 10]])
    with pytest.raises(engine.TphipError, match="no HIP device"):
        engine.townsend_pi_dense([1.0], [0.1])


# ---- multi-rank: loci round-robin + one all-gather (gloo, world_size 2) -------------------------------
_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch
from tapir_amd import dist as tdist, synth
import oracle_engine
rank, world = tdist.init_process_group("gloo")
nloci, ncols, ntaxa = 7, 40, 6
d = synth.simulate(nloci, ncols, ntaxa, 99)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
mine = tdist.shard_loci(nloci, rank, world)
cols = np.concatenate([np.arange(l * ncols, (l + 1) * ncols) for l in mine])
off = np.arange(len(mine) + 1) * ncols
plan = oracle_engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off, d["pi"][mine], d["exch"][mine], pin["T"],
                          [10, 30], [[5, 15]], correction=pin["correction"])
local = torch.from_numpy(plan.run_fused(st[:, cols])["tables"])
full = tdist.gather_tables(local, nloci, rank, world)
np.save(os.path.join(%(out)r, "rank%%d.npy" %% rank), full.numpy())
torch.distributed.destroy_process_group()
'''


def test_sharded_tables_gloo_world2(tmp_path):
    import torch  # noqa: F401
    script = tmp_path / "w.py"
    script.write_text(_WORKER % {"root": ROOT, "out": str(tmp_path)})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29517", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    a, b = np.load(tmp_path / "rank0.npy"), np.load(tmp_path / "rank1.npy")
    assert np.array_equal(a, b) and a.shape[0] == 7
    # single-process answer over all loci, same engine
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_engine
    from tapir_amd import synth
    d = synth.simulate(7, 40, 6, 99)
    pin = synth.plan_inputs(d["root"], d["names"])
    plan = oracle_engine.Plan(6, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"], pin["T"],
                              [10, 30], [[5, 15]], correction=pin["correction"])
    full = plan.run_fused(d["states"].numpy())["tables"]
    assert np.array_equal(a, full)  # bit-identical: a locus' row does not depend on the sharding


def test_fast_rates_writer_is_byte_identical_to_json_dump():
    """pipeline.dumps_rates_json must equal json.dumps(format_rates_json(...), indent=4): that is the text
    tapir leaves on disk after parse_site_rates rewrote HyPhy's file (tapir/compute.py:43)."""
    from tapir_amd import pipeline
    rng = np.random.default_rng(0)
    for n in (0, 1, 7, 500):
        subst = rng.gamma(1, 1, n) * rng.choice([0, 1, 1e-5, 10, 1234.5], n)
        rate = rng.gamma(1, 0.3, n) * rng.choice([0, 1, 1e-6], n)
        ll = -rng.gamma(2, 3, n)
        ll[:n // 7] = -0.00001  # rounds to -0.0
        pi = rng.dirichlet([5] * 4)
        ex = [0.96, 1.0, 0.58, 0.36, 1.87, 0.51]
        site = np.arange(1, n + 1)
        a = json.dumps(pipeline.format_rates_json(pi, ex, site, subst, rate, ll, rate / 100), indent=4)
        assert pipeline.dumps_rates_json(pi, ex, site, subst, rate, ll, rate / 100) == a
        assert json.loads(a)["sites"]["freqs"]["A"] == float(pi[0])
    # values where a vectorised rounding could differ from "%.4f": signed zeros, decimal ties whose binary neighbour decides,
    # magnitudes beyond 2^52 / 10^4 (taken literally), denormals
    v = np.array([0.0, -0.0, 1e-5, -1e-5, 0.00005, 0.00015, 0.12345, 2.5e-5, 1234.56785, 1e15, 1e22, 123456789.123456, 5e-324,
                  0.99995, 2.00005, 1.00015])
    site = np.arange(1, len(v) + 1)
    a = json.dumps(pipeline.format_rates_json(pi, ex, site, v, v, -v, v / 100), indent=4)
    assert pipeline.dumps_rates_json(pi, ex, site, v, v, -v, v / 100) == a
    # sites not numbered 1..n do not fit the cached template of a locus of n sites
    for site in (np.arange(5, 5 + len(v)), np.arange(len(v), 0, -1), np.r_[1, np.arange(3, len(v) + 1), len(v)]):
        a = json.dumps(pipeline.format_rates_json(pi, ex, site, v, v, -v, v / 100), indent=4)
        assert pipeline.dumps_rates_json(pi, ex, site, v, v, -v, v / 100) == a
    # more distinct lengths than the template cache keeps
    for n in range(2, 80):
        x = rng.gamma(1, 1, n)
        a = json.dumps(pipeline.format_rates_json(pi, ex, np.arange(1, n + 1), x, x, -x, x / 7), indent=4)
        assert pipeline.dumps_rates_json(pi, ex, np.arange(1, n + 1), x, x, -x, x / 7) == a
    assert 0 < len(pipeline._TEMPLATES) <= 32


def test_pool_parses_alignments_straight_into_the_batch_array(tmp_path):
    """--multiprocessing: the workers write their parsed rows into the batch array through /dev/shm (HostPool.parse_into)
    instead of returning them through the pool's pipes; same array and offsets as the sequential reader, ragged loci and a
    permuted taxon order included, and files the direct route cannot size (no NCHAR in the header) fall back silently."""
    from tapir_amd import pipeline, synth
    d = synth.simulate(7, 60, 6, 5)
    st = d["states"].numpy()
    off = d["locus_offsets"].copy()
    off[3] -= 11                                  # ragged: locus 2 is shorter, locus 3 longer
    aln = tmp_path / "aln"
    aln.mkdir()
    synth.write_nexus_dir(str(aln), st, off, d["names"], d["root"])
    paths = sorted(str(aln / f) for f in os.listdir(aln) if f.endswith(".nex"))
    # one file lists its taxa in another order
    text = open(paths[1]).read().split("matrix\n")
    rows = text[1].split("\n")
    body, tail = rows[:6], rows[6:]
    open(paths[1], "w").write(text[0] + "matrix\n" + "\n".join(body[::-1] + tail))
    leaf = list(d["names"])
    ref_states, ref_off = pipeline.load_alignments(paths, leaf)
    with pipeline.HostPool(2) as pool:
        got_states, got_off = pipeline.load_alignments(paths, leaf, pool=pool)
        assert np.array_equal(got_states, ref_states) and np.array_equal(got_off, ref_off)
        # a header without NCHAR: the direct route declines, the plain one still reads the file
        t = open(paths[2]).read()
        import re
        open(paths[2], "w").write(re.sub(r"nchar=\d+", "", t))
        assert pool.parse_into(paths, leaf, np.empty) is None
        got_states, got_off = pipeline.load_alignments(paths, leaf, pool=pool)
        assert np.array_equal(got_states, ref_states) and np.array_equal(got_off, ref_off)
        # taxa that do not match the tree: the plain route words the error
        with pytest.raises(pipeline.PipelineError, match="do not match the tree"):
            pipeline.load_alignments(paths, leaf[:-1] + ["nobody"], pool=pool)


def test_shared_memory_files_need_room(tmp_path, monkeypatch):
    """A tmpfs that fills up under a memory map kills the process with SIGBUS (containers often give /dev/shm 64 MB), so
    /dev/shm is used only when it has room: the direct parsing route declines, the writers' hand-over file goes to the
    ordinary temp directory."""
    import collections
    import shutil
    from tapir_amd import pipeline, synth
    usage = collections.namedtuple("usage", "total used free")
    monkeypatch.setattr(shutil, "disk_usage", lambda d: usage(1 << 26, (1 << 26) - 4096, 4096))
    assert pipeline._shared_dir(1 << 20) is None
    if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK):
        assert pipeline._shared_dir() == "/dev/shm"          # no size asked: only presence
    d = synth.simulate(3, 40, 5, 9)
    aln = tmp_path / "aln"
    aln.mkdir()
    synth.write_nexus_dir(str(aln), d["states"].numpy(), d["locus_offsets"], d["names"], d["root"])
    paths = sorted(str(aln / f) for f in os.listdir(aln) if f.endswith(".nex"))
    with pipeline.HostPool(2) as pool:
        assert pool.parse_into(paths, list(d["names"]), np.empty) is None
        states, off = pipeline.load_alignments(paths, list(d["names"]), pool=pool)     # the plain route still works
    assert np.array_equal(states, d["states"].numpy()) and np.array_equal(off, d["locus_offsets"])


def test_cli_multiprocessing_flag_gives_identical_files(golden_dir, tmp_path, oracle):
    """--multiprocessing (reference: Pool(cpu_count()-1), bin/tapir_compute.py:159-164) parallelises only the
    host side; every output file must be identical to the sequential run.  With the pool the sqlite inserts run on a
    second thread while the workers write the `.rates` files (pipeline.run_alignments(during_write=...)): same rows in
    all four tables, and what the side thread raises is raised by the call."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_engine
    from tapir_amd import cli, synth
    d = synth.simulate(6, 40, 5, 3)
    aln = tmp_path / "aln"
    aln.mkdir()
    tree = synth.write_nexus_dir(str(aln), d["states"].numpy(), d["locus_offsets"], d["names"], d["root"])
    shutil.move(tree, tmp_path / "tree.newick")
    outs = []
    for flag in ([], ["--multiprocessing"]):
        out = tmp_path / ("out" + str(len(outs)))
        out.mkdir()
        cli.main([str(aln), str(tmp_path / "tree.newick"), "--output", str(out), "--times", "10,30", "--intervals", "5-15",
                  "--exchangeabilities", "1,1.2,0.8,0.9,1.5,1"] + flag, engine_mod=oracle_engine)
        outs.append(out)
        assert ("during_write" in cli.LAST_TIMINGS) == bool(flag) and cli.LAST_TIMINGS["sqlite"] > 0
    names = sorted(os.listdir(outs[0]))
    assert names == sorted(os.listdir(outs[1])) and len(names) == 8
    for n in names:
        if n.endswith(".rates") or n.endswith(".newick"):
            assert open(outs[0] / n).read() == open(outs[1] / n).read(), n
    q = "select l.locus, n.time, n.pi from loci l join net n on n.id = l.id order by 1, 2"
    a = sqlite3.connect(outs[0] / "phylogenetic-informativeness.sqlite").execute(q).fetchall()
    b = sqlite3.connect(outs[1] / "phylogenetic-informativeness.sqlite").execute(q).fetchall()
    assert a == b and len(a) == 6 * 100
    for q, n in (("select * from loci order by id", 6), ("select * from discrete order by id, time", 12),
                 ("select * from interval order by id, interval", 6)):
        a = sqlite3.connect(outs[0] / "phylogenetic-informativeness.sqlite").execute(q).fetchall()
        b = sqlite3.connect(outs[1] / "phylogenetic-informativeness.sqlite").execute(q).fetchall()
        assert a == b and len(a) == n, q
    # the side thread's failure surfaces in the caller, after the workers are done with the shared file
    from tapir_amd import newick, pipeline
    root = newick.read_tree(str(tmp_path / "tree.newick"), "newick")
    leaf_names = [x.name for x in newick.leaves(root)]
    parent, blen, leaf = newick.to_arrays(root, leaf_names)
    files = sorted(str(aln / f) for f in os.listdir(aln) if f.endswith(".nex"))
    out = tmp_path / "out_fail"
    out.mkdir()

    def fail(tuples):
        assert len(tuples) == 6
        raise RuntimeError("disk full")

    with pipeline.HostPool(2) as pool:
        with pytest.raises(RuntimeError, match="disk full"):
            pipeline.run_alignments(files, leaf_names, parent, blen, leaf, 100, [10], [(5, 15)], 1.0, 3,
                                    np.ones(6), output_dir=str(out), engine_mod=oracle_engine, pool=pool, during_write=fail)
    assert len(os.listdir(out)) == 6
    if os.path.isdir("/dev/shm"):
        assert not [f for f in os.listdir("/dev/shm") if f.startswith("tapir_amd_")]


def test_stage1_model_enumeration():
    """The 203 rate-class models of models_and_rates.bf:544-566: the script's loop order, every set partition of
    the six rates exactly once, and the number of free rates per model (class of AG is fixed at 1)."""
    from oracle import stage1_oracle
    from tapir_amd import stage1
    ms = stage1.model_strings()
    assert ms[0] == "012345" and ms[1] == "000000" and len(ms) == 203
    assert sorted(ms) == sorted(stage1_oracle.partitions6())
    cls, k = stage1.model_design(ms)
    assert k[0] == 5 and k[1] == 0 and k.max() == 5 and (k[1:] <= 4).all()
    i = ms.index("010010")  # HKY85: transitions (AG, CT) vs transversions
    assert k[i] == 1 and list(cls[i]) == [0, -1, 0, 0, -1, 0]
    pi = np.array([0.1, 0.2, 0.3, 0.4])
    assert abs(stage1.total_factor(pi, np.ones(6)) - (1 - (pi ** 2).sum())) < 1e-15


def test_stage1_optimiser_against_independent_restatement(tmp_path):
    """Host logic of stage 1 (batched L-BFGS, finite-difference stencils, Akaike averaging) with the oracle's
    likelihood behind the engine interface, against oracle/stage1_oracle.py (scipy L-BFGS-B, one model at a
    time).  Parity unpinned against HyPhy itself: the reference holds no stage-1 output."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_engine
    from oracle import stage1_oracle
    from tapir_amd import stage1, synth
    d = synth.simulate(2, 120, 5, 11)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()
    pi = np.asarray(d["pi"])
    blen = np.asarray(pin["blen"]) / pin["correction"]
    plan = oracle_engine.Plan(5, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], pi, np.ones((2, 6)), pin["T"],
                              [1], [[0, 1]])
    got = stage1.model_averaged_exchangeabilities(plan, st, pi, pin["parent"], blen)
    assert got["exch"].shape == (2, 6) and np.all(got["exch"][:, 1] == 1.0)
    assert np.allclose(got["weights"].sum(1), 1.0)
    for l in range(2):
        ref = stage1_oracle.model_averaged(st[:, l * 120:(l + 1) * 120], pin["parent"], blen, pin["leaf"], pi[l])
        assert np.max(np.abs(got["exch"][l] - ref["exch"]) / ref["exch"]) < 1e-3
        lnl = np.array([ref["lnl"][m] for m in got["models"]])
        w = np.array([ref["weights"][m] for m in got["models"]])
        keep = w > 1e-9   # models abandoned early (weight < e^-30) are not polished to their optimum
        assert np.max(np.abs(lnl - got["lnl"][l])[keep]) < 1e-3
        assert np.all(got["lnl"][l] <= lnl + 1e-3)
        assert np.max(np.abs(w - got["weights"][l])) < 1e-4


def test_cli_default_runs_model_averaging(tmp_path):
    """Without --exchangeabilities/--subs-model the command line does what the reference's HyPhy script does:
    estimates the exchangeabilities per locus (stage 1) and reports them in every .rates file."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_engine
    from tapir_amd import cli, nexus, stage1, synth, newick, compute
    d = synth.simulate(2, 60, 4, 5)
    aln = tmp_path / "aln"
    aln.mkdir()
    tree = synth.write_nexus_dir(str(aln), d["states"].numpy(), d["locus_offsets"], d["names"], d["root"])
    shutil.move(tree, tmp_path / "tree.newick")
    out = tmp_path / "out"
    out.mkdir()
    cli.main([str(aln), str(tmp_path / "tree.newick"), "--output", str(out), "--times", "10", "--intervals", "5-15"],
             engine_mod=oracle_engine)
    files = sorted(f for f in os.listdir(out) if f.endswith(".rates"))
    assert len(files) == 2
    mats = [json.load(open(out / f))["sites"]["subs_matrix"] for f in files]
    for m in mats:
        assert m["AG"] == 1.0 and all(m[k] > 0 for k in m)
    assert mats[0] != mats[1]  # per-locus estimates, not a shared constant


_CLI_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import oracle_engine
from tapir_amd import cli
cli.main(%(argv)r, engine_mod=oracle_engine)
'''


def test_cli_two_ranks_gloo_matches_single_process(tmp_path):
    """`torchrun --nproc-per-node 2 tapir_compute.py ...`: files dealt round-robin, each rank writes its own .rates
    files into the one output directory, one all-gather of the PI rows, rank 0 writes sqlite in file order.
    Everything on disk must equal the single-process run (5 files on 2 ranks: ragged shards)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_engine
    from tapir_amd import cli, synth
    d = synth.simulate(5, 40, 5, 8)
    aln = tmp_path / "aln"
    aln.mkdir()
    tree = synth.write_nexus_dir(str(aln), d["states"].numpy(), d["locus_offsets"], d["names"], d["root"])
    shutil.move(tree, tmp_path / "tree.newick")
    outs = []
    for name in ("single", "multi"):
        out = tmp_path / name
        out.mkdir()
        outs.append(out)
    argv = [str(aln), str(tmp_path / "tree.newick"), "--times", "10,30", "--intervals", "5-15,20-40",
            "--exchangeabilities", "1,1.2,0.8,0.9,1.5,1", "--output"]
    cli.main(argv + [str(outs[0])], engine_mod=oracle_engine)
    script = tmp_path / "w.py"
    script.write_text(_CLI_WORKER % {"root": ROOT, "argv": argv + [str(outs[1])]})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29531", str(script)],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    names = sorted(os.listdir(outs[0]))
    assert names == sorted(os.listdir(outs[1])) and len(names) == 7    # 5 .rates + tree + sqlite, ONE directory
    for n in names:
        if n.endswith(".rates") or n.endswith(".newick"):
            assert open(outs[0] / n).read() == open(outs[1] / n).read(), n
    for q in ("select l.locus, n.time, n.pi from loci l join net n on n.id = l.id order by l.id, n.time",
              "select l.locus, n.time, n.pi from loci l join discrete n on n.id = l.id order by l.id, n.time",
              "select l.locus, n.interval, n.pi, n.error from loci l join interval n on n.id = l.id order by l.id, n.interval",
              "select sql from sqlite_master order by name"):
        a = sqlite3.connect(outs[0] / "phylogenetic-informativeness.sqlite").execute(q).fetchall()
        b = sqlite3.connect(outs[1] / "phylogenetic-informativeness.sqlite").execute(q).fetchall()
        assert a == b and len(a) > 0


def test_discrete_gamma_categories():
    """Yang's discrete gamma for the opt-in rate mixture: mean 1, ordered, K = 1 is the identity, large alpha
    collapses every category onto 1; known values for alpha = 0.5, K = 4 (Yang 1994, table 1 to 4 decimals)."""
    from tapir_amd import compute
    r, w = compute.discrete_gamma(0.5, 4)
    assert np.allclose(w, 0.25) and abs((r * w).sum() - 1.0) < 1e-12 and np.all(np.diff(r) > 0)
    assert np.allclose(r, [0.0334, 0.2519, 0.8203, 2.8944], atol=5e-5)
    assert np.allclose(compute.discrete_gamma(0.5, 1)[0], [1.0])
    assert np.allclose(compute.discrete_gamma(1e6, 4)[0], 1.0, atol=3e-3)


def test_rate_mixture_in_the_oracle_and_cli(tmp_path):
    """The mixture objective: categories that all equal 1 reproduce the plain model; a real discrete gamma changes
    the optimum; and the command line's --gamma-categories reaches the engine (oracle stand-in here)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_engine
    from oracle import oracle as orc
    from tapir_amd import cli, compute, synth
    d = synth.simulate(2, 60, 6, 4)
    pin = synth.plan_inputs(d["root"], d["names"])
    st = d["states"].numpy()[:, :60]
    a = orc.site_rates(st, pin["parent"], pin["blen"], pin["leaf"], d["pi"][0], d["exch"][0], start_mode=0)   # the mixture's start
    b = orc.site_rates(st, pin["parent"], pin["blen"], pin["leaf"], d["pi"][0], d["exch"][0], [1.0, 1.0, 1.0], [0.2, 0.3, 0.5])
    ok = a["flag"] == 0
    assert np.array_equal(a["flag"], b["flag"]) and np.abs(a["rate"] - b["rate"])[ok].max() < 1e-12
    assert np.abs(a["lnl"] - b["lnl"]).max() < 1e-12
    r, w = compute.discrete_gamma(0.5, 4)
    c = orc.site_rates(st, pin["parent"], pin["blen"], pin["leaf"], d["pi"][0], d["exch"][0], r, w)
    both = ok & (c["flag"] == 0)
    assert both.sum() > 5 and np.abs(c["rate"] - a["rate"])[both].max() > 1e-3
    aln = tmp_path / "aln"
    aln.mkdir()
    tree = synth.write_nexus_dir(str(aln), d["states"].numpy(), d["locus_offsets"], d["names"], d["root"])
    shutil.move(tree, tmp_path / "tree.newick")
    outs = []
    for extra in ([], ["--gamma-categories", "4", "--gamma-alpha", "0.5"]):
        out = tmp_path / ("o%d" % len(outs))
        out.mkdir()
        cli.main([str(aln), str(tmp_path / "tree.newick"), "--output", str(out), "--times", "10", "--intervals", "5-15",
                  "--exchangeabilities", "1,1,1,1,1,1"] + extra, engine_mod=oracle_engine)
        f = sorted(x for x in os.listdir(out) if x.endswith(".rates"))[0]
        outs.append(json.load(open(out / f))["sites"]["rates"])
    assert any(abs(x["rate"] - y["rate"]) > 1e-3 for x, y in zip(*outs))


def test_stage1_blocks_of_loci_give_the_same_rates():
    """pipeline.model_averaged_exchangeabilities fits loci in blocks (bounded candidate batches); loci are
    independent, so the block size must not change any locus' estimate beyond the optimiser's own stopping tolerance
    (the batched L-BFGS shares its iteration counter between the problems of a block: when one of them is moved by the
    boundary escape the others skip that iteration, so the stopping points can differ in the sixth digit; the stage's
    parity tolerance is 1e-3)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_engine
    from tapir_amd import pipeline, synth
    d = synth.simulate(3, 50, 4, 12)
    pin = synth.plan_inputs(d["root"], d["names"])
    args = (oracle_engine, d["states"].numpy(), d["locus_offsets"], d["pi"], 4, pin["parent"], pin["blen"], pin["leaf"], pin["T"],
            [1], [[0, 1]], pin["correction"])
    a = pipeline.model_averaged_exchangeabilities(*args)
    b = pipeline.model_averaged_exchangeabilities(*args, block_loci=2)
    assert a.shape == (3, 6) and np.max(np.abs(a - b) / a) < 2e-5


def test_nexus_sequential_wrapping_and_matchchar(tmp_path):
    """NEXUS variants DendroPy reads and HyPhy's ReadDataFile accepts: a sequential matrix whose sequences wrap over
    several lines (label on its own line or not), and MATCHCHAR."""
    from tapir_amd import nexus
    p = tmp_path / "w.nex"
    p.write_text("#NEXUS\nBEGIN DATA;\nDIMENSIONS NTAX=2 NCHAR=8;\nFORMAT DATATYPE=DNA;\nMATRIX\n'tax a'\n ACG T\n ACGT\nb\nACGTAC\nGT\n;\nEND;\n")
    names, st = nexus.read_states(str(p))
    assert names == ["tax a", "b"] and st.tolist() == [[1, 2, 4, 8, 1, 2, 4, 8]] * 2
    p.write_text("#NEXUS\nBEGIN DATA;\nDIMENSIONS NTAX=2 NCHAR=4;\nFORMAT DATATYPE=DNA MATCHCHAR=.;\nMATRIX\na ACGT\nb ..C.\n;\nEND;\n")
    names, st = nexus.read_states(str(p))
    assert st.tolist() == [[1, 2, 4, 8], [1, 2, 2, 8]]
    p.write_text("#NEXUS\nBEGIN DATA;\nDIMENSIONS NTAX=2 NCHAR=4;\nFORMAT DATATYPE=DNA;\nMATRIX\na ACGT\nb ACG\n;\nEND;\n")
    with pytest.raises(nexus.NexusError):
        nexus.read_states(str(p))


def test_fused_tree_program_expands_to_the_plain_one(tmp_path):
    """tree_program.hpp (host side, compiled with g++): the fused op stream site_rate_kernel interprets -- CHERRY for
    TIP_SET + TIP_MUL on equally long branches, PUSH / POP_MUL riding as flags on their neighbours -- must expand to
    exactly the plain stream classify_kernel and the likelihood kernels read."""
    import subprocess
    from tapir_amd import synth
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "tree_program_dump")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I" + os.path.join(root, "tapir_amd", "csrc"),
                           os.path.join(root, "tests", "native", "tree_program_dump.cpp"), "-o", exe])
    TIP_SET, TIP_MUL, BRANCH, PUSH, POP_MUL, CHERRY = range(6)
    PUSH_BEFORE, POP_AFTER = 0x100, 0x200
    for ntaxa, seed, perturb in [(64, synth.WORKLOAD_SEED["C3"], False), (16, synth.WORKLOAD_SEED["C2"], False), (37, 5, True)]:
        tree_root, names = synth.yule_tree(ntaxa, seed)
        pin = synth.plan_inputs(tree_root, names)
        parent, blen, leaf = np.asarray(pin["parent"]), np.asarray(pin["blen"], dtype=np.float64).copy(), np.asarray(pin["leaf"])
        if perturb:   # a non-ultrametric tree: most cherries lose their equal branch lengths
            blen *= 1.0 + 0.3 * np.random.default_rng(1).random(len(blen))
        text = "%d %d\n" % (ntaxa, len(parent)) + "".join("%d %.17g %d\n" % (parent[i], blen[i], leaf[i]) for i in range(len(parent)))
        out = subprocess.run([exe], input=text, capture_output=True, text=True, check=True).stdout.split("\n")
        plain = [(int(a), int(b), float(c)) for tag, a, b, c in (l.split() for l in out if l.startswith("plain"))]
        fused = [(int(a), int(b), float(c)) for tag, a, b, c in (l.split() for l in out if l.startswith("fused"))]
        tips = [op for op in plain if op[0] <= TIP_MUL]
        expanded, k = [], 0
        for code, taxon, t in fused:
            base, flags = code & 0xff, code & ~0xff
            if flags & PUSH_BEFORE:
                assert base in (TIP_SET, CHERRY)
                expanded.append((PUSH, 0, 0.0))
            if base == CHERRY:
                assert tips[k][0] == TIP_SET and tips[k + 1][0] == TIP_MUL and tips[k][2] == tips[k + 1][2] == t
                expanded += [tips[k], tips[k + 1]]
                k += 2
            else:
                expanded.append((base, taxon, t))
                k += base <= TIP_MUL
            if flags & POP_AFTER:
                assert base == BRANCH
                expanded.append((POP_MUL, 0, 0.0))
        assert expanded == plain
        assert not any(code == PUSH or code == POP_MUL for code, _, _ in fused)   # every one of them found a neighbour
        ncherry = sum(1 for code, _, _ in fused if code & 0xff == CHERRY)
        if not perturb:
            assert ncherry == sum(1 for i in range(len(plain) - 1)
                                  if plain[i][0] == TIP_SET and plain[i + 1][0] == TIP_MUL and plain[i][2] == plain[i + 1][2]) > 0
            if ntaxa == 64:
                assert ncherry == 21 and len(fused) == 105
