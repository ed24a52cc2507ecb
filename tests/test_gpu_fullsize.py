"""Full-size GPU checks: EVERY BASELINE.json config at its real size on one MI355X (C4's 50 000 loci = 3.2 GB of
states and C5's 10 000 loci x 256 taxa = 5.1 GB both fit one 288-GB GPU), through size-independent properties, plus
a sampled comparison with the oracle.  The oracle cannot process 5 x 10^7 columns in test time, so at full size:
  * first-order optimality: at EVERY reported interior maximiser the Newton distance |f'/f''| (evaluated by the
    independent eval_columns diagnostic entry, byte path) is below 1e-6 and the curvature is negative;
  * constant columns give exactly rate 0 and lnL = ln(pi_x); the informative count equals a torch count;
  * determinism: two launches give bit-identical outputs;
  * sharding invariance: a locus' PI row and its per-site outputs are bit-identical whether the batch holds all
    loci or half of them, and net PI is additive over a split of a locus' columns (to rounding);
  * three random 400-column slices agree with the CPU oracle to the usual tolerances.
All property checks run on the device (nothing but the oracle samples crosses PCIe), so the whole file takes minutes.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(workload):
    import torch
    from tapir_amd import engine, synth
    if engine.device_count() < 1:
        pytest.fail("no GPU visible")
    nloci, ncols, ntaxa, times, intervals = synth.WORKLOADS[workload]
    seed = synth.WORKLOAD_SEED[workload]
    tree = synth.yule_tree(ntaxa, seed)
    d = synth.simulate(nloci, ncols, ntaxa, seed, device="cuda", tree=tree)
    pin = synth.plan_inputs(d["root"], d["names"])
    return torch, engine, d, pin, nloci, ncols, ntaxa, times, intervals


def _run(torch, plan, d_states, nloci):
    n, W = plan.ncols, plan.width
    dev = d_states.device
    out = dict(rate=torch.empty(n, dtype=torch.float64, device=dev), subst=torch.empty(n, dtype=torch.float64, device=dev),
               lnl=torch.empty(n, dtype=torch.float64, device=dev), flag=torch.empty(n, dtype=torch.uint8, device=dev),
               nres=torch.empty(n, dtype=torch.int32, device=dev), tables=torch.empty((nloci, W), dtype=torch.float64, device=dev))
    ws = torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=dev)
    plan.run_dev(d_states, out["rate"], out["subst"], out["lnl"], out["flag"], out["nres"], out["tables"], ws,
                 torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("workload", ["C2", "C3", "C4", "C5"])
def test_fullsize_properties(workload, oracle):
    torch, engine, d, pin, nloci, ncols, ntaxa, times, intervals = _setup(workload)
    plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"], pin["T"],
                       times, intervals, correction=pin["correction"], threshold=3, round_decimals=4)
    assert plan.ncols == nloci * ncols
    st = d["states"]
    dev = st.device
    a = _run(torch, plan, st, nloci)
    b = _run(torch, plan, st, nloci)
    for k in a:  # determinism (NaN-free outputs: culled columns are NaN only inside the PI stage)
        assert torch.equal(a[k], b[k]), k
    del b
    flag, rate, lnl, nres = a["flag"], a["rate"], a["lnl"], a["nres"]
    counts = torch.bincount(flag.to(torch.int64), minlength=5).cpu().numpy()
    assert counts[1] == 0 and counts[4] == 0, counts          # no flat columns in these shapes, no iteration limit
    # informative count and constant-column answers against torch counts
    single = (st == 1) | (st == 2) | (st == 4) | (st == 8)
    assert torch.equal(nres, single.sum(dim=0, dtype=torch.int32))
    zero = flag == 3
    assert bool((rate[zero] == 0.0).all())
    kappa = torch.from_numpy(plan.models()[3]).to(dev)
    loc = torch.arange(nloci, device=dev).repeat_interleave(ncols)
    first_res = torch.where(single, st, torch.zeros_like(st)).max(dim=0).values   # the one base of a constant column
    del single
    xz = torch.log2(first_res[zero].to(torch.float64)).to(torch.int64)
    pin_t = torch.from_numpy(d["pi"] / d["pi"].sum(1, keepdims=True)).to(dev)
    assert torch.allclose(lnl[zero], torch.log(pin_t[loc[zero], xz]), rtol=0, atol=1e-14)
    del first_res, xz
    # first-order optimality at EVERY interior maximiser, via the diagnostic evaluation entry point (device pointers)
    ok = flag == 0
    u = torch.zeros(plan.ncols, dtype=torch.float64, device=dev)
    u[ok] = torch.log(rate[ok] / kappa[loc[ok]])
    f, g, h = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
    plan.eval_columns_dev(st, u, f, g, h, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert bool((h[ok] < 0).all())
    # distance to the stationary point, first order: |g / h|.  The optimiser's two exits (a step below 2e-3 predicted by the
    # two-point quartic, a Newton step below 3e-4 with its third-order correction) leave ~2e-7 at worst
    resid = (g[ok] / h[ok]).abs().max().item()
    assert resid < 1e-6, resid
    assert (f[ok] - lnl[ok]).abs().max().item() < 1e-9 * lnl[ok].abs().max().item()
    del u, f, g, h
    # a random sample of loci slices against the oracle
    rng = np.random.default_rng(1)
    for l in rng.choice(nloci, 3, replace=False):
        c0 = int(rng.integers(0, max(1, ncols - 400)))
        sl = slice(int(l) * ncols + c0, int(l) * ncols + min(ncols, c0 + 400))
        ref = oracle.site_rates(st[:, sl].cpu().numpy(), pin["parent"], pin["blen"], pin["leaf"], d["pi"][l], d["exch"][l])
        assert np.array_equal(flag[sl].cpu().numpy(), ref["flag"])
        okk = (ref["flag"] == 0) | (ref["flag"] == 3)
        assert np.allclose(rate[sl].cpu().numpy()[okk], ref["rate"][okk], rtol=1e-6, atol=0)
        assert np.allclose(lnl[sl].cpu().numpy(), ref["lnl"], rtol=0, atol=1e-9)
    # sharding invariance: second half of the loci as its own batch -> bit-identical PI rows and per-site outputs
    h0 = nloci // 2
    c_half = int(d["locus_offsets"][h0])
    off2 = d["locus_offsets"][h0:] - d["locus_offsets"][h0]
    plan2 = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off2, d["pi"][h0:], d["exch"][h0:], pin["T"],
                        times, intervals, correction=pin["correction"], threshold=3, round_decimals=4)
    c = _run(torch, plan2, st[:, c_half:].contiguous(), nloci - h0)
    assert torch.equal(c["tables"], a["tables"][h0:])
    assert torch.equal(c["rate"], rate[c_half:])
    assert torch.equal(c["lnl"], lnl[c_half:])
    del c
    plan2.close()
    # tables are finite and net PI at t=0 is exactly 0
    assert bool(torch.isfinite(a["tables"]).all()) and bool((a["tables"][:, 0] == 0).all())
    # additivity of net PI over a split of one locus' columns (PI stage alone, no cull, no rounding)
    r0 = rate[:ncols].cpu().numpy() / pin["correction"]
    T = pin["T"]
    whole = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], [0, ncols], [d["pi"][0]], [d["exch"][0]], T, times,
                        intervals, round_decimals=-1, threshold=0).pi_tables(r0)[0]
    cut = ncols // 3
    parts = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], [0, cut, ncols], d["pi"][:2], d["exch"][:2], T, times,
                        intervals, round_decimals=-1, threshold=0).pi_tables(r0)
    assert np.allclose(parts.sum(axis=0), whole, rtol=1e-12, atol=0)
    plan.close()
    del a, st, d
    torch.cuda.empty_cache()


def test_deep_tree_spilled_stack_is_bit_identical(monkeypatch):
    """256 taxa: the persistent site-rate kernel keeps three parked partials in LDS and the fourth in a global scratch
    row per wave (8 resident waves per CU instead of 6; tphip.hip plan creation).  Same bits as with the whole stack in
    LDS, on an eighth of C5 (2.5 M columns: the persistent grid, so the spill variant is the one that runs)."""
    import torch
    from tapir_amd import engine, synth
    nloci, ncols, ntaxa, times, intervals = synth.WORKLOADS["C5"]
    nloci //= 8
    seed = synth.WORKLOAD_SEED["C5"]
    d = synth.simulate(nloci, ncols, ntaxa, seed, device="cuda", tree=synth.yule_tree(ntaxa, seed))
    pin = synth.plan_inputs(d["root"], d["names"])
    outs = {}
    for spill in ("0", "1"):
        monkeypatch.setenv("TPHIP_SITE_SPILL", spill)
        plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"], pin["T"],
                           times, intervals, correction=pin["correction"])
        assert plan.stack_depth == 4
        outs[spill] = (_run(torch, plan, d["states"], nloci), plan.workspace_bytes)
        plan.close()
    assert outs["1"][1] > outs["0"][1]          # the scratch rows are part of the workspace: the variant did run
    for k in outs["0"][0]:
        assert torch.equal(outs["0"][0][k], outs["1"][0][k]), k


_PORT = [29630]


def _bench(argv, nproc=None, timeout=900):
    _PORT[0] += 1   # a fresh rendezvous port per launch
    env = dict(os.environ)   # (bench.py sets HSA_ENABLE_IPC_MODE_LEGACY=0 itself: dmabuf IPC for RCCL on this pool)
    if nproc is None:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + argv
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
               "--master-addr", "127.0.0.1", "--master-port", str(_PORT[0]), os.path.join(ROOT, "bench.py")] + argv
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["C4", "C5"])
def test_bench_one_rank_torchrun_full_config(workload):
    """The 8-GPU configs, whole, through bench.py under a 1-rank torchrun: the RCCL process group and the all-gather
    run exactly as in the 8-rank launch (one rank), the timed region covers the full 5 x 10^7 / 2 x 10^7 columns."""
    from tapir_amd import synth
    out = _bench(["--gpus", "1", "--workload", workload, "--steps", "2", "--warmup", "1", "--cpu-seconds", "0",
                  "--stage1-loci", "0"], nproc=1)
    nloci, ncols = synth.WORKLOADS[workload][:2]
    assert out["n_gpus"] == 1 and out["config"]["total_columns"] == nloci * ncols == out["config"]["columns_per_gpu"]
    assert out["value"] > 1e7 and out["flags"]["maxit"] == 0 and len(out["table_sha256"]) == 64


def test_bench_two_rank_rehearsal_matches_single_gpu_table():
    """bench.py's N > 1 path (same seeded batch on every rank, loci dealt round-robin, gather, un-permute, rank 0's
    single-GPU run of the whole config) rehearsed with two ranks sharing this box's one GPU (gloo collective through
    the host; RCCL itself is covered by the one-rank launches above).  The run fails unless the gathered table is
    bit-identical to the single-GPU table; the hash must also equal a plain one-process run's."""
    common = ["--workload", "C4", "--loci", "3001", "--steps", "1", "--warmup", "1", "--cpu-seconds", "0", "--stage1-loci", "0"]
    two = _bench(["--gpus", "2", "--rehearse-on-one-gpu"] + common, nproc=2)
    one = _bench(["--gpus", "1"] + common)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and two["gathered_equals_single_gpu"] is True
    assert two["config"]["loci_per_gpu"] == 1501 and two["config"]["total_columns"] == 3001 * 1000
    assert two["table_sha256"] == one["table_sha256"]
