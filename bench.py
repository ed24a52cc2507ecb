#!/usr/bin/env python3
"""bench.py -- alignment columns/s through site-rate ML + PI tables on MI355X.

A "step" is one pass of the hot path (classify -> compact -> per-site rate ML -> PI tables, plus the
all-gather of PI tables when N > 1) over one batch of synthetic loci already resident in HBM.

  python bench.py                       one GPU, workload C3 (100 loci x 50 000 columns x 64 taxa): the headline
  python bench.py --workload C4         one GPU, the WHOLE 8-GPU config C4 (50 000 loci x 1 000 x 64; 3.2 GB: it fits)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W
      N > 1: STRONG scaling of one fixed config (default C4): every rank generates the same seeded batch, keeps the
      loci i with i mod N == rank (what Pool.map(worker, params) did over cores, bin/tapir_compute.py:159-164), runs
      its share, and one all-gather (RCCL) collects the [L, W] PI table; after the timed region rank 0 runs the whole
      config alone and requires the gathered table to be bit-identical to its own.

Prints ONE JSON line on rank 0 (contract in the round prompt) with the extra objects `roofline` (dominant
kernel = site_rate_kernel; algorithmic bytes = (ntaxa + 24) per column), `fp64` (the bound
that actually binds this path) and `cpu_baseline` (the CPU oracle timed on the box's host cores on a
bounded sample of the same workload, one core and all usable cores; a reported baseline, not the target),
whose `parity_sample` is the GPU-vs-oracle gate SURVEY 8d attaches to every measurement: the run fails on a mismatch.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# This pool's host driver supports dmabuf IPC only: without the variable RCCL's handle exchange between the ranks of one
# node fails with `hipIpcGetMemHandle: invalid argument`.  It is exported on the boxes already; set here (before torch
# loads the runtime) so that a launch from a bare environment behaves like every rehearsal did.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TFLOPS = 78.6  # MI355X FP64 vector peak


def flops_per_eval(oc):
    """FP64 flop count of one likelihood evaluation (value + 1st + 2nd u-derivative), from the op mix of the
    compiled tree program and the arithmetic each op issues in site_rate_kernel.hpp (FMA = 2 flop):
      exp_nonpos_tab                  19  (max, fma, add, 2 fma, 4 fma + mul, fma)
      tip message                    137  (3 exp + 15 + 64 + 1)
      internal branch                244  (84 for U^-1 v/v'/v'', 3 exp + 30, 72 for U, 1)
      product rule (TIP_MUL/POP_MUL)  40  (TIP_SET writes its message into the accumulator: no product)
      fused cherry                   -57  (the second tip reuses the first one's 3 exponentials)
      root: L, L', L'' + log, div   ~ 50"""
    return (oc["tip_set"] * 137 + oc["tip_mul"] * (137 + 40) + oc["pop_mul"] * 40 + oc["branch"] * 244 + 50
            - oc.get("cherry", 0) * 57)


def usable_cpus():
    """Cores this process may really use: the affinity mask, cut by the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


# ---- CPU oracle legs (bench.py's cpu_baseline is one of the three places allowed to call oracle/) ----------------

def _oracle_job(job):
    """One locus sample through the oracle: site-rate ML + rounding / correction / cull + net PI + dqagse integrals,
    i.e. the same stages the GPU step runs.  Returns (seconds, rate, flag, nres, net, integrals, errors)."""
    st, parent, blen, leaf, pi, exch, correction, T, intervals = job
    from oracle import oracle as orc
    orc.lib()
    t0 = time.perf_counter()
    r = orc.site_rates(st, parent, blen, leaf, pi, exch)
    rates = orc.round_dp(r["rate"], 4) / correction
    rates[r["nres"] < 3] = np.nan
    net = orc.net_pi(rates, T)
    si, se = orc.net_integrals(rates, intervals, 0)
    return time.perf_counter() - t0, r["rate"], r["flag"], r["nres"], net, si, se


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=["C2", "C3", "C4", "C5", "R1"],
                    help="default: C3 on one GPU (the headline), C4 dealt over the ranks when --gpus > 1")
    ap.add_argument("--loci", type=int, default=None, help="override the number of loci of the config (experiments)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration per leg (0 = skip)")
    ap.add_argument("--stage1-loci", type=int, default=16,
                    help="also time HyPhy's stage 1 (model-averaged exchangeabilities) on the first N loci (0 = skip)")
    ap.add_argument("--gamma-categories", type=int, default=1,
                    help="K > 1: opt-in discrete-gamma rate mixture in the site-rate stage (alpha 0.5); not the headline config")
    ap.add_argument("--integ-mode", type=int, default=0, help="0 = QUADPACK emulation (reference parity), 1 = closed form")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all on cuda:0 with a gloo process group (tables staged through the host for the "
                         "collective): rehearses the sharded code path on a one-GPU box; not a measurement")
    ap.add_argument("--no-single-gpu-check", action="store_true",
                    help="N > 1: skip rank 0's run of the whole config (the bit-identity check of the gathered table)")
    args = ap.parse_args()

    if args.workload == "R1":
        return real_data_workload(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    workload = args.workload or ("C3" if world == 1 else "C4")

    # The all-cores CPU leg needs worker processes; they are forked HERE, before torch is imported and before anything
    # touches the GPU (a fork of a process that holds a HIP context inherits its KFD state), and sit idle until the
    # GPU part is over.
    cpu_pool, cpu_workers = None, 0
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        import multiprocessing
        from oracle import oracle as orc
        orc.build()
        cpu_workers = max(1, min(usable_cpus() - 1, 64))
        cpu_pool = multiprocessing.get_context("fork").Pool(cpu_workers)

    import torch
    from tapir_amd import dist as tdist
    from tapir_amd import engine, synth

    local_rank = 0 if args.rehearse_on_one_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if engine.device_count() < 1:
        raise SystemExit("bench.py needs a GPU: the engine has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torch.distributed.run (RANK set) the RCCL path is used even with one rank, so that a 1-GPU torchrun
    # exercises exactly the code the 8-GPU run executes
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29500")
            if args.rehearse_on_one_gpu:
                dist.init_process_group(backend="gloo", rank=rank, world_size=world)
            else:
                dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    nloci_full, ncols, ntaxa, times, intervals = synth.WORKLOADS[workload]
    nloci_full = args.loci or nloci_full
    seed = synth.WORKLOAD_SEED[workload]

    # every rank generates the SAME batch (rank-independent seed) and keeps its round-robin share of the loci
    t_gen = time.time()
    tree = synth.yule_tree(ntaxa, seed)
    data = synth.simulate(nloci_full, ncols, ntaxa, seed, device=dev, tree=tree)
    pin = synth.plan_inputs(data["root"], data["names"])
    mine = tdist.shard_loci(nloci_full, rank, world)
    nloci = len(mine)
    if world > 1:
        idx = torch.from_numpy(mine).to(dev)
        d_states = data["states"].view(ntaxa, nloci_full, ncols)[:, idx, :].reshape(ntaxa, nloci * ncols).contiguous()
        if rank != 0 or args.no_single_gpu_check:
            data["states"] = None   # only rank 0 needs the whole batch again (single-GPU check)
    else:
        d_states = data["states"]
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    def make_plan(loci_idx):
        off = np.arange(len(loci_idx) + 1, dtype=np.int64) * ncols
        extra = {}
        if args.gamma_categories > 1:
            from tapir_amd import compute
            extra = dict(zip(("cat_rates", "cat_weights"), compute.discrete_gamma(0.5, args.gamma_categories)))
        return engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], off, data["pi"][loci_idx], data["exch"][loci_idx],
                           pin["T"], times, intervals, correction=pin["correction"], threshold=3, round_decimals=4,
                           integ_mode=args.integ_mode, device=local_rank, **extra)

    def buffers(plan, nl):
        n = plan.ncols
        return dict(rate=torch.empty(n, dtype=torch.float64, device=dev), subst=torch.empty(n, dtype=torch.float64, device=dev),
                    lnl=torch.empty(n, dtype=torch.float64, device=dev), flag=torch.empty(n, dtype=torch.uint8, device=dev),
                    nres=torch.empty(n, dtype=torch.int32, device=dev),
                    tables=torch.empty((nl, plan.width), dtype=torch.float64, device=dev),
                    ws=torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=dev))

    plan = make_plan(mine)
    n = plan.ncols
    B = buffers(plan, nloci)
    gather_buf = {}
    result = {}

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        plan.run_dev(d_states, B["rate"], B["subst"], B["lnl"], B["flag"], B["nres"], B["tables"], B["ws"], stream)
        if use_dist:   # the one collective: [ceil(L/G), W] PI rows per rank -> the [L, W] table in locus order
            if args.rehearse_on_one_gpu:
                result["table"] = tdist.gather_tables(B["tables"].cpu(), nloci_full, rank, world, always=True).to(dev)
            else:
                result["table"] = tdist.gather_tables(B["tables"], nloci_full, rank, world, buffers=gather_buf, always=True)
        else:
            result["table"] = B["tables"]

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    plan.profile_enable(True)
    plan.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    site_ms, pi_ms, launches = plan.profile_read(reset=True)
    plan.profile_enable(False)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if args.rehearse_on_one_gpu else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    evals = plan.last_eval_count()

    total_cols = nloci_full * ncols
    value = total_cols * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    site_avg_ms = site_ms / max(1, launches)
    pi_avg_ms = pi_ms / max(1, launches)
    alg_bytes = n * (ntaxa + 24)  # per launch of the dominant kernel, this rank
    achieved = alg_bytes / (site_avg_ms * 1e-3) / 1e9 if site_avg_ms > 0 else 0.0

    table = result["table"]
    table_sha = hashlib.sha256(table.cpu().numpy().tobytes()).hexdigest() if rank == 0 else None
    single_check = None
    if world > 1 and rank == 0 and not args.no_single_gpu_check:
        # the whole config on this one GPU: the gathered table of the sharded run must carry the same bits
        full_plan = make_plan(np.arange(nloci_full))
        FB = buffers(full_plan, nloci_full)
        full_plan.run_dev(data["states"], FB["rate"], FB["subst"], FB["lnl"], FB["flag"], FB["nres"], FB["tables"], FB["ws"],
                          torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        single_check = bool(torch.equal(FB["tables"], table))
        full_plan.close()
        del FB
        if not single_check:
            raise SystemExit("PARITY FAILURE: the table gathered from %d ranks differs from the single-GPU table" % world)

    # what the HBM roof is on this very device: a device-to-device copy of 1 GiB (read + write counted)
    hbm_copy_gbs = None
    if rank == 0:
        src = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
        dst = torch.empty_like(src)
        dst.copy_(src)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            dst.copy_(src)
        torch.cuda.synchronize()
        hbm_copy_gbs = 5 * 2 * (1 << 30) / (time.perf_counter() - t1) / 1e9
        del src, dst

    # HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC pass of this workload
    # (profiles/: separate --pmc passes, FETCH_SIZE calibrated on classify_kernel's known byte count)
    traffic = None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc_path) and world == 1:
        with open(pmc_path) as fh:
            traffic = json.load(fh).get(workload, {}).get("site_rate_kernel_bytes")

    flags = torch.bincount(B["flag"].to(torch.int64), minlength=5).cpu().numpy().tolist()
    fl_eval = flops_per_eval(plan.op_counts)
    fp64_tflops = evals * fl_eval / (site_avg_ms * 1e-3) / 1e12 if site_avg_ms > 0 else 0.0

    out = {
        "metric": "alignment columns/sec (site-rate+PI)",
        "value": value,
        "unit": "columns/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong" if world > 1 else "none",   # N = 1 is neither weak nor strong scaling
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (seeded Yule tree, GTR-simulated columns, Gamma(0.5) site rates, 5% gaps; SURVEY.md 8d)",
        "config": {
            "workload": "%s: %d loci x %d columns x %d taxa, per-site GTR rate ML + PI "
                        "(T=%d net times, %d --times, %d --intervals, integ_mode=%d)%s"
                        % (workload, nloci_full, ncols, ntaxa, pin["T"], len(times), len(intervals), args.integ_mode,
                           "" if args.gamma_categories <= 1 else ", +G mixture of %d rate categories (extension)" % args.gamma_categories),
            "total_columns": total_cols,
            "columns_per_gpu": n,
            "loci_per_gpu": nloci,
            "parallelism": ("whole config on one GPU" if world == 1 else
                            "one fixed config, locus i -> rank i mod %d, one all-gather of PI tables" % world),
            "stack_depth": plan.stack_depth,
        },
        "roofline": {
            "bound": "hbm", "kernel": "site_rate_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes, "kernel_avg_ms": site_avg_ms,
            "hbm_copy_measured_gbs": hbm_copy_gbs,
            "note": "BASELINE.json mandates the HBM figure; the path is FP64-VALU bound (see fp64); traffic = PMC "
                    "FETCH_SIZE (calibrated) + WRITE_SIZE bytes per launch from profiles/pmc_traffic.json; rank 0's share",
        },
        "fp64": {
            "evals_per_launch": evals, "evals_per_column": evals / max(1, n), "flop_per_eval_model": fl_eval,
            "achieved_tflops": fp64_tflops, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
            "frac": fp64_tflops / FP64_VALU_PEAK_TFLOPS,
        },
        "stages_ms": {"site_rate_kernel": site_avg_ms, "pi_kernels": pi_avg_ms, "step_total": ms_per_step},
        "flags": dict(zip(["ok", "flat", "saturated", "zero", "maxit"], flags)),
        "table_sha256": table_sha,
        "gathered_equals_single_gpu": single_check,
        "gen_seconds": t_gen,
    }
    if args.rehearse_on_one_gpu:
        out["rehearsal"] = "all %d ranks on cuda:0, gloo collective through the host: NOT a measurement" % world

    parity_failure = None
    if rank == 0 and world == 1 and args.stage1_loci > 0:
        out["stage1"] = stage1_sample(data, pin, min(args.stage1_loci, nloci), ncols, ntaxa, times, intervals)
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"], parity_failure = cpu_baseline(cpu_pool, cpu_workers, data, pin, nloci, ncols, ntaxa, times,
                                                           intervals, args, B, plan)
    if cpu_pool is not None:
        cpu_pool.close()
        cpu_pool.join()
    if rank == 0:
        print(json.dumps(out))
        sys.stdout.flush()
    if use_dist:
        # every rank's own rows must sit, bit-identical, at their round-robin places of the gathered table
        assert torch.equal(table[torch.from_numpy(mine).to(dev)], B["tables"])
    plan.close()
    if use_dist:
        dist.destroy_process_group()
    if parity_failure:
        raise SystemExit("PARITY FAILURE (GPU vs CPU oracle on the bench's own columns): " + parity_failure)


def real_data_workload(args):
    """--workload R1: a real-data-shaped batch.  The synthetic configs repeat almost no column (5 % random gaps), so the
    one-fit-per-unique-pattern machinery (HyPhy: GetDataInfo(dupInfo...), alreadyDone[siteMap], models_and_rates.bf:1033-1044;
    here csrc/pattern_kernels.hpp) is idle in them.  R1 = bootstrap resamples of the reference's bundled locus
    (tests/golden/chr1_918.nex, 5 taxa x 226 columns, 57 patterns) on its tree, 1536 loci x 2048 columns = 3.1e6 columns (at the 2^20
    columns from which the automatic mode engages, the batch still runs in a latency-bound small-batch mode where 36 x fewer
    columns to fit are worth 0.04 ms and the four extra launches cost 0.065), with
    the PhyDesign parameters of the known-answer file; the line carries `dedup` = the same pass with de-duplication off and
    automatic (bit-identical outputs are required)."""
    import torch
    from tapir_amd import compute, engine, newick, nexus
    g = os.path.join(ROOT, "tests", "golden")
    names, states = nexus.read_states(os.path.join(g, "chr1_918.nex"))
    root = newick.read_tree(os.path.join(g, "Euteleost.tree"))
    depth, factor = compute.correct_tree(root)
    parent, blen, leaf = newick.to_arrays(root, names)
    kat = json.load(open(os.path.join(g, "chr1_918_phydesign_rates.json")))
    pi = np.array(kat["freqs_ACGT"])
    exch = np.array([kat[k] for k in ("AC", "AG", "AT", "CG", "CT", "GT")])
    L, S = args.loci or 1536, 2048
    rng = np.random.default_rng(20261004)
    st = np.ascontiguousarray(np.concatenate([states[:, rng.integers(0, states.shape[1], S)] for _ in range(L)], axis=1))
    uniq = np.mean([len({st[:, l * S + c].tobytes() for c in range(S)}) for l in range(min(L, 32))]) / S
    off = np.arange(L + 1, dtype=np.int64) * S
    dev = torch.device("cuda", 0)
    d_states = torch.from_numpy(st).to(dev)
    times, intervals = [10, 20, 50], [[0, 10], [20, 100]]
    res = {}
    for name, mode in (("off", engine.DEDUP_OFF), ("auto", engine.DEDUP_AUTO)):
        plan = engine.Plan(len(names), parent, blen, leaf, off, np.tile(pi, (L, 1)), np.tile(exch, (L, 1)), int(depth), times, intervals,
                           correction=factor, pattern_dedup=mode)
        n = plan.ncols
        B = dict(rate=torch.empty(n, dtype=torch.float64, device=dev), subst=torch.empty(n, dtype=torch.float64, device=dev),
                 lnl=torch.empty(n, dtype=torch.float64, device=dev), flag=torch.empty(n, dtype=torch.uint8, device=dev),
                 nres=torch.empty(n, dtype=torch.int32, device=dev), tables=torch.empty((L, plan.width), dtype=torch.float64, device=dev),
                 ws=torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=dev))
        stream = torch.cuda.current_stream().cuda_stream
        for _ in range(args.warmup):
            plan.run_dev(d_states, B["rate"], B["subst"], B["lnl"], B["flag"], B["nres"], B["tables"], B["ws"], stream)
        torch.cuda.synchronize()
        plan.profile_enable(True)
        plan.profile_read(reset=True)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            plan.run_dev(d_states, B["rate"], B["subst"], B["lnl"], B["flag"], B["nres"], B["tables"], B["ws"], stream)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        site_ms, pi_ms, launches = plan.profile_read(reset=True)
        res[name] = dict(ms=1e3 * dt / args.steps, site_ms=site_ms / max(1, launches), evals=plan.last_eval_count(),
                         out={k: B[k].cpu().numpy() for k in ("rate", "subst", "lnl", "flag", "nres", "tables")})
        plan.close()
    for k in res["off"]["out"]:
        if not np.array_equal(res["off"]["out"][k], res["auto"]["out"][k], equal_nan=True):
            raise SystemExit("PARITY FAILURE: de-duplication changed `%s`" % k)
    n = L * S
    a = res["auto"]
    alg = n * (len(names) + 24)
    print(json.dumps({
        "metric": "alignment columns/sec (site-rate+PI)", "value": n / (a["ms"] * 1e-3), "unit": "columns/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": a["ms"], "higher_is_better": True, "scaling": "none",
        "vs_baseline": None, "dtype": "f64",
        "data": "bootstrap resamples of tapir/tests/test-data/chr1_918.nex (committed fixture) on Euteleost.tree, PhyDesign parameters",
        "config": {"workload": "R1: %d loci x %d resampled columns x %d taxa (real-data-shaped: %.1f %% of a locus' columns are "
                               "distinct patterns), T=%d, %d --times, %d --intervals" % (L, S, len(names), 100 * uniq, int(depth),
                                                                                        len(times), len(intervals)),
                   "total_columns": n},
        "roofline": {"bound": "hbm", "kernel": "site_rate_kernel", "achieved": alg / (a["site_ms"] * 1e-3) / 1e9 if a["site_ms"] else 0.0,
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": (alg / (a["site_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if a["site_ms"] else 0.0,
                     "traffic": None, "algorithmic_bytes_per_launch": alg, "kernel_avg_ms": a["site_ms"]},
        "cpu_baseline": None,
        "dedup": {"unique_fraction": uniq, "evals_off": res["off"]["evals"], "evals_auto": a["evals"],
                  "site_ms_off": res["off"]["site_ms"], "site_ms_auto": a["site_ms"], "step_ms_off": res["off"]["ms"],
                  "step_ms_auto": a["ms"], "outputs_bit_identical": True},
    }))


def stage1_sample(data, pin, nl, ncols, ntaxa, times, intervals):
    """Not part of `value`: wall time of the stage that precedes the per-site loop in HyPhy's script (203-model fit +
    Akaike averaging of the exchangeabilities, models_and_rates.bf:405-897) on the first loci of the same batch, through
    the product path (pattern compression, likelihood + gradient kernels, tapir_amd/stage1.py)."""
    from tapir_amd import engine, pipeline
    # the alignment sits in pinned host memory, as the command line's parser leaves it (pipeline.load_alignments)
    st = engine.pinned_empty((ntaxa, nl * ncols), np.uint8)
    st[:] = data["states"][:, :nl * ncols].cpu().numpy()
    off = np.arange(nl + 1, dtype=np.int64) * ncols

    def run():
        t0 = time.perf_counter()
        # one engine call: upload, empirical base frequencies (bf:968), site patterns (bf:960-963), the 203 fits, averaging
        e = pipeline.model_averaged_exchangeabilities(engine, st, off, None, ntaxa, pin["parent"], pin["blen"], pin["leaf"],
                                                      pin["T"], times, intervals, pin["correction"])
        return e, time.perf_counter() - t0

    _, first = run()
    exch, dt = run()
    true = np.asarray(data["exch"][:nl])
    return {"loci": nl, "columns": nl * ncols, "seconds": dt, "columns_per_s": nl * ncols / dt, "first_call_seconds": first,
            "note": "second of two calls; tphip_stage1_fit through the pipeline: host-pointer path from pinned memory incl. the "
                    "upload; estimates vs generating rates differ by design (the simulation has Gamma site rates, stage 1 "
                    "assumes one rate)",
            "max_rel_dev_from_generating_rates": float(np.max(np.abs(exch - true / true[:, 1:2]) / (true / true[:, 1:2])))}


def cpu_baseline(pool, workers, data, pin, nloci, ncols, ntaxa, times, intervals, args, B, plan):
    """The CPU oracle (oracle/tapir_oracle.c) on bounded samples of the same bytes, two legs:
      single thread -- the analogue of single-threaded hyphy2, one locus at a time;
      all cores     -- one oracle worker per locus over the pool forked at start-up, the analogue of
                       Pool(cpu_count() - 1).map(worker, params) (bin/tapir_compute.py:162-163).
    Every column the oracle processes is also compared with what the GPU just produced for it (parity_sample)."""
    st = data["states"]
    T = pin["T"]
    d_rate, d_flag, d_nres, d_tables = B["rate"], B["flag"], B["nres"], B["tables"]
    par = dict(columns=0, loci_with_tables=0, max_rel_rate=0.0, flag_mismatches=0, nres_mismatches=0, tables_max_rel=0.0)

    def job(l, c1):
        sl = st[:, l * ncols:l * ncols + c1].cpu().numpy()
        return (sl, pin["parent"], pin["blen"], pin["leaf"], data["pi"][l], data["exch"][l], pin["correction"], T, intervals)

    def check(l, c1, res):
        _, rate, flag, nres, net, si, se = res
        g_rate = d_rate[l * ncols:l * ncols + c1].cpu().numpy()
        g_flag = d_flag[l * ncols:l * ncols + c1].cpu().numpy()
        g_nres = d_nres[l * ncols:l * ncols + c1].cpu().numpy()
        par["columns"] += c1
        par["flag_mismatches"] += int((g_flag != flag).sum())
        par["nres_mismatches"] += int((g_nres != nres).sum())
        ok = ((flag == 0) | (flag == 3)) & (g_flag == flag)
        if ok.any():
            rel = np.abs(g_rate[ok] - rate[ok]) / np.maximum(np.abs(rate[ok]), 1e-12)
            par["max_rel_rate"] = max(par["max_rel_rate"], float(rel.max()))
        if c1 == ncols and args.gamma_categories <= 1:   # a whole locus: its PI row is comparable too
            row = d_tables[l].cpu().numpy()
            n_t, n_i = len(times), len(intervals)
            ref = np.concatenate([net, net[np.asarray(times, dtype=np.int64)], si])
            got = row[:T + n_t + n_i]
            par["tables_max_rel"] = max(par["tables_max_rel"], float(np.max(np.abs(got - ref)) / max(1e-300, np.abs(ref).max())))
            par["loci_with_tables"] += 1

    # ---- leg 1: one thread
    probe_cols = min(ncols, 512)
    res = _oracle_job(job(0, probe_cols))
    per_col = res[0] / probe_cols
    want = int(max(probe_cols, min(nloci * ncols, args.cpu_seconds / per_col)))
    done, t_total, l = 0, 0.0, 0
    while done < want and l < nloci:
        c1 = min(ncols, want - done)
        res = _oracle_job(job(l, c1))
        check(l, c1, res)
        t_total += res[0]
        done += c1
        l += 1
    single = done / t_total
    out = {"value": single, "unit": "columns/s", "cores": 1, "kind": "port",
           "sample": "first %d columns of the same synthetic batch (%d loci), oracle/tapir_oracle.c single thread, "
                     "site-rate ML + net PI + dqagse integrals, %.1f s" % (done, l, t_total),
           "host_cpus": os.cpu_count(), "usable_cpus": usable_cpus()}
    # ---- leg 2: all usable cores, one locus sample per task (different loci than leg 1 where the batch has them)
    if pool is not None and workers >= 1:
        per_task = int(max(64, min(ncols, args.cpu_seconds * single / 2)))   # ~cpu_seconds / 2 of work per task, 2 rounds
        first = l if l + 2 * workers <= nloci else 0
        loci = [(first + k) % nloci for k in range(min(2 * workers, nloci))]
        jobs = [job(x, per_task) for x in loci]
        t0 = time.perf_counter()
        results = pool.map(_oracle_job, jobs, chunksize=1)
        wall = time.perf_counter() - t0
        for x, r in zip(loci, results):
            check(x, per_task, r)
        out["all_cores"] = {"value": len(jobs) * per_task / wall, "unit": "columns/s", "cores": workers,
                            "sample": "%d tasks of %d columns (one locus each) over %d forked oracle workers, %.1f s wall, "
                                      "%.1f s of CPU work" % (len(jobs), per_task, workers, wall, sum(r[0] for r in results))}
    # ---- leg 3: the cost structure of the reference's own PI stage -- a dense (T, S) matrix,
    # nansum, and one scipy.integrate.quad per (site, interval) summed in Python (tapir/compute.py:46-52, 76-94) -- written
    # here with numpy / scipy on the rates the GPU just produced for one whole locus (so its row of the GPU's PI table is
    # comparable); reported per column, an extrapolation from that locus.  Skipped where scipy is not installed.
    try:
        from scipy import integrate as _integrate
    except ImportError:
        _integrate = None
    if _integrate is not None and args.gamma_categories <= 1:
        from tapir_amd import compute
        l3 = min(nloci - 1, 1)
        c3 = ncols
        raw = d_rate[l3 * ncols:l3 * ncols + c3].cpu().numpy()
        nres = d_nres[l3 * ncols:l3 * ncols + c3].cpu().numpy()
        rates = compute.round_like_hyphy(raw, 4) / pin["correction"]
        rates = np.where(nres >= 3, rates, np.nan)

        def townsend(t, r):
            return 16.0 * r * r * t * np.exp(-4.0 * r * t)

        t0 = time.perf_counter()
        tv = np.arange(T, dtype=np.float64)[:, None]
        net = np.nansum(townsend(tv, rates[None, :]), axis=1)
        fin = rates[np.isfinite(rates)]
        epochs = []
        budget = max(2.0, args.cpu_seconds)          # seconds for the quad calls; beyond it the leg extrapolates further
        used_sites = len(fin)
        for (a, b) in intervals:
            vals = []
            for k, r in enumerate(fin):
                vals.append(_integrate.quad(townsend, a, b, args=(r,)))
                if k % 1024 == 1023 and time.perf_counter() - t0 > budget * (len(epochs) + 1) / len(intervals):
                    used_sites = min(used_sites, k + 1)
                    break
            epochs.append((sum(v[0] for v in vals), sum(v[1] for v in vals), len(vals)))
        dt3 = time.perf_counter() - t0
        frac = sum(e[2] for e in epochs) / max(1, len(intervals) * len(fin))
        leg3 = {"value": c3 * frac / dt3, "unit": "columns/s", "cores": 1, "kind": "port",
                "sample": "locus %d of the same batch (%d columns, %d finite rates), %.0f %% of its (site, interval) quad calls in "
                          "%.1f s: extrapolated to the whole batch; numpy (T, S) matrix + nansum + scipy.integrate.quad per site "
                          "and interval (PI stage only, no site-rate ML)" % (l3, c3, len(fin), 100 * frac, dt3)}
        if frac == 1.0:   # every call made: this is what the reference's PI stage would store for the locus
            row = d_tables[l3].cpu().numpy()
            n_t, n_i = len(times), len(intervals)
            ref = np.concatenate([net, net[np.asarray(times, dtype=np.int64)], [e[0] for e in epochs]])
            rel = float(np.max(np.abs(row[:T + n_t + n_i] - ref)) / max(1e-300, np.abs(ref).max()))
            leg3["gpu_table_row_max_rel_diff"] = rel
            par["scipy_table_max_rel"] = rel
        out["numpy_scipy_pi"] = leg3
    out["parity_sample"] = par
    failure = None
    if par["flag_mismatches"] or par["nres_mismatches"]:
        failure = "%d flag and %d informative-count mismatches" % (par["flag_mismatches"], par["nres_mismatches"])
    elif par["max_rel_rate"] > 1e-6:
        failure = "site rates differ by %.3g relative (tolerance 1e-6)" % par["max_rel_rate"]
    elif par["tables_max_rel"] > 1e-9:
        failure = "PI table rows differ by %.3g relative (tolerance 1e-9)" % par["tables_max_rel"]
    elif par.get("scipy_table_max_rel", 0.0) > 1e-9:
        failure = "PI table row differs from numpy + scipy.integrate.quad by %.3g relative (tolerance 1e-9)" % par["scipy_table_max_rel"]
    return out, failure


if __name__ == "__main__":
    main()
