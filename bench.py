#!/usr/bin/env python3
"""bench.py -- alignment columns/s through site-rate ML + PI tables on MI355X.

A "step" is one pass of the hot path (classify -> compact -> per-site rate ML -> PI tables, plus the
all-gather of PI tables when N > 1) over one batch of synthetic loci already resident in HBM.

  python bench.py                       one GPU, workload C3 (100 loci x 50 000 columns x 64 taxa)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W        weak scaling: every rank owns one C3-sized shard

Prints ONE JSON line on rank 0 (contract in the round prompt) with the extra objects `roofline` (dominant
kernel = site_rate_kernel; algorithmic bytes = (ntaxa + 24) per column), `fp64` (the bound
that actually binds this path) and `cpu_baseline` (the CPU oracle timed on the box's host cores on a
bounded sample of the same workload; a reported baseline, not the target).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C3", choices=["C2", "C3", "C4", "C5"])
    ap.add_argument("--loci", type=int, default=None, help="override loci per GPU")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU-baseline duration (0 = skip)")
    ap.add_argument("--stage1-loci", type=int, default=4,
                    help="also time HyPhy's stage 1 (model-averaged exchangeabilities) on the first N loci (0 = skip)")
    ap.add_argument("--gamma-categories", type=int, default=1,
                    help="K > 1: opt-in discrete-gamma rate mixture in the site-rate stage (alpha 0.5); not the headline config")
    ap.add_argument("--integ-mode", type=int, default=0, help="0 = QUADPACK emulation (reference parity), 1 = closed form")
    args = ap.parse_args()

    import torch
    from tapir_amd import dist as tdist
    from tapir_amd import engine, synth

    rank, world = tdist.rank_world()
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
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
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    nloci_full, ncols, ntaxa, times, intervals = synth.WORKLOADS[args.workload]
    # per-GPU share: C2/C3 are single-GPU configs (whole config per GPU); C4/C5 are 8-GPU configs (1/8 per GPU)
    share = {"C2": nloci_full, "C3": nloci_full, "C4": nloci_full // 8, "C5": nloci_full // 8}[args.workload]
    nloci = args.loci or share
    seed = synth.WORKLOAD_SEED[args.workload]

    t_gen = time.time()
    tree = synth.yule_tree(ntaxa, seed)
    data = synth.simulate(nloci, ncols, ntaxa, seed + 1000 * rank, device=dev, tree=tree)
    pin = synth.plan_inputs(data["root"], data["names"])
    torch.cuda.synchronize()
    t_gen = time.time() - t_gen

    plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], data["locus_offsets"], data["pi"], data["exch"],
                       pin["T"], times, intervals, correction=pin["correction"], threshold=3, round_decimals=4,
                       integ_mode=args.integ_mode, device=local_rank,
                       **({} if args.gamma_categories <= 1 else
                          dict(zip(("cat_rates", "cat_weights"), __import__("tapir_amd.compute", fromlist=["x"]).discrete_gamma(0.5, args.gamma_categories)))))
    n = plan.ncols
    W = plan.width
    d_states = data["states"]
    d_rate = torch.empty(n, dtype=torch.float64, device=dev)
    d_subst = torch.empty(n, dtype=torch.float64, device=dev)
    d_lnl = torch.empty(n, dtype=torch.float64, device=dev)
    d_flag = torch.empty(n, dtype=torch.uint8, device=dev)
    d_nres = torch.empty(n, dtype=torch.int32, device=dev)
    d_tables = torch.empty((nloci, W), dtype=torch.float64, device=dev)
    d_ws = torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=dev)
    gathered = torch.empty((world * nloci, W), dtype=torch.float64, device=dev) if use_dist else None

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        plan.run_dev(d_states, d_rate, d_subst, d_lnl, d_flag, d_nres, d_tables, d_ws, stream)
        if use_dist:
            dist.all_gather_into_tensor(gathered, d_tables)  # the one collective: [L/G, W] PI tables per rank

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
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    evals = plan.last_eval_count()

    total_cols = n * world
    value = total_cols * args.steps / elapsed
    ms_per_step = 1e3 * elapsed / args.steps
    site_avg_ms = site_ms / max(1, launches)
    pi_avg_ms = pi_ms / max(1, launches)
    alg_bytes = n * (ntaxa + 24)  # per launch of the dominant kernel, this rank
    achieved = alg_bytes / (site_avg_ms * 1e-3) / 1e9 if site_avg_ms > 0 else 0.0

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
    if os.path.exists(pmc_path):
        with open(pmc_path) as fh:
            traffic = json.load(fh).get(args.workload, {}).get("site_rate_kernel_bytes")

    flags = torch.bincount(d_flag.to(torch.int64), minlength=5).cpu().numpy().tolist()
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
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic (seeded Yule tree, GTR-simulated columns, Gamma(0.5) site rates, 5% gaps; SURVEY.md 8d)",
        "config": {
            "workload": "%s shape per GPU: %d loci x %d columns x %d taxa, per-site GTR rate ML + PI "
                        "(T=%d net times, %d --times, %d --intervals, integ_mode=%d)%s"
                        % (args.workload, nloci, ncols, ntaxa, pin["T"], len(times), len(intervals), args.integ_mode,
                           "" if args.gamma_categories <= 1 else ", +G mixture of %d rate categories (extension)" % args.gamma_categories),
            "columns_per_gpu": n,
            "loci_per_gpu": nloci,
            "parallelism": "loci sharded over %d rank(s), one all-gather of PI tables" % world,
            "stack_depth": plan.stack_depth,
        },
        "roofline": {
            "bound": "hbm", "kernel": "site_rate_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "algorithmic_bytes_per_launch": alg_bytes, "kernel_avg_ms": site_avg_ms,
            "hbm_copy_measured_gbs": hbm_copy_gbs,
            "note": "BASELINE.json mandates the HBM figure; the path is FP64-VALU bound (see fp64); traffic = PMC "
                    "FETCH_SIZE (calibrated) + WRITE_SIZE bytes per launch from profiles/pmc_traffic.json",
        },
        "fp64": {
            "evals_per_launch": evals, "evals_per_column": evals / max(1, n), "flop_per_eval_model": fl_eval,
            "achieved_tflops": fp64_tflops, "peak_tflops": FP64_VALU_PEAK_TFLOPS,
            "frac": fp64_tflops / FP64_VALU_PEAK_TFLOPS,
        },
        "stages_ms": {"site_rate_kernel": site_avg_ms, "pi_kernels": pi_avg_ms, "step_total": ms_per_step},
        "flags": dict(zip(["ok", "flat", "saturated", "zero", "maxit"], flags)),
        "gen_seconds": t_gen,
    }

    if rank == 0 and world == 1 and args.stage1_loci > 0:
        out["stage1"] = stage1_sample(data, pin, min(args.stage1_loci, nloci), ncols, ntaxa, times, intervals)
    if rank == 0 and world == 1 and args.cpu_seconds > 0:
        out["cpu_baseline"] = cpu_baseline(plan, data, pin, nloci, ncols, ntaxa, times, intervals, args, d_rate, d_nres)
    if rank == 0:
        print(json.dumps(out))
    if use_dist:
        # every rank's rows are in `gathered`; rank r's block must be bit-identical to its own table
        assert torch.equal(gathered[rank * nloci:(rank + 1) * nloci], d_tables)
    plan.close()
    if use_dist:
        dist.destroy_process_group()


def stage1_sample(data, pin, nl, ncols, ntaxa, times, intervals):
    """Not part of `value`: wall time of the stage that precedes the per-site loop in HyPhy's script (203-model fit +
    Akaike averaging of the exchangeabilities, models_and_rates.bf:405-897) on the first loci of the same batch, through
    the product path (pattern compression, likelihood + gradient kernels, tapir_amd/stage1.py)."""
    import time
    import numpy as np
    from tapir_amd import engine, nexus, pipeline
    st = data["states"][:, :nl * ncols].cpu().numpy()
    off = np.arange(nl + 1, dtype=np.int64) * ncols
    t0 = time.perf_counter()
    pi = nexus.base_frequencies_from_histogram(engine.state_histogram(st, off))
    exch = pipeline.model_averaged_exchangeabilities(engine, st, off, pi, ntaxa, pin["parent"], pin["blen"], pin["leaf"],
                                                     pin["T"], times, intervals, pin["correction"])
    dt = time.perf_counter() - t0
    true = np.asarray(data["exch"][:nl])
    return {"loci": nl, "columns": nl * ncols, "seconds": dt, "columns_per_s": nl * ncols / dt,
            "note": "host-pointer path incl. copies; estimates vs generating rates differ by design (the simulation has "
                    "Gamma site rates, stage 1 assumes one rate)",
            "max_rel_dev_from_generating_rates": float(np.max(np.abs(exch - true / true[:, 1:2]) / (true / true[:, 1:2])))}


def cpu_baseline(plan, data, pin, nloci, ncols, ntaxa, times, intervals, args, d_rate, d_nres):
    """The CPU oracle (oracle/tapir_oracle.c, single thread) on a bounded sample of the same bytes:
    site-rate ML + net PI + QUADPACK interval integrals, i.e. the same stages the GPU step runs."""
    from oracle import oracle as orc
    orc.lib()
    st = data["states"]

    def run(l0, c0, c1):
        sl = st[:, l0 * ncols + c0:l0 * ncols + c1].cpu().numpy()
        t0 = time.perf_counter()
        r = orc.site_rates(sl, pin["parent"], pin["blen"], pin["leaf"], data["pi"][l0], data["exch"][l0])
        rates = orc.round_dp(r["rate"], 4) / pin["correction"]
        rates[r["nres"] < 3] = np.nan
        orc.net_pi(rates, pin["T"])
        orc.net_integrals(rates, intervals, 0)
        return time.perf_counter() - t0

    probe_cols = min(ncols, 512)
    t_probe = run(0, 0, probe_cols)
    per_col = t_probe / probe_cols
    want = int(max(probe_cols, min(nloci * ncols, args.cpu_seconds / per_col)))
    done, t_total, l = 0, 0.0, 0
    while done < want and l < nloci:
        c1 = min(ncols, want - done)
        t_total += run(l, 0, c1)
        done += c1
        l += 1
    return {"value": done / t_total, "unit": "columns/s", "cores": 1, "kind": "port",
            "sample": "first %d columns of the same synthetic batch (%d loci), oracle/tapir_oracle.c single thread, "
                      "site-rate ML + net PI + dqagse integrals, %.1f s" % (done, l, t_total),
            "host_cpus": os.cpu_count()}


if __name__ == "__main__":
    main()
