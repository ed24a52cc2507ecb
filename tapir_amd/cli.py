"""tapir_compute.py, GPU edition: the reference's command line (bin/tapir_compute.py:18-53, 125-177) with the
per-locus HyPhy loop replaced by one batched call into the HIP engine.

Same positionals, same required/optional flags and defaults, same output directory contents
(Tree_<factor>_<depth>.newick, <alignment>.rates JSON per locus, phylogenetic-informativeness.sqlite).
`--hyphy` and `--template` are accepted for compatibility and ignored (there is no subprocess);
`--multiprocessing` parallelises the host side only (NEXUS parsing, .rates files).  New, opt-in flags only: --device, --exchangeabilities / --subs-model,
--integral-mode, --full-precision-rates, --gamma-categories / --gamma-alpha.

Several GPUs: launch it with `python -m torch.distributed.run --nproc-per-node G bin/tapir_compute.py ...` (one process
per GPU).  The files are dealt round-robin over the ranks (what `Pool.map(worker, params)` did over cores,
bin/tapir_compute.py:159-164); every rank writes the .rates files of its own loci into the one output directory
rank 0 created; the per-locus PI rows are collected with a single all-gather (tapir_amd/dist.py: RCCL, or gloo
without GPUs) and rank 0 writes the sqlite file in the original file order.

Stage 1 of the HyPhy script (203-model fit + model averaging of the GTR exchangeabilities,
models_and_rates.bf:405-897) runs on the GPU too (tapir_amd/stage1.py) unless the exchangeabilities are given
with --exchangeabilities / --subs-model.
"""
import argparse
import os
import sys

import numpy as np

from . import base, compute, db, newick, pipeline
from . import dist as tdist


LAST_TIMINGS = {}   # seconds per stage of the last main() call on this rank (tools/e2e_cli_timing.py)


def get_args(argv=None):
    """Get CLI arguments and options (mirrors bin/tapir_compute.py:18-53)."""
    parser = argparse.ArgumentParser(description="""tapir:  compute the
            phylogenetic informativeness of DNA loci""")
    parser.add_argument('alignments', help="The folder of alignments", action=base.FullPaths, type=base.is_dir)
    parser.add_argument('tree', help="The input tree", action=base.FullPaths)
    required = parser.add_argument_group("required arguments")
    required.add_argument('--times', help="""Comma-separated list of start
        times of interest (MYA)""", type=base.get_list_from_ints, required=True)
    required.add_argument('--intervals', help="""Comma-separated list of
        interval ranges of interest (i,e. in MYA)""", type=base.get_list_from_ranges, required=True)
    parser.add_argument('--tree-format', help="The format of the tree", dest='tree_format',
                        choices=['nexus', 'newick'], default='newick')
    parser.add_argument('--output', help="The path to the output directory", default=os.getcwd(),
                        action=base.FullPaths, type=base.is_dir)
    parser.add_argument('--hyphy', help="Ignored (kept for compatibility): there is no hyphy subprocess",
                        default="hyphy2")
    parser.add_argument('--template', help="Ignored (kept for compatibility)", default=None)
    parser.add_argument('--threshold', help="""Minimum number of taxa without
        a gap for a site to be considered informative""", default=3, type=int)
    parser.add_argument('--multiprocessing', help="""Enable parallel
        reading of alignments and writing of site-rate files (the rates themselves are always batched on the GPU)""",
                        default=False, action='store_true')
    parser.add_argument('--site-rates', default=False, action='store_true',
                        help="Use previously calculated site rates")
    parser.add_argument('--subset-pi-map-file', help="""Calculate PI for a
        subset of sites. If specified, this should be a tab-delimited file
        with the alignment file name in the 1st column, the start of the
        interval (0-offset) in the 2nd column, and the end of the interval in
        the 3rd column.""")
    new = parser.add_argument_group("MI355X engine options (not in the reference)")
    new.add_argument('--device', type=int, default=0, help="HIP device ordinal")
    new.add_argument('--exchangeabilities', type=_six_floats, default=None,
                     help="AC,AG,AT,CG,CT,GT used for every locus (default: model-averaged estimates per locus, "
                          "as the HyPhy script computes them)")
    new.add_argument('--subs-model', default=None,
                     help="tab-delimited file: alignment file name, AC, AG, AT, CG, CT, GT [, A, C, G, T frequencies]")
    new.add_argument('--integral-mode', choices=['quadpack', 'closed'], default='quadpack',
                     help="quadpack = emulate scipy.integrate.quad incl. its error column; closed = analytic")
    new.add_argument('--gamma-categories', type=int, default=1,
                     help="K > 1: discrete-gamma mixture of K rate categories on top of each site's rate (GTR+G; the "
                          "reference's HyPhy script has none, so K = 1 is the drop-in setting)")
    new.add_argument('--gamma-alpha', type=float, default=0.5, help="shape of that gamma distribution")
    new.add_argument('--reference-start', action='store_true',
                     help="start every site's optimiser at siteRate = 1 as the HyPhy script does (models_and_rates.bf:1050) "
                          "instead of at the site's parsimony rate: identical on unimodal sites, and on the rare multimodal "
                          "ones the optimum uphill of HyPhy's start; costs about one likelihood evaluation per site")
    new.add_argument('--full-precision-rates', action='store_true',
                     help="do not round site rates to 4 decimals before PI (the reference rounds through its JSON file)")
    return parser.parse_args(argv)


def _six_floats(string):
    try:
        v = [float(x) for x in string.split(',')]
        assert len(v) == 6
    except Exception as e:
        raise argparse.ArgumentTypeError("Cannot convert exchangeabilities to six numbers: {0}".format(e))
    return v


def welcome_message():
    return '''
    ***************************************************
    *                                                 *
    * tapir: high-throughput estimates of             *
    * phylogenetic informativeness                    *
    *                                                 *
    * MI355X engine (tapir_amd): site-rate ML and PI  *
    * in one HIP pipeline.  Method credits:           *
    *                                                 *
    *   - J.P. Townsend, 2007. Profiling              *
    *     phylogenetic informativeness. Systematic    *
    *     Biology, 56(2), 222-231.                    *
    *                                                 *
    *   - Pond, S.L.K., Frost, S.D.W., and S.V. Muse, *
    *     2005. Hyphy: hypothesis testing using       *
    *     phylogenies. Bioinformatics, 21(5), 676-9.  *
    *                                                 *
    *   - B.C. Faircloth, J. Chang, M.E. Alfaro, 2012.*
    *     TAPIR (the program this engine drops into). *
    *                                                 *
    ***************************************************\n\n'''


def read_subs_model(path, alignments):
    table = {}
    with open(path) as fh:
        for line in fh:
            parts = line.rstrip("\n").split("\t")
            if len(parts) >= 7 and not line.startswith("#"):
                table[parts[0]] = [float(x) for x in parts[1:]]
    exch, pi = [], []
    for a in alignments:
        b = os.path.basename(a)
        if b not in table:
            raise IOError("no substitution model for {0} in {1}".format(b, path))
        exch.append(table[b][:6])
        pi.append(table[b][6:10] if len(table[b]) >= 10 else None)
    pis = None if any(p is None for p in pi) else np.array(pi)
    return np.array(exch), pis


def _broadcast(obj, rank, world):
    if world == 1:
        return obj
    import torch.distributed as dist
    box = [obj if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def _gather_rows(tables, nfiles, rank, world, on_gpu):
    """[n_local, W] rows of this rank's files (file i -> rank i mod world) -> [nfiles, W] in file order."""
    if world == 1:
        return tables
    import torch
    t = torch.from_numpy(np.ascontiguousarray(tables, dtype=np.float64))
    if on_gpu:
        t = t.cuda()
    return tdist.gather_tables(t, nfiles, rank, world).cpu().numpy()


def main(argv=None, engine_mod=None):
    """Main loop (mirrors bin/tapir_compute.py:125-177)."""
    args = get_args(argv)
    rank, world = tdist.rank_world()
    on_gpu = engine_mod is None
    pool = None
    if args.multiprocessing:  # the reference: Pool(processes = cpu_count() - 1), bin/tapir_compute.py:162-163
        from multiprocessing import cpu_count
        # forked here, before torch.distributed / RCCL and before the first libtphip call: see pipeline.HostPool
        # (forking hundreds of workers costs more than it saves: at most 16 per rank)
        pool = pipeline.HostPool(max(1, min(16, (cpu_count() - 1) // world)))
    try:
        return _main(args, rank, world, on_gpu, engine_mod, pool)
    finally:
        if pool is not None:
            pool.close()


def _main(args, rank, world, on_gpu, engine_mod, pool):
    if world > 1:
        tdist.init_process_group(None if on_gpu else "gloo")
        if on_gpu:
            args.device = int(os.environ.get("LOCAL_RANK", rank))
    if rank == 0:
        print(welcome_message())
        args.output = base.create_unique_dir(args.output)
        # correct branch lengths
        setup = (args.output,) + tuple(compute.correct_branch_lengths(args.tree, args.tree_format, d=args.output))
    else:
        setup = None
    args.output, tree_depth, correction, tree = _broadcast(setup, rank, world)
    # generate a vector of times given start and stops
    T = int(tree_depth)
    subset_pi = dict()
    if args.subset_pi_map_file:
        subset_pi = dict(base.parse_subset_map_file(args.subset_pi_map_file))
    root = newick.read_tree(tree, 'newick')
    leaf_names = [n.name for n in newick.leaves(root)]
    parent, blen, leaf = newick.to_arrays(root, leaf_names)
    integ_mode = 0 if args.integral_mode == 'quadpack' else 1
    progress = pipeline.dot_progress if rank == 0 else None
    cat_rates, cat_weights = (compute.discrete_gamma(args.gamma_alpha, args.gamma_categories)
                              if args.gamma_categories > 1 else (None, None))
    W = T + len(args.times) + 2 * len(args.intervals)
    db_name = os.path.join(args.output, 'phylogenetic-informativeness.sqlite')
    stored = False

    def store(pis):   # tapir_compute.py's last step; single process: may run while the pool writes the .rates files
        import time
        t_db = time.perf_counter()
        conn, c = db.create_probe_db(db_name)
        db.insert_pi_data(conn, c, pis)
        conn.commit()
        c.close()
        conn.close()
        LAST_TIMINGS["sqlite"] = time.perf_counter() - t_db

    class _TableSink:
        """sqlite rows block by block, in file order, while later blocks are still being computed (pipeline._run_streamed)"""

        def __init__(self):
            self.conn = self.c = None
            self.seconds = 0.0

        def add(self, names, tables):
            import time
            t0 = time.perf_counter()
            if self.conn is None:
                self.conn, self.c = db.create_probe_db(db_name)
            db.insert_tables(self.conn, self.c, names, tables, T, args.times, args.intervals)
            self.seconds += time.perf_counter() - t0

        def close(self):
            import time
            t0 = time.perf_counter()
            if self.conn is None:
                self.conn, self.c = db.create_probe_db(db_name)
            self.conn.commit()
            self.c.close()
            self.conn.close()
            self.seconds += time.perf_counter() - t0
            LAST_TIMINGS["sqlite"] = self.seconds

    if not args.site_rates:
        if rank == 0:
            print("\nEstimating site rates and PI for files:")
        files = base.get_files(args.alignments, '*.nex,*.nexus')
        mine = [files[i] for i in tdist.shard_loci(len(files), rank, world)]
        exch, pi = None, None  # None: fit and model-average the exchangeabilities per locus (HyPhy stage 1)
        if args.exchangeabilities is not None:
            exch = np.array(args.exchangeabilities)
        if args.subs_model:
            exch, pi = read_subs_model(args.subs_model, mine)
        if mine:
            pis, out = pipeline.run_alignments(mine, leaf_names, parent, blen, leaf, T, args.times, args.intervals,
                                               correction, args.threshold, exch, pi=pi, subsets=subset_pi,
                                               output_dir=args.output, device=args.device, integ_mode=integ_mode,
                                               round_decimals=-1 if args.full_precision_rates else 4,
                                               engine_mod=engine_mod, progress=progress, pool=pool,
                                               cat_rates=cat_rates, cat_weights=cat_weights,
                                               start_rule=1 if args.reference_start else 0,
                                               during_write=store if world == 1 else None,
                                               table_sink=_TableSink() if world == 1 else None)
            tables = out["final_tables"]
            stored = bool(out.get("during_write_done"))
            sqlite_seconds = LAST_TIMINGS.get("sqlite")
            LAST_TIMINGS.clear()
            LAST_TIMINGS.update(out.get("timings", {}))
            if stored:
                LAST_TIMINGS["sqlite"] = sqlite_seconds
        else:
            pis, tables = [], np.zeros((0, W))
    else:
        if rank == 0:
            print("Estimating PI for files (--site-rate option):")
        files = base.get_files(args.alignments, '*.rates')
        mine = [files[i] for i in tdist.shard_loci(len(files), rank, world)]
        if mine:
            pis, tables = pipeline.run_rate_files(mine, leaf_names, parent, blen, leaf, T, args.times, args.intervals,
                                                  correction, subsets=subset_pi, device=args.device, integ_mode=integ_mode,
                                                  engine_mod=engine_mod, progress=progress, return_tables=True)
        else:
            pis, tables = [], np.zeros((0, W))
    # the one collective: every rank's PI rows -> all rows in file order
    all_tables = _gather_rows(tables, len(files), rank, world, on_gpu)
    if rank == 0:
        # store results somewhere
        sys.stdout.write("\nStoring results in {0}...".format(db_name))
        sys.stdout.flush()
        if world == 1:
            if not stored:
                store(pis)
        else:
            import time
            t_db = time.perf_counter()
            conn, c = db.create_probe_db(db_name)
            db.insert_tables(conn, c, files, all_tables, T, args.times, args.intervals)
            conn.commit()
            c.close()
            conn.close()
            LAST_TIMINGS["sqlite"] = time.perf_counter() - t_db
        sys.stdout.write("DONE")
        sys.stdout.flush()
        print("\n")
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    return args.output


if __name__ == '__main__':
    main()
