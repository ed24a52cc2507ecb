# end-to-end CLI timing on a synthetic directory of NEXUS files (ingest + engine + JSON + sqlite), with the per-stage
# breakdown the pipeline records.
# usage: python tools/e2e_cli_timing.py LOCI COLS TAXA [--model-averaging] [cli flags ...]
#   --model-averaging: let the CLI estimate the exchangeabilities (HyPhy stage 1) instead of fixing them
# The alignments are generated (on the GPU when there is one) and written by a CHILD process, so that this process has
# not touched the GPU when cli.main() forks its --multiprocessing pool.
import os, sys, time, tempfile, shutil, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if sys.argv[1] == "--generate":
    tmp, nloci, ncols, ntaxa = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    import torch
    from tapir_amd import synth
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    t0 = time.time()
    d = synth.simulate(nloci, ncols, ntaxa, 77, device=dev)
    st = d["states"].cpu().numpy()
    print("generated on %s in %.1f s" % (dev, time.time() - t0), flush=True)
    t0 = time.time()
    aln = os.path.join(tmp, "aln")
    os.mkdir(aln)
    step = 5000
    for l0 in range(0, nloci, step):   # in slices, with a progress line (a silent minute-long loop looks hung)
        l1 = min(nloci, l0 + step)
        off = d["locus_offsets"][l0:l1 + 1]
        tree = synth.write_nexus_dir(aln, st[:, off[0]:off[-1]], off - off[0], d["names"], d["root"], prefix="locus%02d_" % (l0 // step))
        print("  wrote %d files, %.1f s" % (l1, time.time() - t0), flush=True)
    shutil.move(tree, os.path.join(tmp, "tree.newick"))
    sys.exit(0)

nloci, ncols, ntaxa = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
need = nloci * ncols * (ntaxa + 400)          # input text + .rates JSON (~300 B per site) + sqlite
shm_ok = os.path.isdir("/dev/shm") and shutil.disk_usage("/dev/shm").free > 1.5 * need
tmp = tempfile.mkdtemp(dir="/dev/shm" if shm_ok else None)
try:
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--generate", tmp, str(nloci), str(ncols), str(ntaxa)])
    from tapir_amd import cli
    try:
        import torch  # noqa: F401  (stage 1's device-resident optimisers import it lazily: on a fresh box the first import pages
    except ImportError:              #  the wheel in for ~10 s, which is the machine's cost, not the pipeline's; importing does not touch the GPU)
        pass
    out = os.path.join(tmp, "out")
    os.mkdir(out)
    extra = [a for a in sys.argv[4:] if a != "--model-averaging"]
    fixed = [] if "--model-averaging" in sys.argv[4:] else ["--exchangeabilities", "1,1.2,0.8,0.9,1.5,1"]
    t0 = time.time()
    cli.main([os.path.join(tmp, "aln"), os.path.join(tmp, "tree.newick"), "--output", out, "--times", "10,30,50,90",
              "--intervals", "5-15,25-35,45-55,85-95"] + extra + fixed)
    dt = time.time() - t0
    nbytes = sum(os.path.getsize(os.path.join(dp, f)) for dp, _, fs in os.walk(out) for f in fs)
    print("CLI end to end (%s): %.2f s for %d loci x %d columns x %d taxa = %.3g columns/s; %.2f GB written under %s"
          % (" ".join(extra + (["fixed exchangeabilities"] if fixed else ["model averaging"])), dt, nloci, ncols, ntaxa,
             nloci * ncols / dt, nbytes / 1e9, "/dev/shm" if shm_ok else tempfile.gettempdir()))
    print("stages (s): " + ", ".join("%s %.2f" % kv for kv in cli.LAST_TIMINGS.items()))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
