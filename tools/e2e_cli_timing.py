# end-to-end CLI timing on a synthetic directory of NEXUS files (ingest + engine + JSON + sqlite)
# usage: python tools/e2e_cli_timing.py LOCI COLS TAXA [--model-averaging] [cli flags ...]
#   --model-averaging: let the CLI estimate the exchangeabilities (HyPhy stage 1) instead of fixing them
import os, sys, time, tempfile, shutil, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tapir_amd import synth, cli
nloci, ncols, ntaxa = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d = synth.simulate(nloci, ncols, ntaxa, 77)
tmp = tempfile.mkdtemp()
aln = os.path.join(tmp, "aln"); os.mkdir(aln); out = os.path.join(tmp, "out"); os.mkdir(out)
t0 = time.time()
tree = synth.write_nexus_dir(aln, d["states"].numpy(), d["locus_offsets"], d["names"], d["root"])
shutil.move(tree, os.path.join(tmp, "tree.newick"))
print("wrote %d files in %.1f s" % (nloci, time.time() - t0))
pr = cProfile.Profile()
t0 = time.time()
pr.enable()
extra = [a for a in sys.argv[4:] if a != "--model-averaging"]
fixed = [] if "--model-averaging" in sys.argv[4:] else ["--exchangeabilities", "1,1.2,0.8,0.9,1.5,1"]
cli.main([aln, os.path.join(tmp, "tree.newick"), "--output", out, "--times", "10,30,50,90", "--intervals", "5-15,25-35,45-55,85-95"] + extra + fixed)
pr.disable()
dt = time.time() - t0
print("CLI end to end: %.2f s for %d loci x %d columns x %d taxa = %.3g columns/s" % (dt, nloci, ncols, ntaxa, nloci * ncols / dt))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
shutil.rmtree(tmp)
