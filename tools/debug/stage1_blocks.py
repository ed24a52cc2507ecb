import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tapir_amd import engine, nexus, pipeline, synth
L, n, nt = (int(x) for x in sys.argv[1:4])
d = synth.simulate(L, n, nt, 77)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
off = d["locus_offsets"]
pi = nexus.base_frequencies_from_histogram(engine.state_histogram(st, off))
for blk in [int(b) for b in sys.argv[4:]] * 2:
    t0 = time.perf_counter()
    pipeline.model_averaged_exchangeabilities(engine, st, off, pi, nt, pin["parent"], pin["blen"], pin["leaf"], pin["T"], [10], [[5, 15]], pin["correction"], block_loci=blk)
    print("block %d: %.2f s" % (blk, time.perf_counter() - t0), flush=True)
