import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tapir_amd import engine, nexus, pipeline, synth
L, n, nt = (int(x) for x in sys.argv[1:4])
d = synth.simulate(L, n, nt, 20261006)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
off = d["locus_offsets"]
def run():
    pi = nexus.base_frequencies_from_histogram(engine.state_histogram(st, off))
    return pipeline.model_averaged_exchangeabilities(engine, st, off, pi, nt, pin["parent"], pin["blen"], pin["leaf"], pin["T"], [10], [[5, 15]], pin["correction"])
run()
t0 = time.perf_counter(); run(); print("second call: %.3f s" % (time.perf_counter() - t0))
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
