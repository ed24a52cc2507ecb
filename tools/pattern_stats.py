"""Unique site patterns in the synthetic bench workloads (how much HyPhy-style pattern compression would save).
usage: python tools/pattern_stats.py WORKLOAD [LOCI]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tapir_amd import engine, synth

L0, cols, taxa = synth.WORKLOADS[sys.argv[1]][:3]
loci = int(sys.argv[2]) if len(sys.argv) > 2 else L0
d = synth.simulate(loci, cols, taxa, synth.WORKLOAD_SEED[sys.argv[1]], device="cuda")
st = d["states"].cpu().numpy()
t0 = time.time()
pst, poff, w, cmap = engine.compress_columns(st, d["locus_offsets"])
t1 = time.time()
norm = np.where(st == 0, 15, st & 15)
const = (norm == norm[0:1]).all(0)
print("%s: %d loci x %d cols x %d taxa: %d columns -> %d patterns (%.3f), %.1f%% columns constant, call %.3f s (incl. copies)"
      % (sys.argv[1], loci, cols, taxa, st.shape[1], pst.shape[1], pst.shape[1] / st.shape[1], 100 * const.mean(), t1 - t0))
ncc = int((~const).sum())
pc = (pst == pst[0:1]).all(0)
print("non-constant columns %d -> non-constant patterns %d (%.3f)" % (ncc, int((~pc).sum()), (~pc).sum() / max(ncc, 1)))
