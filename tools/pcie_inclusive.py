# PCIe-inclusive rate: host-pointer tphip_run_fused on the C3 shape (numpy in, numpy out)
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tapir_amd import engine, synth
nloci, ncols, ntaxa, times, intervals = synth.WORKLOADS["C3"]
seed = synth.WORKLOAD_SEED["C3"]
tree = synth.yule_tree(ntaxa, seed)
d = synth.simulate(nloci, ncols, ntaxa, seed, device="cuda", tree=tree)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].cpu().numpy()
plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"], pin["T"], times,
                   intervals, correction=pin["correction"])
plan.run_fused(st)
t0 = time.perf_counter(); n = 3
for _ in range(n): plan.run_fused(st)
dt = (time.perf_counter() - t0) / n
print("PCIe-inclusive host-pointer run_fused C3: %.1f ms per pass, %.3g columns/s (pageable numpy buffers)" % (dt * 1e3, st.shape[1] / dt))
