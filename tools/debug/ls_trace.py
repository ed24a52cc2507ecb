# how many backtracking rounds the general model's line search takes (numpy optimiser), per iteration
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tapir_amd import engine, nexus, stage1, synth
L, n, nt = (int(x) for x in sys.argv[1:4])
d = synth.simulate(L, n, nt, 5)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
pi = nexus.base_frequencies_from_histogram(engine.state_histogram(st, d["locus_offsets"]))
plan = engine.Plan(nt, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], pi, np.ones((L, 6)), pin["T"], [1], [[0, 1]], correction=pin["correction"])
s1 = stage1.Stage1(plan, st, pi, pin["parent"], np.asarray(pin["blen"]) / pin["correction"], device_fit=False)
calls = []
orig = s1._grm_value
def value(X, idx):
    calls.append(len(idx))
    return orig(X, idx)
s1._grm_value = value
og = s1._grm_value_and_grad
def vg(X, idx):
    calls.append(-len(idx))
    return og(X, idx)
s1._grm_value_and_grad = vg
ge, gt, gl = s1.fit_grm()
print("sequence of calls (negative = gradient, value = problems evaluated):")
print(calls)
print("iterations", s1.grm_iters)
