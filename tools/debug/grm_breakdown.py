# where the general model's device-resident fit spends its time (synchronised timers around the objective calls)
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tapir_amd import engine, nexus, stage1, stage1_device, synth
L, n, nt = (int(x) for x in sys.argv[1:4])
d = synth.simulate(L, n, nt, 5)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].numpy()
pi = nexus.base_frequencies_from_histogram(engine.state_histogram(st, d["locus_offsets"]))
plan = engine.Plan(nt, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], pi, np.ones((L, 6)), pin["T"], [1], [[0, 1]], correction=pin["correction"])
T = {"value": [0.0, 0, 0], "value_and_grad": [0.0, 0, 0], "escape": [0.0, 0, 0]}
for name in T:
    orig = getattr(stage1_device.DeviceGrmFitter, name)
    def wrap(orig=orig, name=name):
        def f(self, X, idx, *a):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            out = orig(self, X, idx, *a)
            torch.cuda.synchronize(); T[name][0] += time.perf_counter() - t0; T[name][1] += 1; T[name][2] += X.shape[0]
            return out
        return f
    setattr(stage1_device.DeviceGrmFitter, name, wrap())
for rep in range(2):
    for k in T: T[k][:] = [0.0, 0, 0]
    s1 = stage1.Stage1(plan, st, pi, pin["parent"], np.asarray(pin["blen"]) / pin["correction"], device_fit="always")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    s1.fit_grm()
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s1.close()
print("general model %.3f s, iterations max %d" % (dt, s1.grm_iters.max()))
for k, v in T.items():
    print("  %-15s %.3f s in %d calls, %d problems" % (k, v[0], v[1], v[2]))
print("  optimiser's own tensor work: %.3f s" % (dt - sum(v[0] for v in T.values())))
