# PCIe-inclusive rate: host-pointer tphip_run_fused on the C3 shape (numpy in, numpy out), with pageable buffers and
# with pinned ones (engine.pinned_empty: direct DMA, per-column results copied while the PI kernels run).
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tapir_amd import engine, synth
wl = sys.argv[1] if len(sys.argv) > 1 else "C3"
nloci, ncols, ntaxa, times, intervals = synth.WORKLOADS[wl]
seed = synth.WORKLOAD_SEED[wl]
tree = synth.yule_tree(ntaxa, seed)
d = synth.simulate(nloci, ncols, ntaxa, seed, device="cuda", tree=tree)
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"].cpu().numpy()
plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"], pin["T"], times,
                   intervals, correction=pin["correction"])
ref = plan.run_fused(st)
n = 5
t0 = time.perf_counter()
for _ in range(n): plan.run_fused(st)
dt = (time.perf_counter() - t0) / n
print("PCIe-inclusive host-pointer run_fused %s: %.1f ms per pass, %.3g columns/s (pageable numpy buffers)" % (wl, dt * 1e3, st.shape[1] / dt))
stp = engine.pinned_empty(st.shape, np.uint8)
stp[...] = st
got = plan.run_fused(stp, pinned=True)
for k in ref:
    assert np.array_equal(ref[k], got[k], equal_nan=True), k
t0 = time.perf_counter()
for _ in range(n): plan.run_fused(stp, pinned=True)
dt2 = (time.perf_counter() - t0) / n
# the allocation of the pinned result arrays is part of run_fused(pinned=True); time the bare library call too
lib, h = plan._lib, plan._h
t0 = time.perf_counter()
for _ in range(n):
    engine._check(lib.tphip_run_fused(h, stp.ctypes.data, got["rate"].ctypes.data, got["subst"].ctypes.data, got["lnl"].ctypes.data,
                                      got["flag"].ctypes.data, got["nres"].ctypes.data, got["tables"].ctypes.data))
dt3 = (time.perf_counter() - t0) / n
print("  pinned buffers: %.1f ms per pass incl. allocating pinned result arrays; %.1f ms with the buffers reused, %.3g columns/s"
      % (dt2 * 1e3, dt3 * 1e3, st.shape[1] / dt3))

# the pipeline of locus groups behind the pinned path, by number of groups (1 = off)
for k in (1, 2, 3, 4, 6):
    os.environ["TPHIP_HOST_SPLIT"] = str(k)
    pk = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"], pin["T"], times,
                     intervals, correction=pin["correction"])
    out = pk.run_fused(stp, pinned=True)
    for key in ref:
        assert np.array_equal(ref[key], out[key], equal_nan=True), (k, key)
    t0 = time.perf_counter()
    for _ in range(n):
        engine._check(lib.tphip_run_fused(pk._h, stp.ctypes.data, out["rate"].ctypes.data, out["subst"].ctypes.data, out["lnl"].ctypes.data,
                                          out["flag"].ctypes.data, out["nres"].ctypes.data, out["tables"].ctypes.data))
    print("  %d locus group(s): %.1f ms per pass (bit-identical results)" % (k, (time.perf_counter() - t0) / n * 1e3))
    pk.close()
del os.environ["TPHIP_HOST_SPLIT"]
