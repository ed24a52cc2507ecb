# GPU: the largest |f'/f''| at the reported maximisers of a whole BASELINE config, with the columns' curvature -- which
# columns set the margin of the full-size residual test.   usage: python tools/debug/residual_gpu.py C4 [TPHIP_LIB path ...]
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tapir_amd import engine, synth
wl = sys.argv[1]
nloci, ncols, ntaxa, times, intervals = synth.WORKLOADS[wl]
seed = synth.WORKLOAD_SEED[wl]
tree = synth.yule_tree(ntaxa, seed)
d = synth.simulate(nloci, ncols, ntaxa, seed, device="cuda", tree=tree, chunk_loci=max(1, (1 << 26) // (ncols * ntaxa)))
pin = synth.plan_inputs(d["root"], d["names"])
st = d["states"]
dev = st.device
plan = engine.Plan(ntaxa, pin["parent"], pin["blen"], pin["leaf"], d["locus_offsets"], d["pi"], d["exch"], pin["T"], times, intervals,
                   correction=pin["correction"], threshold=3, round_decimals=4)
n, W = plan.ncols, plan.width
out = dict(rate=torch.empty(n, dtype=torch.float64, device=dev), subst=torch.empty(n, dtype=torch.float64, device=dev),
           lnl=torch.empty(n, dtype=torch.float64, device=dev), flag=torch.empty(n, dtype=torch.uint8, device=dev),
           nres=torch.empty(n, dtype=torch.int32, device=dev), tables=torch.empty((nloci, W), dtype=torch.float64, device=dev))
ws = torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=dev)
plan.run_dev(st, out["rate"], out["subst"], out["lnl"], out["flag"], out["nres"], out["tables"], ws, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ok = out["flag"] == 0
kappa = torch.from_numpy(plan.models()[3]).to(dev)
loc = torch.arange(nloci, device=dev).repeat_interleave(ncols)
u = torch.zeros(n, dtype=torch.float64, device=dev)
u[ok] = torch.log(out["rate"][ok] / kappa[loc[ok]])
f, g, h = torch.empty_like(u), torch.empty_like(u), torch.empty_like(u)
plan.eval_columns_dev(st, u, f, g, h, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
r = torch.where(ok, (g / h).abs(), torch.zeros_like(g))
top = torch.topk(r, 12)
print("%s: %d optimised columns, evaluations %d (%.3f per optimised column); residual > 1e-7: %d, > 3e-7: %d, > 1e-6: %d"
      % (wl, int(ok.sum()), plan.last_eval_count(), plan.last_eval_count() / int(ok.sum()), int((r > 1e-7).sum()), int((r > 3e-7).sum()), int((r > 1e-6).sum())))
for v, i in zip(top.values.tolist(), top.indices.tolist()):
    print("  residual %.3e  |h| %.3e  |g| %.3e  u %.4f  column %d" % (v, abs(h[i].item()), abs(g[i].item()), u[i].item(), i))

idx = top.indices.cpu().numpy()
os.makedirs("gpurun_out", exist_ok=True)
np.savez("gpurun_out/resid_top_%s.npz" % wl, cols=idx, states=st[:, top.indices].cpu().numpy(), locus=idx // ncols, pi=d["pi"][idx // ncols], exch=d["exch"][idx // ncols],
         parent=np.asarray(pin["parent"]), blen=np.asarray(pin["blen"]), leaf=np.asarray(pin["leaf"]), rate=out["rate"][top.indices].cpu().numpy(),
         resid=top.values.cpu().numpy(), kappa=kappa[loc[top.indices]].cpu().numpy())
