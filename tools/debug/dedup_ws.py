import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tapir_amd import engine, newick, nexus
from tapir_amd.compute import correct_tree
g = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
names, st = nexus.read_states(os.path.join(g, "chr1_918.nex"))
root = newick.read_tree(os.path.join(g, "Euteleost.tree"))
depth, factor = correct_tree(root)
parent, blen, leaf = newick.to_arrays(root, names)
kat = json.load(open(os.path.join(g, "chr1_918_phydesign_rates.json")))
pi = np.array(kat["freqs_ACGT"]); exch = np.array([kat[k] for k in ("AC", "AG", "AT", "CG", "CT", "GT")])
T, n_i, n, L = 174, 1, 226, 1
plan = engine.Plan(5, parent, blen, leaf, [0, n], [pi], [exch], T, [10], [[0, 10]], correction=factor, pattern_dedup=2)
dev = "cuda"
d_st = torch.from_numpy(st).to(dev)
o = [torch.zeros(n, dtype=torch.float64, device=dev) for _ in range(3)]
fl = torch.zeros(n, dtype=torch.uint8, device=dev); nr = torch.zeros(n, dtype=torch.int32, device=dev)
ws = torch.zeros(plan.workspace_bytes, dtype=torch.uint8, device=dev)
plan.site_rates_dev(d_st, o[0], o[1], o[2], fl, nr, ws, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
al = lambda x: (x + 255) // 256 * 256
off = 0
lay = {}
for name, size in [("work_cols", 4 * n), ("work_count", 4 * L), ("work_prefix", 8 * (L + 1)), ("slice_prefix", 8 * (L + 1)),
                   ("partial", 8 * 1 * (T + 2 * n_i)), ("packed", 4 * 1 * n), ("hash", 8 * n), ("dup_of", 4 * n), ("tab_key", 16 * n),
                   ("tab_val", 8 * n), ("on", 4 * L)]:
    lay[name] = off
    off = al(off + size)
print("ws bytes", plan.workspace_bytes, "computed", off + 256)
w = ws.cpu().numpy()
dup = w[lay["dup_of"]:lay["dup_of"] + 4 * n].view(np.int32)
on = w[lay["on"]:lay["on"] + 4].view(np.int32)
print("on", on, "dup_of >= 0:", np.flatnonzero(dup >= 0).tolist()[:20], dup[[21, 28, 32]])
print("flags", fl.cpu().numpy()[[5, 21, 26, 28, 32]], "work_count", w[lay["work_count"]:lay["work_count"] + 4].view(np.int32))
