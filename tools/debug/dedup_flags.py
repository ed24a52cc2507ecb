import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from tapir_amd import engine, newick, nexus
from tapir_amd.compute import correct_tree
g = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden")
names, st = nexus.read_states(os.path.join(g, "chr1_918.nex"))
root = newick.read_tree(os.path.join(g, "Euteleost.tree"))
depth, factor = correct_tree(root)
parent, blen, leaf = newick.to_arrays(root, names)
kat = json.load(open(os.path.join(g, "chr1_918_phydesign_rates.json")))
pi = np.array(kat["freqs_ACGT"]); exch = np.array([kat[k] for k in ("AC", "AG", "AT", "CG", "CT", "GT")])
out = {}
for mode in (1, 2):
    plan = engine.Plan(5, parent, blen, leaf, [0, 226], [pi], [exch], 174, [10], [[0, 10]], correction=factor, pattern_dedup=mode)
    out[mode] = plan.site_rates(st)
    print("mode", mode, "evals", plan.last_eval_count())
    plan.close()
bad = np.flatnonzero(out[2]["flag"] == 64)
print("flag 64 at", bad.tolist())
for c in bad[:10]:
    same = [int(k) for k in range(226) if (st[:, k] == st[:, c]).all()]
    print(c, "pattern", st[:, c].tolist(), "same pattern cols", same, "their flags on/off", out[2]["flag"][same].tolist(), out[1]["flag"][same].tolist())
for c in bad[:4]:
    same = [int(k) for k in range(226) if (st[:, k] == st[:, c]).all()]
    print(c, "rates on", out[2]["rate"][same].tolist(), "off", out[1]["rate"][same].tolist(), "lnl on", out[2]["lnl"][same].tolist())
