"""CPU-only timing of the tail of the command line (round/cull, .rates files through the pool, sqlite) with a stub engine
that returns random site results: what does running the sqlite inserts beside the writers save?"""
import os, sys, time, shutil, types, numpy as np
sys.path.insert(0, '/root/repo')
from tapir_amd import pipeline, synth, newick, db
L, S, NT = int(sys.argv[1]), 1000, 16
work = "/tmp/overlap_bench"; shutil.rmtree(work, ignore_errors=True); os.makedirs(work + "/aln"); os.makedirs(work + "/out")
d = synth.simulate(4, S, NT, 3)
tree = synth.write_nexus_dir(work + "/aln", d["states"].numpy(), d["locus_offsets"], d["names"], d["root"])
one = sorted(f for f in os.listdir(work + "/aln") if f.endswith(".nex"))[0]
files = []
for i in range(L):
    p = work + "/aln/l%06d.nex" % i
    os.link(work + "/aln/" + one, p) if not os.path.exists(p) else None
    files.append(p)
root = newick.read_tree(tree, "newick"); leaf_names = [x.name for x in newick.leaves(root)]
parent, blen, leaf = newick.to_arrays(root, leaf_names)
T, times, intervals = 100, [10, 20, 30, 40], [(0, 10), (10, 50)]
W = T + len(times) + 2 * len(intervals)
rng = np.random.default_rng(1)
class Plan:
    def __init__(self, ntaxa, parent, blen, leaf, off, *a, **k): self.n = int(off[-1]); self.L = len(off) - 1
    def close(self): pass
    def run_fused(self, states):
        n = self.n
        return dict(rate=rng.random(n) * 3, nres=np.full(n, NT), subst=rng.integers(0, 9, n).astype(float), lnl=-rng.random(n) * 30,
                    tables=rng.random((self.L, W)))
eng = types.SimpleNamespace(Plan=Plan, state_histogram=lambda st, off, device=0: np.tile(np.arange(16), (len(off) - 1, 1)))
def store(pis):
    conn, c = db.create_probe_db(work + "/out/x.sqlite"); db.insert_pi_data(conn, c, pis); conn.commit(); c.close(); conn.close()
with pipeline.HostPool(int(sys.argv[2])) as pool:
    for mode in ("sequential", "overlapped"):
        t0 = time.perf_counter()
        pis, out = pipeline.run_alignments(files, leaf_names, parent, blen, leaf, T, times, intervals, 1.0, 3, np.ones(6),
                                           output_dir=work + "/out", engine_mod=eng, pool=pool,
                                           during_write=store if mode == "overlapped" else None)
        t1 = time.perf_counter()
        if not out.get("during_write_done"): store(pis)
        t2 = time.perf_counter()
        print(mode, "total %.2f" % (t2 - t0), "store after %.2f" % (t2 - t1), {k: round(v, 2) for k, v in out["timings"].items()})
shutil.rmtree(work)
