"""Seeded synthetic alignments of the BASELINE.json shapes.

One ultrametric pure-birth tree per configuration (root-to-tip = 100 time units, so T = 100 net-PI points),
per-locus GTR parameters, Gamma-distributed true site rates (the "+Gamma" of the config names: the
reference itself estimates one free rate per site), columns simulated down the tree under
GTR(pi_l, R_l) at the site's rate, 5 % of cells replaced by gaps.

The generator runs in torch so the big shapes can be produced directly in HBM (device="cuda"); the random
streams of the CPU and GPU generators differ, so parity tests and the CPU baseline always take the bytes
one generator made and hand the SAME bytes to both sides.
"""

import numpy as np

from . import newick

WORKLOADS = {
    # name: (loci, columns per locus, taxa, times, intervals)
    "C2": (1000, 500, 16, [10, 30, 50, 90], [[5, 15], [25, 35], [45, 55], [85, 95]]),
    "C3": (100, 50000, 64, [10, 30, 50, 90], [[5, 15], [25, 35], [45, 55], [85, 95]]),
    "C4": (50000, 1000, 64, [10, 30, 50, 90], [[5, 15], [25, 35], [45, 55], [85, 95]]),
    "C5": (10000, 2000, 256, [10, 30, 50, 90], [[3 * k, 3 * k + 4] for k in range(32)]),
}
WORKLOAD_SEED = {"C2": 20261005, "C3": 20261006, "C4": 20261007, "C5": 20261008}


def yule_tree(ntaxa, seed, depth=100.0):
    """Pure-birth topology with sorted-uniform node heights scaled to `depth`; taxa named t0000...

    Returns (root Node, names).  Built backwards: start from ntaxa lineages at height 0 and merge two
    random lineages at each successive node height."""
    rng = np.random.default_rng(seed)
    heights = np.sort(rng.uniform(0.0, 1.0, ntaxa - 1))
    heights = heights / heights[-1] * depth  # the last coalescence is the root at height `depth`
    names = ["t%04d" % i for i in range(ntaxa)]
    active = []
    for nm in names:
        n = newick.Node()
        n.name = nm
        active.append((n, 0.0))
    for h in heights:
        i, j = sorted(rng.choice(len(active), size=2, replace=False))
        (a, ha), (b, hb) = active[i], active[j]
        p = newick.Node()
        a.length, b.length = float(h - ha), float(h - hb)
        a.parent = b.parent = p
        p.children = [a, b]
        active = [x for k, x in enumerate(active) if k not in (i, j)] + [(p, h)]
    return active[0][0], names


def correction_factor(root):
    """tapir/compute.py:62-67 on a parsed tree."""
    nleaves = len(newick.leaves(root))
    mean_bl = newick.tree_length(root) / (2 * nleaves - 3)
    string_len = len(str(int(mean_bl + 0.5)))
    return 10 ** string_len if string_len > 1 else 1


def locus_parameters(nloci, seed):
    """pi ~ Dirichlet(10,10,10,10); (AC,AT,CG,CT,GT) ~ LogNormal(0, 0.5), AG = 1.
    Returns pi [L,4], exch [L,6] in AC,AG,AT,CG,CT,GT order."""
    rng = np.random.default_rng(seed + 1)
    pi = rng.dirichlet([10.0] * 4, size=nloci)
    r = rng.lognormal(0.0, 0.5, size=(nloci, 5))
    exch = np.stack([r[:, 0], np.ones(nloci), r[:, 1], r[:, 2], r[:, 3], r[:, 4]], axis=1)
    return pi, exch


def _eigen(pi, exch):
    """numpy eigen-systems of Q = R o pi for all loci: lam [L,4], U [L,4,4], Ui [L,4,4]."""
    L = pi.shape[0]
    R = np.zeros((L, 4, 4))
    idx = [(0, 1), (0, 2), (0, 3), (1, 2), (1, 3), (2, 3)]
    for k, (i, j) in enumerate(idx):
        R[:, i, j] = R[:, j, i] = exch[:, k]
    Q = R * pi[:, None, :]
    Q[:, range(4), range(4)] = -Q.sum(axis=2)
    sq = np.sqrt(pi)
    S = sq[:, :, None] * Q / sq[:, None, :]
    S = 0.5 * (S + np.transpose(S, (0, 2, 1)))
    lam, V = np.linalg.eigh(S)
    U = V / sq[:, :, None]
    Ui = np.transpose(V, (0, 2, 1)) * sq[:, None, :]
    return lam, U, Ui


def simulate(nloci, ncols, ntaxa, seed, device="cpu", gap_frac=0.05, rate_shape=0.5, rate_mean=0.004,
             tree=None, chunk_loci=None):
    """Simulate `nloci` loci x `ncols` columns x `ntaxa` taxa.

    Returns dict(states [ntaxa, nloci*ncols] uint8 torch tensor on `device`, locus_offsets int64[L+1],
    pi [L,4], exch [L,6], root, names, true_rates (numpy, per time unit))."""
    import torch
    if tree is None:
        root, names = yule_tree(ntaxa, seed)
    else:
        root, names = tree
    pi, exch = locus_parameters(nloci, seed)
    lam, U, Ui = _eigen(pi, exch)
    order = newick.postorder(root)
    leaf_row = {id(n): names.index(n.name) for n in order if n.is_leaf()}
    dev = torch.device(device)
    total = nloci * ncols
    states = torch.empty((ntaxa, total), dtype=torch.uint8, device=dev)
    rates_out = np.empty(total)
    if chunk_loci is None:
        chunk_loci = max(1, min(nloci, (1 << 22) // max(1, ncols)))
    gen = torch.Generator(device=dev)
    for l0 in range(0, nloci, chunk_loci):
        l1 = min(nloci, l0 + chunk_loci)
        n = (l1 - l0) * ncols
        gen.manual_seed(seed * 1000003 + l0)
        rng = np.random.default_rng(seed * 7919 + l0)
        lam_site = rng.gamma(rate_shape, rate_mean / rate_shape, size=n)  # substitution rate per time unit
        rates_out[l0 * ncols:l1 * ncols] = lam_site
        rate_t = torch.from_numpy(lam_site).to(dev)
        loc = torch.arange(l0, l1, device=dev).repeat_interleave(ncols) - l0
        lam_t = torch.from_numpy(lam[l0:l1]).to(dev)[loc]            # [n,4]
        U_t = torch.from_numpy(U[l0:l1]).to(dev)[loc]                # [n,4,4]
        Ui_t = torch.from_numpy(Ui[l0:l1]).to(dev)[loc]
        pi_t = torch.from_numpy(pi[l0:l1]).to(dev)[loc]
        node_state = {}
        u = torch.rand((n,), generator=gen, dtype=torch.float64, device=dev)
        cdf = torch.cumsum(pi_t, dim=1)
        node_state[id(root)] = torch.clamp((u[:, None] > cdf).sum(dim=1), max=3)
        # pre-order: parents before children
        for node in reversed(order):
            if node is root:
                continue
            ps = node_state[id(node.parent)]
            t = float(node.length or 0.0)
            e = torch.exp(lam_t * (rate_t * t)[:, None])                                  # [n,4]
            Urow = torch.gather(U_t, 1, ps[:, None, None].expand(n, 1, 4)).squeeze(1)     # U[ps, :]
            P = torch.einsum("nk,nkj->nj", Urow * e, Ui_t).clamp_min(0.0)                 # P[ps, :]
            cdf = torch.cumsum(P, dim=1)
            u = torch.rand((n,), generator=gen, dtype=torch.float64, device=dev)
            st = torch.clamp((u[:, None] * cdf[:, 3:4] > cdf).sum(dim=1), max=3)
            node_state[id(node)] = st
            if node.is_leaf():
                g = torch.rand((n,), generator=gen, dtype=torch.float64, device=dev) < gap_frac
                mask = torch.where(g, torch.full_like(st, 15), torch.bitwise_left_shift(torch.ones_like(st), st))
                states[leaf_row[id(node)], l0 * ncols:l1 * ncols] = mask.to(torch.uint8)
        del node_state
    offsets = np.arange(nloci + 1, dtype=np.int64) * ncols
    return dict(states=states, locus_offsets=offsets, pi=pi, exch=exch, root=root, names=names, true_rates=rates_out)


def plan_inputs(root, names):
    """Tree arrays the engine wants for a synthetic tree: depth, correction factor and the post-order
    arrays with branch lengths already divided by the factor (tapir/compute.py:59-74)."""
    depth = newick.distance_from_tip(root)
    factor = correction_factor(root)
    parent, blen, leaf = newick.to_arrays(root, names)
    return dict(depth=depth, T=int(depth), correction=factor, parent=parent, blen=blen / factor, leaf=leaf)


_CODE2CHAR = {1: "A", 2: "C", 4: "G", 8: "T", 15: "-", 5: "R", 10: "Y", 6: "S", 9: "W", 12: "K", 3: "M",
              14: "B", 13: "D", 11: "H", 7: "V"}


def write_nexus_dir(directory, states, locus_offsets, names, root, prefix="locus"):
    """Write a simulated batch as tapir's input: one NEXUS alignment per locus plus `tree.newick`
    (used by end-to-end tests and the ingest/egress timing)."""
    import os
    lut = np.zeros(16, dtype="S1")
    for k, v in _CODE2CHAR.items():
        lut[k] = v.encode()
    states = np.asarray(states)
    chars = lut[states]  # [ntaxa, ncols] bytes
    for l in range(len(locus_offsets) - 1):
        a, b = int(locus_offsets[l]), int(locus_offsets[l + 1])
        with open(os.path.join(directory, "%s%05d.nex" % (prefix, l)), "w") as fh:
            fh.write("#NEXUS\nbegin data;\n\tdimensions ntax=%d nchar=%d;\n\tformat datatype=dna missing=? gap=-;\nmatrix\n"
                     % (len(names), b - a))
            for t, nm in enumerate(names):
                fh.write("%s %s\n" % (nm, chars[t, a:b].tobytes().decode()))
            fh.write(";\nend;\n")
    tree_path = os.path.join(directory, "tree.newick")
    with open(tree_path, "w") as fh:
        fh.write(newick.write(root) + "\n")
    return tree_path
