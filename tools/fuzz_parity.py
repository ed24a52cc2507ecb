"""Randomised GPU-vs-oracle parity sweep (not part of the test suite: minutes of CPU oracle time).
Random trees (binary and with polytomies), taxa 2..90, ragged loci, gaps / IUPAC codes, extreme rates and frequencies.
usage: python tools/fuzz_parity.py [NCASES] [SEED] [MAX_TAXA] [ZERO_FRACTION] [EQUAL_CHERRY_FRACTION]

EQUAL_CHERRY_FRACTION (default 0.5): share of the two-tip joins whose tips get the SAME branch length, as in a
chronogram -- those are the pairs site_rate_kernel executes as one fused CHERRY op.

ZERO_FRACTION > 0 sets that share of the internal branches to length 0 (exploratory): together with very short tips and
noisy data it produces columns whose likelihood is at the rounding level of its own terms (a change would have to
happen on a zero-length branch); there an eigen-decomposition implementation returns noise -- this one, the oracle and
HyPhy alike -- and stage-1 values differ from the oracle by 1e-8 .. 1e-5 relative on such loci."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import oracle as orc
from tapir_amd import engine, newick

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
MAX_TAXA = int(sys.argv[3]) if len(sys.argv) > 3 else 90          # optional third argument: largest tree
ZERO_BRANCHES = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0   # optional fourth: fraction of zero-length INTERNAL branches
EQUAL_CHERRIES = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5  # optional fifth: fraction of cherries with equal tip lengths
CODES = np.array([1, 2, 4, 8, 15, 15, 5, 10, 3, 12, 7, 0], dtype=np.uint8)


def random_tree(n):
    """random rooted tree as newick, some polytomies"""
    def length():
        return 0.0 if rng.random() < ZERO_BRANCHES else rng.gamma(1.0, 1.0) * 10 ** rng.uniform(-3, 0.5)
    # tips keep positive lengths: two zero-length tips with different states make a column impossible (L = 0), where
    # every implementation, HyPhy included, returns rounding noise
    nodes = ["t%d:%g" % (i, max(length(), 1e-4)) for i in range(n)]
    while len(nodes) > 1:
        k = 2 if (len(nodes) < 3 or rng.random() < 0.85) else min(len(nodes), int(rng.integers(3, 5)))
        idx = rng.choice(len(nodes), size=k, replace=False)
        kids = [nodes[i] for i in idx]
        if k == 2 and all("(" not in x for x in kids) and rng.random() < EQUAL_CHERRIES:
            kids[1] = kids[1].rsplit(":", 1)[0] + ":" + kids[0].rsplit(":", 1)[1]   # a cherry of a chronogram
        nodes = [x for j, x in enumerate(nodes) if j not in set(idx.tolist())]
        nodes.append("(%s):%g" % (",".join(kids), length()))
    return nodes[0].rsplit(":", 1)[0] + ";"


bad = 0
for case in range(ncases):
    nt = int(rng.integers(2, MAX_TAXA + 1))
    root = newick.parse(random_tree(nt))
    names = [x.name for x in newick.leaves(root)]
    parent, blen, leaf = newick.to_arrays(root, names)
    L = int(rng.integers(1, 4))
    sizes = rng.integers(0, 120, L)
    off = np.concatenate([[0], np.cumsum(sizes)])
    if off[-1] == 0:
        continue
    ncol = int(off[-1])
    base = (1 << rng.integers(0, 4, size=(nt, ncol))).astype(np.uint8)
    sticky = rng.random(ncol) < rng.uniform(0.2, 0.95)
    base[:, sticky] = base[0, sticky]
    noise = rng.random((nt, ncol)) < rng.uniform(0.0, 0.3)
    st = np.where(noise, rng.choice(CODES, size=(nt, ncol)), base).astype(np.uint8)
    pi = rng.dirichlet(np.full(4, rng.choice([0.5, 3.0, 30.0])), size=L)
    pi = np.maximum(pi, 1e-3); pi /= pi.sum(1, keepdims=True)
    exch = np.exp(rng.normal(0, rng.choice([0.3, 1.5]), (L, 6)))
    try:
        plan = engine.Plan(nt, parent, blen, leaf, off, pi, exch, 5, [1], [[0, 2]])
    except engine.TphipError as e:
        print("case", case, "plan refused:", e); bad += 1; continue
    got = plan.site_rates(st)
    msg = []
    for l in range(L):
        sl = slice(off[l], off[l + 1])
        if sl.stop == sl.start:
            continue
        ref = orc.site_rates(st[:, sl], parent, blen, leaf, pi[l], exch[l])
        if not np.array_equal(got["flag"][sl], ref["flag"]):
            w = np.flatnonzero(got["flag"][sl] != ref["flag"])[:4]
            msg.append("flags differ at %s: gpu %s oracle %s, rates gpu %s oracle %s, lnl gpu %s oracle %s" % (
                w, got["flag"][sl][w], ref["flag"][w], got["rate"][sl][w], ref["rate"][w], got["lnl"][sl][w], ref["lnl"][w]))
            continue
        ok = (ref["flag"] == 0) | (ref["flag"] == 3)
        rel = np.abs(got["rate"][sl][ok] - ref["rate"][ok]) / np.maximum(np.abs(ref["rate"][ok]), 1e-12)
        dl = np.abs(got["lnl"][sl] - ref["lnl"])
        okl = np.isfinite(ref["lnl"])
        if rel.size and rel.max() > 1e-6 and dl[ok][rel > 1e-6].max() > 1e-9:
            msg.append("rate rel %.2e (lnl diff there %.2e)" % (rel.max(), dl[ok][rel > 1e-6].max()))
        if dl[okl].size and dl[okl].max() > 1e-8 * max(1.0, np.abs(ref["lnl"][okl]).max()):
            msg.append("lnl diff %.2e" % dl[okl].max())
    # stage-1 kernels at a random point
    cand = rng.integers(0, L, 3)
    ce = np.exp(rng.normal(0, 0.5, (3, 6)))
    cb = np.asarray(blen)[None, :] * np.exp(rng.normal(0, 0.5, (3, len(parent)))) * 10 ** rng.uniform(-2, 0)
    val = plan.locus_loglik(st, cb, cand, ce)
    lnl, dex, dlt, sdl = plan.locus_gradient(st, cb, cand, ce)
    for c in range(3):
        l = int(cand[c]); sl = slice(off[l], off[l + 1])
        ref = orc.locus_loglik(st[:, sl], parent, cb[c], leaf, pi[l], ce[c]) if sl.stop > sl.start else 0.0
        if not np.isfinite(ref):
            continue
        for name, v in (("value", val[c]), ("grad-lnl", lnl[c])):
            if abs(v - ref) > 1e-9 * max(1.0, abs(ref)):
                msg.append("locus %s %.12g vs oracle %.12g" % (name, v, ref))
        if sl.stop > sl.start:
            b = int(rng.choice(np.flatnonzero(np.asarray(parent) >= 0)))
            h = 1e-5
            up, dn = cb[c].copy(), cb[c].copy(); up[b] *= np.exp(h); dn[b] *= np.exp(-h)
            fd = (orc.locus_loglik(st[:, sl], parent, up, leaf, pi[l], ce[c]) - orc.locus_loglik(st[:, sl], parent, dn, leaf, pi[l], ce[c])) / (2 * h)
            if abs(dlt[c, b] - fd) > 2e-5 * max(1.0, abs(fd), np.abs(dlt[c]).max()):
                msg.append("dlogt[%d] %.8g vs fd %.8g" % (b, dlt[c, b], fd))
    # PI tables on random rates (zeros, NaNs, tiny and huge values) with this case's loci as the column ranges
    T = int(rng.integers(2, 60))
    times = sorted(set(int(x) for x in rng.integers(0, T, int(rng.integers(1, 4)))))
    ivs = []
    for _ in range(int(rng.integers(1, 4))):
        a = int(rng.integers(0, 80)); ivs.append([a, a + int(rng.integers(1, 60))])
    mode = int(rng.integers(0, 2))
    pp = engine.Plan(nt, parent, blen, leaf, off, pi, exch, T, times, ivs, correction=1.0, threshold=0, round_decimals=-1,
                     integ_mode=mode)
    rates = rng.gamma(0.5, 1.0, ncol) * 10 ** rng.uniform(-6, 0.5, ncol)
    rates[rng.random(ncol) < 0.1] = 0.0
    rates[rng.random(ncol) < 0.1] = np.nan
    tab = pp.pi_tables(rates, None)
    pp.close()
    for l in range(L):
        r = rates[off[l]:off[l + 1]]
        fin = r[~np.isnan(r)]
        net = orc.net_pi(fin, T) if fin.size else np.zeros(T)
        si, se = orc.net_integrals(fin, ivs, mode) if fin.size else (np.zeros(len(ivs)), np.zeros(len(ivs)))
        want = np.concatenate([net, net[times], si])
        gotrow = tab[l][:T + len(times) + len(ivs)]
        err = np.abs(gotrow - want) / np.maximum(np.abs(want), 1e-300)
        err = np.where(np.abs(want) < 1e-300, np.abs(gotrow), err)
        if err.size and err.max() > 1e-9:
            msg.append("PI table locus %d rel %.2e at %d (T %d, times %s, ivs %s, mode %d)" % (l, err.max(), int(err.argmax()), T, times, ivs, mode))
    # rate mixture (opt-in extension) on the first locus
    if off[1] > off[0] and case % 3 == 0:
        K = int(rng.integers(2, 6))
        cr = np.exp(rng.normal(0, 0.7, K)); cw = rng.dirichlet(np.ones(K))
        pm = engine.Plan(nt, parent, blen, leaf, off[:2], pi[:1], exch[:1], 5, [1], [[0, 2]], cat_rates=cr, cat_weights=cw)
        gm = pm.site_rates(st[:, :off[1]])
        pm.close()
        rm = orc.site_rates(st[:, :off[1]], parent, blen, leaf, pi[0], exch[0], cr, cw)
        if not np.array_equal(gm["flag"], rm["flag"]):
            msg.append("mixture flags differ at %s" % np.flatnonzero(gm["flag"] != rm["flag"])[:4])
        else:
            okm = (rm["flag"] == 0) | (rm["flag"] == 3)
            relm = np.abs(gm["rate"][okm] - rm["rate"][okm]) / np.maximum(np.abs(rm["rate"][okm]), 1e-12)
            dlm = np.abs(gm["lnl"] - rm["lnl"])
            if relm.size and relm.max() > 1e-6 and dlm[okm][relm > 1e-6].max() > 1e-9:
                msg.append("mixture rate rel %.2e" % relm.max())
    # site-pattern compression against numpy
    pst, poff, pw, cmap = engine.compress_columns(st, off)
    norm = np.where(st == 0, 15, st & 15)
    if pw.sum() != ncol or not np.array_equal(pst[:, cmap], norm):
        msg.append("compress_columns: map does not reproduce the columns")
    for l in range(L):
        uniq = np.unique(norm[:, off[l]:off[l + 1]], axis=1).shape[1] if off[l + 1] > off[l] else 0
        if poff[l + 1] - poff[l] != uniq:
            msg.append("compress_columns: locus %d has %d patterns, numpy finds %d" % (l, poff[l + 1] - poff[l], uniq))
    plan.close()
    if msg:
        bad += 1
        print("case %d (taxa %d, loci %s): %s" % (case, nt, sizes.tolist(), "; ".join(msg)))
print("%d cases, %d with discrepancies" % (ncases, bad))
