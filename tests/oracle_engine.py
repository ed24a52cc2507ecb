"""TEST-ONLY stand-in for tapir_amd.engine, backed by the CPU oracle.

Lets the `-m "not gpu"` suite exercise the host logic above the C ABI (pipeline, CLI, JSON and sqlite
writers) on a machine without a GPU.  It lives under tests/ because only tests may use the oracle; the
product never imports it and has no CPU path."""
import numpy as np

from oracle import oracle as orc

INTEG_QUADPACK, INTEG_CLOSED = 0, 1


def state_histogram(states, locus_offsets, device=0):
    states = np.asarray(states)
    L = len(locus_offsets) - 1
    hist = np.zeros((L, 16), np.int64)
    for l in range(L):
        blk = states[:, locus_offsets[l]:locus_offsets[l + 1]].ravel()
        blk = np.where(blk == 0, 15, blk & 15)
        hist[l] = np.bincount(blk, minlength=16)[:16]
    return hist


def compress_columns(states, locus_offsets, device=0, want_map=True):
    """Stand-in: no compression (every column its own pattern, weight 1)."""
    states = np.asarray(states, np.uint8)
    n = states.shape[1]
    return states, np.asarray(locus_offsets, np.int64), np.ones(n), (np.arange(n) if want_map else None)


class Plan:
    def __init__(self, ntaxa, parent, branch_len, leaf_taxon, locus_offsets, pi, exch, T, times, intervals,
                 correction=1.0, threshold=3, round_decimals=4, integ_mode=0, device=0, cat_rates=None, cat_weights=None,
                 start_rule=0):
        self.cat_rates, self.cat_weights = cat_rates, cat_weights
        # the engine's start_rule: 0 = TPHIP_START_AUTO (the oracle's default), 1 = REFERENCE (oracle mode 2), 2 = PARSIMONY (mode 0)
        self.start_mode = {0: None, 1: 2, 2: 0}[int(start_rule)]
        self.ntaxa = ntaxa
        self.parent = np.asarray(parent, np.int32)
        self.blen = np.asarray(branch_len, np.float64)
        self.leaf = np.asarray(leaf_taxon, np.int32)
        self.off = np.asarray(locus_offsets, np.int64)
        self.nloci = len(self.off) - 1
        self.pi = np.asarray(pi, np.float64).reshape(self.nloci, 4)
        self.exch = np.asarray(exch, np.float64).reshape(self.nloci, 6)
        self.T, self.times = int(T), [int(t) for t in np.asarray(times).reshape(-1)]
        self.intervals = np.asarray(intervals, np.int32).reshape(-1, 2)
        self.correction, self.threshold, self.round_decimals, self.integ_mode = correction, threshold, round_decimals, integ_mode
        self.ncols = int(self.off[-1])
        self.width = self.T + len(self.times) + 2 * len(self.intervals)
        self.chrono_length = float(self.blen[self.parent >= 0].sum())
        for t in self.times:
            if not 0 <= t < self.T:
                raise IndexError("index %d is out of bounds for axis 0 with size %d" % (t, self.T))

    def close(self):
        pass

    class _Cache:
        def release(self):
            pass

    def device_cache(self):
        return Plan._Cache()

    def set_column_weights(self, weights):
        assert weights is None or np.all(np.asarray(weights) == 1.0)

    def locus_loglik(self, states, blen_vecs, cand_locus, cand_exch, cand_vec=None, cand_scale=None, cand_pidx=None,
                     cand_pfac=None, cache=None):
        states = np.asarray(states, np.uint8)
        bv = np.asarray(blen_vecs, np.float64)
        bv = bv.reshape(-1, bv.shape[-1])
        n = len(cand_locus)
        ce = np.asarray(cand_exch, np.float64).reshape(n, 6)
        out = np.empty(n)
        for c in range(n):
            b = bv[c if cand_vec is None else cand_vec[c]].copy()
            if cand_scale is not None:
                b *= cand_scale[c]
            if cand_pidx is not None and cand_pidx[c] >= 0:
                b[cand_pidx[c]] *= cand_pfac[c]
            l = cand_locus[c]
            out[c] = orc.locus_loglik(states[:, self.off[l]:self.off[l + 1]], self.parent, b, self.leaf, self.pi[l], ce[c])
        return out

    def site_rates(self, states):
        states = np.asarray(states, np.uint8)
        out = dict(rate=np.empty(self.ncols), subst=np.empty(self.ncols), lnl=np.empty(self.ncols),
                   flag=np.empty(self.ncols, np.uint8), nres=np.empty(self.ncols, np.int32))
        for l in range(self.nloci):
            sl = slice(self.off[l], self.off[l + 1])
            if sl.stop == sl.start:
                continue
            r = orc.site_rates(states[:, sl], self.parent, self.blen, self.leaf, self.pi[l] / self.pi[l].sum(), self.exch[l], self.cat_rates, self.cat_weights,
                               start_mode=self.start_mode)
            for k in out:
                out[k][sl] = r[k]
        return out

    def pi_tables(self, rates, nres=None):
        rates = np.asarray(rates, np.float64).copy()
        if self.round_decimals >= 0:
            rates = orc.round_dp(rates, self.round_decimals)
        rates = rates / self.correction
        if nres is not None:
            rates[np.asarray(nres) < self.threshold] = np.nan
        tab = np.zeros((self.nloci, self.width))
        n_t, n_i = len(self.times), len(self.intervals)
        for l in range(self.nloci):
            r = rates[self.off[l]:self.off[l + 1]]
            net = orc.net_pi(r, self.T)
            tab[l, :self.T] = net
            tab[l, self.T:self.T + n_t] = net[self.times] if n_t else []
            fin = r[np.isfinite(r)]
            si, se = orc.net_integrals(fin, self.intervals, self.integ_mode)
            tab[l, self.T + n_t:self.T + n_t + n_i] = si
            tab[l, self.T + n_t + n_i:] = se
        return tab

    def run_fused(self, states):
        out = self.site_rates(states)
        out["tables"] = self.pi_tables(out["rate"], out["nres"])
        return out
