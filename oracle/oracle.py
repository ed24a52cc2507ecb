"""CPU parity oracle (Python side) -- TEST INFRASTRUCTURE ONLY.

Restates the reference hot path on the CPU so GPU results can be checked against it:

* numpy restatements of tapir/compute.py (cited per function below); the interval integrals call
  scipy.integrate.quad exactly as the reference does (tapir/compute.py:50-52) -- scipy is a third-party
  library present on the GPU box, not reference code;
* ctypes wrappers over oracle/tapir_oracle.c (site-rate ML = HyPhy stage 2, the QUADPACK dqagse
  restatement, PI sums) for sizes where Python loops would be too slow.

Pinned by (tests/test_oracle_golden.py): the reference's own golden vectors (PhyDesign known answers
test_compute.py:44-71, the five .npy files, chr1_918.subsmodel.phydesign.rates) and outputs of the
reference's compute.py captured by tests/golden/make_golden.py.

Nothing in tapir_amd/ imports this module.
"""
import ctypes
import os
import re
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_c_dp = ctypes.POINTER(ctypes.c_double)
_dp_t = _c_dp
_c_u8p = ctypes.POINTER(ctypes.c_uint8)
_c_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force=False):
    """Compile oracle/tapir_oracle.c -> oracle/libtapir_oracle.so with gcc (plain C, no GPU)."""
    so = os.path.join(_HERE, "libtapir_oracle.so")
    src = os.path.join(_HERE, "tapir_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-ffp-contract=off", "-shared", "-o", so, src, "-lm"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.orc_townsend_pi.restype = ctypes.c_double
        L.orc_townsend_pi.argtypes = [ctypes.c_double, ctypes.c_double]
        L.orc_integral_closed.restype = ctypes.c_double
        L.orc_integral_closed.argtypes = [ctypes.c_double] * 3
        L.orc_quad_townsend.restype = ctypes.c_int
        L.orc_quad_townsend.argtypes = [ctypes.c_double] * 3 + [_c_dp, _c_dp, _c_i32p]
        L.orc_net_pi.restype = None
        L.orc_net_pi.argtypes = [_c_dp, ctypes.c_int64, ctypes.c_int32, _c_dp]
        L.orc_net_integrals.restype = None
        L.orc_net_integrals.argtypes = [_c_dp, ctypes.c_int64, _c_i32p, ctypes.c_int32, ctypes.c_int32, _c_dp, _c_dp]
        L.orc_site_rates.restype = ctypes.c_int64
        L.orc_site_rates.argtypes = [_c_u8p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _c_i32p, _c_dp, _c_i32p,
                                     _c_dp, _c_dp, _c_dp, _c_dp, _c_dp, _c_u8p, _c_i32p]
        L.orc_column_curve.restype = None
        L.orc_column_curve.argtypes = [_c_u8p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _c_i32p, _c_dp, _c_i32p,
                                       _c_dp, _c_dp, ctypes.c_int64, _c_dp, ctypes.c_int32, _c_dp, _c_dp, _c_dp]
        L.orc_gtr_eigen.restype = None
        L.orc_gtr_eigen.argtypes = [_c_dp] * 6
        L.orc_locus_loglik.restype = ctypes.c_double
        L.orc_locus_loglik.argtypes = [_c_u8p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, _c_i32p, _dp_t, _c_i32p, _dp_t, _dp_t]
        L.orc_informative_counts.restype = None
        L.orc_informative_counts.argtypes = [_c_u8p, ctypes.c_int64, ctypes.c_int32, _c_i32p]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(_c_dp)


# --------------------------------------------------------------------------------------------------
# numpy restatements of tapir/compute.py
# --------------------------------------------------------------------------------------------------

def get_townsend_pi(time, rates):
    """tapir/compute.py:46-48 (Townsend 2007 eq. 10 as coded: 16 r^2 t exp(-4 r t))."""
    return 16 * (rates ** 2) * time * np.exp(-(4 * rates * time))


def get_time(start, stop, step=1):
    """tapir/compute.py:54-57."""
    return np.reshape(np.array(range(start, stop, step)), (-1, 1))


def get_net_pi_for_periods(pi, times):
    """tapir/compute.py:76-79."""
    sums = np.nansum(pi, axis=1)[times]
    return dict(zip(times, sums))


def get_integral_over_times(start, stop, rate):
    """tapir/compute.py:50-52 -- scipy.integrate.quad (QUADPACK dqagse), defaults."""
    from scipy import integrate
    return integrate.quad(get_townsend_pi, start, stop, args=(rate))


def get_net_integral_for_epochs(rates, epochs):
    """tapir/compute.py:81-94 (Python's sequential sum of per-site integrals and abserrs)."""
    vec = np.vectorize(get_integral_over_times)
    out = {}
    for span in epochs:
        name = "{0}-{1}".format(span[0], span[1])
        assert span[0] < span[1], "Start time [{0}] is sooner than end time [{1}]".format(span[0], span[1])
        integral, error = vec(span[0], span[1], rates)
        out[name] = {"sum(integral)": sum(integral), "sum(error)": sum(error)}
    return out


def cull_uninformative_rates(rates, inform):
    """tapir/compute.py:108-110."""
    return rates * inform


def informative_mask_from_chars(rows, threshold=4):
    """tapir/compute.py:96-106 on already-parsed sequences (list of equal-length strings):
    1.0 where at least `threshold` cells are one of A/T/G/C (case-insensitive), else NaN."""
    n = len(rows[0])
    counts = np.zeros(n, dtype=np.int64)
    for seq in rows:
        assert len(seq) == n
        for idx, cell in enumerate(seq):
            counts[idx] += 1 if (len(cell) == 1 and cell.upper() in "ATGC") else 0
    return np.array([1 if counts[x] >= threshold else np.nan for x in range(n)])


def worker_tables(rates, T, times, epochs):
    """bin/tapir_compute.py:114-122 for one locus: net PI per time, discrete PI, interval sums."""
    tv = get_time(0, T)
    pi = get_townsend_pi(tv, rates)
    pi_net = np.nansum(pi, axis=1)
    pi_times = get_net_pi_for_periods(pi, times)
    pi_epochs = get_net_integral_for_epochs(rates[np.isfinite(rates)], epochs)
    return pi_net, pi_times, pi_epochs


# --------------------------------------------------------------------------------------------------
# minimal NEXUS / Newick readers (oracle-private; the product has its own in tapir_amd/)
# --------------------------------------------------------------------------------------------------

_IUPAC = {"A": 1, "C": 2, "G": 4, "T": 8, "U": 8, "R": 5, "Y": 10, "S": 6, "W": 9, "K": 12, "M": 3,
          "B": 14, "D": 13, "H": 11, "V": 7, "N": 15, "?": 15, "-": 15, "X": 15}


def read_nexus_matrix(path):
    """Returns (names, rows) from the MATRIX block of a simple (non-interleaved) NEXUS data file."""
    txt = open(path).read()
    m = re.search(r"matrix\s*(.*?);", txt, flags=re.I | re.S)
    names, rows = [], []
    for line in m.group(1).strip().splitlines():
        line = line.strip()
        if not line:
            continue
        name, seq = line.split(None, 1)
        names.append(name)
        rows.append(seq.replace(" ", ""))
    return names, rows


def encode_rows(rows):
    """list of sequences -> uint8 [ntaxa, ncols] bit masks (A=1,C=2,G=4,T=8, gap/?/N=15)."""
    return np.array([[_IUPAC[c.upper()] for c in seq] for seq in rows], dtype=np.uint8)


class _Node:
    def __init__(self):
        self.children, self.name, self.length = [], None, None


def parse_newick(text):
    """Recursive-descent Newick reader -> root _Node (names, branch lengths; comments [..] dropped)."""
    text = re.sub(r"\[[^\]]*\]", "", text).strip()
    pos = [0]

    def node():
        n = _Node()
        if text[pos[0]] == "(":
            pos[0] += 1
            while True:
                n.children.append(node())
                if text[pos[0]] == ",":
                    pos[0] += 1
                    continue
                assert text[pos[0]] == ")"
                pos[0] += 1
                break
        m = re.match(r"[^:,;()\s]*", text[pos[0]:])
        if m.group(0):
            n.name = m.group(0)
        pos[0] += m.end()
        if pos[0] < len(text) and text[pos[0]] == ":":
            m = re.match(r":\s*([0-9eE+\-.]+)", text[pos[0]:])
            n.length = float(m.group(1))
            pos[0] += m.end()
        return n

    return node()


def tree_arrays(root, taxon_names):
    """Post-order arrays for tapir_oracle.c: parent[], blen[], leaf_taxon[] (root last)."""
    order = []

    def walk(n):
        for c in n.children:
            walk(c)
        order.append(n)

    walk(root)
    idx = {id(n): i for i, n in enumerate(order)}
    parent = np.full(len(order), -1, dtype=np.int32)
    blen = np.zeros(len(order), dtype=np.float64)
    leaf = np.full(len(order), -1, dtype=np.int32)
    for n in order:
        for c in n.children:
            parent[idx[id(c)]] = idx[id(n)]
        if n is not root:
            blen[idx[id(n)]] = n.length if n.length else 0.0
        if not n.children:
            leaf[idx[id(n)]] = taxon_names.index(n.name)
    return parent, blen, leaf


def correct_branch_lengths_values(newick_text):
    """tapir/compute.py:59-74 without the file write: returns (depth, factor, root with scaled lengths).

    depth = root -> tip distance (DendroPy distance_from_tip: max over leaves);
    mean = tree_length / (2 * n_leaves - 3); factor = 10**len(str(int(mean + 0.5))) if that length > 1."""
    root = parse_newick(newick_text)

    def depth(n):
        return max([(c.length or 0.0) + depth(c) for c in n.children], default=0.0)

    def length(n):
        return sum((c.length or 0.0) + length(c) for c in n.children)

    def leaves(n):
        return 1 if not n.children else sum(leaves(c) for c in n.children)

    d = depth(root)
    mean_bl = length(root) / (2 * leaves(root) - 3)
    string_len = len(str(int(mean_bl + 0.5)))
    factor = 10 ** string_len if string_len > 1 else 1

    def scale(n):
        for c in n.children:
            if c.length:
                c.length /= factor
            scale(c)

    scale(root)
    return d, factor, root


# --------------------------------------------------------------------------------------------------
# ctypes wrappers over tapir_oracle.c
# --------------------------------------------------------------------------------------------------

AUTO_START_MIN_TAXA = 32   # = kFirstStepMinTaxa of the engine (csrc/site_rate_params.hpp): TPHIP_START_AUTO


def site_rates(states, parent, blen, leaf_taxon, pi, exch, cat_rates=None, cat_weights=None, start_mode=None):
    """HyPhy stage 2 restatement for one locus. states: uint8 [ntaxa, ncols] masks.
    cat_rates / cat_weights: optional discrete rate mixture on top of the site rate (not in the reference).
    start_mode: 0 = the product's optimiser from the parsimony start (accelerated exits); 1 = reference-faithful:
    every column starts at siteRate = 1 (models_and_rates.bf:1050), plain safeguarded Newton to 1e-12; 2 = the
    product's step rule and exits started at siteRate = 1 (the engine's TPHIP_START_REFERENCE); None = what the engine
    does by default (TPHIP_START_AUTO): 2 on trees of fewer than 32 taxa, 0 from 32 on.
    Returns dict(rate, subst, lnl, flag, nres, nevals)."""
    states = np.ascontiguousarray(states, dtype=np.uint8)
    ntaxa, ncols = states.shape
    if start_mode is None:
        start_mode = 0 if ntaxa >= AUTO_START_MIN_TAXA else 2
    parent = np.ascontiguousarray(parent, dtype=np.int32)
    blen = np.ascontiguousarray(blen, dtype=np.float64)
    leaf_taxon = np.ascontiguousarray(leaf_taxon, dtype=np.int32)
    pi = np.ascontiguousarray(pi, dtype=np.float64)
    exch = np.ascontiguousarray(exch, dtype=np.float64)
    rate = np.empty(ncols); subst = np.empty(ncols); lnl = np.empty(ncols)
    flag = np.empty(ncols, dtype=np.uint8); nres = np.empty(ncols, dtype=np.int32)
    if cat_rates is not None and len(cat_rates) > 1:
        cr = np.ascontiguousarray(cat_rates, dtype=np.float64)
        cw = np.ascontiguousarray(cat_weights if cat_weights is not None else np.full(len(cr), 1.0 / len(cr)), dtype=np.float64)
        fn = lib().orc_site_rates_mix
        fn.restype = ctypes.c_int64
        ne = fn(states.ctypes.data_as(_c_u8p), ctypes.c_int64(ncols), ctypes.c_int32(ntaxa), ctypes.c_int32(len(parent)),
                parent.ctypes.data_as(_c_i32p), _dp(blen), leaf_taxon.ctypes.data_as(_c_i32p), _dp(pi), _dp(exch),
                ctypes.c_int32(len(cr)), _dp(cr), _dp(cw), _dp(rate), _dp(subst), _dp(lnl),
                flag.ctypes.data_as(_c_u8p), nres.ctypes.data_as(_c_i32p))
        return dict(rate=rate, subst=subst, lnl=lnl, flag=flag, nres=nres, nevals=int(ne))
    if start_mode:
        fn = lib().orc_site_rates_mode
        fn.restype = ctypes.c_int64
        ne = fn(states.ctypes.data_as(_c_u8p), ctypes.c_int64(ncols), ctypes.c_int32(ntaxa), ctypes.c_int32(len(parent)),
                parent.ctypes.data_as(_c_i32p), _dp(blen), leaf_taxon.ctypes.data_as(_c_i32p), _dp(pi), _dp(exch),
                ctypes.c_int32(int(start_mode)), _dp(rate), _dp(subst), _dp(lnl), flag.ctypes.data_as(_c_u8p),
                nres.ctypes.data_as(_c_i32p))
        return dict(rate=rate, subst=subst, lnl=lnl, flag=flag, nres=nres, nevals=int(ne))
    ne = lib().orc_site_rates(states.ctypes.data_as(_c_u8p), ncols, ntaxa, len(parent),
                              parent.ctypes.data_as(_c_i32p), _dp(blen), leaf_taxon.ctypes.data_as(_c_i32p),
                              _dp(pi), _dp(exch), _dp(rate), _dp(subst), _dp(lnl),
                              flag.ctypes.data_as(_c_u8p), nres.ctypes.data_as(_c_i32p))
    return dict(rate=rate, subst=subst, lnl=lnl, flag=flag, nres=nres, nevals=int(ne))


def column_curve(states, parent, blen, leaf_taxon, pi, exch, col, u):
    states = np.ascontiguousarray(states, dtype=np.uint8)
    ntaxa, ncols = states.shape
    parent = np.ascontiguousarray(parent, dtype=np.int32)
    blen = np.ascontiguousarray(blen, dtype=np.float64)
    leaf_taxon = np.ascontiguousarray(leaf_taxon, dtype=np.int32)
    pi = np.ascontiguousarray(pi, dtype=np.float64)
    exch = np.ascontiguousarray(exch, dtype=np.float64)
    u = np.ascontiguousarray(u, dtype=np.float64)
    f = np.empty_like(u); g = np.empty_like(u); h = np.empty_like(u)
    lib().orc_column_curve(states.ctypes.data_as(_c_u8p), ncols, ntaxa, len(parent), parent.ctypes.data_as(_c_i32p),
                           _dp(blen), leaf_taxon.ctypes.data_as(_c_i32p), _dp(pi), _dp(exch), col, _dp(u), len(u),
                           _dp(f), _dp(g), _dp(h))
    return f, g, h


def locus_loglik(states, parent, blen, leaf_taxon, pi, exch):
    """HyPhy stage-1 objective: sum over columns of log L at site rate 1 (models_and_rates.bf:487-520)."""
    states = np.ascontiguousarray(states, dtype=np.uint8)
    ntaxa, ncols = states.shape
    parent = np.ascontiguousarray(parent, dtype=np.int32)
    blen = np.ascontiguousarray(blen, dtype=np.float64)
    leaf_taxon = np.ascontiguousarray(leaf_taxon, dtype=np.int32)
    pi = np.ascontiguousarray(pi, dtype=np.float64)
    exch = np.ascontiguousarray(exch, dtype=np.float64)
    return lib().orc_locus_loglik(states.ctypes.data_as(_c_u8p), ncols, ntaxa, len(parent), parent.ctypes.data_as(_c_i32p),
                                  _dp(blen), leaf_taxon.ctypes.data_as(_c_i32p), _dp(pi), _dp(exch))


def gtr_eigen(pi, exch):
    pi = np.ascontiguousarray(pi, dtype=np.float64)
    exch = np.ascontiguousarray(exch, dtype=np.float64)
    lam = np.empty(4); U = np.empty((4, 4)); Ui = np.empty((4, 4)); kappa = np.empty(1)
    lib().orc_gtr_eigen(_dp(pi), _dp(exch), _dp(lam), _dp(U), _dp(Ui), _dp(kappa))
    return lam, U, Ui, float(kappa[0])


def quad_townsend(a, b, rate):
    """QUADPACK dqagse restatement -> (integral, abserr, neval, ier)."""
    r = ctypes.c_double(); e = ctypes.c_double(); n = ctypes.c_int32()
    ier = lib().orc_quad_townsend(float(a), float(b), float(rate), ctypes.byref(r), ctypes.byref(e), ctypes.byref(n))
    return r.value, e.value, n.value, ier


def net_pi(rates, T):
    rates = np.ascontiguousarray(rates, dtype=np.float64)
    out = np.empty(T)
    lib().orc_net_pi(_dp(rates), rates.size, T, _dp(out))
    return out


def net_integrals(rates, intervals, mode=0):
    """mode 0: dqagse restatement (integral and abserr sums); mode 1: closed form (error = 0)."""
    rates = np.ascontiguousarray(rates, dtype=np.float64)
    iv = np.ascontiguousarray(intervals, dtype=np.int32).reshape(-1, 2)
    si = np.empty(len(iv)); se = np.empty(len(iv))
    lib().orc_net_integrals(_dp(rates), rates.size, iv.ctypes.data_as(_c_i32p), len(iv), mode, _dp(si), _dp(se))
    return si, se


def informative_counts(states):
    states = np.ascontiguousarray(states, dtype=np.uint8)
    ntaxa, ncols = states.shape
    out = np.empty(ncols, dtype=np.int32)
    lib().orc_informative_counts(states.ctypes.data_as(_c_u8p), ncols, ntaxa, out.ctypes.data_as(_c_i32p))
    return out


def round_dp(x, decimals=4):
    """HyPhy Format(x,0,4) (bf:1093-1095) followed by json parse: the double nearest to the 4-dp decimal."""
    return np.array([float("%.*f" % (decimals, v)) if np.isfinite(v) else v for v in np.atleast_1d(x)])
