"""Host-side mirror of tapir/compute.py: same function names, argument meaning and error behaviour, with
the arithmetic of the hot path done by the HIP engine (tapir_amd.engine).  No DendroPy, no scipy.

Reference functions mirrored (file:line in /root/reference/tapir/compute.py):
  parse_site_rates :24-44, get_townsend_pi :46-48, get_integral_over_times :50-52, get_time :54-57,
  correct_branch_lengths :59-74, get_net_pi_for_periods :76-79, get_net_integral_for_epochs :81-94,
  get_informative_sites :96-106, cull_uninformative_rates :108-110.
"""
import json
import os

import numpy as np

from . import engine, newick, nexus


def parse_site_rates(rate_file, correction=1, test=False, count=0):
    """Parse the site-rate JSON (schema of models_and_rates.bf:1018-1104) to a vector of rates / correction.

    Unless `test`, the file is rewritten with an added "corrected_rates" list (tapir/compute.py:40-43).
    The reference's sleep-and-retry on IOError (:31-37) is not mirrored: it discards the retried result
    and exists only because HyPhy wrote the file from another process."""
    with open(rate_file, "r") as fh:
        data = json.load(fh)
    rates = np.array([line["rate"] for line in data["sites"]["rates"]])
    corrected = rates / correction
    if not test:
        data["sites"]["corrected_rates"] = [{"site": k + 1, "rate": v} for k, v in enumerate(corrected)]
        with open(rate_file, "w") as fh:
            json.dump(data, fh, indent=4)
    return corrected


def round_like_hyphy(x, decimals=4):
    """The value tapir reads back from a number HyPhy wrote with Format(x, 0, decimals) (models_and_rates.bf:1093-1095):
    the double nearest to the decimal that printf-style rounding of the EXACT binary value gives (= float("%.4f" % x)),
    vectorised.  numpy.round(x * 1e4) / 1e4 is not that: the product is rounded, so it can sit exactly on a
    half-integer the true product only comes close to, and the answer is then off by 1e-4.  Only those elements (a
    half-integer product) are ambiguous; they are settled by the exact decimal conversion itself.  Same rule as
    round_like_printf in csrc/pi_kernels.hpp."""
    x = np.asarray(x, dtype=np.float64)
    scale = 10.0 ** decimals
    src = x.reshape(-1)
    p = src * scale
    out = np.rint(p)
    np.subtract(p, out, out=p)   # in place: at C4 scale every temporary is 400 MB of pages touched for the first time
    np.abs(p, out=p)
    tie = np.flatnonzero(p == 0.5)
    np.divide(out, scale, out=out)
    for i in tie:
        out[i] = float("%.*f" % (decimals, src[i]))
    return out.reshape(x.shape) if x.ndim else out[0]


def get_townsend_pi(time, rates, device=0):
    """Townsend et al. equation 10 as coded: 16 * rates**2 * time * exp(-4 * rates * time), on the GPU.

    `time` may be a scalar, or the (T, 1) column get_time() returns, in which case the result is the
    (T, S) matrix numpy broadcasting would give; `rates` a scalar or a vector."""
    t = np.asarray(time, dtype=np.float64)
    r = np.asarray(rates, dtype=np.float64)
    out = engine.townsend_pi_dense(t.reshape(-1), r.reshape(-1), device=device)
    if t.ndim == 0 and r.ndim == 0:
        return out[0, 0]
    if t.ndim == 0:
        return out[0].reshape(r.shape)
    if r.ndim == 0:
        return out[:, 0].reshape(t.shape)
    return out.reshape(t.shape[0], -1) if t.ndim == 2 and t.shape[1] == 1 else out


def get_integral_over_times(start, stop, rate, device=0):
    """(integral, abserr) of scipy.integrate.quad(get_townsend_pi, start, stop, args=(rate)), from the GPU's
    QUADPACK dqagse emulation.  `rate` may be a scalar or a vector (the reference vectorises it, :84)."""
    r = np.asarray(rate, dtype=np.float64)
    integral, abserr = engine.quad_townsend(start, stop, r.reshape(-1), device=device)
    if r.ndim == 0:
        return integral[0], abserr[0]
    return integral.reshape(r.shape), abserr.reshape(r.shape)


def get_time(start, stop, step=1):
    """Column of times (tapir/compute.py:54-57)."""
    return np.reshape(np.array(range(start, stop, step)), (-1, 1))


def correct_tree(root):
    """tapir/compute.py:61-70 on a parsed tree, in place: returns (depth, correction_factor)."""
    depth = newick.distance_from_tip(root)
    nleaves = len(newick.leaves(root))
    mean_branch_length = newick.tree_length(root) / (2 * nleaves - 3)
    string_len = len(str(int(mean_branch_length + 0.5)))
    correction_factor = 10 ** string_len if string_len > 1 else 1
    for node in newick.postorder(root):
        if node is not root and node.length:  # `if edge.length:` skips None and zero
            node.length /= correction_factor
    return depth, correction_factor


def correct_branch_lengths(tree_file, format, d=""):
    """Scale branch lengths to values shorter than 100; writes Tree_<factor>_<depth>.newick into `d`.
    Returns (depth, correction_factor, path) like tapir/compute.py:59-74."""
    root = newick.read_tree(tree_file, format)
    depth, correction_factor = correct_tree(root)
    pth = os.path.join(d, "Tree_{0}_{1}.newick".format(correction_factor, depth))
    with open(pth, "w") as fh:
        fh.write(newick.write(root) + "\n")
    return depth, correction_factor, pth


def get_net_pi_for_periods(pi, times):
    """Sum across sites of the PI matrix at the requested times (tapir/compute.py:76-79).
    A time >= T raises IndexError exactly as numpy indexing does in the reference."""
    sums = np.nansum(pi, axis=1)[times]
    return dict(zip(times, sums))


def get_net_integral_for_epochs(rates, epochs, device=0):
    """Per interval, the sum over sites of the quad integral and of its error bound
    (tapir/compute.py:81-94); per-site values come from the GPU, the sums are Python's sequential sum."""
    epochs_results = {}
    rates = np.asarray(rates, dtype=np.float64)
    for span in epochs:
        name = "{0}-{1}".format(span[0], span[1])
        assert span[0] < span[1], \
            "Start time [{0}] is sooner than end time [{1}]".format(span[0], span[1])
        integral, error = engine.quad_townsend(span[0], span[1], rates, device=device)
        epochs_results[name] = {"sum(integral)": sum(integral), "sum(error)": sum(error)}
    return epochs_results


def informative_counts(states):
    """Per column, the number of cells that are exactly one of A/T/G/C (tapir/compute.py:104)."""
    states = np.asarray(states, dtype=np.uint8)
    return ((states == 1) | (states == 2) | (states == 4) | (states == 8)).sum(axis=0).astype(np.int32)


def get_informative_sites(alignment, threshold=4):
    """1.0 where a column has at least `threshold` unambiguous bases, NaN elsewhere
    (tapir/compute.py:96-106; note the function default is 4 while the CLI passes 3)."""
    _, states = nexus.read_states(alignment)
    counts = informative_counts(states)
    return np.array([1 if c >= threshold else np.nan for c in counts])


def cull_uninformative_rates(rates, inform):
    """NaN-out rates of uninformative sites (tapir/compute.py:108-110)."""
    return rates * inform


def discrete_gamma(alpha, ncat):
    """Yang's (1994) discrete gamma: `ncat` equiprobable categories of a Gamma(alpha, alpha) rate distribution (mean 1),
    each represented by its mean.  Returns (rates[ncat], weights[ncat]).  Host-side helper for the opt-in rate mixture
    of the site-rate stage (tphip_plan_desc.ncat); the reference's script has no such mixture."""
    from scipy.special import gammainc, gammaincinv
    ncat = int(ncat)
    if ncat <= 1:
        return np.ones(1), np.ones(1)
    if not alpha > 0:
        raise ValueError("alpha must be positive")
    cuts = gammaincinv(alpha, np.arange(1, ncat) / ncat)          # category boundaries of Gamma(alpha, 1)
    upper = np.concatenate([gammainc(alpha + 1.0, cuts), [1.0]])
    lower = np.concatenate([[0.0], gammainc(alpha + 1.0, cuts)])
    rates = ncat * (upper - lower)                                # E[X / alpha | category] with X ~ Gamma(alpha, 1)
    return rates, np.full(ncat, 1.0 / ncat)
