"""ctypes binding of libtphip.so (include/tphip.h) -- the MI355X site-rate + PI engine.

This replaces, for every locus at once, what `worker()` does per locus in the reference
(bin/tapir_compute.py:84-123): the HyPhy subprocess (Popen + JSON file round trip) and the numpy/scipy
PI arithmetic.  There is deliberately no CPU fallback: if the shared library is missing, or no GPU is
visible, every entry point raises.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# TPHIP_LIB: another build of the library (same-box A/B comparisons of kernel variants, tools/ only)
LIB_PATH = os.environ.get("TPHIP_LIB") or os.path.join(_HERE, "libtphip.so")

FLAG_OK, FLAG_FLAT, FLAG_SATURATED, FLAG_ZERO, FLAG_MAXIT = 0, 1, 2, 3, 4
START_AUTO, START_REFERENCE, START_PARSIMONY = 0, 1, 2
DEDUP_AUTO, DEDUP_OFF, DEDUP_ON = 0, 1, 2
INTEG_QUADPACK, INTEG_CLOSED = 0, 1

_vp = ctypes.c_void_p
_i32, _i64, _f64 = ctypes.c_int32, ctypes.c_int64, ctypes.c_double


class TphipError(Exception):
    """An error reported by libtphip (message from tphip_last_error)."""


class PlanDesc(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("device", _i32), ("ntaxa", _i32), ("nnodes", _i32), ("parent", _vp), ("branch_len", _vp),
                ("leaf_taxon", _vp), ("nloci", _i64), ("locus_offsets", _vp), ("pi", _vp), ("exch", _vp),
                ("T", _i32), ("times", _vp), ("n_t", _i32), ("intervals", _vp), ("n_i", _i32),
                ("integ_mode", _i32), ("correction", _f64), ("threshold", _i32), ("round_decimals", _i32),
                ("ncat", _i32), ("cat_rate", _vp), ("cat_weight", _vp), ("start_rule", _i32),
                ("pattern_dedup", _i32)]


class Stage1Opts(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("maxit_grm", _i32), ("maxit_sub", _i32), ("no_prune", _i32),
                ("fd_step", _f64), ("free_root_pair", _i32), ("compress_patterns", _i32), ("empirical_pi", _i32),
                ("row_pitch", _i64)]


# every symbol include/tphip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("tphip_version", ctypes.c_int, []),
    ("tphip_last_error", ctypes.c_char_p, []),
    ("tphip_device_count", ctypes.c_int, []),
    ("tphip_plan_create", ctypes.c_int, [ctypes.POINTER(PlanDesc), ctypes.POINTER(_vp)]),
    ("tphip_plan_destroy", ctypes.c_int, [_vp]),
    ("tphip_plan_table_width", _i32, [_vp]),
    ("tphip_plan_ncols", _i64, [_vp]),
    ("tphip_plan_workspace_bytes", ctypes.c_size_t, [_vp]),
    ("tphip_plan_chrono_length", _f64, [_vp]),
    ("tphip_plan_stack_depth", _i32, [_vp]),
    ("tphip_plan_op_counts", ctypes.c_int, [_vp, _vp]),
    ("tphip_plan_cherry_count", _i32, [_vp]),
    ("tphip_plan_get_models", ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    ("tphip_site_rates_dev", ctypes.c_int, [_vp] * 8 + [ctypes.c_size_t, _vp]),
    ("tphip_pi_tables_dev", ctypes.c_int, [_vp] * 5 + [ctypes.c_size_t, _vp]),
    ("tphip_run_dev", ctypes.c_int, [_vp] * 9 + [ctypes.c_size_t, _vp]),
    ("tphip_townsend_pi_dense_dev", ctypes.c_int, [_i32, _vp, _i64, _vp, _i32, _vp, _vp]),
    ("tphip_quad_townsend_dev", ctypes.c_int, [_i32, _vp, _i64, _f64, _f64, _i32, _vp, _vp, _vp]),
    ("tphip_locus_loglik_dev", ctypes.c_int, [_vp, _vp, _i64] + [_vp] * 9),
    ("tphip_locus_gradient_dev", ctypes.c_int, [_vp, _vp, _i64] + [_vp] * 13),
    ("tphip_state_histogram_dev", ctypes.c_int, [_i32, _vp, _i64, _i32, _vp, _i64, _vp, _vp]),
    ("tphip_profile_enable", ctypes.c_int, [_vp, _i32]),
    ("tphip_profile_read", ctypes.c_int, [_vp, ctypes.POINTER(_f64), ctypes.POINTER(_f64), ctypes.POINTER(_i64), _i32]),
    ("tphip_last_eval_count", ctypes.c_int, [_vp, ctypes.POINTER(_i64)]),
    ("tphip_host_alloc", _vp, [ctypes.c_size_t]),
    ("tphip_host_free", ctypes.c_int, [_vp]),
    ("tphip_site_rates", ctypes.c_int, [_vp] * 7),
    ("tphip_pi_tables", ctypes.c_int, [_vp] * 4),
    ("tphip_run_fused", ctypes.c_int, [_vp] * 8),
    ("tphip_run_fused_pitched", ctypes.c_int, [_vp, _vp, _i64] + [_vp] * 6),
    ("tphip_townsend_pi_dense", ctypes.c_int, [_i32, _vp, _i64, _vp, _i32, _vp]),
    ("tphip_quad_townsend", ctypes.c_int, [_i32, _vp, _i64, _f64, _f64, _i32, _vp, _vp]),
    ("tphip_state_histogram", ctypes.c_int, [_i32, _vp, _i64, _i32, _vp, _i64, _vp]),
    ("tphip_eval_columns", ctypes.c_int, [_vp] * 6),
    ("tphip_eval_columns_dev", ctypes.c_int, [_vp] * 7),
    ("tphip_corrected_rates_dev", ctypes.c_int, [_vp] * 5),
    ("tphip_corrected_rates", ctypes.c_int, [_vp] * 4),
    ("tphip_locus_loglik", ctypes.c_int, [_vp, _vp, ctypes.POINTER(_vp), _i64, _vp, _i64] + [_vp] * 7),
    ("tphip_plan_set_column_weights", ctypes.c_int, [_vp, _vp]),
    ("tphip_compress_columns", ctypes.c_int, [ctypes.c_int32, _vp, _i64, ctypes.c_int32, _vp, _i64, _vp, _vp, _vp, _vp, _vp]),
    ("tphip_locus_gradient", ctypes.c_int, [_vp, _vp, ctypes.POINTER(_vp), _i64, _vp, _i64] + [_vp] * 11),
    ("tphip_free_device", ctypes.c_int, [_vp, _vp]),
    ("tphip_stage1_fit_dev", ctypes.c_int, [_vp, _vp, ctypes.POINTER(Stage1Opts)] + [_vp] * 10),
    ("tphip_stage1_fit", ctypes.c_int, [_vp, _vp, ctypes.POINTER(_vp), ctypes.POINTER(Stage1Opts)] + [_vp] * 9),
    ("tphip_plan_set_models", ctypes.c_int, [_vp, _vp, _vp]),
]

_lib = None


def _preload_torch_hip_runtime():
    """Make sure the process ends up with ONE HIP/ROCr runtime whatever the import order.

    A process can host one ROCr: a second copy fails to acquire the GPU's VM from KFD and its hipGetDeviceCount
    reports no devices.  PyTorch-ROCm wheels bundle their own runtime (torch/lib/libamdhip64.so, SONAME
    libamdhip64.so.7) and request it by the UN-versioned file name; libtphip.so requests `libamdhip64.so.7`.  When
    torch is imported first the loader satisfies libtphip's request by SONAME from torch's copy; when libtphip.so is
    loaded first it binds /opt/rocm's copy, torch's later request for "libamdhip64.so" matches neither that name nor
    that SONAME, a second runtime is mapped, and torch then raises "No HIP GPUs are available" (seen in round 1 when
    GPU tests touched the engine before torch).  Loading torch's copy by path first makes both orders the first one.
    torch itself is not imported here."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return None
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if not os.path.exists(path):
        return None
    return ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)


def load():
    """Load libtphip.so (built in-tree by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TphipError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                             "(the engine has no CPU fallback)" % LIB_PATH)
        _preload_torch_hip_runtime()
        lib = ctypes.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        raise TphipError("libtphip error %d: %s" % (rc, load().tphip_last_error().decode("utf-8", "replace")))


def device_count():
    return load().tphip_device_count()


def _np(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _ptr(a):
    """Device pointer of a torch tensor, host pointer of a numpy array, or None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        if not a.flags["C_CONTIGUOUS"]:
            raise TphipError("array passed to libtphip is not C-contiguous")
        return a.ctypes.data
    if not a.is_contiguous():   # the library sees only the pointer: a strided tensor would be read as garbage
        raise TphipError("tensor passed to libtphip is not contiguous")
    return a.data_ptr()


class _Pinned:
    def __init__(self, ptr):
        self.ptr = ptr

    def __del__(self):
        try:
            if self.ptr:
                load().tphip_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """numpy array in pinned host memory (tphip_host_alloc): the host-pointer calls then copy by direct DMA and overlap
    the result copies with the PI kernels.  Falls back to ordinary memory when pinning fails."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    ptr = load().tphip_host_alloc(max(n, 1))
    if not ptr:
        return np.empty(shape, dtype)
    holder = _Pinned(ptr)
    buf = (ctypes.c_char * max(n, 1)).from_address(ptr)
    arr = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)
    _PINNED_HOLDERS[id(buf)] = holder   # keep the allocation alive as long as the ctypes buffer (= the array's base) lives
    import weakref
    weakref.finalize(buf, _PINNED_HOLDERS.pop, id(buf), None)
    return arr


_PINNED_HOLDERS = {}


class Plan:
    """Tree program + per-locus GTR models + PI schedule on one device (tphip_plan_*).

    parent / branch_len / leaf_taxon: post-order tree arrays (branch lengths already / correction).
    locus_offsets: [L+1] column ranges; pi [L,4]; exch [L,6] in AC,AG,AT,CG,CT,GT order.
    times / intervals: the --times and --intervals of bin/tapir_compute.py:27-33.
    """

    def __init__(self, ntaxa, parent, branch_len, leaf_taxon, locus_offsets, pi, exch, T, times, intervals,
                 correction=1.0, threshold=3, round_decimals=4, integ_mode=INTEG_QUADPACK, device=0, cat_rates=None,
                 cat_weights=None, start_rule=START_AUTO, pattern_dedup=DEDUP_AUTO):
        lib = load()
        self._lib = lib
        self._h = _vp()
        self._keep = dict(parent=_np(parent, np.int32), blen=_np(branch_len, np.float64),
                          leaf=_np(leaf_taxon, np.int32), off=_np(locus_offsets, np.int64),
                          pi=_np(pi, np.float64).reshape(-1), exch=_np(exch, np.float64).reshape(-1),
                          times=_np(times, np.int32).reshape(-1), iv=_np(intervals, np.int32).reshape(-1))
        k = self._keep
        ncat = 0 if cat_rates is None else len(cat_rates)
        if ncat > 1:   # opt-in rate mixture on top of the site rate (not in the reference; see include/tphip.h)
            k["cr"] = _np(cat_rates, np.float64).reshape(-1)
            k["cw"] = _np(np.full(ncat, 1.0 / ncat) if cat_weights is None else cat_weights, np.float64).reshape(-1)
            if k["cw"].size != ncat:
                raise TphipError("cat_weights must match cat_rates")
        self.nloci = len(k["off"]) - 1
        if k["pi"].size != 4 * self.nloci or k["exch"].size != 6 * self.nloci:
            raise TphipError("pi must be [L,4] and exch [L,6] for L = len(locus_offsets) - 1")
        if k["iv"].size % 2:
            raise TphipError("intervals must be (start, stop) pairs")
        d = PlanDesc(struct_size=ctypes.sizeof(PlanDesc), device=device, ntaxa=ntaxa, nnodes=len(k["parent"]), parent=k["parent"].ctypes.data,
                     branch_len=k["blen"].ctypes.data, leaf_taxon=k["leaf"].ctypes.data, nloci=self.nloci,
                     locus_offsets=k["off"].ctypes.data, pi=k["pi"].ctypes.data, exch=k["exch"].ctypes.data, T=int(T),
                     times=k["times"].ctypes.data, n_t=k["times"].size, intervals=k["iv"].ctypes.data,
                     n_i=k["iv"].size // 2, integ_mode=integ_mode, correction=float(correction),
                     threshold=int(threshold), round_decimals=int(round_decimals),
                     ncat=ncat if ncat > 1 else 0, cat_rate=k["cr"].ctypes.data if ncat > 1 else None,
                     cat_weight=k["cw"].ctypes.data if ncat > 1 else None, start_rule=int(start_rule),
                     pattern_dedup=int(pattern_dedup))
        _check(lib.tphip_plan_create(ctypes.byref(d), ctypes.byref(self._h)))
        self.device = device
        self.ntaxa = ntaxa
        self.T, self.n_t, self.n_i = int(T), k["times"].size, k["iv"].size // 2
        self.ncols = lib.tphip_plan_ncols(self._h)
        self.width = lib.tphip_plan_table_width(self._h)
        self.workspace_bytes = lib.tphip_plan_workspace_bytes(self._h)
        self.chrono_length = lib.tphip_plan_chrono_length(self._h)
        self.stack_depth = lib.tphip_plan_stack_depth(self._h)
        oc = np.zeros(5, np.int32)
        _check(lib.tphip_plan_op_counts(self._h, oc.ctypes.data))
        self.op_counts = dict(zip(("tip_set", "tip_mul", "branch", "push", "pop_mul"), oc.tolist()))
        self.op_counts["cherry"] = int(lib.tphip_plan_cherry_count(self._h))

    def close(self):
        if self._h:
            self._lib.tphip_plan_destroy(self._h)
            self._h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- host-pointer path (numpy in, numpy out; the library does the PCIe copies) -------------
    def site_rates(self, states):
        """states: uint8 [ntaxa, ncols] masks.  Returns dict(rate, subst, lnl, flag, nres)."""
        states = _np(states, np.uint8)
        assert states.shape == (self.ntaxa, self.ncols), (states.shape, self.ntaxa, self.ncols)
        n = self.ncols
        out = dict(rate=np.empty(n), subst=np.empty(n), lnl=np.empty(n), flag=np.empty(n, np.uint8),
                   nres=np.empty(n, np.int32))
        _check(self._lib.tphip_site_rates(self._h, states.ctypes.data, out["rate"].ctypes.data, out["subst"].ctypes.data,
                                          out["lnl"].ctypes.data, out["flag"].ctypes.data, out["nres"].ctypes.data))
        return out

    def pi_tables(self, rates, nres=None):
        """rates: float64 [ncols] raw rates (rounding, /correction and cull are applied by the kernel).
        Returns tables [L, W]."""
        rates = _np(rates, np.float64)
        assert rates.shape == (self.ncols,)
        nres = None if nres is None else _np(nres, np.int32)
        tables = np.empty((self.nloci, self.width))
        _check(self._lib.tphip_pi_tables(self._h, rates.ctypes.data, _ptr(nres), tables.ctypes.data))
        return tables

    def run_fused(self, states, pinned=False):
        """pinned=True: the result arrays live in pinned memory (direct DMA, copies overlapped with the PI kernels; when
        `states` is pinned too -- pinned_empty -- a big batch runs as a pipeline of locus groups, upload under compute)."""
        states = _np(states, np.uint8)
        assert states.shape == (self.ntaxa, self.ncols), (states.shape, self.ntaxa, self.ncols)
        n = self.ncols
        new = pinned_empty if pinned else np.empty
        out = dict(rate=new(n, np.float64), subst=new(n, np.float64), lnl=new(n, np.float64), flag=new(n, np.uint8),
                   nres=new(n, np.int32), tables=new((self.nloci, self.width), np.float64))
        _check(self._lib.tphip_run_fused(self._h, states.ctypes.data, out["rate"].ctypes.data, out["subst"].ctypes.data,
                                         out["lnl"].ctypes.data, out["flag"].ctypes.data, out["nres"].ctypes.data,
                                         out["tables"].ctypes.data))
        return out

    def run_fused_into(self, states, out, col0=0, locus0=0):
        """tphip_run_fused_pitched: `states` may be a column range of a bigger C-contiguous [ntaxa, N] array (a view
        states[:, a:b]); the results go into the caller's arrays `out` (rate, subst, lnl, flag, nres: 1-D over the big batch;
        tables [loci, W]) at column `col0` and locus `locus0` -- how pipeline.py streams a batch block by block."""
        if states.dtype != np.uint8 or states.ndim != 2 or states.strides[1] != 1:
            states = _np(states, np.uint8)
        assert states.shape == (self.ntaxa, self.ncols), (states.shape, self.ntaxa, self.ncols)
        pitch = int(states.strides[0]) if self.ntaxa > 1 else self.ncols
        a, b = int(col0), int(col0) + self.ncols
        views = [out[k][a:b] for k in ("rate", "subst", "lnl", "flag", "nres")] + [out["tables"][locus0:locus0 + self.nloci]]
        for v, dt in zip(views, (np.float64, np.float64, np.float64, np.uint8, np.int32, np.float64)):
            if v.dtype != dt or not v.flags["C_CONTIGUOUS"]:
                raise TphipError("output arrays must be C-contiguous with the documented dtypes")
        assert len(views[0]) == self.ncols and views[5].shape == (self.nloci, self.width)
        _check(self._lib.tphip_run_fused_pitched(self._h, states.ctypes.data, pitch, *[v.ctypes.data for v in views]))

    def eval_columns(self, states, u):
        """Diagnostic: (f, g, h) = log L and its u-derivatives for every column at u[ncols]."""
        states = _np(states, np.uint8)
        u = _np(u, np.float64)
        assert u.shape == (self.ncols,)
        f, g, h = np.empty(self.ncols), np.empty(self.ncols), np.empty(self.ncols)
        _check(self._lib.tphip_eval_columns(self._h, states.ctypes.data, u.ctypes.data, f.ctypes.data, g.ctypes.data,
                                            h.ctypes.data))
        return f, g, h

    def eval_columns_dev(self, d_states, d_u, d_f, d_g, d_h, stream=0):
        _check(self._lib.tphip_eval_columns_dev(self._h, _ptr(d_states), _ptr(d_u), _ptr(d_f), _ptr(d_g), _ptr(d_h), stream))

    def corrected_rates(self, rates, nres=None):
        """parse_site_rates (+ cull when nres is given) as the PI stage applies them: round4(rate) / correction."""
        rates = _np(rates, np.float64)
        assert rates.shape == (self.ncols,)
        nres = None if nres is None else _np(nres, np.int32)
        out = np.empty(self.ncols)
        _check(self._lib.tphip_corrected_rates(self._h, rates.ctypes.data, _ptr(nres), out.ctypes.data))
        return out

    def corrected_rates_dev(self, d_rates, d_nres, d_out, stream=0):
        _check(self._lib.tphip_corrected_rates_dev(self._h, _ptr(d_rates), _ptr(d_nres), _ptr(d_out), stream))

    def locus_loglik(self, states, blen_vecs, cand_locus, cand_exch, cand_vec=None, cand_scale=None, cand_pidx=None,
                     cand_pfac=None, cache=None):
        """Sum over columns of log L (site rate 1) for candidate parameter sets (tphip_locus_loglik).
        Branch lengths of candidate c = blen_vecs[cand_vec[c]] * cand_scale[c], with branch cand_pidx[c]
        additionally multiplied by cand_pfac[c].  Defaults: vec = arange, scale = 1, no perturbation.
        cache: device_cache() object that keeps the alignment on the device between calls."""
        states = _np(states, np.uint8)
        bv = _np(blen_vecs, np.float64)
        bv = bv.reshape(-1, bv.shape[-1])
        cl = _np(cand_locus, np.int32).reshape(-1)
        n = len(cl)
        ce = _np(cand_exch, np.float64).reshape(n, 6)
        cv = _np(np.arange(n) if cand_vec is None else cand_vec, np.int32).reshape(n)
        cs = _np(np.ones(n) if cand_scale is None else cand_scale, np.float64).reshape(n)
        ci = _np(np.full(n, -1) if cand_pidx is None else cand_pidx, np.int32).reshape(n)
        cf = _np(np.ones(n) if cand_pfac is None else cand_pfac, np.float64).reshape(n)
        out = np.empty(n)
        ref = ctypes.byref(cache.ptr) if cache is not None else None
        _check(self._lib.tphip_locus_loglik(self._h, states.ctypes.data, ref, bv.shape[0], bv.ctypes.data, n, cl.ctypes.data,
                                            ce.ctypes.data, cv.ctypes.data, cs.ctypes.data, ci.ctypes.data, cf.ctypes.data,
                                            out.ctypes.data))
        return out

    def locus_loglik_dev(self, d_states_ptr, ncand, d_cand_locus, d_cand_exch, d_blen_vecs, d_cand_vec, d_cand_scale,
                         d_cand_pidx, d_cand_pfac, d_out, stream=0):
        """tphip_locus_loglik_dev: every array is a device tensor (int32 / float64), nothing is copied or synchronised.
        d_states_ptr: device address of the alignment (int, e.g. the pointer a device_cache() holds)."""
        _check(self._lib.tphip_locus_loglik_dev(self._h, d_states_ptr, int(ncand), _ptr(d_cand_locus), _ptr(d_cand_exch),
                                                _ptr(d_blen_vecs), _ptr(d_cand_vec), _ptr(d_cand_scale), _ptr(d_cand_pidx),
                                                _ptr(d_cand_pfac), _ptr(d_out), stream))

    def locus_gradient_dev(self, d_states_ptr, ncand, d_cand_locus, d_cand_exch, d_blen_vecs, d_cand_vec, d_cand_scale,
                           d_cand_pidx, d_cand_pfac, d_lnl, d_dexch, d_dlogt, d_sum_dlogt, d_d2logt, stream=0):
        """tphip_locus_gradient_dev on device tensors (d_dlogt / d_d2logt may be None); nothing is copied or synchronised."""
        _check(self._lib.tphip_locus_gradient_dev(self._h, d_states_ptr, int(ncand), _ptr(d_cand_locus), _ptr(d_cand_exch),
                                                  _ptr(d_blen_vecs), _ptr(d_cand_vec), _ptr(d_cand_scale), _ptr(d_cand_pidx),
                                                  _ptr(d_cand_pfac), _ptr(d_lnl), _ptr(d_dexch), _ptr(d_dlogt),
                                                  _ptr(d_sum_dlogt), _ptr(d_d2logt), stream))

    def locus_gradient(self, states, blen_vecs, cand_locus, cand_exch, cand_vec=None, cand_scale=None, cand_pidx=None,
                       cand_pfac=None, cache=None, per_branch=True, curvature=False):
        """locus_loglik plus its derivatives (tphip_locus_gradient): returns (lnl[n], dexch[n, 6], dlogt[n, nnodes] or
        None, sum_dlogt[n]) and, with curvature=True, a fifth item d2logt[n, nnodes]; dexch holds branch lengths
        fixed, dlogt is d lnL / d log t_b, d2logt the matching second derivatives (Hessian diagonal)."""
        states = _np(states, np.uint8)
        bv = _np(blen_vecs, np.float64)
        bv = bv.reshape(-1, bv.shape[-1])
        cl = _np(cand_locus, np.int32).reshape(-1)
        n = len(cl)
        ce = _np(cand_exch, np.float64).reshape(n, 6)
        cv = _np(np.arange(n) if cand_vec is None else cand_vec, np.int32).reshape(n)
        cs = _np(np.ones(n) if cand_scale is None else cand_scale, np.float64).reshape(n)
        ci = _np(np.full(n, -1) if cand_pidx is None else cand_pidx, np.int32).reshape(n)
        cf = _np(np.ones(n) if cand_pfac is None else cand_pfac, np.float64).reshape(n)
        lnl, dex, st = np.empty(n), np.empty((n, 6)), np.empty(n)
        dlt = np.empty((n, bv.shape[1])) if per_branch else None
        d2 = np.empty((n, bv.shape[1])) if curvature else None
        ref = ctypes.byref(cache.ptr) if cache is not None else None
        _check(self._lib.tphip_locus_gradient(self._h, states.ctypes.data, ref, bv.shape[0], bv.ctypes.data, n, cl.ctypes.data,
                                              ce.ctypes.data, cv.ctypes.data, cs.ctypes.data, ci.ctypes.data, cf.ctypes.data,
                                              lnl.ctypes.data, dex.ctypes.data, dlt.ctypes.data if per_branch else None,
                                              st.ctypes.data, d2.ctypes.data if curvature else None))
        return (lnl, dex, dlt, st, d2) if curvature else (lnl, dex, dlt, st)

    def stage1_fit(self, states, cache=None, details=True, maxit_grm=0, maxit_sub=0, prune_models=True, fd_step=0.0,
                   free_root_pair=False, compress_patterns=False, empirical_pi=False):
        """HyPhy's stage 1 for every locus of the plan in one engine call (tphip_stage1_fit): model-averaged
        exchangeabilities [L, 6], the base frequencies used [L, 4] and, with details, weights / lnl [L, 203], model_exch
        [L, 203, 6], grm_blen [L, nnodes], iteration counts and counters.  The optimisers run on the device
        (csrc/stage1_opt_kernels.hpp).  `states` may be a column range of a bigger C-contiguous [ntaxa, N] array (a view
        states[:, a:b]): it is uploaded with one 2-D copy, not copied on the host.  compress_patterns: collapse the columns
        into site patterns with counts on the device first (HyPhy's dupInfo); empirical_pi: HarvestFrequencies on the device."""
        if states.dtype != np.uint8 or states.ndim != 2 or states.strides[1] != 1:
            states = _np(states, np.uint8)
        assert states.shape == (self.ntaxa, self.ncols), (states.shape, self.ntaxa, self.ncols)
        pitch = int(states.strides[0]) if self.ntaxa > 1 else self.ncols
        if pitch != self.ncols and cache is not None:
            raise ValueError("a device cache cannot hold a column range of a bigger array")
        L, nn = self.nloci, len(self._keep["parent"])
        opts = Stage1Opts(struct_size=ctypes.sizeof(Stage1Opts), maxit_grm=int(maxit_grm), maxit_sub=int(maxit_sub),
                          no_prune=0 if prune_models else 1, fd_step=float(fd_step),
                          free_root_pair=1 if free_root_pair else 0, compress_patterns=1 if compress_patterns else 0,
                          empirical_pi=1 if empirical_pi else 0, row_pitch=0 if pitch == self.ncols else pitch)
        out = dict(exch=np.empty((L, 6)), pi=np.empty((L, 4)))
        if details:
            out.update(weights=np.empty((L, 203)), lnl=np.empty((L, 203)), model_exch=np.empty((L, 203, 6)),
                       grm_blen=np.empty((L, nn)), grm_iters=np.empty(L, np.int32), sub_iters=np.empty((L, 202), np.int32))
        stats = np.zeros(8, np.int64)
        ref = ctypes.byref(cache.ptr) if cache is not None else None
        g = lambda k: out[k].ctypes.data if k in out else None  # noqa: E731
        _check(self._lib.tphip_stage1_fit(self._h, states.ctypes.data, ref, ctypes.byref(opts), out["exch"].ctypes.data,
                                          out["pi"].ctypes.data, g("weights"), g("lnl"), g("model_exch"), g("grm_blen"),
                                          g("grm_iters"), g("sub_iters"), stats.ctypes.data))
        out["stats"] = dict(zip(("nevals", "ngrads", "grm_evals", "grm_grads", "pruned", "fitted", "grm_outer_iterations",
                                 "sub_outer_iterations"), stats.tolist()))
        return out

    def set_models(self, pi=None, exch=None):
        """Replace the per-locus base frequencies [L, 4] and / or exchangeabilities [L, 6] of the plan (tphip_plan_set_models)."""
        a = None if pi is None else _np(pi, np.float64).reshape(-1)
        b = None if exch is None else _np(exch, np.float64).reshape(-1)
        if (a is not None and a.size != 4 * self.nloci) or (b is not None and b.size != 6 * self.nloci):
            raise TphipError("pi must be [L,4] and exch [L,6]")
        _check(self._lib.tphip_plan_set_models(self._h, _ptr(a), _ptr(b)))

    def set_column_weights(self, weights):
        """Column multiplicities for locus_loglik / locus_gradient (site-pattern counts); None removes them."""
        if weights is None:
            _check(self._lib.tphip_plan_set_column_weights(self._h, None))
            return
        w = _np(weights, np.float64).reshape(-1)
        if w.size != self.ncols:
            raise ValueError("weights must have one entry per column of the plan")
        _check(self._lib.tphip_plan_set_column_weights(self._h, w.ctypes.data))

    def device_cache(self):
        """Holder that keeps the alignment on the device across locus_loglik calls; release() frees it."""
        return _DeviceCache(self)

    def models(self):
        L = self.nloci
        lam, U, Ui, kappa = np.empty((L, 4)), np.empty((L, 4, 4)), np.empty((L, 4, 4)), np.empty(L)
        _check(self._lib.tphip_plan_get_models(self._h, lam.ctypes.data, U.ctypes.data, Ui.ctypes.data, kappa.ctypes.data))
        return lam, U, Ui, kappa

    # ---- device-pointer path (torch tensors resident in HBM; nothing is copied or synchronised) --
    def run_dev(self, d_states, d_rate, d_subst, d_lnl, d_flag, d_nres, d_tables, d_ws, stream=0):
        _check(self._lib.tphip_run_dev(self._h, _ptr(d_states), _ptr(d_rate), _ptr(d_subst), _ptr(d_lnl), _ptr(d_flag),
                                       _ptr(d_nres), _ptr(d_tables), _ptr(d_ws), d_ws.numel() * d_ws.element_size(),
                                       stream))

    def site_rates_dev(self, d_states, d_rate, d_subst, d_lnl, d_flag, d_nres, d_ws, stream=0):
        _check(self._lib.tphip_site_rates_dev(self._h, _ptr(d_states), _ptr(d_rate), _ptr(d_subst), _ptr(d_lnl),
                                              _ptr(d_flag), _ptr(d_nres), _ptr(d_ws),
                                              d_ws.numel() * d_ws.element_size(), stream))

    def pi_tables_dev(self, d_rates, d_nres, d_tables, d_ws, stream=0):
        _check(self._lib.tphip_pi_tables_dev(self._h, _ptr(d_rates), _ptr(d_nres), _ptr(d_tables), _ptr(d_ws),
                                             d_ws.numel() * d_ws.element_size(), stream))

    def profile_enable(self, on=True):
        _check(self._lib.tphip_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, reset=True):
        a, b, n = _f64(), _f64(), _i64()
        _check(self._lib.tphip_profile_read(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(n), 1 if reset else 0))
        return a.value, b.value, n.value

    def last_eval_count(self):
        n = _i64()
        _check(self._lib.tphip_last_eval_count(self._h, ctypes.byref(n)))
        return n.value


class _DeviceCache:
    def __init__(self, plan):
        self.plan, self.ptr = plan, _vp()

    def release(self):
        if self.ptr:
            _check(self.plan._lib.tphip_free_device(self.plan._h, self.ptr))
            self.ptr = _vp()


def state_histogram(states, locus_offsets, device=0):
    """Per-locus histogram [L, 16] of the state masks (HarvestFrequencies, bf:968), counted on the GPU."""
    states = _np(states, np.uint8)
    off = _np(locus_offsets, np.int64)
    ntaxa, ncols = states.shape
    hist = np.empty((len(off) - 1, 16), np.int64)
    _check(load().tphip_state_histogram(device, states.ctypes.data, ncols, ntaxa, off.ctypes.data, len(off) - 1,
                                        hist.ctypes.data))
    return hist


def compress_columns(states, locus_offsets, device=0, want_map=True):
    """Unique site patterns per locus (tphip_compress_columns): returns (pattern_states [ntaxa, P] uint8,
    pattern_offsets int64[L+1], weights float64[P], column -> pattern map int64[ncols] or None)."""
    states = _np(states, np.uint8)
    off = _np(locus_offsets, np.int64)
    ntaxa, ncols = states.shape
    buf = np.empty(ntaxa * max(ncols, 1), np.uint8)
    new_off = np.empty(len(off), np.int64)
    w = np.empty(max(ncols, 1))
    cmap = np.empty(max(ncols, 1), np.int64) if want_map else None
    npat = _i64()
    _check(load().tphip_compress_columns(device, states.ctypes.data, ncols, ntaxa, off.ctypes.data, len(off) - 1,
                                         buf.ctypes.data, new_off.ctypes.data, w.ctypes.data,
                                         cmap.ctypes.data if want_map else None, ctypes.byref(npat)))
    P = npat.value
    return buf[:ntaxa * P].reshape(ntaxa, P).copy(), new_off, w[:P].copy(), (cmap[:ncols] if want_map else None)


def townsend_pi_dense(times, rates, device=0):
    """tapir/compute.py:46-48 for a vector of times and a vector of rates -> (n_times, n) matrix (GPU)."""
    rates = _np(rates, np.float64).reshape(-1)
    times = _np(times, np.float64).reshape(-1)
    out = np.empty((times.size, rates.size))
    _check(load().tphip_townsend_pi_dense(device, rates.ctypes.data, rates.size, times.ctypes.data, times.size,
                                          out.ctypes.data))
    return out


def quad_townsend(a, b, rates, integ_mode=INTEG_QUADPACK, device=0):
    """tapir/compute.py:50-52 over a vector of rates -> (integral[n], abserr[n]) (GPU, dqagse emulation)."""
    rates = _np(rates, np.float64).reshape(-1)
    integral, abserr = np.empty(rates.size), np.empty(rates.size)
    _check(load().tphip_quad_townsend(device, rates.ctypes.data, rates.size, float(a), float(b), integ_mode,
                                      integral.ctypes.data, abserr.ctypes.data))
    return integral, abserr
