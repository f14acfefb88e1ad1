"""Engine: thin object wrapper over the C ABI (include/idhmc.h).  All compute happens in libidhmc.so's
HIP kernels; numpy arrays are only the host side of the getters/setters."""
import ctypes as C
import numpy as np

from . import _lib
from ._lib import ModelDesc, Options, check

# reference TreeStatisticsNUTS (src/NUTS.jl:229-242), 32 bytes
TREE_STATS_DTYPE = np.dtype([("pi", "<f8"), ("acceptance_rate", "<f8"), ("term_left", "<i4"),
                             ("term_right", "<i4"), ("depth", "<i4"), ("steps", "<i4")])
assert TREE_STATS_DTYPE.itemsize == 32

MODEL_ISO_GAUSSIAN, MODEL_DIAG_GAUSSIAN, MODEL_DENSE_MVN, MODEL_CUSTOM = 0, 1, 2, 3
EPS_PER_CHAIN, EPS_GLOBAL = 0, 1
METRIC_PER_CHAIN, METRIC_SHARED, METRIC_POOLED = 0, 1, 2
GRAD_STORE, GRAD_RECOMPUTE = 0, 1
T_ADAPT_EPS, T_ACCUM_METRIC, T_ACCUM_MOMENTS, T_KEEP_P, T_USE_DIRECTIONS, T_ACCUM_DIAG = 1, 2, 4, 8, 16, 32
XCHG_DOUBLES, XCHG_ACCEPT, XCHG_LOGEPS = 4, 0, 1
POOL_SEGMENT = 1024
# status codes of include/idhmc.h (IdhmcError.code)
(ERR_BAD_ARG, ERR_HIP, ERR_EPS_UNDERFLOW, ERR_STEPSIZE_SEARCH, ERR_NONFINITE_START, ERR_NO_DEVICE, ERR_ALLOC, ERR_OPTIMIZATION,
 ERR_PEER) = range(1, 10)


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def xchg_accumulate(kind, values, record=None):
    """Host side of the global-stepsize exchange (include/idhmc.h): add the fixed-point record of `values` to
    `record` (4 doubles: hi-limb sum, lo-limb sum, count, errors).  Integer-valued, so records of shards add exactly."""
    rec = np.zeros(XCHG_DOUBLES) if record is None else record
    v = np.ascontiguousarray(values, dtype=np.float64).ravel()
    check(_lib.load().idhmc_xchg_accumulate(int(kind), _dp(v), v.size, _dp(rec)))
    return rec


def xchg_mean(kind, record):
    rec = np.ascontiguousarray(record, dtype=np.float64)
    out = C.c_double()
    check(_lib.load().idhmc_xchg_mean(int(kind), _dp(rec), C.byref(out)))
    return out.value


class Model:
    """The user log density handed to the engine (reference: an AbstractProbabilityModel{D} with
    logdensity_and_gradient!, src/kinetic_energy.jl:73): a built-in device density or HIP source."""

    def __init__(self, kind, D, mu=None, tau=None, prec=None, source=None, params=None):
        self.kind, self.D = int(kind), int(D)
        self.source = None if source is None else source.encode("utf-8")
        self.params = None if params is None else np.ascontiguousarray(params, dtype=np.float64).ravel()
        self.mu = None if mu is None else np.ascontiguousarray(mu, dtype=np.float64)
        self.tau = None if tau is None else np.ascontiguousarray(tau, dtype=np.float64)
        self.prec = None if prec is None else np.ascontiguousarray(prec, dtype=np.float64)
        for name, arr, shape in (("mu", self.mu, (self.D,)), ("tau", self.tau, (self.D,)),
                                 ("prec", self.prec, (self.D, self.D))):
            if arr is not None and arr.shape != shape:
                raise ValueError("%s must have shape %s" % (name, shape))

    def desc(self):
        d = ModelDesc(kind=self.kind, D=self.D)
        if self.mu is not None:
            d.mu = _dp(self.mu)
        if self.tau is not None:
            d.tau = _dp(self.tau)
        if self.prec is not None:
            d.prec = _dp(self.prec)
        if self.source is not None:
            d.source = self.source
        if self.params is not None and self.params.size:
            d.params = _dp(self.params)
            d.nparams = self.params.size
        return d


def IsoGaussian(D):
    """l(q) = -1/2 |q|^2"""
    return Model(MODEL_ISO_GAUSSIAN, D)


def DiagGaussian(mu, sigma=None, tau=None):
    """l(q) = -1/2 sum (q-mu)^2 / sigma^2"""
    mu = np.asarray(mu, dtype=np.float64)
    if tau is None:
        tau = 1.0 / np.asarray(sigma, dtype=np.float64) ** 2
    return Model(MODEL_DIAG_GAUSSIAN, mu.shape[0], mu=mu, tau=tau)


def DenseMVN(mu, prec):
    """l(q) = -1/2 (q-mu)' prec (q-mu)"""
    mu = np.asarray(mu, dtype=np.float64)
    return Model(MODEL_DENSE_MVN, mu.shape[0], mu=mu, prec=prec)


def CustomDensity(D, source, params=None):
    """A user-supplied density as HIP device source (include/idhmc.h, IDHMC_MODEL_CUSTOM): `source` defines
    template <int NCH> __device__ double logdensity_and_gradient(const Vec<NCH>&, Vec<NCH>&, const UserCtx&);
    it is compiled with hipRTC against the engine's kernels when the Engine is created."""
    return Model(MODEL_CUSTOM, D, source=source, params=params)


def default_options(**kw):
    o = Options()
    _lib.load().idhmc_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError("idhmc_options has no field %r" % k)
        setattr(o, k, v)
    return o


class Engine:
    """All chains of one device (one context per device, include/idhmc.h)."""

    def __init__(self, model, nchains, options=None, seed=1, first_chain=0, device=0):
        self.lib = _lib.load()
        self.model = model
        self.opt = options if options is not None else default_options()
        self.C, self.D = int(nchains), model.D
        h = C.c_void_p()
        desc = model.desc()
        check(self.lib.idhmc_create(C.byref(h), device, self.C, first_chain, C.byref(desc),
                                    C.byref(self.opt), seed))
        self.h = h
        self._hook = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.idhmc_destroy(self.h)
            self.h = None

    __del__ = close

    # ---- state ------------------------------------------------------------------------------------
    def _get_mat(self, fn):
        out = np.empty((self.C, self.D))
        check(fn(self.h, _dp(out)))
        return out

    def _get_vec(self, fn):
        out = np.empty(self.C)
        check(fn(self.h, _dp(out)))
        return out

    def _mat(self, a):
        a = np.ascontiguousarray(a, dtype=np.float64)
        if a.shape != (self.C, self.D):
            raise ValueError("expected shape (%d, %d), got %s" % (self.C, self.D, a.shape))
        return a

    def set_q(self, q):
        check(self.lib.idhmc_set_q(self.h, _dp(self._mat(q))))

    def set_p(self, p):
        check(self.lib.idhmc_set_p(self.h, _dp(self._mat(p))))

    def random_position(self):
        check(self.lib.idhmc_random_position(self.h))

    def set_minv(self, minv):
        minv = np.ascontiguousarray(minv, dtype=np.float64)
        if minv.shape == (self.D,):
            check(self.lib.idhmc_set_minv(self.h, _dp(minv), 0))
        else:
            check(self.lib.idhmc_set_minv(self.h, _dp(self._mat(minv)), 1))

    def set_eps(self, eps):
        if np.ndim(eps) == 0:
            check(self.lib.idhmc_set_eps(self.h, float(eps)))
        else:
            e = np.ascontiguousarray(eps, dtype=np.float64)
            if e.shape != (self.C,):
                raise ValueError("eps must be a scalar or have shape (%d,)" % self.C)
            check(self.lib.idhmc_set_eps_per_chain(self.h, _dp(e)))

    q = property(lambda s: s._get_mat(s.lib.idhmc_get_q))
    p = property(lambda s: s._get_mat(s.lib.idhmc_get_p))
    grad = property(lambda s: s._get_mat(s.lib.idhmc_get_grad))
    minv = property(lambda s: s._get_mat(s.lib.idhmc_get_minv))
    lq = property(lambda s: s._get_vec(s.lib.idhmc_get_lq))
    eps = property(lambda s: s._get_vec(s.lib.idhmc_get_eps))

    def logdensity(self):
        return self._get_vec(self.lib.idhmc_logdensity)

    def placement_info(self):
        """(probe rate in GB/s of the placement of the state arrays that was kept, candidates tried); (0.0, 1) when not probed"""
        g, n = C.c_double(0.0), C.c_int32(0)
        check(self.lib.idhmc_placement_info(self.h, C.byref(g), C.byref(n)))
        return float(g.value), int(n.value)

    def placement_cost(self):
        """what the placement search of idhmc_create cost: dict(create_ms, peak_transient_bytes, single_array_GBps, kind)"""
        ms, pk, sg, kd = C.c_double(0.0), C.c_int64(0), C.c_double(0.0), C.c_int32(0)
        check(self.lib.idhmc_placement_cost(self.h, C.byref(ms), C.byref(pk), C.byref(sg), C.byref(kd)))
        return {"create_ms": float(ms.value), "peak_transient_bytes": int(pk.value), "single_array_GBps": float(sg.value),
                "kind": ("separate allocations", "one allocation, arrays 2050 MiB apart", "one mapped physical allocation (VMM)",
                         "separate allocations found by the pair walk")[int(kd.value)]}

    def lanes_info(self):
        """(lanes of the dense single-step sweep in use, of them on different hardware queues)"""
        a, b = C.c_int32(0), C.c_int32(0)
        check(self.lib.idhmc_lanes_info(self.h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def device_bytes(self):
        return int(self.lib.idhmc_device_bytes(self.h))

    def padded_dim(self):
        return int(self.lib.idhmc_padded_dim(self.h))

    def synchronize(self):
        check(self.lib.idhmc_synchronize(self.h))

    def set_stream(self, stream_ptr):
        check(self.lib.idhmc_set_stream(self.h, C.c_void_p(stream_ptr)))

    # ---- hot path ----------------------------------------------------------------------------------
    def refresh_momentum(self, it):
        check(self.lib.idhmc_refresh_momentum(self.h, it))

    def set_leapfrog_grad_mode(self, mode):
        """GRAD_STORE (0) or GRAD_RECOMPUTE (1): see idhmc_options.leapfrog_grad_mode"""
        check(self.lib.idhmc_set_leapfrog_grad_mode(self.h, int(mode)))

    def leapfrog(self, eps=None, n_steps=1):
        if eps is None:
            check(self.lib.idhmc_leapfrog_own_eps(self.h, n_steps))
        else:
            check(self.lib.idhmc_leapfrog(self.h, float(eps), n_steps))

    def nuts_transition(self, it, flags=0, directions=None):
        if directions is not None:
            d = np.ascontiguousarray(directions, dtype=np.uint32)
            if d.shape != (self.C,):
                raise ValueError("directions must have shape (%d,)" % self.C)
            check(self.lib.idhmc_set_directions(self.h, d.ctypes.data_as(C.POINTER(C.c_uint32))))
            flags |= T_USE_DIRECTIONS
        check(self.lib.idhmc_nuts_transition(self.h, it, flags))

    def nuts_transitions(self, it, n, flags=0):
        """n transitions of every chain (numbers it .. it + n - 1) in one launch; same state as n nuts_transition calls (include/idhmc.h)"""
        check(self.lib.idhmc_nuts_transitions(self.h, int(it), int(n), int(flags)))

    def fused_launch_info(self):
        """(possible on this device, used by the library's drivers) -- include/idhmc.h"""
        a, b = C.c_int32(), C.c_int32()
        check(self.lib.idhmc_fused_launch_info(self.h, C.byref(a), C.byref(b)))
        return bool(a.value), bool(b.value)

    def poll_abort(self, lag=0):
        """abort code (0 / IDHMC_ERR_EPS_UNDERFLOW) raised up to the transition `lag` launches back (include/idhmc.h)"""
        code = C.c_int32()
        check(self.lib.idhmc_poll_abort(self.h, int(lag), C.byref(code)))
        return code.value

    def tree_stats(self):
        out = np.empty(self.C, dtype=TREE_STATS_DTYPE)
        check(self.lib.idhmc_get_tree_stats(self.h, out.ctypes.data))
        return out

    def total_steps(self):
        v = C.c_int64()
        check(self.lib.idhmc_total_steps(self.h, C.byref(v)))
        return v.value

    def debug_counters(self):
        out = np.zeros(32, dtype=np.uint64)
        check(self.lib.idhmc_debug_counters(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    # ---- adaptation --------------------------------------------------------------------------------
    def find_local_optimum(self, magnitude_penalty=1e-4, iterations=50):
        check(self.lib.idhmc_find_local_optimum(self.h, float(magnitude_penalty), int(iterations)))

    def find_initial_stepsize(self, per_chain_only=False):
        """per-chain searches; in global-eps mode followed by the pooled exp(mean log eps) unless per_chain_only"""
        check((self.lib.idhmc_find_initial_stepsize_per_chain if per_chain_only else self.lib.idhmc_find_initial_stepsize)(self.h))

    def da_init(self):
        check(self.lib.idhmc_da_init(self.h))

    def da_finalize(self):
        check(self.lib.idhmc_da_finalize(self.h))

    def accept_sum(self, dev_ptr):
        check(self.lib.idhmc_accept_sum(self.h, C.c_void_p(dev_ptr)))

    def logeps_sum(self, dev_ptr):
        check(self.lib.idhmc_logeps_sum(self.h, C.c_void_p(dev_ptr)))

    def set_eps_from_logeps(self, dev_ptr):
        check(self.lib.idhmc_set_eps_from_logeps(self.h, C.c_void_p(dev_ptr)))

    def da_adapt_global(self, dev_ptr):
        check(self.lib.idhmc_da_adapt_global(self.h, C.c_void_p(dev_ptr)))

    def set_allreduce_hook(self, fn, dev_ptr):
        """fn(dev_ptr) -> None must SUM-all-reduce the XCHG_DOUBLES doubles at dev_ptr across ranks."""
        if fn is None:
            self._hook = None
            check(self.lib.idhmc_set_allreduce_hook(self.h, C.cast(None, _lib.ALLREDUCE_FN), None, None))
            return

        def _cb(buf, user):
            try:
                fn(buf)
                return 0
            except Exception:  # an exception must not unwind through C
                import traceback
                traceback.print_exc()
                return 1
        self._hook = _lib.ALLREDUCE_FN(_cb)
        check(self.lib.idhmc_set_allreduce_hook(self.h, self._hook, None, C.c_void_p(dev_ptr)))

    # native RCCL communicator for the global-eps exchange (include/idhmc.h: idhmc_comm_*)
    @staticmethod
    def comm_unique_id():
        buf = C.create_string_buffer(128)
        check(_lib.load().idhmc_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks, rank, unique_id):
        if len(unique_id) != 128:
            raise ValueError("unique id must be 128 bytes")
        check(self.lib.idhmc_comm_init(self.h, int(nranks), int(rank), C.c_char_p(bytes(unique_id))))

    def comm_destroy(self):
        check(self.lib.idhmc_comm_destroy(self.h))

    def comm_allreduce(self, dev_ptr, n=XCHG_DOUBLES):
        check(self.lib.idhmc_comm_allreduce(self.h, C.c_void_p(dev_ptr), int(n)))

    def comm_info(self):
        """(ranks, this rank, all-reduces enqueued) of the context's RCCL communicator; zeros without one"""
        nr, r, n = C.c_int32(), C.c_int32(), C.c_int64()
        check(self.lib.idhmc_comm_info(self.h, C.byref(nr), C.byref(r), C.byref(n)))
        return nr.value, r.value, n.value

    def metric_begin(self):
        check(self.lib.idhmc_metric_begin(self.h))

    def metric_update(self, lam):
        check(self.lib.idhmc_metric_update(self.h, float(lam)))

    def pool_partials(self, pass_, dev_ptr, seg_lo, seg_hi):
        """pooled metric by hand: per-segment partial sums of a pass into a device table (include/idhmc.h)"""
        check(self.lib.idhmc_pool_partials(self.h, int(pass_), C.c_void_p(dev_ptr), int(seg_lo), int(seg_hi)))

    def pool_consume(self, pass_, dev_ptr, nseg, lam):
        check(self.lib.idhmc_pool_consume(self.h, int(pass_), C.c_void_p(dev_ptr), int(nseg), float(lam)))

    def moments_reset(self):
        check(self.lib.idhmc_moments_reset(self.h))

    def moments(self):
        mean = np.empty((self.C, self.D))
        var = np.empty((self.C, self.D))
        cnt = np.empty(self.C, dtype=np.int64)
        check(self.lib.idhmc_get_moments(self.h, _dp(mean), _dp(var), cnt.ctypes.data_as(C.POINTER(C.c_int64))))
        return mean, var, cnt

    # ---- diagnostics reduced on the device (reference src/diagnostics.jl:28-32, 61-101) --------------
    def diag_reset(self):
        """start (or restart) the device-side diagnostics; mcmc() then adds every draw"""
        check(self.lib.idhmc_diag_reset(self.h))

    def diag_counters(self):
        """the context's integer counters (include/idhmc.h); counters of several ranks add"""
        out = np.zeros(_lib.DIAG_COUNTERS, dtype=np.uint64)
        check(self.lib.idhmc_get_diag_counters(self.h, out.ctypes.data_as(C.POINTER(C.c_uint64))))
        return out

    def tree_summary(self, counters=None):
        """reference summarize_tree_statistics from the device counters (of this context, or summed ones)"""
        from .diagnostics import summary_from_counters
        return summary_from_counters(self.diag_counters() if counters is None else counters)

    def ebfmi(self):
        out = np.empty(self.C)
        check(self.lib.idhmc_get_ebfmi(self.h, _dp(out)))
        return out

    # ---- drivers -----------------------------------------------------------------------------------
    def _bufs(self, N, store_draws, store_stats):
        draws = np.empty((N, self.C, self.D)) if store_draws else None
        stats = np.empty((N, self.C), dtype=TREE_STATS_DTYPE) if store_stats else None
        return draws, stats, (_dp(draws) if store_draws else None), (stats.ctypes.data if store_stats else None)

    def tuning_stage(self, N, adapt_metric, iter0, store_draws=False, store_stats=True):
        draws, stats, dpp, sp = self._bufs(N, store_draws, store_stats)
        check(self.lib.idhmc_tuning_stage(self.h, N, int(adapt_metric), iter0, dpp, sp))
        return draws, stats

    def mcmc(self, N, iter0, store_draws=True, store_stats=True):
        draws, stats, dpp, sp = self._bufs(N, store_draws, store_stats)
        check(self.lib.idhmc_mcmc(self.h, N, iter0, dpp, sp))
        return draws, stats

    def mcmc_with_warmup(self, N, store_draws=True, store_stats=True):
        draws, stats, dpp, sp = self._bufs(N, store_draws, store_stats)
        check(self.lib.idhmc_mcmc_with_warmup(self.h, N, dpp, sp))
        return draws, stats

    # ---- measurement -------------------------------------------------------------------------------
    def time_leapfrog(self, eps, sweeps):
        ms = C.c_float()
        check(self.lib.idhmc_time_leapfrog(self.h, float(eps), sweeps, C.byref(ms)))
        return ms.value

    def time_transitions(self, n, iter0):
        ms = C.c_float()
        check(self.lib.idhmc_time_transitions(self.h, n, iter0, C.byref(ms)))
        return ms.value

    def time_transitions_fused(self, n, iter0):
        """the same n transitions as one idhmc_nuts_transitions launch"""
        ms = C.c_float()
        check(self.lib.idhmc_time_transitions_fused(self.h, n, iter0, C.byref(ms)))
        return ms.value
