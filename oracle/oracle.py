"""TEST INFRASTRUCTURE: ctypes binding of the CPU oracle (oracle/idhmc_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product (inplacedhmc.jl_amd) never does.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# IDHMC_ORACLE_LIB: a build of the same source made elsewhere (bench.py's cpu_baseline compiles one with -march=native
# on the machine it times; same bits -- -ffp-contract=off, every fma explicit -- different instruction selection)
_SO = os.environ.get("IDHMC_ORACLE_LIB", os.path.join(_HERE, "libidhmc_oracle.so"))


def build_native(outdir):
    """gcc -O3 -march=native build of the oracle for the host this runs on; returns (path, flags)"""
    flags = ["-O3", "-march=native", "-ffp-contract=off", "-fno-math-errno", "-fPIC", "-pthread"]
    so = os.path.join(outdir, "libidhmc_oracle_native.so")
    subprocess.check_call(["gcc"] + flags + ["-shared", "-o", so, os.path.join(_HERE, "idhmc_oracle.c"), "-lm"])
    return so, " ".join(flags)


def build(force=False):
    """Compile the oracle with gcc (Makefile in this directory)."""
    if "IDHMC_ORACLE_LIB" in os.environ:
        return _SO
    src = [os.path.join(_HERE, f) for f in ("idhmc_oracle.c", "idhmc_oracle.h", "orc_math.h")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in src)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libidhmc_oracle.so"], stdout=subprocess.DEVNULL)
    return _SO


class Model(C.Structure):
    _fields_ = [("kind", C.c_int32), ("D", C.c_int32), ("L", C.c_int32),
                ("mu", C.POINTER(C.c_double)), ("tau", C.POINTER(C.c_double)),
                ("prec", C.POINTER(C.c_double)), ("fn", C.c_void_p), ("params", C.POINTER(C.c_double))]


class TreeStats(C.Structure):
    _fields_ = [("pi", C.c_double), ("acceptance_rate", C.c_double),
                ("term_left", C.c_int32), ("term_right", C.c_int32),
                ("depth", C.c_int32), ("steps", C.c_int32)]


STATS_DTYPE = np.dtype([("pi", "<f8"), ("acceptance_rate", "<f8"), ("term_left", "<i4"),
                        ("term_right", "<i4"), ("depth", "<i4"), ("steps", "<i4")])


class Options(C.Structure):
    _fields_ = [("max_depth", C.c_int32), ("min_delta", C.c_double),
                ("da_delta", C.c_double), ("da_gamma", C.c_double), ("da_kappa", C.c_double),
                ("da_t0", C.c_int32),
                ("ss_a_min", C.c_double), ("ss_a_max", C.c_double), ("ss_eps0", C.c_double),
                ("ss_C", C.c_double), ("ss_maxiter_crossing", C.c_int32),
                ("ss_maxiter_bisect", C.c_int32),
                ("init_steps", C.c_int32), ("middle_steps", C.c_int32),
                ("doubling_stages", C.c_int32), ("terminating_steps", C.c_int32),
                ("adapt_metric", C.c_int32), ("stepsize_search", C.c_int32),
                ("eps_init", C.c_double),
                ("local_opt_iterations", C.c_int32), ("local_opt_penalty", C.c_double)]


class DAState(C.Structure):
    _fields_ = [("mu", C.c_double), ("m", C.c_int64), ("Hbar", C.c_double),
                ("logeps", C.c_double), ("logeps_bar", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_SO)
    dp = C.POINTER(C.c_double)
    vp = C.c_void_p
    L.orc_default_options.argtypes = [C.POINTER(Options)]
    L.orc_chain_create.restype = vp
    L.orc_chain_create.argtypes = [C.POINTER(Model), C.POINTER(Options), C.c_uint64, C.c_uint32]
    L.orc_chain_destroy.argtypes = [vp]
    L.orc_chain_L.argtypes = [vp]
    for name in ("q", "p", "grad", "minv", "w"):
        f = getattr(L, "orc_chain_" + name)
        f.restype = dp
        f.argtypes = [vp]
    L.orc_chain_lq.restype = C.c_double
    L.orc_chain_lq.argtypes = [vp]
    L.orc_chain_set_q.argtypes = [vp, dp]
    L.orc_chain_set_minv.argtypes = [vp, dp]
    L.orc_chain_random_position.argtypes = [vp]
    L.orc_model_logdensity_and_gradient.restype = C.c_double
    L.orc_model_logdensity_and_gradient.argtypes = [C.POINTER(Model), dp, dp]
    L.orc_kinetic_energy.restype = C.c_double
    L.orc_kinetic_energy.argtypes = [dp, dp, C.c_int]
    L.orc_rand_p.argtypes = [vp, C.c_uint32]
    L.orc_chain_logdensity.restype = C.c_double
    L.orc_chain_logdensity.argtypes = [vp]
    L.orc_chain_leapfrog.argtypes = [vp, C.c_double]
    L.orc_sample_tree.argtypes = [vp, C.c_double, C.c_uint32, C.POINTER(TreeStats)]
    L.orc_sample_tree_ex.argtypes = [vp, C.c_double, C.c_uint32, C.c_int, C.c_uint32, C.c_int,
                                     C.POINTER(TreeStats)]
    L.orc_chain_last_margin.restype = C.c_double
    L.orc_chain_last_margin.argtypes = [vp]
    L.orc_da_init.argtypes = [C.POINTER(DAState), C.c_double]
    L.orc_da_adapt.argtypes = [C.POINTER(Options), C.POINTER(DAState), C.c_double]
    L.orc_da_current_eps.restype = C.c_double
    L.orc_da_current_eps.argtypes = [C.POINTER(DAState)]
    L.orc_da_final_eps.restype = C.c_double
    L.orc_da_final_eps.argtypes = [C.POINTER(DAState)]
    L.orc_metric_from_draws.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_int, C.c_double]
    L.orc_find_initial_stepsize.argtypes = [vp, dp]
    L.orc_find_local_optimum.argtypes = [vp, C.c_double, C.c_int]
    L.orc_mcmc_with_warmup.argtypes = [vp, C.c_int, dp, C.c_void_p, dp]
    L.orc_num_stored.argtypes = [C.POINTER(Options), C.c_int]
    L.orc_threaded_mcmc.argtypes = [C.POINTER(Model), C.POINTER(Options), C.c_uint64, C.c_uint32,
                                    C.c_int, C.c_int, C.c_int, dp, C.c_void_p, dp]
    L.orc_bench_leapfrog.restype = C.c_double
    L.orc_bench_leapfrog.argtypes = [C.POINTER(Model), C.c_uint64, C.c_int, C.c_int, C.c_double, dp, C.c_int]
    L.orc_bench_nuts.restype = C.c_double
    L.orc_bench_nuts.argtypes = [C.POINTER(Model), C.c_uint64, C.c_int, C.c_int, C.c_double, dp, dp, C.c_int, C.POINTER(C.c_long)]
    for name in ("log", "exp", "log1p"):
        f = getattr(L, "orc_%s_export" % name)
        f.restype = C.c_double
        f.argtypes = [C.c_double]
    L.orc_sincos2pi_export.argtypes = [C.c_double, dp, dp]
    L.orc_logaddexp_export.restype = C.c_double
    L.orc_logaddexp_export.argtypes = [C.c_double, C.c_double]
    L.orc_philox_export.argtypes = [C.POINTER(C.c_uint32)] * 3
    L.orc_dot_export.restype = C.c_double
    L.orc_dot_export.argtypes = [dp, dp, C.c_int]
    L.orc_randexp_export.restype = C.c_double
    L.orc_randexp_export.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32]
    L.orc_rand_directions_export.restype = C.c_uint32
    L.orc_rand_directions_export.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    L.orc_randn_export.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, dp]
    L.orc_logprob2_export.restype = C.c_double
    L.orc_logprob2_export.argtypes = [C.c_int, C.c_double, C.c_double]
    L.orc_is_turning_export.argtypes = [dp, dp, dp, C.c_int]
    L.orc_acceptance_rate_export.restype = C.c_double
    L.orc_acceptance_rate_export.argtypes = [C.c_double, C.c_int]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def padded_len(D):
    return (D + 127) // 128 * 128


def default_options(**kw):
    o = Options()
    lib().orc_default_options(C.byref(o))
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class OracleModel:
    """Keeps the padded parameter arrays alive next to the C struct."""

    def __init__(self, kind, D, mu=None, tau=None, prec=None):
        L = padded_len(D)
        self.D, self.L, self.kind = D, L, kind
        self.mu = self.tau = self.prec = None
        m = Model(kind=kind, D=D, L=L)
        if mu is not None:
            self.mu = np.zeros(L)
            self.mu[:D] = mu
            m.mu = _dp(self.mu)
        if tau is not None:
            self.tau = np.zeros(L)
            self.tau[:D] = tau
            m.tau = _dp(self.tau)
        if prec is not None:
            self.prec = np.zeros((L, L))
            self.prec[:D, :D] = prec
            m.prec = _dp(self.prec)
        self.c = m

    @staticmethod
    def iso(D):
        return OracleModel(0, D)

    @staticmethod
    def diag(mu, tau):
        return OracleModel(1, len(mu), mu=np.asarray(mu, float), tau=np.asarray(tau, float))

    @staticmethod
    def dense(mu, prec):
        return OracleModel(2, len(mu), mu=np.asarray(mu, float), prec=np.asarray(prec, float))

    @staticmethod
    def custom(D, c_source, params, workdir):
        """A user density for the oracle: `c_source` (C, may include "orc_math.h") must define
        double logdensity_and_gradient(const double *q, double *grad, int D, int L, const double *params);
        compiled here with the oracle's own flags."""
        src = os.path.join(workdir, "user_density.c")
        so = os.path.join(workdir, "user_density.so")
        with open(src, "w") as f:
            f.write(c_source)
        subprocess.check_call(["gcc", "-O2", "-mavx2", "-mfma", "-ffp-contract=off", "-fPIC", "-shared", "-I", _HERE,
                               "-o", so, src, "-lm"])
        m = OracleModel(3, D)
        m._userlib = C.CDLL(so)
        m.params = np.ascontiguousarray(params, dtype=np.float64)
        m.c.fn = C.cast(m._userlib.logdensity_and_gradient, C.c_void_p)
        m.c.params = _dp(m.params)
        return m

    def logdensity_and_gradient(self, q):
        qq = np.zeros(self.L)
        qq[:self.D] = q
        g = np.zeros(self.L)
        lq = lib().orc_model_logdensity_and_gradient(C.byref(self.c), _dp(qq), _dp(g))
        return lq, g[:self.D].copy()


class OracleChain:
    """One chain = one reference thread (src/mcmc.jl:150-157)."""

    def __init__(self, model, options=None, seed=1, chain_id=0):
        self.model = model
        self.opt = options if options is not None else default_options()
        self.h = lib().orc_chain_create(C.byref(model.c), C.byref(self.opt), seed, chain_id)
        self.L, self.D = model.L, model.D

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_chain_destroy(self.h)
            self.h = None

    def _vec(self, name):
        p = getattr(lib(), "orc_chain_" + name)(self.h)
        return np.ctypeslib.as_array(p, shape=(self.L,))

    q = property(lambda s: s._vec("q"))
    p = property(lambda s: s._vec("p"))
    grad = property(lambda s: s._vec("grad"))
    minv = property(lambda s: s._vec("minv"))
    w = property(lambda s: s._vec("w"))
    lq = property(lambda s: lib().orc_chain_lq(s.h))

    def set_q(self, q):
        qq = np.zeros(self.L)
        qq[:self.D] = np.asarray(q, float)[:self.D]
        lib().orc_chain_set_q(self.h, _dp(qq))

    def set_p(self, p):
        self.p[:] = 0.0
        self.p[:self.D] = np.asarray(p, float)[:self.D]

    def set_minv(self, minv):
        mm = np.ones(self.L)
        mm[:self.D] = np.asarray(minv, float)[:self.D]
        lib().orc_chain_set_minv(self.h, _dp(mm))

    def random_position(self):
        lib().orc_chain_random_position(self.h)

    def rand_p(self, it):
        lib().orc_rand_p(self.h, it)

    def logdensity(self):
        return lib().orc_chain_logdensity(self.h)

    def leapfrog(self, eps):
        lib().orc_chain_leapfrog(self.h, eps)

    def sample_tree(self, eps, it, directions=None, refresh_p=True):
        st = TreeStats()
        lib().orc_sample_tree_ex(self.h, eps, it, int(directions is not None),
                                 int(directions or 0), int(refresh_p), C.byref(st))
        return st

    def last_margin(self):
        return lib().orc_chain_last_margin(self.h)

    def find_initial_stepsize(self):
        e = C.c_double()
        rc = lib().orc_find_initial_stepsize(self.h, C.byref(e))
        return rc, e.value

    def find_local_optimum(self, magnitude_penalty=1e-4, iterations=50):
        """FindLocalOptimum (src/warmup.jl:137-187) with the engine's own L-BFGS; returns 0 or -5"""
        return lib().orc_find_local_optimum(self.h, float(magnitude_penalty), int(iterations))

    def mcmc_with_warmup(self, N):
        NS = lib().orc_num_stored(C.byref(self.opt), N)
        chain = np.zeros((NS, self.L))
        stats = np.zeros(NS, dtype=STATS_DTYPE)
        e = C.c_double()
        rc = lib().orc_mcmc_with_warmup(self.h, N, _dp(chain), stats.ctypes.data, C.byref(e))
        return rc, chain, stats, e.value


def threaded_mcmc(model, N, nchains, options=None, seed=1, first_chain=0, nthreads=None):
    opt = options if options is not None else default_options()
    NS = lib().orc_num_stored(C.byref(opt), N)
    chains = np.zeros((nchains, NS, model.L))
    stats = np.zeros((nchains, NS), dtype=STATS_DTYPE)
    eps = np.zeros(nchains)
    nthreads = nthreads or os.cpu_count() or 1
    rc = lib().orc_threaded_mcmc(C.byref(model.c), C.byref(opt), seed, first_chain, nchains, N,
                                 nthreads, _dp(chains), stats.ctypes.data, _dp(eps))
    return rc, chains, stats, eps


def da_trace(eps0, accepts, options=None):
    opt = options if options is not None else default_options()
    s = DAState()
    lib().orc_da_init(C.byref(s), eps0)
    out = []
    for a in accepts:
        lib().orc_da_adapt(C.byref(opt), C.byref(s), float(a))
        out.append((s.mu, s.m, s.Hbar, s.logeps, s.logeps_bar,
                    lib().orc_da_current_eps(C.byref(s)), lib().orc_da_final_eps(C.byref(s))))
    return out


def metric_from_draws(draws, D, lam):
    """draws: (N, L) row per draw."""
    N, L = draws.shape
    minv = np.zeros(L)
    w = np.zeros(L)
    d = np.ascontiguousarray(draws)
    lib().orc_metric_from_draws(_dp(minv), _dp(w), _dp(d), L, D, N, lam)
    return minv, w


def bench_nuts(model, nchains, transitions, eps, minv=None, q0=None, seed=1, nthreads=1):
    """NUTS transitions at fixed eps on `nthreads` host threads: (seconds, leapfrog steps taken)"""
    mm = None
    if minv is not None:
        mm = np.ones(model.L)
        mm[:model.D] = minv
    qq = None
    if q0 is not None:
        qq = np.zeros((nchains, model.L))
        qq[:, :model.D] = q0
    steps = C.c_long()
    t = lib().orc_bench_nuts(C.byref(model.c), seed, nchains, transitions, eps, _dp(mm) if mm is not None else None,
                             _dp(qq) if qq is not None else None, nthreads, C.byref(steps))
    return t, steps.value


def bench_leapfrog(model, nchains, sweeps, eps, minv=None, seed=1, nthreads=1):
    mm = None
    if minv is not None:
        mm = np.ones(model.L)
        mm[:model.D] = minv
    return lib().orc_bench_leapfrog(C.byref(model.c), seed, nchains, sweeps, eps,
                                    _dp(mm) if mm is not None else None, nthreads)


# ---- the global-stepsize exchange (include/idhmc.h "the global-stepsize exchange"), restated with Python integers ----
# A value x is the integer v = rint(clamp(x) * 2^S); limbs hi = v >> B (arithmetic), lo = v & (2^B - 1); limb sums are
# integers.  kind 0 = acceptance rates (S = 52, B = 26, clamp [0, 1]); kind 1 = log eps (S = 40, B = 25, clamp +-1024).
XCHG_ACCEPT, XCHG_LOGEPS = 0, 1
_XCHG = {XCHG_ACCEPT: (52, 26), XCHG_LOGEPS: (40, 25)}


def xchg_record(kind, values):
    S, B = _XCHG[kind]
    hi = lo = 0
    n = 0
    for x in np.asarray(values, dtype=np.float64).ravel():
        if kind == XCHG_ACCEPT:
            x = 0.0 if not (x >= 0.0) else min(float(x), 1.0)
        else:
            x = 0.0 if x != x else max(-1024.0, min(float(x), 1024.0))
        v = int(np.rint(np.float64(x) * np.float64(2.0 ** S)))
        hi += v >> B
        lo += v & ((1 << B) - 1)
        n += 1
    return [hi, lo, n]


def xchg_mean(kind, rec):
    S, B = _XCHG[kind]
    v = np.float64(rec[0]) * np.float64(2.0 ** B) + np.float64(rec[1])
    return float((v * np.float64(2.0 ** -S)) / np.float64(rec[2]))


def global_initial_eps(chains):
    """global-eps mode after the per-chain searches (momentum already drawn): exp(pooled mean of log eps)"""
    L = lib()
    logs = []
    for ch in chains:
        rc, e = ch.find_initial_stepsize()
        assert rc == 0
        logs.append(L.orc_log_export(e))
    return L.orc_exp_export(xchg_mean(XCHG_LOGEPS, xchg_record(XCHG_LOGEPS, logs)))


def global_eps_stage(chains, N, iter0, eps0, options=None):
    """warmup!(TuningNUTS{Nothing}) with ONE dual-averaging state fed by the pooled mean acceptance of `chains`
    (src/warmup.jl:269-314 with the exchange of include/idhmc.h): returns (eps used per transition, final eps)."""
    opt = options if options is not None else default_options()
    L = lib()
    s = DAState()
    L.orc_da_init(C.byref(s), eps0)
    used = []
    for n in range(N):
        eps = L.orc_da_current_eps(C.byref(s))
        used.append(eps)
        acc = [ch.sample_tree(eps, iter0 + 1 + n).acceptance_rate for ch in chains]
        L.orc_da_adapt(C.byref(opt), C.byref(s), xchg_mean(XCHG_ACCEPT, xchg_record(XCHG_ACCEPT, acc)))
    return np.array(used), L.orc_da_final_eps(C.byref(s))
