"""ctypes binding of libidhmc.so (include/idhmc.h).  There is NO fallback: if the HIP library is
missing or does not load, importing the engine fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("IDHMC_LIB", os.path.join(_HERE, "libidhmc.so"))  # IDHMC_LIB: diagnostic builds


class IdhmcError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("idhmc error %d: %s" % (code, msg))
        self.code = code


class ModelDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("D", C.c_int32), ("mu", C.POINTER(C.c_double)),
                ("tau", C.POINTER(C.c_double)), ("prec", C.POINTER(C.c_double)),
                ("source", C.c_char_p), ("params", C.POINTER(C.c_double)), ("nparams", C.c_int64)]


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
                ("eps_init", C.c_double), ("eps_mode", C.c_int32), ("metric_mode", C.c_int32),
                ("local_opt_iterations", C.c_int32), ("leapfrog_grad_mode", C.c_int32), ("local_opt_penalty", C.c_double)]


class TreeSummary(C.Structure):
    _fields_ = [("N", C.c_int64), ("a_mean", C.c_double), ("a_quantiles", C.c_double * 5),
                ("max_depth", C.c_int64), ("divergence", C.c_int64), ("turning", C.c_int64),
                ("depth_counts", C.c_int64 * 33)]


DIAG_COUNTERS = 39 + 1024
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p)

# every symbol include/idhmc.h declares: name -> (restype, argtypes)
_vp, _dp, _i32, _u32, _i64, _u64, _dbl = (C.c_void_p, C.POINTER(C.c_double), C.c_int32, C.c_uint32,
                                          C.c_int64, C.c_uint64, C.c_double)
SYMBOLS = {
    "idhmc_default_options": (None, [C.POINTER(Options)]),
    "idhmc_last_error": (C.c_char_p, []),
    "idhmc_version": (C.c_int, []),
    "idhmc_build_digest": (C.c_char_p, []),
    "idhmc_create": (C.c_int, [C.POINTER(_vp), C.c_int, _i64, _i64, C.POINTER(ModelDesc), C.POINTER(Options), _u64]),
    "idhmc_destroy": (C.c_int, [_vp]),
    "idhmc_set_stream": (C.c_int, [_vp, _vp]),
    "idhmc_synchronize": (C.c_int, [_vp]),
    "idhmc_nchains": (_i64, [_vp]),
    "idhmc_dim": (_i32, [_vp]),
    "idhmc_padded_dim": (_i32, [_vp]),
    "idhmc_device_bytes": (_i64, [_vp]),
    "idhmc_placement_info": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "idhmc_lanes_info": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "idhmc_placement_cost": (C.c_int, [_vp, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "idhmc_set_q": (C.c_int, [_vp, _dp]),
    "idhmc_random_position": (C.c_int, [_vp]),
    "idhmc_set_p": (C.c_int, [_vp, _dp]),
    "idhmc_set_minv": (C.c_int, [_vp, _dp, C.c_int]),
    "idhmc_set_eps": (C.c_int, [_vp, _dbl]),
    "idhmc_set_eps_per_chain": (C.c_int, [_vp, _dp]),
    "idhmc_get_q": (C.c_int, [_vp, _dp]),
    "idhmc_get_p": (C.c_int, [_vp, _dp]),
    "idhmc_get_grad": (C.c_int, [_vp, _dp]),
    "idhmc_get_minv": (C.c_int, [_vp, _dp]),
    "idhmc_get_lq": (C.c_int, [_vp, _dp]),
    "idhmc_get_eps": (C.c_int, [_vp, _dp]),
    "idhmc_logdensity": (C.c_int, [_vp, _dp]),
    "idhmc_refresh_momentum": (C.c_int, [_vp, _u32]),
    "idhmc_leapfrog": (C.c_int, [_vp, _dbl, _i32]),
    "idhmc_set_leapfrog_grad_mode": (C.c_int, [_vp, _i32]),
    "idhmc_leapfrog_own_eps": (C.c_int, [_vp, _i32]),
    "idhmc_nuts_transition": (C.c_int, [_vp, _u32, _u32]),
    "idhmc_nuts_transitions": (C.c_int, [_vp, _u32, _i32, _u32]),
    "idhmc_fused_launch_info": (C.c_int, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "idhmc_set_directions": (C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    "idhmc_get_tree_stats": (C.c_int, [_vp, _vp]),
    "idhmc_poll_abort": (C.c_int, [_vp, _i32, C.POINTER(C.c_int32)]),
    "idhmc_find_local_optimum": (C.c_int, [_vp, _dbl, _i32]),
    "idhmc_find_initial_stepsize": (C.c_int, [_vp]),
    "idhmc_find_initial_stepsize_per_chain": (C.c_int, [_vp]),
    "idhmc_da_init": (C.c_int, [_vp]),
    "idhmc_da_finalize": (C.c_int, [_vp]),
    "idhmc_accept_sum": (C.c_int, [_vp, _vp]),
    "idhmc_logeps_sum": (C.c_int, [_vp, _vp]),
    "idhmc_da_adapt_global": (C.c_int, [_vp, _vp]),
    "idhmc_set_eps_from_logeps": (C.c_int, [_vp, _vp]),
    "idhmc_xchg_accumulate": (C.c_int, [_i32, _dp, _i64, _dp]),
    "idhmc_xchg_mean": (C.c_int, [_i32, _dp, _dp]),
    "idhmc_set_allreduce_hook": (C.c_int, [_vp, ALLREDUCE_FN, _vp, _vp]),
    "idhmc_comm_unique_id": (C.c_int, [_vp]),
    "idhmc_comm_init": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp]),
    "idhmc_comm_destroy": (C.c_int, [_vp]),
    "idhmc_comm_allreduce": (C.c_int, [_vp, _vp, _i32]),
    "idhmc_comm_info": (C.c_int, [_vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "idhmc_metric_begin": (C.c_int, [_vp]),
    "idhmc_metric_update": (C.c_int, [_vp, _dbl]),
    "idhmc_pool_partials": (C.c_int, [_vp, _i32, _vp, _i64, _i64]),
    "idhmc_pool_consume": (C.c_int, [_vp, _i32, _vp, _i64, _dbl]),
    "idhmc_moments_reset": (C.c_int, [_vp]),
    "idhmc_get_moments": (C.c_int, [_vp, _dp, _dp, C.POINTER(C.c_int64)]),
    "idhmc_diag_reset": (C.c_int, [_vp]),
    "idhmc_get_diag_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "idhmc_tree_summary_from_counters": (C.c_int, [C.POINTER(C.c_uint64), C.POINTER(TreeSummary)]),
    "idhmc_get_ebfmi": (C.c_int, [_vp, _dp]),
    "idhmc_tuning_stage": (C.c_int, [_vp, _i32, _i32, _u32, _dp, _vp]),
    "idhmc_mcmc": (C.c_int, [_vp, _i32, _u32, _dp, _vp]),
    "idhmc_mcmc_with_warmup": (C.c_int, [_vp, _i32, _dp, _vp]),
    "idhmc_total_steps": (C.c_int, [_vp, C.POINTER(C.c_int64)]),
    "idhmc_time_leapfrog": (C.c_int, [_vp, _dbl, _i32, C.POINTER(C.c_float)]),
    "idhmc_time_transitions": (C.c_int, [_vp, _i32, _u32, C.POINTER(C.c_float)]),
    "idhmc_time_transitions_fused": (C.c_int, [_vp, _i32, _u32, C.POINTER(C.c_float)]),
    "idhmc_debug_counters": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
}

_lib = None


def load():
    """Load libidhmc.so and bind every symbol of include/idhmc.h.  Raises if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "inplacedhmc.jl_amd: the HIP library %s is missing. Build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` (hipcc --offload-arch=gfx950). "
            "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        f = getattr(lib, name)  # AttributeError here = header/library mismatch
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise IdhmcError(rc, load().idhmc_last_error().decode("utf-8", "replace"))
