"""inplacedhmc.jl_amd -- MI355X-native many-chain NUTS leapfrog/gradient engine.

Host-side mirror (Python, ctypes) of the hot-path API of chriselrod/InplaceDHMC.jl over the C ABI in
include/idhmc.h.  The compute lives in libidhmc.so (hand-written HIP for gfx950); nothing here
computes on the CPU and nothing falls back to a CPU path.
"""
from ._lib import IdhmcError, LIB_PATH, load as load_library  # noqa: F401
from .engine import (Engine, Model, IsoGaussian, DiagGaussian, DenseMVN, CustomDensity, default_options,  # noqa: F401
                     TREE_STATS_DTYPE, EPS_PER_CHAIN, EPS_GLOBAL, METRIC_PER_CHAIN, METRIC_SHARED, METRIC_POOLED, GRAD_STORE, GRAD_RECOMPUTE,
                     T_ADAPT_EPS, T_ACCUM_METRIC, T_ACCUM_MOMENTS, T_KEEP_P, T_USE_DIRECTIONS, T_ACCUM_DIAG,
                     XCHG_DOUBLES, XCHG_ACCEPT, XCHG_LOGEPS, POOL_SEGMENT, xchg_accumulate, xchg_mean,
                     ERR_BAD_ARG, ERR_HIP, ERR_EPS_UNDERFLOW, ERR_STEPSIZE_SEARCH, ERR_NONFINITE_START, ERR_NO_DEVICE, ERR_ALLOC,
                     ERR_OPTIMIZATION, ERR_PEER)
from .api import (NUTS, DualAveraging, FixedStepsize, InitialStepsizeSearch, FindLocalOptimum, TuningNUTS,  # noqa: F401,E402
                  NoProgressReport, LogProgressReport, GaussianKineticEnergy, default_warmup_stages,
                  fixed_stepsize_warmup_stages, mcmc_with_warmup, threaded_mcmc, run_stages, num_stored)
from . import diagnostics as Diagnostics  # noqa: F401,E402
from .diagnostics import (EBFMI, summarize_tree_statistics, summary_from_counters, ess, rhat_from_moments,  # noqa: F401,E402
                          ess_from_moments)
from . import distributed  # noqa: F401,E402

TreeStatisticsNUTS = TREE_STATS_DTYPE  # reference name (src/NUTS.jl:229)
