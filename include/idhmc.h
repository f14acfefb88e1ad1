/*
 * idhmc.h -- C ABI of the MI355X-native many-chain NUTS leapfrog/gradient engine.
 *
 * This is the drop-in boundary for the hot path of chriselrod/InplaceDHMC.jl
 * (src/kinetic_energy.jl, src/hamiltonian.jl, src/tree.jl, src/NUTS.jl,
 * src/stepsize.jl and the two caller loops of src/warmup.jl).  The reference is
 * pure Julia and has no FFI of its own; each entry point below names the
 * reference function (file:line under /root/reference) whose work it replaces
 * for ALL chains of a context at once.  A Julia host binds these with `ccall`
 * (INTEGRATION.md shows the stub); this repository's own hosts are the C++
 * drivers inside the library (idhmc_warmup / idhmc_mcmc) and the Python
 * ctypes mirror in inplacedhmc.jl_amd/.
 *
 * Conventions
 *  - plain pointers and sizes only; no C++ or torch types cross this boundary.
 *  - every function returns an int status (IDHMC_OK = 0); idhmc_last_error()
 *    gives the text.  Numerical trouble (non-finite log density, divergence) is
 *    DATA, reported through idhmc_tree_stats.termination, never an error
 *    (reference convention: src/kinetic_energy.jl:80-84,107-112, src/NUTS.jl:179-180).
 *  - host arrays are chain-major and UNPADDED: element (c, d) of an nchains x D
 *    array is at [c*D + d].  The library owns all device memory.
 *  - one context per device; calls on a context are stream-ordered and not
 *    re-entrant; distinct contexts may be used from distinct host threads
 *    (reference threading model: one chain per thread, src/mcmc.jl:150-157).
 *  - RNG streams are keyed by (seed, GLOBAL chain id, transition number), and the one pooled statistic of the
 *    global-stepsize mode is exchanged as exact integers (see "the global-stepsize exchange"), so results do not
 *    depend on how chains are sharded over devices: bit-identical for 1, 2, 4, 8 ranks (the pooled metric too, for
 *    shards aligned to IDHMC_POOL_SEGMENT chains).
 *  - all arithmetic is IEEE fp64 (the reference is Float64-only,
 *    src/warmup.jl:108-120, src/mcmc.jl:118,143).
 */
#ifndef IDHMC_H
#define IDHMC_H
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define IDHMC_VERSION 4

enum {
    IDHMC_OK = 0,
    IDHMC_ERR_BAD_ARG = 1,        /* programmer error (reference: AssertionError / MethodError) */
    IDHMC_ERR_HIP = 2,            /* a HIP runtime call failed */
    IDHMC_ERR_EPS_UNDERFLOW = 3,  /* dual-averaging eps < 1e-10 (reference src/warmup.jl:291-296) */
    IDHMC_ERR_STEPSIZE_SEARCH = 4,/* bracketing/bisection ran out of iterations (src/stepsize.jl:71,101) */
    IDHMC_ERR_NONFINITE_START = 5,/* starting point has non-finite density (src/stepsize.jl:152-153) */
    IDHMC_ERR_NO_DEVICE = 6,
    IDHMC_ERR_ALLOC = 7,
    IDHMC_ERR_OPTIMIZATION = 8,   /* FindLocalOptimum: no finite optimum after 100 restarts (src/warmup.jl:172) */
    IDHMC_ERR_PEER = 9            /* sharded run: a chain of ANOTHER rank raised one of the errors above; this rank's chains are
                                     fine, but the stage fails on every rank together (the ranks have agreed on it through the
                                     exchange record's slot [3], so none is left waiting in a collective) */
};

/* ---- downward boundary: the user log density -----------------------------
 * Reference contract: logdensity_and_gradient!(grad, model, q, sptr) -> l(q)
 * (src/kinetic_energy.jl:73,89), dimension(model) (src/warmup.jl:102).
 * Built-in device densities needed by BASELINE.json's configs, and the plug-in form:
 *
 * IDHMC_MODEL_CUSTOM -- the user's density as HIP device source, compiled at idhmc_create with hipRTC
 * against the engine's kernel templates (the device counterpart of handing the reference an
 * AbstractProbabilityModel).  `source` must define, at namespace scope,
 *
 *   template <int NCH>
 *   __device__ double logdensity_and_gradient(const Vec<NCH> &q, Vec<NCH> &grad, const UserCtx &ctx);
 *
 * called by the 64 lanes of the chain's wavefront.  Lane l holds elements {128 j + 2 l, 128 j + 2 l + 1},
 * j < NCH, of every vector (q.c[j].x, q.c[j].y); elements >= ctx.D are padding: they arrive as 0 and their
 * gradient must be returned as 0.  It returns l(q) (the same value in every lane) and fills grad l(q);
 * a non-finite return value is treated as -Inf (src/kinetic_energy.jl:80-84).  ctx gives `params`
 * (the `params` array below, in device memory), `nparams`, `D`, `lane`, and `lds` -- one L-double
 * scratch vector in LDS private to the wavefront.  Everything in inplacedhmc.jl_amd/csrc/idhmc_math.hpp
 * and idhmc_device.hpp is in scope (wave_sum, dfma, dlog, dexp, ...).  Compile errors are returned
 * through idhmc_last_error(). */
enum {
    IDHMC_MODEL_ISO_GAUSSIAN = 0,   /* l(q) = -1/2 |q|^2                       */
    IDHMC_MODEL_DIAG_GAUSSIAN = 1,  /* l(q) = -1/2 sum tau_d (q_d - mu_d)^2    */
    IDHMC_MODEL_DENSE_MVN = 2,      /* l(q) = -1/2 (q-mu)' P (q-mu), P = Sigma^-1 (fp64 MFMA) */
    IDHMC_MODEL_CUSTOM = 3          /* user HIP source, see above */
};
typedef struct {
    int32_t kind;
    int32_t D;              /* dimension(model): 1 <= D <= 1024; ISO and DIAG up to 2048 in every metric mode, CUSTOM up to
                               2048 with a SHARED or POOLED metric */
    const double *mu;       /* host, D  (DIAG, DENSE) */
    const double *tau;      /* host, D  (DIAG) */
    const double *prec;     /* host, D*D row-major, symmetric (DENSE) */
    const char *source;     /* CUSTOM: NUL-terminated HIP device source */
    const double *params;   /* CUSTOM: host, nparams doubles copied to the device (may be NULL) */
    int64_t nparams;
} idhmc_model_desc;

/* ---- options: the reference's keyword structs, flattened ------------------
 * NUTS(max_depth, min_D)            src/NUTS.jl:214-219
 * DualAveraging(d, g, k, t0)        src/stepsize.jl:191-193
 * InitialStepsizeSearch(...)        src/stepsize.jl:29-37
 * default_warmup_stages(...)        src/warmup.jl:361-372                    */
enum { IDHMC_EPS_PER_CHAIN = 0, IDHMC_EPS_GLOBAL = 1 };
enum { IDHMC_METRIC_PER_CHAIN = 0, IDHMC_METRIC_SHARED = 1, IDHMC_METRIC_POOLED = 2 };
typedef struct {
    int32_t max_depth;               /* 10 */
    double  min_delta;               /* -1000.0 */
    double  da_delta, da_gamma, da_kappa; /* 0.8, 0.05, 0.75 */
    int32_t da_t0;                   /* 10 */
    double  ss_a_min, ss_a_max, ss_eps0, ss_C; /* 0.25, 0.75, 1.0, 2.0 */
    int32_t ss_maxiter_crossing, ss_maxiter_bisect; /* 400, 400 */
    int32_t init_steps, middle_steps, doubling_stages, terminating_steps; /* 75, 25, 5, 50 */
    int32_t adapt_metric;            /* 1: TuningNUTS{Diagonal} in the doubling stages; 0: TuningNUTS{Nothing} */
    int32_t stepsize_search;         /* 1: InitialStepsizeSearch stage; 0: start from eps_init */
    double  eps_init;                /* initialization = (eps = ...), src/warmup.jl:87-92 */
    int32_t eps_mode;                /* IDHMC_EPS_PER_CHAIN = reference semantics (src/warmup.jl:284-303);
                                        IDHMC_EPS_GLOBAL = one dual-averaging state fed by the mean
                                        acceptance of all chains of all ranks (the RCCL all-reduce hook) */
    int32_t metric_mode;             /* IDHMC_METRIC_PER_CHAIN = reference semantics (src/warmup.jl:309);
                                        IDHMC_METRIC_SHARED = one fixed M^-1 for all chains, never adapted;
                                        IDHMC_METRIC_POOLED = one M^-1 for all chains, adapted from the pooled windows of
                                        every chain (of every rank, through the idhmc_comm_* communicator: 2 all-reduces
                                        of D + 1 doubles per window) -- an addition for the many-chain regime, like the
                                        global stepsize; not reference semantics.  Rank-count-invariant like the stepsize:
                                        partial sums are formed per segment of IDHMC_POOL_SEGMENT GLOBAL chain ids and
                                        added in segment order on every rank (see idhmc_pool_partials) */
    int32_t local_opt_iterations;    /* FindLocalOptimum stage of idhmc_mcmc_with_warmup (src/warmup.jl:137-150,
                                        362): 0 = skipped (default at this level), reference default 50 */
    int32_t leapfrog_grad_mode;      /* IDHMC_GRAD_STORE (default): idhmc_leapfrog(eps, 1) streams q, p, grad l in and out
                                        (6 D 8 bytes per chain-step, the reference's data movement);
                                        IDHMC_GRAD_RECOMPUTE: a separable density re-derives grad l(q) from q and does not
                                        write grad l(q') -- 4 D 8 bytes; idhmc_get_grad and every call that needs the
                                        array re-evaluate it first, results are bit-identical */
    double  local_opt_penalty;       /* magnitude_penalty, reference default 1e-4 */
} idhmc_options;

/* reference TreeStatisticsNUTS, src/NUTS.jl:229-242: exactly 32 bytes */
typedef struct {
    double  pi;                 /* logdensity(H, zeta) */
    double  acceptance_rate;
    int32_t term_left, term_right; /* InvalidTree (src/tree.jl:278-300): left==right divergence,
                                      (1,0) REACHED_MAX_DEPTH, otherwise turning */
    int32_t depth;
    int32_t steps;              /* leapfrog steps evaluated */
} idhmc_tree_stats;

enum { IDHMC_GRAD_STORE = 0, IDHMC_GRAD_RECOMPUTE = 1 };

typedef struct idhmc_ctx idhmc_ctx;

void idhmc_default_options(idhmc_options *opt);
const char *idhmc_last_error(void);
int idhmc_version(void);
/* 16 hex digits: SHA-256 prefix of the sources this library was built from (csrc/Makefile, `make print-digest` prints the tree's);
 * a build check compares the two, so a stale libidhmc.so next to edited sources does not pass for the current one */
const char *idhmc_build_digest(void);

/* Create the engine for `nchains` chains on HIP device `device`.  Chain c of
 * this context has global id first_chain_id + c.  Replaces the per-thread set-up
 * of threaded_mcmc / initialize_warmup_state (src/mcmc.jl:130-157,
 * src/warmup.jl:100-129): kappa = I, Tree arena, state vectors. */
int idhmc_create(idhmc_ctx **out, int device, int64_t nchains, int64_t first_chain_id,
                 const idhmc_model_desc *model, const idhmc_options *opt, uint64_t seed);
int idhmc_destroy(idhmc_ctx *ctx);
/* run on a caller-owned hipStream_t (e.g. a torch.cuda.Stream); NULL = the library's own non-blocking stream.
 * NB the legacy default stream's handle IS NULL: it cannot be selected, and it does not order itself against the
 * library's stream -- hand over a created stream when other work must be ordered with the engine's. */
int idhmc_set_stream(idhmc_ctx *ctx, void *hip_stream);
int idhmc_synchronize(idhmc_ctx *ctx);
int64_t idhmc_nchains(const idhmc_ctx *ctx);
int32_t idhmc_dim(const idhmc_ctx *ctx);
int32_t idhmc_padded_dim(const idhmc_ctx *ctx);
int64_t idhmc_device_bytes(const idhmc_ctx *ctx);
/* Contexts with state arrays of 64 MiB or more try several placements of q, p, grad l (and a per-chain M^-1) in HBM when they are
 * created and keep the one on which the single-step sweep's access pattern runs fastest (the same kernel differs by 10 % between
 * placements).  Reports the probe's rate on the placement kept (GB/s; 0 when nothing was probed) and how many
 * candidates were tried.  IDHMC_PLACEMENT_TRIES=1 in the environment takes the first placement. */
int idhmc_placement_info(const idhmc_ctx *ctx, double *probe_GBps, int32_t *candidates);
/* What that search cost and against what it judged: wall time of the search inside idhmc_create (ms), the most device bytes held at
 * one time while candidates were compared (the pair walk that comes first: spacers included, bounded by IDHMC_PLACEMENT_WALK_BYTES,
 * default 64 GiB, half of the free memory and 250 ms; the walk over whole sets: IDHMC_PLACEMENT_MAX_BYTES, default 16 GiB, and a quarter of it;
 * everything but the set kept is given back before idhmc_create returns),
 * the rate of ONE array alone in the same probe (GB/s; a candidate set is "good" at >= 1.10 x that), and the kind of placement
 * kept: 0 = separate allocations, 1 = one allocation with the arrays 2050 MiB apart, 2 = one physical allocation mapped with the
 * virtual-memory API, 3 = separate allocations found by the pair walk (IDHMC_PLACEMENT_PAIRS=0 skips it).  Any pointer may be NULL. */
int idhmc_placement_cost(const idhmc_ctx *ctx, double *create_ms, int64_t *peak_transient_bytes, double *single_array_GBps, int32_t *kind);
/* The dense density's single-step sweep runs as up to four lanes of kernels on four streams (lane 0 = the context's stream); lanes
 * overlap only on different hardware queues, so the library picks streams that do (an idle-kernel test at the first such sweep).
 * Reports the lanes in use (0 before the first sweep or when the sweep is one kernel) and how many of them were found on
 * different hardware queues (fewer than `lanes` under a profiler that serialises streams). */
int idhmc_lanes_info(const idhmc_ctx *ctx, int32_t *lanes, int32_t *on_distinct_queues);

/* ---- state (PhasePoint / EvaluatedLogDensity, src/hamiltonian.jl:237-276) - */
/* q <- host[nchains*D]; evaluates l(q), grad l(q) (evaluate_l!, src/kinetic_energy.jl:72-85) */
int idhmc_set_q(idhmc_ctx *ctx, const double *q);
/* q ~ U[-2,2)^D per chain, then evaluate (random_position!, src/warmup.jl:73,119-124) */
int idhmc_random_position(idhmc_ctx *ctx);
int idhmc_set_p(idhmc_ctx *ctx, const double *p);
/* M^-1 <- host; per_chain = 0: D values shared by all chains, 1: nchains*D.
 * W = 1/sqrt(M^-1) (GaussianKineticEnergy, src/hamiltonian.jl:50-57) */
int idhmc_set_minv(idhmc_ctx *ctx, const double *minv, int per_chain);
int idhmc_set_eps(idhmc_ctx *ctx, double eps);                /* all chains */
int idhmc_set_eps_per_chain(idhmc_ctx *ctx, const double *eps);
int idhmc_get_q(idhmc_ctx *ctx, double *q);                   /* nchains*D */
int idhmc_get_p(idhmc_ctx *ctx, double *p);
int idhmc_get_grad(idhmc_ctx *ctx, double *g);   /* grad l(q) of the current state; re-evaluated first when the device copy is stale: a NUTS transition of a
                                                     separable density does not write it back (nothing on the sampling path reads it), nor does the
                                                     IDHMC_GRAD_RECOMPUTE leapfrog */
int idhmc_get_minv(idhmc_ctx *ctx, double *minv);             /* nchains*D (shared metric is broadcast) */
int idhmc_get_lq(idhmc_ctx *ctx, double *lq);                 /* nchains: l(q) */
int idhmc_get_eps(idhmc_ctx *ctx, double *eps);               /* nchains */
/* pi = l(q) - K(p) per chain: logdensity(H, z), src/kinetic_energy.jl:107-112 (uses kinetic_energy :14-24) */
int idhmc_logdensity(idhmc_ctx *ctx, double *pi);

/* ---- hot path -------------------------------------------------------------- */
/* p <- W .* randn for transition number `iter` (rand_p!, src/kinetic_energy.jl:63; called
 * from src/NUTS.jl:254 and src/warmup.jl:195) */
int idhmc_refresh_momentum(idhmc_ctx *ctx, uint32_t iter);
/* n_steps fused leapfrog steps of size eps (eps < 0: backward, src/NUTS.jl:20) for every chain:
 * loop A, gradient, loop B (leapfrog, src/kinetic_energy.jl:126-163), leaving l(q') and
 * pi' = l(q') - K(p') per chain.  n_steps = 1 streams the state through HBM once.
 * Dense density, n_steps = 1, from 12 288 chains on, on the library's own stream: the sweep runs as several kernels on
 * up to four internal streams (ranges of chains; they fork from the context's stream and join it again inside the next
 * idhmc_* call of any other kind, so every call still sees the effects of all earlier ones).  With a caller-owned
 * stream (idhmc_set_stream) the sweep is one kernel on that stream, so that work the caller enqueues there is
 * ordered after it without going through the library. */
int idhmc_leapfrog(idhmc_ctx *ctx, double eps, int32_t n_steps);
/* the same with every chain's own eps (as set by idhmc_set_eps*, adaptation or the search) */
int idhmc_leapfrog_own_eps(idhmc_ctx *ctx, int32_t n_steps);
/* switch idhmc_options.leapfrog_grad_mode on a live context */
int idhmc_set_leapfrog_grad_mode(idhmc_ctx *ctx, int32_t mode);
/* one NUTS transition per chain with that chain's current eps: sample_tree (src/NUTS.jl:251-264)
 * = directions, rand_p!, sample_trajectory/adjacent_tree (src/tree.jl:321-444), leaf / turn /
 * acceptance / proposal bookkeeping (src/NUTS.jl:32-191).  iter >= 1 numbers the transition
 * (RNG address).  flags: see below.  Afterwards q, grad, l(q) hold the new draw; the momentum
 * array is unspecified until it is refreshed (the reference's next sample_tree overwrites it
 * first thing, src/NUTS.jl:254; no caller reads it in between). */
enum {
    IDHMC_T_ADAPT_EPS = 1,      /* adapt_stepsize after the transition (src/warmup.jl:303) */
    IDHMC_T_ACCUM_METRIC = 2,   /* add the new draw to the running metric window (src/warmup.jl:299,309) */
    IDHMC_T_ACCUM_MOMENTS = 4,  /* add the new draw to the running posterior mean / variance */
    IDHMC_T_KEEP_P = 8,         /* do not refresh p (reference kwarg p=..., src/NUTS.jl:251-258) */
    IDHMC_T_USE_DIRECTIONS = 16,/* use injected directions (reference kwarg directions=...) */
    IDHMC_T_ACCUM_DIAG = 32     /* add the transition to the device-side diagnostics (idhmc_diag_reset first) */
};
int idhmc_nuts_transition(idhmc_ctx *ctx, uint32_t iter, uint32_t flags);
/* n transitions of every chain, numbered iter, iter + 1, ..., iter + n - 1, in ONE launch (src/warmup.jl:288-305 and :324-330 run
 * a chain's transitions back to back; chains are independent, src/mcmc.jl:150-157).  The state afterwards is bit for bit that of n
 * calls of idhmc_nuts_transition -- every random number is addressed by (seed, chain, transition) -- but the device hands out
 * (transition, chain) pairs from one queue, so a chain's next transition starts as soon as its previous one is done and a
 * wavefront is free, instead of when the slowest tree of the whole launch is: with few chains per resident wavefront (configs[3]:
 * four) the end of every single-transition launch is a quarter of its time.  Only the records of the last transition are kept
 * (idhmc_get_tree_stats); IDHMC_T_USE_DIRECTIONS and IDHMC_T_KEEP_P are not allowed, nor IDHMC_T_ADAPT_EPS with the global stepsize (its exchange
 * sits between transitions).  If a chain raises the abort code no further transitions are started.
 * The library's own drivers (idhmc_tuning_stage, idhmc_mcmc) use it where no per-transition record leaves the device and the
 * chains are few per wavefront; IDHMC_FUSE=0 / 1 in the environment forbids / forces that. */
int idhmc_nuts_transitions(idhmc_ctx *ctx, uint32_t iter, int32_t n, uint32_t flags);
/* *possible: the device places workgroups b and b + 8 on one XCD (probed when the context was created), which the hand-over of a
 * chain inside a launch relies on -- otherwise idhmc_nuts_transitions makes n launches; *used_by_drivers: and IDHMC_FUSE does not forbid it */
int idhmc_fused_launch_info(idhmc_ctx *ctx, int32_t *possible, int32_t *used_by_drivers);
int idhmc_set_directions(idhmc_ctx *ctx, const uint32_t *directions); /* nchains, for IDHMC_T_USE_DIRECTIONS */
/* The reference aborts a warm-up the moment a chain's stepsize falls below 1e-10 (src/warmup.jl:291-296).  Every
 * transition launch is followed by an asynchronous copy of the device's abort word into pinned host memory;
 * idhmc_poll_abort waits for the word of the launch `lag` launches back (0 = the newest: a synchronisation) and
 * returns its code (0 or IDHMC_ERR_EPS_UNDERFLOW) without touching the stream otherwise.  The library's own drivers
 * poll with lag 8, i.e. stop within 8 transitions of the underflow and then fail with the code. */
int idhmc_poll_abort(idhmc_ctx *ctx, int32_t lag, int32_t *code);
int idhmc_get_tree_stats(idhmc_ctx *ctx, idhmc_tree_stats *stats);    /* nchains records of the last transition */

/* ---- adaptation (callers of the hot path, src/warmup.jl:188-314) ----------- */
/* find_initial_stepsize per chain with the momentum now in p (src/stepsize.jl:111-164,
 * src/warmup.jl:188-200); result becomes each chain's eps.  Global mode: exp(mean of log eps over all chains of all
 * ranks), the IDHMC_XCHG_LOGEPS record all-reduced through the hook / communicator -- every rank gets the same bits */
int idhmc_find_initial_stepsize(idhmc_ctx *ctx);
/* the per-chain searches alone, in every mode: for a host that pools the stepsizes itself (idhmc_logeps_sum, its own
 * exchange of the record, idhmc_set_eps_from_logeps) */
int idhmc_find_initial_stepsize_per_chain(idhmc_ctx *ctx);
/* FindLocalOptimum (src/warmup.jl:137-187): per chain, maximise l(q) - magnitude_penalty/2 * sum(q^2) for at
 * most `iterations` quasi-Newton iterations from the current q; a non-finite result restarts from a new random
 * position with the penalty doubled, at most 100 times, else the call fails with IDHMC_ERR_OPTIMIZATION
 * (:162-172).  Leaves q, l(q), grad l(q) at the optimum.  The reference calls
 * QuasiNewtonMethods.proptimize! (external, source absent); the iteration here is the engine's own
 * L-BFGS (5 pairs, Armijo backtracking), identical in oracle/ and on the device. */
int idhmc_find_local_optimum(idhmc_ctx *ctx, double magnitude_penalty, int32_t iterations);
/* initial_adaptation_state from each chain's eps (src/stepsize.jl:208-212) */
int idhmc_da_init(idhmc_ctx *ctx);
/* eps <- final_eps = exp(logeps_bar) (src/stepsize.jl:241, src/warmup.jl:313) */
int idhmc_da_finalize(idhmc_ctx *ctx);
/* ---- the global-stepsize exchange (IDHMC_EPS_GLOBAL) --------------------------------------------------
 * The reference adapts eps per chain and has no exchange (src/warmup.jl:284-303); north_star adds ONE dual-averaging
 * state fed by the mean acceptance of all chains of all ranks.  For the result not to depend on how the chains are
 * sharded, the pooled statistic is carried as EXACT integers: every chain's value x is rounded to a fixed-point
 * integer v = rint(x * 2^S), split into two limbs hi = v >> B (arithmetic), lo = v & (2^B - 1), and the limbs are
 * summed as integers (in doubles: every partial sum stays below 2^53, so any SUM all-reduce of doubles -- RCCL, gloo,
 * MPI -- in any association is exact).  The mean is ((sum hi * 2^B + sum lo) * 2^-S) / count, evaluated from the
 * totals on every rank alike.  Two statistics use it:
 *   IDHMC_XCHG_ACCEPT  acceptance rates in [0, 1]:        S = 52, B = 26   (adapt_stepsize's `a`, src/stepsize.jl:220)
 *   IDHMC_XCHG_LOGEPS  log(eps) of the per-chain searches: S = 40, B = 25   (|log eps| clamped to 1024)
 * both exact for up to 2^26 chains per job.  The exchange buffer is IDHMC_XCHG_DOUBLES doubles in device memory:
 *   [0] sum of hi limbs  [1] sum of lo limbs  [2] number of chains  [3] number of chains with a pending error status
 * so eps is bit-identical for 1, 2, 4, 8 ... ranks. */
#define IDHMC_XCHG_DOUBLES 4
enum { IDHMC_XCHG_ACCEPT = 0, IDHMC_XCHG_LOGEPS = 1,
       IDHMC_XCHG_STATUS = 2 /* library-internal: a record whose sums are zero, only slots [2] and [3] carry data (error agreement) */ };
/* device: the exchange record of the last transition's acceptance rates / of the chains' current log(eps) */
int idhmc_accept_sum(idhmc_ctx *ctx, double *dev_xchg);
int idhmc_logeps_sum(idhmc_ctx *ctx, double *dev_xchg);
/* device: adapt_stepsize (src/stepsize.jl:220-229) with the pooled mean acceptance of an (all-reduced) record */
int idhmc_da_adapt_global(idhmc_ctx *ctx, const double *dev_xchg);
/* device: every chain's eps <- exp(pooled mean of log eps) of an (all-reduced) IDHMC_XCHG_LOGEPS record */
int idhmc_set_eps_from_logeps(idhmc_ctx *ctx, const double *dev_xchg);
/* host helpers of the same protocol (no device needed): add the record of n values to xchg4 (zero it first), and
 * the mean of a record; what a caller that exchanges by other means (MPI, files) or checks a run needs */
int idhmc_xchg_accumulate(int32_t kind, const double *values, int64_t n, double *xchg4);
int idhmc_xchg_mean(int32_t kind, const double *xchg4, double *mean);
/* Let the library's own drivers (idhmc_find_initial_stepsize, idhmc_tuning_stage, idhmc_mcmc_with_warmup) run the
 * exchange: after the record has been enqueued on the context's stream the library calls fn(dev_xchg, user); fn
 * must SUM-all-reduce the IDHMC_XCHG_DOUBLES doubles at dev_xchg over all ranks, ordered after the work already on
 * the stream and before what is enqueued next (RCCL on the same stream, or torch.distributed.all_reduce on a tensor
 * aliasing dev_xchg with the context on torch's stream).  dev_xchg is caller-owned device memory.
 * fn == NULL: single-rank (no exchange).
 * Errors are agreed on before they are returned: the drivers enqueue the exchange FIRST and then fail on every rank together
 * when the all-reduced slot [3] is non-zero (local code on the rank that owns the failing chain, IDHMC_ERR_PEER on the others);
 * idhmc_tuning_stage ends with one more such record (zeros but slot [3]) before its pooled-metric collectives.
 * The hook carries ONLY this record.  IDHMC_METRIC_POOLED needs table-sized all-reduces, which go through the context's own
 * communicator (idhmc_comm_init); under a hook WITHOUT a communicator idhmc_metric_update pools the chains of this rank only
 * (rank-local metric) -- exchange the tables yourself with idhmc_pool_partials / idhmc_pool_consume for a job-wide one. */
typedef int (*idhmc_allreduce_fn)(double *dev_xchg, void *user);
int idhmc_set_allreduce_hook(idhmc_ctx *ctx, idhmc_allreduce_fn fn, void *user, double *dev_xchg);
/* Native exchange: an RCCL communicator owned by the context (one rank per GPU, xGMI inside a node).
 * Rank 0 obtains the 128-byte id with idhmc_comm_unique_id and distributes it by any host channel; every
 * rank then calls idhmc_comm_init (collective, on the context's device).  From then on the library's
 * drivers enqueue ncclAllReduce(SUM, IDHMC_XCHG_DOUBLES x fp64) on the context's stream between the record and its
 * consumer -- the only collective of the path (the reference has none: per-chain adaptation,
 * src/warmup.jl:284-303); a hook set with idhmc_set_allreduce_hook takes precedence.
 * RCCL is loaded with dlopen at the first of these calls; without it they return IDHMC_ERR_HIP. */
#define IDHMC_COMM_ID_BYTES 128
int idhmc_comm_unique_id(void *id128);
int idhmc_comm_init(idhmc_ctx *ctx, int32_t nranks, int32_t rank, const void *id128);
int idhmc_comm_destroy(idhmc_ctx *ctx);
/* for callers that drive the transitions themselves: SUM-all-reduce n doubles at dev_buf in place on the context's stream */
int idhmc_comm_allreduce(idhmc_ctx *ctx, double *dev_buf, int32_t n);
/* what the communicator has done: ranks, this rank, all-reduces enqueued since idhmc_comm_init (measurement: a
 * multi-GPU bench line reports these so that the record shows RCCL saw N ranks).  Without a communicator: 0, 0, 0. */
int idhmc_comm_info(idhmc_ctx *ctx, int32_t *nranks, int32_t *rank, int64_t *allreduces);
/* start / finish a metric window: GaussianKineticEnergy!(kappa, chain, lambda)
 * (src/hamiltonian.jl:117-189, src/warmup.jl:308-311), computed from running sums instead of a stored chain */
int idhmc_metric_begin(idhmc_ctx *ctx);
int idhmc_metric_update(idhmc_ctx *ctx, double lambda);
/* IDHMC_METRIC_POOLED by hand (what idhmc_metric_update does with the context's communicator): the column sums of a
 * pass are formed per segment of IDHMC_POOL_SEGMENT consecutive GLOBAL chain ids (ascending chain order inside) into a
 * table [seg_hi - seg_lo][padded_dim + 1] (last column: draw counts); segments this context holds no chain of are
 * written as zeros.  Tables of several contexts ADD (exact when every segment lies on one context: shards aligned to
 * the segment size); idhmc_pool_consume adds the table's rows in order -- pass 0 yields the pooled mean, pass 1 the
 * metric (reference regularisation, src/hamiltonian.jl:156-158 with the pooled count).  Call pass 0 partials, exchange,
 * consume, then the same for pass 1. */
#define IDHMC_POOL_SEGMENT 1024
int idhmc_pool_partials(idhmc_ctx *ctx, int32_t pass, double *dev_table, int64_t seg_lo, int64_t seg_hi);
int idhmc_pool_consume(idhmc_ctx *ctx, int32_t pass, const double *dev_table, int64_t nseg, double lambda);
/* running posterior moments over draws accumulated with IDHMC_T_ACCUM_MOMENTS */
int idhmc_moments_reset(idhmc_ctx *ctx);
int idhmc_get_moments(idhmc_ctx *ctx, double *mean, double *var, int64_t *count); /* nchains*D each */

/* ---- diagnostics reduced on the device (reference src/diagnostics.jl:28-32, 61-101) -----------------------
 * The reference computes EBFMI and summarize_tree_statistics from the stored TreeStatisticsNUTS records; at 65 536
 * chains the records of a run are N x 2 MiB, so the transition kernel reduces them as it goes (IDHMC_T_ACCUM_DIAG):
 * per chain the running sums EBFMI needs, and for all chains of the context integer counters -- every one an exact
 * integer, so counters of several contexts (ranks) simply add:
 *   [0] transitions  [1], [2] hi / lo limb sums of the acceptance rates (IDHMC_XCHG_ACCEPT fixed point)
 *   [3] REACHED_MAX_DEPTH  [4] divergent  [5] turning  (InvalidTree classes, src/tree.jl:278-300)
 *   [6 .. 6+32] trees of depth 0..32  [39 .. 39+1023] acceptance-rate histogram, 1024 equal bins on [0, 1]
 * idhmc_mcmc adds every draw once idhmc_diag_reset has been called. */
#define IDHMC_DIAG_COUNTERS (39 + 1024)
#define IDHMC_DIAG_ACC_BINS 1024
typedef struct {                    /* reference TreeStatisticsSummary, src/diagnostics.jl:44-55 */
    int64_t N;
    double  a_mean;                 /* exact to 2^-52 per record */
    double  a_quantiles[5];         /* 5/25/50/75/95 %, interpolated inside the histogram bin: within 1/1024 of the sample quantile */
    int64_t max_depth, divergence, turning;
    int64_t depth_counts[33];       /* first element is for depth 0 */
} idhmc_tree_summary;
int idhmc_diag_reset(idhmc_ctx *ctx);
int idhmc_get_diag_counters(idhmc_ctx *ctx, uint64_t *counters);            /* IDHMC_DIAG_COUNTERS values */
/* host only, no context: the summary of (summed) counters */
int idhmc_tree_summary_from_counters(const uint64_t *counters, idhmc_tree_summary *out);
/* EBFMI per chain = mean(abs2, diff(pi)) / var(pi) over the accumulated transitions (src/diagnostics.jl:28-32) */
int idhmc_get_ebfmi(idhmc_ctx *ctx, double *ebfmi);                          /* nchains */

/* ---- drivers: the reference's caller loops, run by the library -------------- */
/* warmup!(TuningNUTS) (src/warmup.jl:269-314): N transitions with dual averaging, optional
 * metric update at the end.  draws (host, N*nchains*D) / stats (host, N*nchains) may be NULL.
 * iter0 = number of transitions already made (RNG address continues at iter0+1). */
int idhmc_tuning_stage(idhmc_ctx *ctx, int32_t N, int32_t adapt_metric, uint32_t iter0,
                       double *draws, idhmc_tree_stats *stats);
/* mcmc! (src/warmup.jl:316-332): N transitions at fixed eps */
int idhmc_mcmc(idhmc_ctx *ctx, int32_t N, uint32_t iter0, double *draws, idhmc_tree_stats *stats);
/* mcmc_with_warmup! for all chains (src/mcmc.jl:94-105 under threaded_mcmc :130-159):
 * random start, [stepsize search], default_warmup_stages, then N draws.
 * draws: host N*nchains*D or NULL; stats: host N*nchains or NULL. */
int idhmc_mcmc_with_warmup(idhmc_ctx *ctx, int32_t N, double *draws, idhmc_tree_stats *stats);
/* total leapfrog steps taken by NUTS transitions since creation (sum of stats.steps) */
int idhmc_total_steps(idhmc_ctx *ctx, int64_t *steps);

/* ---- measurement helper ------------------------------------------------------
 * `sweeps` back-to-back idhmc_leapfrog(eps, 1) launches bracketed by HIP events on the
 * context's stream; returns the mean kernel+gap time per sweep in milliseconds. */
int idhmc_time_leapfrog(idhmc_ctx *ctx, double eps, int32_t sweeps, float *ms_per_sweep);
int idhmc_time_transitions(idhmc_ctx *ctx, int32_t n, uint32_t iter0, float *ms_total);      /* n idhmc_nuts_transition launches */
int idhmc_time_transitions_fused(idhmc_ctx *ctx, int32_t n, uint32_t iter0, float *ms_total); /* one idhmc_nuts_transitions(n) launch */
/* 32 device counters: [0] = total leapfrog steps; [1] = pending abort code; [2..] = per-phase shader-cycle sums of the
 * NUTS kernel, filled only by the diagnostic build (-DIDHMC_STAMPS, tools/stamps.sh), zero otherwise. */
int idhmc_debug_counters(idhmc_ctx *ctx, uint64_t *out32);

#ifdef __cplusplus
}
#endif
#endif
